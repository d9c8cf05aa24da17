"""Builds tests/rccl_stub/rccl_stub.cpp (a shared-memory stand-in for the six librccl entry points csrc/comm.inl binds)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.abspath(os.path.join(HERE, "..", "_build", "librccl_stub.so"))


def build_stub(force: bool = False) -> str:
    src = os.path.join(HERE, "rccl_stub.cpp")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(src):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-format-truncation", src, "-o", OUT, "-lrt", "-pthread"], check=True)
    return OUT
