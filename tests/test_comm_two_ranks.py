"""pf_comm_unique_id / pf_broadcast_weights driven from TWO processes (round-2 verdict: the rank != 0 branch of csrc/comm.inl
had never executed anywhere).  The engine is the SIMT-emulator flavour, librccl is tests/rccl_stub (the six entry points comm.inl
binds, over shared memory, selected with PEPPA_RCCL_LIBRARY) -- everything above the transport is the product's own code:
communicator set-up, the size exchange, the receive-capacity check, the copy into the caller's buffer and the program load on
the receiving rank."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import helpers
from tests.rccl_stub import build_stub

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_broadcast_weights_two_ranks_stub_rccl(emu_library, tmp_path, student_weights):
    stub = build_stub()
    env = dict(os.environ, PEPPA_RCCL_LIBRARY=stub, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("PF_RCCL_STUB_TIMEOUT", None)
    procs = [subprocess.Popen([sys.executable, "-m", "tests.comm_rank_worker", str(r), "2", str(tmp_path), emu_library],
                              cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank hung in the collective")
    assert [p.returncode for p in procs] == [0, 0], "\n".join(o[-3000:] for o in outs)
    rep = [json.load(open(tmp_path / ("report_rank%d.json" % r))) for r in range(2)]
    a = [np.load(tmp_path / ("out_a_rank%d.npz" % r)) for r in range(2)]

    # A: rank 1 received rank 0's bytes, loaded them, and computes the same landmarks -- which are the oracle's
    assert rep[0]["a"]["bytes"] == rep[1]["a"]["bytes"] > 1 << 20 and rep[1]["a"]["rccl_version"] == 99900
    assert np.array_equal(a[0]["blob"], a[1]["blob"])
    assert np.array_equal(a[0]["loc"], a[1]["loc"]) and np.array_equal(a[0]["score"], a[1]["score"])
    from oracle import synth_weights as sw
    oloc, _, taps = helpers.oracle_student(student_weights, sw.smooth_blob_images(2, 64, seed=909))
    safe = helpers.heat_margins(taps) > 2e-3
    assert np.abs(a[1]["loc"] - oloc).reshape(2, 98, 2).max(2)[safe].max() < 1e-4

    # B: too small a receive buffer fails on rank 1 only, AFTER the exchange, and the communicator survives
    assert rep[0]["b"] == "ok", rep[0]["b"]
    assert rep[1]["b"].startswith("error") and "exceeds the receive capacity 4096" in rep[1]["b"], rep[1]["b"]
    assert rep[0]["b_after"] and rep[1]["b_after"]

    # C: bad rank / world are refused before any communication, on every rank
    for r in rep:
        for tag in ("c_rank_eq_world", "c_negative", "c_world0"):
            assert r[tag].startswith("error") and "bad rank" in r[tag], (tag, r[tag])

    # D: ranks that disagree about the world size get an error each instead of a hang
    assert rep[0]["d"].startswith("error") and rep[1]["d"].startswith("error"), (rep[0]["d"], rep[1]["d"])
    assert "CommInitRank" in rep[0]["d"] and "CommInitRank" in rep[1]["d"]
