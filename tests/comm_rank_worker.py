"""One rank of tests/test_comm_two_ranks.py (run as a subprocess: ``python -m tests.comm_rank_worker RANK WORLD DIR LIB``).

Drives the engine's OWN collective entry points -- pf_comm_unique_id / pf_broadcast_weights, i.e. csrc/comm.inl -- on the
SIMT-emulator flavour of the library, with PEPPA_RCCL_LIBRARY pointing at tests/rccl_stub (set by the parent).  The unique id
travels out of band through a file, as bench.py sends it through torch.distributed."""
import json
import os
import sys
import time

import numpy as np


def _wait_for(path, timeout=120.0):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise TimeoutError(path)
        time.sleep(0.01)
    time.sleep(0.02)
    with open(path, "rb") as f:
        return f.read()


def _publish(path, data):
    with open(path + ".tmp", "wb") as f:
        f.write(data)
    os.replace(path + ".tmp", path)


def main():
    rank, world, work, lib = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    from oracle import synth_weights as sw
    from peppa_pig_face_landmark_amd import _native
    from peppa_pig_face_landmark_amd._native import Engine
    from peppa_pig_face_landmark_amd.graph.student import build_student_program

    report = {"rank": rank}
    eng = Engine(0, lib)
    blob = None
    if rank == 0:
        blob, _ = build_student_program(sw.student_weights(), 64, "f32")
    crops = sw.smooth_blob_images(2, 64, seed=909)

    def uid(tag):
        path = os.path.join(work, "uid_" + tag)
        if rank == 0:
            u = Engine.comm_unique_id(lib)
            _publish(path, u)
            return u
        return _wait_for(path)

    # ---- case A: the broadcast proper ------------------------------------------------------------------------------
    u = uid("a")
    got, ms = eng.broadcast_weights(u, rank, world, _native.PF_NET_LANDMARK, blob, 2)
    loc, score = eng.landmark_forward(crops)
    np.savez(os.path.join(work, "out_a_rank%d.npz" % rank), loc=loc, score=score, blob=np.frombuffer(got, np.uint8))
    report["a"] = {"bytes": len(got), "rccl_version": eng.rccl_version()}

    # ---- case B: the receive buffer of rank 1 is too small: rank 1 fails, the others must come back (no deadlock) ----
    # same communicator (same id) on purpose: the collective sequence continues after the failed call
    try:
        eng.broadcast_weights(u, rank, world, _native.PF_NET_LANDMARK, blob, 2, capacity=4096)
        report["b"] = "ok"
    except _native.PeppaHipError as e:
        report["b"] = "error: " + str(e)
    # ... and the communicator is still usable afterwards
    got2, _ = eng.broadcast_weights(u, rank, world, _native.PF_NET_LANDMARK, blob, 2)
    loc2, _ = eng.landmark_forward(crops)
    report["b_after"] = bool(len(got2) == len(got) and np.array_equal(loc2, loc))

    # ---- case C: argument validation happens before any communication ----------------------------------------------------
    for tag, (r_, w_) in (("c_rank_eq_world", (world, world)), ("c_negative", (-1, world)), ("c_world0", (0, 0))):
        try:
            eng.broadcast_weights(u, r_, w_, _native.PF_NET_LANDMARK, blob if blob is not None else b"x" * 64, 2)
            report[tag] = "ok"
        except (_native.PeppaHipError, AssertionError) as e:
            report[tag] = "error: " + str(e)

    # ---- case D: ranks that disagree about the world size: nobody hangs, everybody gets an error ----------------------------
    u = uid("d")
    os.environ["PF_RCCL_STUB_TIMEOUT"] = "3"
    try:
        eng.broadcast_weights(u, rank, world if rank == 0 else world + 1, _native.PF_NET_LANDMARK, blob, 2)
        report["d"] = "ok"
    except _native.PeppaHipError as e:
        report["d"] = "error: " + str(e)
    eng.close()
    with open(os.path.join(work, "report_rank%d.json" % rank), "w") as f:
        json.dump(report, f)


if __name__ == "__main__":
    main()
