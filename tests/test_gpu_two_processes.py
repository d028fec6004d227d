"""Two REAL processes -- own HIP contexts, a process group between them, DeviceBackend in each -- driving the multi-GPU
functions of sourmash_amd.parallel through their own entry points, on the one GPU a test box has (VERDICT r03
item 1b).  What differs from two GPUs of a node: the collectives are gloo's (device tensors staged through the host), and the
two ranks' resident gather loops share the CUs (SMG_GATHER_LOOP_WGS workgroups each instead of one per CU).  Everything else is
what rank r of N runs: open_exchange (name broadcast, both ranks map and register the POSIX segment, every step agreed), the
loop kernels agreeing on each round's winner through that memory, the record protocol, the tile dealing + ONE all-gather of
compare, the overlap pass per shard + ONE all-gather of search / prefetch, the sketch union.  Run with -m gpu."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_pair(extra_env):
    port = 29650 + os.getpid() % 300
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", **extra_env)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "two_process_worker.py"), str(r), "2", str(port)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    res = []
    for rc, o, e in outs:
        lines = [ln for ln in o.splitlines() if ln.startswith("RESULT ")]
        assert rc == 0 and lines, (rc, o[-1500:], e[-3000:])
        res.append(json.loads(lines[-1][7:]))
    return res


def test_two_processes_drive_every_distributed_entry_point():
    res = _run_pair({"SMG_GATHER_LOOP_WGS": "64"})
    for r in res:
        assert "error" not in r, r["error"]
        assert r["compare"] and r["overlaps"] and r["sketch_union"], r
        for thr in (0, 30_000):
            for mode in ("device", "shared", "records"):
                g = r["gather_%s_thr%d" % (mode, thr)]
                assert g["ok"] and g["rounds"] > 20, (mode, thr, g)       # whichever protocol ran: the oracle's ordered picks
            assert r["gather_records_thr%d" % thr]["protocol"] == "candidate records"
    # the resident loops of both processes agreeing every round -- through per-rank device memory mapped with hipIpc (the
    # default: what ranks on the GPUs of one node use over xGMI) and through the shared host segment -- must have been what
    # answered (a run in which the two grids were not resident together falls back, correctly, and is repeated up to 4 times)
    # WHICH of the two threshold runs used the resident transport is part of the record (a fall-back on one of the two passes the
    # `any`, correctly -- two grids sharing one GPU are not always resident together -- but it must not pass silently)
    used = {}
    for rank, r in enumerate(res):
        for mode, needle in (("device", "hipIpc"), ("shared", "shared host memory")):
            hits = [thr for thr in (0, 30_000) if needle in r["gather_%s_thr%d" % (mode, thr)]["protocol"]]
            used["rank%d_%s" % (rank, mode)] = {"resident_transport_at_threshold_bp": hits,
                                                "fell_back_at_threshold_bp": [thr for thr in (0, 30_000) if thr not in hits]}
            assert hits, (mode, r)
    print("two-process transports:", json.dumps(used))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "two_process_transports.json"), "w") as fh:
        json.dump(used, fh, indent=1)
    # both ranks run the same protocol in every run (every step is agreed by a MIN all-reduce)
    for mode in ("device", "shared"):
        assert used["rank0_" + mode] == used["rank1_" + mode], used


def test_two_full_size_grids_on_one_gpu_fall_back_instead_of_failing():
    """One workgroup per CU per rank (the default for a GPU of one's own) cannot be resident twice on one GPU: each rank's loop
    gives up at its gate with nothing touched, the ranks agree, and the record protocol answers -- never an error, never a
    hang (round 3 threw `Internal` here)."""
    res = _run_pair({"SMG_GATHER_LOOP_WGS": "0", "SMG_GATHER_GATE_US": "5000"})
    for r in res:
        assert "error" not in r, r["error"]
        for thr in (0, 30_000):
            for mode in ("device", "shared", "records"):
                assert r["gather_%s_thr%d" % (mode, thr)]["ok"], r
