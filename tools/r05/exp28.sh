#!/bin/bash
# round 5: the bench line's watchdog (a line even when a secondary metric never returns) and a two-rank rehearsal on one GPU
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_exp28; mkdir -p $O
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --extras-timeout 4 > $O/watchdog.json 2> $O/watchdog.err; echo "rc=$?" >> $O/watchdog.err
SMG_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --bases 1e9 --no-io > $O/two_ranks.json 2> $O/two_ranks.err; echo "rc=$?" >> $O/two_ranks.err
SMG_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 2 --warmup 1 --bases 1e9 --no-io --extras-timeout 3 > $O/two_ranks_watchdog.json 2> $O/two_ranks_watchdog.err; echo "rc=$?" >> $O/two_ranks_watchdog.err
for f in watchdog two_ranks two_ranks_watchdog; do echo "== $f"; tail -1 $O/$f.err; python - <<PY
import json
try:
    l=[x for x in open("$O/$f.json") if x.startswith("{")]
    d=json.loads(l[-1]); print(len(l), "line(s); value", d["value"], "n_gpus", d["n_gpus"], "extra keys", list(d["extra"].keys())[:8], "timed_out" in d["extra"], d["summary"].get("dist_gather_c5_collective"))
except Exception as e:
    print("no line:", repr(e))
PY
done
