#!/bin/bash
# round 5, experiment 6: the state as it stands -- full GPU suite, the driver's bench command, the round's profiles
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_exp6; mkdir -p $O
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/pytest_gpu.txt
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "rc $?" >> $O/bench.err )
( SMG_BENCH_FORCE_COLLECTIVES=1 timeout 900 python bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --no-io > $O/bench_forced_collectives.json 2> $O/bench_fc.err; echo "rc $?" >> $O/bench_fc.err )
bash tools/prof_r05.sh all > $O/prof.log 2>&1
