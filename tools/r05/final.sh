#!/bin/bash
# round 5, validation pass: whole GPU suite, sketch counters for the present sources, the bench line with them, kernel stats of the
# compare-ext / protein benches, the sketch rate by k
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_final; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/pytest_gpu.txt
timeout 900 bash tools/prof_r05.sh sketch > $O/prof_sketch.log 2>&1
cp gpurun_out/r05_prof/r05_pmc.txt profiles/r05_pmc.txt      # (on the box: the bench below quotes these counters)
cp gpurun_out/r05_prof/r05_kernel_stats.txt profiles/r05_kernel_stats.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
timeout 600 bash tools/prof_r05.sh ext > $O/prof_ext.log 2>&1
timeout 600 python tools/bench_sketch_k.py long > $O/long_k.json 2> $O/long_k.err
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/pf/lk -o p -- python $GRAFT_REPO_ROOT/tools/bench_sketch_k.py long > /dev/null 2> /tmp/pf_lk.log ) || tail -3 /tmp/pf_lk.log
python profiles/summarize.py $(find /tmp/pf/lk -name "*.db" | head -1) > $O/long_k_kernels.txt
timeout 600 python tools/bench_sketch_k.py > $O/sketch_k.json 2>> $O/long_k.err
