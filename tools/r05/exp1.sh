#!/bin/bash
# round 5, experiment 1: the ring walk of overlap_lean_kernel -- correctness, then A/B against the round-4 library
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_exp1; mkdir -p $O
P=$GRAFT_REPO_ROOT/sourmash_amd
( timeout 900 python -m pytest tests/test_gpu_gather.py -x -q 2>&1 | tail -15 ) > $O/pytest_gather.txt
for v in base "" g1 g16; do
  lib=$P/libsourmash_amd${v:+_$v}.so
  echo "== lib ${v:-g8}" >> $O/bench.txt
  SMG_LIBRARY=$lib timeout 300 python tools/bench_gather.py >> $O/bench.txt 2>&1
  for qpr in 9600 8800 8000; do
    echo "-- qpr $qpr" >> $O/bench.txt
    SMG_LIBRARY=$lib SMG_OVERLAP_QPR=$qpr timeout 120 python tools/bench_overlap.py --reps 7 >> $O/bench.txt 2>&1
  done
done
# traffic of the overlap pass: FETCH_SIZE per dispatch, base vs ring
for v in base "" g16; do
  lib=$P/libsourmash_amd${v:+_$v}.so
  ( cd /tmp && SMG_LIBRARY=$lib rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf_${v:-g8} -o p -- python $GRAFT_REPO_ROOT/tools/bench_gather.py > /dev/null 2> /tmp/pf.log ) || tail -3 /tmp/pf.log >> $O/fetch.txt
  echo "== lib ${v:-g8}" >> $O/fetch.txt
  python profiles/summarize.py $(find /tmp/pf_${v:-g8} -name "*.db" | head -1) | grep -i "overlap_lean\|counter\|build_scatter\|build_count_runs" >> $O/fetch.txt
done
