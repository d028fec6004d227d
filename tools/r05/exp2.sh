#!/bin/bash
# round 5, experiment 2: full GPU suite on the new host paths, the gather loop's look-ahead A/B, bench.py with the API extras
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_exp2; mkdir -p $O
P=$GRAFT_REPO_ROOT/sourmash_amd
for v in base ""; do
  lib=$P/libsourmash_amd${v:+_$v}.so
  for pf in 1 2 3 5; do
    [ "$v" = base ] && [ $pf != 1 ] && continue
    echo "== lib ${v:-new} prefetch $pf" >> $O/gather_ab.txt
    SMG_LIBRARY=$lib SMG_GATHER_PREFETCH=$pf timeout 300 python tools/bench_gather.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d[k] for k in ('index_build_ms','rounds','loop_ms','us_per_round','overlap_pass_ms')}, all(d['checks'].values()))" >> $O/gather_ab.txt 2>&1
  done
done
SMG_GATHER_TRACE=1 timeout 300 python tools/bench_gather.py 2>&1 | grep "persistent loop\|of which" | tail -4 >> $O/gather_ab.txt
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $O/pytest_gpu.txt
( timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc $?" >> $O/bench.err; tail -5 $O/bench.err ) 
