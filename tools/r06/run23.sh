#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for z in 0; do echo -n "Z=$z "; timeout 300 python tools/bench_compare_ext.py 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['abund_c3_u32']['ms'], d['abund_c3_u64']['ms'], d['abund_core_u32']['ms'])"; done
