# kernel times of the gather index build at C5 (GPU box): bash tools/prof_build.sh [SMG_GATHER_PASS1 value]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
[ -n "$1" ] && export SMG_GATHER_PASS1=$1
( cd /tmp && rm -rf /tmp/pb && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pb -o p -- python $GRAFT_REPO_ROOT/tools/bench_gather.py > /tmp/pb.json 2> /tmp/pb.log ) || tail -3 /tmp/pb.log
echo "# SMG_GATHER_PASS1=${SMG_GATHER_PASS1:-default}"
python -c "
import json
d=json.loads(open('/tmp/pb.json').read().strip().splitlines()[-1])
print({k:v for k,v in d.items() if 'build' in k or k in ('rounds','total_ms','loop_ms','checks','first','last')})"
python profiles/summarize.py $(find /tmp/pb -name "*.db" | head -1) | grep "smg::build\|smg::overlap_lean\|^kernel\|exclusive_scan\|scan" | cut -c1-150
