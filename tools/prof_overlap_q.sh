# kernel times and instruction counters of the overlap pass for a given number of query hashes per range (GPU box):
#   bash tools/prof_overlap_q.sh <SMG_OVERLAP_QPR> ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for Q in "$@"; do
  export SMG_OVERLAP_QPR=$Q
  ( cd /tmp && rm -rf /tmp/pq_$Q && timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d /tmp/pq_$Q -o p -- python $GRAFT_REPO_ROOT/tools/bench_overlap.py --reps 2 > /dev/null 2> /tmp/pq_$Q.log ) || tail -3 /tmp/pq_$Q.log
  echo "# SMG_OVERLAP_QPR=$Q"
  python profiles/summarize.py $(find /tmp/pq_$Q -name "*.db" | head -1) | grep "overlap_\|stream_\|^kernel\|build_range" | cut -c1-160
done
