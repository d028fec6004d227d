# Round-6 profiles (GPU box).  $1 selects a part (sketch | gather | ext | all).  Text summaries go to gpurun_out/r06_prof/;
# the rocpd databases stay in /tmp.  Counters are collected in their own runs, one --pmc group per pass, with
# --kernel-trace only (MI355X_MICROARCH.md: TCC slots; no hip / hsa tracing next to --pmc).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_prof; mkdir -p $OUT
S=$GRAFT_REPO_ROOT/profiles/summarize.py
PART=${1:-all}
HDR="$(python $GRAFT_REPO_ROOT/profiles/pmcfile.py header)"   # source hashes: bench.py refuses counters of changed kernels
pass() {   # pass <dir tag> <counters...> -- <command...>
  local tag=$1; shift
  local ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  ( cd /tmp && rocprofv3 --kernel-trace --pmc "${ctr[@]}" -d /tmp/p6/$tag -o p -- "$@" > /dev/null 2> /tmp/p6_$tag.log ) || tail -3 /tmp/p6_$tag.log
}
db() { find /tmp/p6/$1 -name "*.db" | head -1; }
if [ "$PART" = sketch ] || [ "$PART" = all ]; then
  B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-compare"
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p6/stats -o p -- python $GRAFT_REPO_ROOT/bench.py > $OUT/r06_bench_under_profiler.json 2> /tmp/p6_stats.log ) || tail -3 /tmp/p6_stats.log
  python $S $(db stats) > $OUT/r06_kernel_stats.txt
  pass FETCH FETCH_SIZE -- $B
  pass WRITE WRITE_SIZE -- $B
  pass VALU1 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- $B
  pass VALU2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- $B
  { echo "$HDR"; python $S $(db FETCH) $(db WRITE) $(db VALU1) $(db VALU2); } > $OUT/r06_pmc.txt
  grep "sketch_dna_kernel" $OUT/r06_pmc.txt | cut -c1-170
fi
if [ "$PART" = gather ] || [ "$PART" = all ]; then
  G="python $GRAFT_REPO_ROOT/tools/bench_gather.py"
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p6/gstats -o p -- $G > $OUT/r06_gather_under_profiler.json 2> /tmp/p6_gstats.log ) || tail -3 /tmp/p6_gstats.log
  python $S $(db gstats) > $OUT/r06_gather_kernels.txt
  pass GFETCH FETCH_SIZE -- $G
  pass GWRITE WRITE_SIZE -- $G
  { echo "$HDR"; python $S $(db GFETCH) $(db GWRITE); } > $OUT/r06_gather_pmc.txt
  grep -i "build_\|overlap\|scatter" $OUT/r06_gather_pmc.txt | grep -i "SIZE" | cut -c1-170
fi
if [ "$PART" = ext ] || [ "$PART" = all ]; then
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p6/xstats -o p -- python $GRAFT_REPO_ROOT/tools/bench_compare_ext.py > /dev/null 2> /tmp/p6_xstats.log ) || tail -3 /tmp/p6_xstats.log
  python $S $(db xstats) > $OUT/r06_compare_ext_kernels.txt
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p6/pstats -o p -- python $GRAFT_REPO_ROOT/tools/bench_protein.py > $OUT/r06_protein.json 2> /tmp/p6_pstats.log ) || tail -3 /tmp/p6_pstats.log
  python $S $(db pstats) > $OUT/r06_protein_kernels.txt
fi
