# Round-4 profiles (GPU box).  $1 selects a part (sketch | gather | compare | all).  Text summaries go to gpurun_out/;
# the rocpd databases stay in /tmp.  Counters are collected in their own runs, one --pmc group per pass, with
# --kernel-trace only (MI355X_MICROARCH.md: TCC slots; no hip / hsa tracing next to --pmc).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
S=$GRAFT_REPO_ROOT/profiles/summarize.py
PART=${1:-all}
HDR="$(python $GRAFT_REPO_ROOT/profiles/pmcfile.py header)"   # source hashes: bench.py refuses counters of changed kernels
pass() {   # pass <dir tag> <output name> <counters...> -- <command...>
  local tag=$1; shift
  local ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  ( cd /tmp && rocprofv3 --kernel-trace --pmc "${ctr[@]}" -d /tmp/p4/$tag -o p -- "$@" > /dev/null 2> /tmp/p4_$tag.log ) || tail -3 /tmp/p4_$tag.log
}
db() { find /tmp/p4/$1 -name "*.db" | head -1; }
if [ "$PART" = sketch ] || [ "$PART" = all ]; then
  B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-compare"
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p4/stats -o p -- python $GRAFT_REPO_ROOT/bench.py > $OUT/r04_bench_under_profiler.json 2> /tmp/p4_stats.log ) || tail -3 /tmp/p4_stats.log
  python $S $(db stats) > $OUT/r04_kernel_stats.txt
  pass FETCH FETCH_SIZE -- $B
  pass WRITE WRITE_SIZE -- $B
  pass VALU1 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- $B            # VALU-active cycles on their own
  pass VALU2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- $B
  pass VALU3 SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -- $B
  { echo "$HDR"; python $S $(db FETCH) $(db WRITE) $(db VALU1) $(db VALU2) $(db VALU3); } > $OUT/r04_pmc.txt
  grep "sketch_dna_kernel" $OUT/r04_pmc.txt | cut -c1-170
fi
if [ "$PART" = gather ] || [ "$PART" = all ]; then
  G="python $GRAFT_REPO_ROOT/tools/bench_gather.py"
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p4/gstats -o p -- $G > $OUT/r04_gather_under_profiler.json 2> /tmp/p4_gstats.log ) || tail -3 /tmp/p4_gstats.log
  python $S $(db gstats) > $OUT/r04_gather_kernels.txt
  pass GFETCH FETCH_SIZE -- $G
  pass GWRITE WRITE_SIZE -- $G
  pass GSQ SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -- $G
  { echo "$HDR"; python $S $(db GFETCH) $(db GWRITE) $(db GSQ); } > $OUT/r04_gather_pmc.txt
  grep -i "build_\|overlap\|scatter" $OUT/r04_gather_pmc.txt | grep -i "SIZE" | cut -c1-170
fi
if [ "$PART" = compare ] || [ "$PART" = all ]; then
  C="python $GRAFT_REPO_ROOT/tools/bench_compare.py c4"
  pass CMP SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- $C
  ( cd /tmp && SMG_COMPARE_KERNEL=walk rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/p4/CMPW -o p -- $C > /dev/null 2> /tmp/p4_CMPW.log ) || tail -3 /tmp/p4_CMPW.log
  { echo "$HDR"; python $S $(db CMP) $(db CMPW); } > $OUT/r04_compare_pmc.txt
  grep "compare_hash\|compare_tile" $OUT/r04_compare_pmc.txt | cut -c1-170
fi
