# End-of-round profiles (GPU box): kernel stats of the default bench, PMC passes of the sketch kernel, counters of the
# merge kernel at C4.  Text summaries go to gpurun_out/ (the rocpd databases stay in /tmp).
set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
S=$GRAFT_REPO_ROOT/profiles/summarize.py
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py > $OUT/prof_bench_stdout.txt 2> /tmp/prof_stats.log || tail -3 /tmp/prof_stats.log
python $S $(find /tmp/prof/stats -name "*.db" | head -1) > $OUT/r01_end_kernel_stats.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d /tmp/prof/$C -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-compare > /dev/null 2> /tmp/prof_$C.log || tail -3 /tmp/prof_$C.log
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY -d /tmp/prof/SQ -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-compare > /dev/null 2> /tmp/prof_SQ.log || tail -3 /tmp/prof_SQ.log
python $S $(find /tmp/prof/FETCH_SIZE -name "*.db" | head -1) $(find /tmp/prof/WRITE_SIZE -name "*.db" | head -1) $(find /tmp/prof/SQ -name "*.db" | head -1) > $OUT/r01_end_pmc.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/prof/CMP -o cmp -- python $GRAFT_REPO_ROOT/tools/bench_compare.py c4 > $OUT/prof_cmp_stdout.txt 2> /tmp/prof_CMP.log || tail -3 /tmp/prof_CMP.log
python $S $(find /tmp/prof/CMP -name "*.db" | head -1) > $OUT/r01_end_compare_pmc.txt
tail -1 $OUT/prof_bench_stdout.txt | cut -c1-300
head -8 $OUT/r01_end_kernel_stats.txt
grep -A12 "counter" $OUT/r01_end_compare_pmc.txt | grep compare_tile
