# HBM traffic and wait counters of the gather kernels at C5 (GPU box): separate --pmc passes, summaries to gpurun_out/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d /tmp/pg/$C -o g -- python $GRAFT_REPO_ROOT/tools/bench_gather.py > /dev/null 2> /tmp/pg_$C.log || tail -3 /tmp/pg_$C.log
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE -d /tmp/pg/SQ -o g -- python $GRAFT_REPO_ROOT/tools/bench_gather.py > /dev/null 2> /tmp/pg_SQ.log || tail -3 /tmp/pg_SQ.log
python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/pg/FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pg/WRITE_SIZE -name "*.db" | head -1) $(find /tmp/pg/SQ -name "*.db" | head -1) > $OUT/r01_gather_pmc.txt
grep "smg::" $OUT/r01_gather_pmc.txt | grep -i "FETCH_SIZE\|WRITE_SIZE" | cut -c1-130
