cd $GRAFT_REPO_ROOT
bash tools/prof_r02.sh all > gpurun_out/run26_prof.txt 2>&1
bash tools/prof_pmc.sh g1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" python $GRAFT_REPO_ROOT/tools/bench_gather.py
bash tools/prof_pmc.sh g2 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAVES" python $GRAFT_REPO_ROOT/tools/bench_gather.py
cat gpurun_out/pmc_g1.txt gpurun_out/pmc_g2.txt > gpurun_out/r02_gather_sq.txt
python tools/bench_gather.py > gpurun_out/r02_gather_c5.json 2>/dev/null
SMG_BENCH_FORCE_COLLECTIVES=1 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_forced_collectives.json 2>/dev/null
tail -30 gpurun_out/run26_prof.txt | cut -c1-170
