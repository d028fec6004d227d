cd $GRAFT_REPO_ROOT
bash tools/prof_pmc.sh g1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" python $GRAFT_REPO_ROOT/tools/bench_gather.py
bash tools/prof_pmc.sh g2 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAVES" python $GRAFT_REPO_ROOT/tools/bench_gather.py
grep "build_range\|stream_lookup\|build_partition\|build_scatter\|apply" gpurun_out/pmc_g1.txt gpurun_out/pmc_g2.txt | cut -c1-200
