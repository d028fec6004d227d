# counters of the overlap pass at C5 (GPU box): bash tools/prof_overlap.sh <tag>; summary to gpurun_out/pmc_overlap_<tag>.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-x}
HDR="$(python profiles/pmcfile.py header)"
run() { ( cd /tmp && rocprofv3 --kernel-trace --pmc $2 -d /tmp/po_$TAG/$1 -o p -- python $GRAFT_REPO_ROOT/tools/bench_gather.py > /dev/null 2> /tmp/po_$1.log ) || tail -3 /tmp/po_$1.log; }
run SQ1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
run SQ2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_BUSY_CYCLES"
run FETCH "FETCH_SIZE"
{ echo "$HDR"; python profiles/summarize.py $(find /tmp/po_$TAG/SQ1 -name "*.db" | head -1) $(find /tmp/po_$TAG/SQ2 -name "*.db" | head -1) $(find /tmp/po_$TAG/FETCH -name "*.db" | head -1); } | grep "sources\|overlap_wide\|stream_lookup\|^kernel\|counter" > gpurun_out/pmc_overlap_$TAG.txt
