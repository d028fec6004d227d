# counters of the overlap pass alone at C5 (GPU box): bash tools/prof_overlap2.sh <tag> [SMG_OVERLAP_WIDE value]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-x}; export SMG_OVERLAP_WIDE=${2:-lean}
HDR="$(python profiles/pmcfile.py header)"
run() { ( cd /tmp && rm -rf /tmp/po_$TAG/$1 && rocprofv3 --kernel-trace --pmc $2 -d /tmp/po_$TAG/$1 -o p -- python $GRAFT_REPO_ROOT/tools/bench_overlap.py --reps 3 > /dev/null 2> /tmp/po_$1.log ) || tail -3 /tmp/po_$1.log; }
run SQ1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
run SQ2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_INSTS_SMEM"
run FETCH "FETCH_SIZE"
run TCC "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run TCP "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
{ echo "$HDR"; echo "# SMG_OVERLAP_WIDE=$SMG_OVERLAP_WIDE"; for d in SQ1 SQ2 FETCH TCC TCP; do python profiles/summarize.py $(find /tmp/po_$TAG/$d -name "*.db" | head -1); done; } | grep "sources\|SMG_OVERLAP\|overlap_\|stream_lookup\|^kernel\|counter" > gpurun_out/pmc_overlap_$TAG.txt
rocprofv3 -L 2>/dev/null | grep -o "TC[CP]_[A-Z0-9_a-z]*" | sort -u | tr '\n' ' ' > gpurun_out/counters_tc.txt
cat gpurun_out/pmc_overlap_$TAG.txt | cut -c1-150
