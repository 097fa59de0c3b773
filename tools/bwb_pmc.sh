#!/bin/bash
# Memory-side counters of the head weight gradient's harness at several batch sizes (which level the tile transfers hit, what a
# request waits): usage (GPU box, repo root): bash tools/bwb_pmc.sh <binary> <out.json prefix> N...
BIN=$1; OUT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for N in "$@"; do
    i=0
    for C in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum" \
             "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum" \
             "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
        i=$((i+1))
        rocprofv3 --pmc $C -d /tmp/bp_${N}_$i -o r -- $R/$BIN $N > /dev/null 2>&1
    done
    python3 $R/tools/rocpd_pmc.py /tmp/bp_${N}_1/r_results.db /tmp/bp_${N}_2/r_results.db /tmp/bp_${N}_3/r_results.db /tmp/bp_${N}_4/r_results.db --match k_mvn --json ${OUT}_$N.json > /dev/null
    rm -rf /tmp/bp_${N}_*
done
