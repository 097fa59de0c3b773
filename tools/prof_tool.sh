#!/bin/bash
# PMC profile of a stand-alone tool binary: tools/prof_tool.sh <tag> <binary> [args]; summaries -> gpurun_out/<tag>/
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
BIN=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 5 120 rocprofv3 --pmc $C -d /tmp/pmct_${TAG}_$i -o r -- $BIN "$@" > $OUT/pmc_$i.log 2>&1
done
timeout -k 5 120 rocprofv3 --kernel-trace --stats -d /tmp/pmct_${TAG}_k -o r -- $BIN "$@" > $OUT/trace.log 2>&1
python3 $R/tools/rocpd_stats.py /tmp/pmct_${TAG}_k/r_results.db $OUT/kernel_stats.csv > /dev/null
python3 $R/tools/rocpd_pmc.py /tmp/pmct_${TAG}_1/r_results.db /tmp/pmct_${TAG}_2/r_results.db /tmp/pmct_${TAG}_3/r_results.db /tmp/pmct_${TAG}_4/r_results.db --match k_ --json $OUT/pmc_counters.json > /dev/null
rm -rf /tmp/pmct_${TAG}_*
cat $OUT/pmc_counters.json; cat $OUT/kernel_stats.csv
