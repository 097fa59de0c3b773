#!/bin/bash
# TA / TCP counters of a stand-alone tool binary: tools/prof_ta.sh <tag> <binary> [args]; summary -> gpurun_out/<tag>/ta_counters.json
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
BIN=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
# few counters a pass (a set the hardware cannot collect together aborts rocprofv3, which then hangs in its signal handler:
# every pass runs under its own timeout)
for C in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES"; do
    i=$((i+1))
    timeout -k 5 90 rocprofv3 --pmc $C -d /tmp/ta_${TAG}_$i -o r -- $BIN "$@" > $OUT/ta_$i.log 2>&1 || echo "pass $i failed" >> $OUT/ta_fail.log
done
python3 $R/tools/rocpd_pmc.py /tmp/ta_${TAG}_1/r_results.db /tmp/ta_${TAG}_2/r_results.db /tmp/ta_${TAG}_3/r_results.db /tmp/ta_${TAG}_4/r_results.db --match k_ --json $OUT/ta_counters.json > /dev/null
rm -rf /tmp/ta_${TAG}_*
cat $OUT/ta_counters.json
