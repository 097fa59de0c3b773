#!/bin/bash
# L1-TLB (UTCL1) requests / misses and L2 read latency per kernel of one bench step: bash tools/tlb_pmc.sh <out.json> [bench args]
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum -d /tmp/tlb1 -o r -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" > /dev/null 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum -d /tmp/tlb2 -o r -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" > /dev/null 2>&1
python3 $R/tools/rocpd_pmc.py /tmp/tlb1/r_results.db /tmp/tlb2/r_results.db --match k_ --json $OUT > /dev/null
rm -rf /tmp/tlb1 /tmp/tlb2
