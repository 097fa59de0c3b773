#!/bin/bash
# Collect the per-round profile artefacts on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats, HBM traffic counters (separate passes), SQ counters; only the small summaries are kept.
# usage: tools/profile_round.sh <tag> [bench.py args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o r -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" > $OUT/trace.log 2>&1
python3 $R/tools/rocpd_stats.py /tmp/prof_$TAG/r_results.db $OUT/kernel_stats.csv > /dev/null
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $C -d /tmp/pmc_${TAG}_$i -o r -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" > $OUT/pmc_$i.log 2>&1
done
python3 $R/tools/rocpd_pmc.py /tmp/pmc_${TAG}_1/r_results.db /tmp/pmc_${TAG}_2/r_results.db /tmp/pmc_${TAG}_3/r_results.db /tmp/pmc_${TAG}_4/r_results.db /tmp/pmc_${TAG}_5/r_results.db --match k_ --json $OUT/pmc_counters.json > /dev/null
# HBM bytes per launch (MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE count KB; on gfx950 FETCH_SIZE reports
# half of the bytes of wide streaming reads, so it is doubled)
python3 - "$OUT/pmc_counters.json" "$OUT/hbm_traffic.json" <<'PY'
import json, sys
c = json.load(open(sys.argv[1]))
import os
head = None
try:        # the box has no .git: the caller leaves `git rev-parse --short HEAD` (+ "-dirty") in .git_head before gpurun
    head = open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), ".git_head")).read().strip() or None
except OSError:
    pass
out = {"note": "per launch; hbm_read_bytes_corrected = FETCH_SIZE KB x 1024 x 2 (gfx950), hbm_write_bytes = WRITE_SIZE KB x 1024",
       "git_head": head, "kernels": {}}
for k, v in c.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        out["kernels"][k] = {"FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"],
                             "hbm_read_bytes_corrected": v["FETCH_SIZE"] * 1024 * 2,
                             "hbm_write_bytes": v["WRITE_SIZE"] * 1024}
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
PY
rm -rf /tmp/prof_$TAG /tmp/pmc_${TAG}_*
# the graphics clock each large kernel ran at (docs/HARDWARE.md rule 41: the step runs at the board's power cap)
CLK_JSON=$OUT/clock.json bash $R/tools/clk_probe.sh k_ python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 2 "$@" > $OUT/clock.txt 2>&1
