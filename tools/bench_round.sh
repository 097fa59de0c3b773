#!/bin/bash
# The round's bench lines on the GPU box (run through gpurun from the repo root): headline (default run) and the secondary
# workloads, one JSON each under gpurun_out/<tag>/; then the B = 100 step's kernel trace (tools/minibatch_probe.py).
# usage: tools/bench_round.sh <tag>
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 600 python3 bench.py > $OUT/headline_bench.json 2> $OUT/headline.err || exit 1
for w in "cfg2 irt4pl_1d_bbvi_100kx100" "cfg4 irt2pl_1d_bbvi_missing90_1Mx500" "cfg4ai irt2pl_1d_amortized_missing90_1Mx500" "cfg5 hodina_1Mx30x8" "dense1d irt2pl_1d_bbvi_dense_1Mx500"; do
    set -- $w
    timeout -k 10 300 python3 bench.py --workload $2 --steps 200 --warmup 5 > $OUT/$1_bench.json 2> $OUT/$1.err || exit 1
done
for n in 125 250 500; do
    timeout -k 10 300 python3 bench.py --persons ${n}000 --steps 200 --warmup 5 --no-cpu-baseline > $OUT/shard_${n}k_bench.json 2> $OUT/shard.err || exit 1
done
timeout -k 10 120 python3 tools/minibatch_probe.py --steps 1000 > $OUT/minibatch_probe.txt 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/mb_$TAG -o r -- python3 $R/tools/minibatch_probe.py --modes graph4 --steps 1000 > $OUT/mb_trace.log 2>&1 || exit 1
DB=$(find /tmp/mb_$TAG -name "*.db" | tail -1)
python3 $R/tools/rocpd_stats.py $DB $OUT/minibatch_kernel_stats.csv > /dev/null
python3 $R/tools/timeline.py $DB 20 > $OUT/minibatch_timeline.txt
rm -rf /tmp/mb_$TAG
