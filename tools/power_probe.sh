#!/bin/bash
# Board power and graphics clock WHILE the bench's steps run (rocm-smi samples once a second beside a long run of bench.py):
# what the step draws against the board's cap.    usage (GPU box, repo root): bash tools/power_probe.sh [bench args] > out.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
rocm-smi --showmaxpower 2>&1 | grep -i "max graphics"
python3 $R/bench.py --no-cpu-baseline --steps 2500 --warmup 5 --event-every 100 "$@" > /tmp/power_probe_bench.json 2>/dev/null &
BP=$!
for i in $(seq 1 40); do
    S=$(rocm-smi --showpower --showclocks 2>&1)
    P=$(echo "$S" | grep -i "package power" | sed 's/.*: //')
    C=$(echo "$S" | grep "sclk" | sed 's/.*(\(.*\))/\1/')
    echo "t=${i}s power_W=$P sclk=$C"
    kill -0 $BP 2>/dev/null || break
    sleep 1
done
wait $BP
python3 -c "
import json; d = json.loads(open('/tmp/power_probe_bench.json').read().strip().splitlines()[-1])
print('bench: %.2f steps/s, %.3f ms/step over %d steps' % (d['value'], d['ms_per_step'], d['steps']))"
