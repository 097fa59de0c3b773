# usage: kt.sh <kernel-substring> [bench.py args...] : average kernel times from a short bench under rocprofv3
K=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pk -o r -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/pk/r_results.db | grep "$K" | awk -F, '{print $1, "avg ns", $(NF-3)}' | cut -c1-90
