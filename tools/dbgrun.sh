for d in 0 16 28; do
  cd /tmp && export TMPDIR=/tmp
  VX_DBG=$d rocprofv3 --kernel-trace --stats -d /tmp/pd$d -o r -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
  python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/pd$d/r_results.db | grep "$1" | awk -F, -v d=$d '{print "VX_DBG=" d, "avg ns", $(NF-3)}'
done
