# usage: kt2.sh <n_kernels> <python script + args> : timeline of the last kernels of a run under rocprofv3
N=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/pk2 -o r -- python3 "$@" > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/timeline.py /tmp/pk2/r_results.db $N
