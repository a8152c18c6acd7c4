# usage: r4_ab.sh <outfile> <ab_libs args...>
ulimit -c 0
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4f
mkdir -p $O
out=$1; shift
timeout 1200 python tools/ab_libs.py "$@" > $O/$out 2>&1
cat $O/$out
