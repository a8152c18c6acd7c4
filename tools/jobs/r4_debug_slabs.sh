ulimit -c 0
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4d
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_slabs.py -x -q --durations=8 > $O/slabs.log 2>&1
grep -v "^  File\|^Thread\|^$" $O/slabs.log | tail -c 3500
