set -x
ulimit -c 0
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4d
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_slabs.py -x -q -s -k "two_sets_asymmetric" > $O/two_sets.log 2>&1
tail -c 2500 $O/two_sets.log
AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 timeout 600 python -m pytest tests/test_gpu_slabs.py -x -q -s -k "two_sets_asymmetric" > $O/two_sets_serial.log 2>&1
grep -v "^  File\|^Thread\|^$" $O/two_sets_serial.log | tail -c 2000
