set -x
ulimit -c 0
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6fuzz2}
mkdir -p $O
timeout 400 python tools/fuzz_slabs.py --minutes 4 > $O/fuzz_slabs.txt 2>&1; tail -3 $O/fuzz_slabs.txt
timeout 700 python tools/fuzz_gpu.py --minutes 4 --devices 3 --seed 9 > $O/fuzz_multi.txt 2>&1; tail -3 $O/fuzz_multi.txt
timeout 300 python tools/dropin_rate.py > $O/dropin.txt 2>&1; tail -3 $O/dropin.txt
