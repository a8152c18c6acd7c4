set -x
ulimit -c 0
export TMPDIR=/tmp
for L in ab_libs/libtnsx_nocull.so ab_libs/libtnsx_rc5.so treensearch_amd/lib/libtnsx.so; do
  timeout 120 python tools/c2_loop.py $L 6 2>&1 | grep -v amdgpu | tail -3
  timeout 120 python tools/c2_loop.py $L 6 1000000 2>&1 | grep -v amdgpu | tail -3
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small_cases" 2>&1 | tail -15
