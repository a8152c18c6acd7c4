set -x
ulimit -c 0
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "register_cull or small_cases" 2>&1 | tail -3
bash tools/prof_gpu.sh r6
