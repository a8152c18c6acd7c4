set -x
ulimit -c 0
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6cold2}
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_reuse.py tests/test_gpu_api.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1
tail -8 $O/pytest.log
{ echo "# tools/cold_probe.py: engine 0 also pays the one-off costs of the process (code objects, first uses of runtime paths), engines 1.. only their own"; 
  timeout 300 python tools/cold_probe.py 10000000 3; timeout 300 python tools/cold_probe.py 1000000 3; } 2>&1 | grep -v amdgpu.ids > $O/cold.txt
cat $O/cold.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-pmc > $O/bench_c2_short.json 2> $O/bench.err
python -c "
import json
d=json.loads(open('$O/bench_c2_short.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['cold_run'])"
