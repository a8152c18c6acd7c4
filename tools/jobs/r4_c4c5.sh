# usage: bash tools/jobs/r4_c4c5.sh <label>   -- full-size tests of configs[3] / configs[4] + their bench lines, into gpurun_out/<label>/
ulimit -c 0
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4q}
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_reuse.py -x -q -m gpu 2>&1 | tail -4
timeout 1500 python bench.py --workload c4 --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > $O/bench_c4_50m.json 2> $O/bench_c4.err
python - <<PY
import json
for f in ("bench_c4_50m.json",):
    d=json.loads([l for l in open("$O/"+f) if l.startswith("{")][-1])
    print(f, {k:d[k] for k in ("value","ms_per_step")}, d["stage_ms"], d.get("steady_state"))
PY
timeout 1500 python bench.py --workload c5 --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > $O/bench_c5_200m_1gpu.json 2> $O/bench_c5.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_c5_200m_1gpu.json") if l.startswith("{")][-1])
print("c5", {k:d[k] for k in ("value","ms_per_step")}, d["stage_ms"], d.get("steady_state"))
PY
tail -3 $O/bench_c5.err
