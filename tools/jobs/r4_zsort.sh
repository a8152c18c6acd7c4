ulimit -c 0
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4h
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_fullsize.py -x -q -k "zsort or c4 or configs3 or dam" 2>&1 | tail -5
timeout 600 python bench.py --workload c4 --points 10000000 --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > $O/bench_c4_10m_zs.json 2>/dev/null
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4h/bench_c4_10m_zs.json") if l.startswith("{")][-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["stage_ms"])
PY
