mkdir -p gpurun_out/r2
pj() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', d['ms_per_step'], d['stage_ms'])"; }
python bench.py --workload c2 --no-pmc --no-cpu-baseline 2>/dev/null | pj c2_random
python bench.py --workload c2 --no-pmc --no-cpu-baseline --zsort-input 2>/dev/null | pj c2_zsorted
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 TNSX_BENCH_FORCE_SLAB=1
python bench.py --workload c5 --points 25000000 --no-pmc 2>/dev/null | pj c5_25M_random
python bench.py --workload c5 --points 25000000 --no-pmc --zsort-input 2>/dev/null | pj c5_25M_zsorted
