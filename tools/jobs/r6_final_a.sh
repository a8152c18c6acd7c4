set -x
ulimit -c 0
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6fa}
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1
tail -14 $O/pytest.log
timeout 700 python tools/fuzz_gpu.py --minutes 8 --seed 6 > $O/fuzz.txt 2>&1
tail -5 $O/fuzz.txt
SECONDS=0; timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench.py wall seconds: $SECONDS"
python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); rf=d['roofline']
print(d['value'], d['ms_per_step'], rf['frac'], rf['avg_launch_ms'], rf['traffic'], rf['whole_run']['frac'], d['cold_run'])
for k,v in d['secondary'].items(): print(k, v.get('ms_per_step'), v.get('query_ms'), v.get('query_frac'), v.get('traffic_over_algorithmic'))"
