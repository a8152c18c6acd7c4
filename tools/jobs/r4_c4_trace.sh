ulimit -c 0
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4h
mkdir -p $O
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c4 -o kt -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --points 10000000 --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-stage-pass > $O/c4_traced.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/kt_c4/.. 2>/dev/null | head -5
python - <<'PY'
import csv,glob
for f in glob.glob("gpurun_out/r4h/kt_c4/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:22]:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>4} avg_us {float(r['AverageNs'])/1e3:9.1f} total% {r['Percentage']}")
PY
