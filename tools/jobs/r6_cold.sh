set -x
ulimit -c 0
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6cold}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "nan or torch_csr or zsort or neighborlist" > $O/pytest.log 2>&1
tail -8 $O/pytest.log
timeout 300 python tools/cold_probe.py 10000000 3 > $O/cold_c2.txt 2>&1
cat $O/cold_c2.txt
timeout 300 python tools/cold_probe.py 1000000 3 > $O/cold_1m.txt 2>&1
cat $O/cold_1m.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d $O/trace -o t -- python $GRAFT_REPO_ROOT/tools/cold_probe.py 10000000 2 > $O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
ls $O/trace/*/ 2>/dev/null | head
python - <<PY
import csv, glob
for f in glob.glob("$O/trace/**/*hip_api_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:14]:
        print(f'{r["Name"]:40s} calls {r["Calls"]:>6s} total ms {float(r["TotalDurationNs"]) / 1e6:10.3f}')
PY
