ulimit -c 0
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4z}
mkdir -p $O

timeout 300 python tools/zsort_probe.py 10000000 8 2>&1 | tail -1 | tee $O/zsort_10m.txt

cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $GRAFT_REPO_ROOT/tools/zsort_probe.py 10000000 8 > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/prof_summary.py $O 2>/dev/null | grep -v "at::native\|k_query" | head -24 | tee $O/kt_summary.txt
