# usage: bash tools/jobs/r4_suite_bench.sh <label> [pytest-args...]   -- GPU suite + default bench line + kernel timeline, into gpurun_out/<label>/
set -x
ulimit -c 0
export TMPDIR=/tmp
L=${1:-r4}; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$L
mkdir -p $O
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest ${@:-tests} -m gpu -x -q > $O/pytest.log 2>&1
  tail -15 $O/pytest.log
fi
timeout 900 python bench.py ${BENCH_ARGS:-} > $O/bench.json 2> $O/bench.err
tail -c 2500 $O/bench.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-stage-pass > $O/bench_traced.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline.py $O/kt > $O/timeline.txt 2>&1
cat $O/timeline.txt
