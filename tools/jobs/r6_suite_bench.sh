# usage: bash tools/jobs/r6_suite_bench.sh <label> [pytest-args...]   -- GPU suite + default bench line (+ the opt-in group-formulation tests when WITH_GROUP=1), into gpurun_out/<label>/
set -x
ulimit -c 0
export TMPDIR=/tmp
L=${1:-r6}; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$L
mkdir -p $O
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1800 python -m pytest ${@:-tests} -m gpu -x -q --durations=15 > $O/pytest.log 2>&1
  tail -25 $O/pytest.log
fi
if [ "${WITH_GROUP:-0}" = "1" ]; then
  timeout 900 python -m pytest tools/test_group_formulation.py -m gpu -x -q > $O/pytest_group.log 2>&1
  tail -5 $O/pytest_group.log
fi
if [ "${SKIP_BENCH:-0}" != "1" ]; then
  SECONDS=0; timeout 1500 python bench.py ${BENCH_ARGS:-} > $O/bench.json 2> $O/bench.err
  tail -c 6000 $O/bench.json
  echo "bench.py wall seconds: $SECONDS"
fi
