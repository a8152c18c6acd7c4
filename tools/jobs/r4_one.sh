ulimit -c 0
export TMPDIR=/tmp
timeout 900 python -m pytest "$@" -x -q 2>&1 | tail -40
