set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r4a
./tools/ubench/atomic_scatter > gpurun_out/r4a/atomic_scatter.txt 2>&1
cat gpurun_out/r4a/atomic_scatter.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4a/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/r4a/bench_traced.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline.py gpurun_out/r4a/kt > gpurun_out/r4a/timeline.txt 2>&1
cat gpurun_out/r4a/timeline.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-secondary > gpurun_out/r4a/bench_plain.log 2>&1
tail -c 3000 gpurun_out/r4a/bench_plain.log
