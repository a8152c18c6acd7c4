ulimit -c 0
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4h
mkdir -p $O
timeout 1500 python bench.py --workload c4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c4_50m.json 2> $O/bench_c4.err
tail -c 1800 $O/bench_c4_50m.json
timeout 1500 python bench.py --workload c5 --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > $O/bench_c5_200m_1gpu.json 2> $O/bench_c5.err
tail -c 1500 $O/bench_c5_200m_1gpu.json; tail -3 $O/bench_c5.err
