# round 6, verdict item 1: where does the query's fabric traffic come from, and does a ticket order with a small working set per XCD shorten the kernel?
# A/B of ab_libs/libtnsx_base.so against ab_libs/libtnsx_yz.so (-DTNSX_YZ_TICKETS=1) in one process + counter passes of each.
set -x
ulimit -c 0
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6yz}
mkdir -p $O
LIBS="ab_libs/libtnsx_base.so ab_libs/libtnsx_yz.so ${EXTRA_LIBS:-}"
timeout 900 python tools/ab_libs.py $LIBS --check --zsort --move --rounds 6 --steps 15 > $O/ab_rounds.txt 2>&1
cat $O/ab_rounds.txt
timeout 900 python tools/ab_libs.py $LIBS --zsort --move --recreate 6 --steps 15 > $O/ab_recreate.txt 2>&1
cat $O/ab_recreate.txt
cd /tmp
for L in $LIBS; do
  N=$(basename $L .so)
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_READ_sum TCC_WRITE_sum"; do
    T=$(echo $C | tr ' ' '_' | cut -c1-40)
    timeout 300 rocprofv3 --pmc $C --kernel-include-regex "k_query_pool_fast" --output-format csv -d $O/pmc_${N}/$T -o pmc -- python $GRAFT_REPO_ROOT/tools/c2_loop.py $GRAFT_REPO_ROOT/$L 8 > $O/pmc_${N}_$T.log 2>&1 < /dev/null
  done
  python $GRAFT_REPO_ROOT/tools/pmc_variant.py $O/pmc_${N} > $O/pmc_${N}.txt 2>&1
  echo "== $N"; cat $O/pmc_${N}.txt
  rm -rf $O/pmc_${N}
done
cd $GRAFT_REPO_ROOT
timeout 900 python tools/ab_libs.py $LIBS --check --workload c4 --points 10000000 --rounds 5 --steps 10 > $O/ab_c4_rounds.txt 2>&1
cat $O/ab_c4_rounds.txt
timeout 600 python bench.py --workload c4 --points 10000000 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c4_10m.json 2> $O/bench_c4_10m.err
python -c "
import json
d=json.loads(open('$O/bench_c4_10m.json').read().strip().splitlines()[-1]); rf=d['roofline']
print('c4 10M: ms/step', d['ms_per_step'], 'fill', d['stage_ms']['fill'], 'traffic', rf['traffic'], 'algorithmic', rf['bytes_per_launch']); print(json.dumps(rf['traffic_detail'], indent=1))"
