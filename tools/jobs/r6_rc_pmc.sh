set -x
ulimit -c 0
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6rcpmc}
mkdir -p $O
cd /tmp
for L in ${LIBS:-ab_libs/libtnsx_nocull.so ab_libs/libtnsx_rc4nosym.so}; do
  N=$(basename $L .so)
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
    T=$(echo $C | tr ' ' '_' | cut -c1-40)
    timeout 300 rocprofv3 --pmc $C --kernel-include-regex "k_query_pool_fast" --output-format csv -d $O/pmc_${N}/$T -o pmc -- python $GRAFT_REPO_ROOT/tools/c2_loop.py $GRAFT_REPO_ROOT/$L 8 > $O/pmc_${N}_$T.log 2>&1 < /dev/null
  done
  python $GRAFT_REPO_ROOT/tools/pmc_variant.py $O/pmc_${N} > $O/pmc_${N}.txt 2>&1
  echo "== $N"; cat $O/pmc_${N}.txt
  rm -rf $O/pmc_${N}
done
