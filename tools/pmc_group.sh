export TMPDIR=/tmp
OUT=gpurun_out/gp_pmc; mkdir -p $OUT
CMD="python tools/group_probe.py 2 10000000 5"   # (group_probe.py loads ab_libs/libtnsx_group.so: tools/build_group_variant.sh)
pass() { local name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "k_query_groups" --output-format csv -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1 < /dev/null; }
pass A SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
pass B SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES
pass C SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_UNALIGNED_STALL
python tools/prof_summary.py $OUT $OUT/pmc.json 2>&1 | head -60
