#!/bin/bash
# Runs ON THE GPU BOX: timing + counter passes of tools/ubench/fine_query (the stand-alone fine-cell / LDS-tile query formulation of round 5).
# usage: tools/pmc_fine.sh <label> [points]
export TMPDIR=/tmp
LABEL=${1:-fine}; N=${2:-10000000}
OUT=gpurun_out/$LABEL; mkdir -p $OUT
BIN=tools/ubench/fine_query
$BIN $N SAB 10 > $OUT/timing.txt 2>&1
CMD="$BIN $N AB 3"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $CMD > $OUT/kt.log 2>&1 < /dev/null
pass() { local name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "k_fine_query" --output-format csv -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1 < /dev/null; }
pass A SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
pass B SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH
pass C SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES
pass D FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
cat $OUT/timing.txt
python tools/prof_summary.py $OUT $OUT/pmc.json 2>&1 | head -80
