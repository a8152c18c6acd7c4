#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + PMC passes (each in its own run) of bench.py.
# usage: tools/prof_gpu.sh <label> [kernel-regex] [extra bench args]
set -u
LABEL=${1:-run}; REGEX=${2:-"k_query|k_cs_|k_cell|k_table|k_set_checksum|k_sort_records|k_bucket|k_occ|k_run_begin"}
if [ $# -ge 2 ]; then shift 2; else shift $#; fi
OUT=gpurun_out/prof_$LABEL
mkdir -p $OUT
export TMPDIR=/tmp
export TNSX_BENCH_INNER=1   # the bench run under the profiler is the timed loop only: no copy-ceiling measurement, no random-order variant behind it
BENCH="python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-pmc --no-secondary --no-stage-pass $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --no-stage-pass $* > $OUT/kt.log 2>&1 < /dev/null
pass() {  # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "$REGEX" --output-format csv -d $OUT/pmc_$name -o pmc -- $BENCH > $OUT/pmc_$name.log 2>&1 < /dev/null
}
pass A SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
pass B SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD
pass C FETCH_SIZE GRBM_GUI_ACTIVE
pass D WRITE_SIZE GRBM_GUI_ACTIVE
pass E TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_INSTS_BRANCH SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES
find $OUT -name "*.csv" | head -40
python tools/prof_summary.py $OUT $OUT/pmc.json > $OUT/summary.txt 2>&1 < /dev/null
cat $OUT/summary.txt
