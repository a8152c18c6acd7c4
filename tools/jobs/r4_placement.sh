ulimit -c 0
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4f
mkdir -p $O
timeout 300 python tools/placement_probe.py --engines 5 --steps 6 --lib ab_libs/libtnsx_dbgpool.so > $O/placement_dbg.txt 2>&1
grep "engine\|region 0:\|region 8:" $O/placement_dbg.txt | tail -120
