"""CPU tests of the parts of bench.py that turn profiler output into the fields of the bench line, and of the on-device point generator the bench uses."""
import csv
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HEADER = ('"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name","Workgroup_Size","LDS_Block_Size",'
          '"Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"')
FAST = "void tnsx::k_query_pool_fast<0, false, false, true, false>(tnsx::QueryArgs, tnsx::PoolArgs)"
FAT = "void tnsx::k_query_pool_fast<0, false, false, true, true>(tnsx::QueryArgs, tnsx::PoolArgs)"
GENERAL = "void tnsx::k_query<2, false, false, false>(tnsx::QueryArgs, tnsx::PoolArgs)"
BEGIN = "tnsx::k_run_begin(tnsx::RunBeginArgs)"


def _canned(dispatches, counter="WRITE_SIZE"):
    """rocprofv3 counter_collection.csv text for [(kernel name, value), ...] in dispatch order (the format of ROCm 7.2's rocprofv3 --output-format csv)"""
    lines = [HEADER]
    for k, (name, val) in enumerate(dispatches, start=1):
        lines.append(f'{k},{k},"Agent 2",1,453,453,625152,62,"{name}",256,4096,0,84,0,96,"{counter}",{val:.6f},{1000 * k},{1000 * k + 500}')
        lines.append(f'{k},{k},"Agent 2",1,453,453,625152,62,"{name}",256,4096,0,84,0,96,"GRBM_GUI_ACTIVE",12345.000000,{1000 * k},{1000 * k + 500}')
    return "\n".join(lines) + "\n"


def _rows(text):
    return list(csv.DictReader(io.StringIO(text)))


def test_steady_state_counters_drop_the_dry_pass_and_sum_the_tiers():
    import bench
    # c2-shaped: the cold step = dry pass (writes nothing) + sized pass with the two heavy tiers, then warm-up and timed steps of ONE launch each
    c2 = [(BEGIN, 3.0), (FAST, 12.0), (FAST, 2_600_000.0), (FAT, 0.5), (GENERAL, 0.25),
          (BEGIN, 3.0), (FAST, 2_650_000.0), (BEGIN, 3.0), (FAST, 2_660_000.0), (BEGIN, 3.0), (FAST, 2_658_000.0), (BEGIN, 3.0), (FAST, 2_662_000.0)]
    got, det = bench.steady_state_counters(_rows(_canned(c2)), ["WRITE_SIZE"], 3)
    assert abs(got["WRITE_SIZE"] - (2_660_000.0 + 2_658_000.0 + 2_662_000.0) / 3) < 1e-6
    assert det["query_dispatches_per_step"] == 1.0 and list(det["per_kernel_per_step"]) == ["k_query_pool_fast<0, false, false, true, false>"]
    # what round 5 computed (mean over every dispatch of the kernel name): diluted by the dry pass
    naive = np.mean([v for n, v in c2 if n == FAST])
    assert naive < 0.9 * got["WRITE_SIZE"]
    # c4-shaped: three tiers per step -- the step's traffic is their SUM, not the largest per-name mean
    c4 = [(BEGIN, 1.0), (FAST, 5.0), (FAT, 1.0), (GENERAL, 1.0), (FAST, 5_000_000.0), (FAT, 4_000_000.0), (GENERAL, 900_000.0)]
    for _ in range(3):
        c4 += [(BEGIN, 1.0), (FAST, 5_100_000.0), (FAT, 4_050_000.0), (GENERAL, 910_000.0)]
    got, det = bench.steady_state_counters(_rows(_canned(c4)), ["WRITE_SIZE"], 3)
    assert abs(got["WRITE_SIZE"] - (5_100_000.0 + 4_050_000.0 + 910_000.0)) < 1e-6 and det["query_dispatches_per_step"] == 3.0
    assert set(det["per_kernel_per_step"]) == {"k_query_pool_fast<0, false, false, true, false>", "k_query_pool_fast<0, false, false, true, true>", "k_query<2, false, false, false>"}
    # c3-shaped: two pairs per step (two launches of the same kernel)
    c3 = [(BEGIN, 1.0), (FAST, 1.0), (FAST, 1.0)] + [(BEGIN, 1.0), (FAST, 2_000_000.0), (FAST, 300_000.0)] * 4
    got, det = bench.steady_state_counters(_rows(_canned(c3)), ["WRITE_SIZE"], 3)
    assert abs(got["WRITE_SIZE"] - 2_300_000.0) < 1e-6 and det["query_dispatches_per_step"] == 2.0
    # other counters in the file are ignored; too few steps say so instead of returning a number
    got, det = bench.steady_state_counters(_rows(_canned(c2[:5])), ["WRITE_SIZE"], 3)
    assert got == {} and "k_run_begin" in det["note"]


def test_device_generator_equals_the_numpy_generator():
    import torch
    from treensearch_amd import datagen as D
    for n, seed, start in [(1000, 12345, 0), (100003, 7, 123456789), (5, 12346, 2 ** 31 + 5), (0, 1, 0)]:
        a = D.uniform_cloud(n, seed, start)
        b = D.uniform_cloud_torch(n, seed, start, device="cpu", chunk=4097)
        assert b.dtype == torch.float32 and tuple(b.shape) == (n, 3) and np.array_equal(a, b.numpy())
