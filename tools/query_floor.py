#!/usr/bin/env python3
"""Instruction-count lower bound of the 27-cell query at BASELINE.json configs[1] next to what the compiled kernel issues.

  1. the chunk histogram of the workload (CPU, numpy): for every occupied cell of the engine's grid the number of query points
     and of candidates in its 27 neighbour cells -> NC = ceil(candidates / 64) register chunks; weights = query points
  2. the static instruction mix of the query loop of k_query_pool_fast for every NC (gfx950 ISA of the build as it is)
  3. the MINIMAL mix a wave64 formulation of "test every candidate of the 27 cells bit-exactly, compact the hits" needs per chunk
     and per query (listed below), evaluated with the same histogram
  4. both turned into time with the issue costs measured on MI355X (tools/ubench: plain VALU 2.6 cycles, packed fp32 4.7, anything
     that touches an SGPR / VCC / lane select 4.3, SALU 4.3, LDS / VMEM issue 4), for 1024 SIMDs at 2.4 GHz, assuming PERFECT overlap
     of the scalar unit with the vector pipe (lower bound) and NO overlap (upper bound of the issue model)

usage: python tools/query_floor.py [--points 10000000] [--pmc profiles/<file>.json]   (no GPU needed)"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from treensearch_amd import datagen as D   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=10_000_000)
ap.add_argument("--pmc", default=None, help="pmc.json of tools/prof_gpu.sh: measured instruction counts of the fast kernel")
args = ap.parse_args()
n = args.points

# ---------------------------------------------------------------- 1. chunk histogram
pts = D.uniform_cloud(n, 12345)
r = float(D.radius_for_neighbors(n))
lo = pts.min(axis=0) - np.float32(2 * r)
h = r * (1.0 + 1e-5)
ijk = np.floor((pts - lo) / np.float32(h)).astype(np.int64)
dims = ijk.max(axis=0) + 3
key = (ijk[:, 2] * dims[1] + ijk[:, 1]) * dims[0] + ijk[:, 0]
cnt = np.bincount(key, minlength=int(dims.prod())).reshape(dims[2], dims[1], dims[0]).astype(np.int64)
cand = np.zeros_like(cnt)
P = np.pad(cnt, 1)
for dz in range(3):
    for dy in range(3):
        for dx in range(3):
            cand += P[dz:dz + cnt.shape[0], dy:dy + cnt.shape[1], dx:dx + cnt.shape[2]]
occ = cnt > 0
nq = cnt[occ]
nc = (cand[occ] + 63) // 64
n_cells = int(occ.sum())
hist = {}
for k in range(1, int(nc.max()) + 1):
    sel = nc == k
    if sel.any():
        hist[k] = (int(sel.sum()), int(nq[sel].sum()))
chunks_per_query = float((nc * nq).sum()) / n
print(f"workload: {n} uniform points, r = {r:.6f}, grid {tuple(int(d) - 2 for d in dims)}, {n_cells} occupied cells, "
      f"{n / n_cells:.2f} query points and {float(cand[occ].mean()):.1f} candidates per cell")
print(f"chunks of 64 candidates per query (weighted by query points): {chunks_per_query:.3f}   (candidates / 64 = {float((cand[occ] * nq).sum()) / n / 64:.3f})")
print("  NC   cells      queries    share")
for k, (c, q) in hist.items():
    print(f"  {k:2d} {c:8d} {q:11d}   {q / n * 100:5.1f} %")

# ---------------------------------------------------------------- 2. static mix of the compiled query loops
tmp = tempfile.mkdtemp()
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-x", "hip", "-I" + ROOT + "/include",
                "-S", "--cuda-device-only", ROOT + "/treensearch_amd/csrc/tnsx_query.hip", "-o", tmp + "/q.s"], check=True, capture_output=True)
lines = open(tmp + "/q.s").read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN4tnsx17k_query_pool_fastILi0ELb0ELb0ELb1ELb0EEEvNS_9QueryArgsE:"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]


def is_instr(l):
    return l.startswith("\t") and not l.startswith("\t.") and not l.strip().startswith(";")


# LLVM annotates every basic block of a loop with "in Loop: Header=BBx_y Depth=d"; the query loop is the depth-2 loop (cells are
# depth 1) that holds the NC staged writes (ds_write_b32).  Blocks of its child loop (records > 64 entries) are annotated with the
# child's header and are left out, as is the out-of-line slab change (a call).
blocks, cur = [], None
for l in body:
    mlab = re.match(r"^(\.LBB\d+_\d+):(.*)$", l)
    if mlab:
        cur = {"label": mlab.group(1), "note": mlab.group(2), "ins": []}
        blocks.append(cur)
    elif l.startswith("; %bb.") and cur is not None:
        cur = {"label": l.split()[1], "note": l, "ins": []}
        blocks.append(cur)
    elif cur is not None:
        if "Loop Header" in l or "Parent Loop" in l or "Child Loop" in l:
            cur["note"] += " " + l
        elif is_instr(l):
            cur["ins"].append(l.strip())
loops = {}   # NC -> instructions of the query loop (the first loop of each NC is the un-culled path)
headers = [b["label"].replace(".L", "") for b in blocks if "Loop Header: Depth=2" in b["note"]]
for hname in headers:
    ins = []
    for b in blocks:
        own = b["label"].replace(".L", "") == hname
        if own or re.search(r"Header=" + re.escape(hname) + r"\b", b["note"]):
            # the rare blocks stay out: the slab change (a call: s_getpc / s_swappc) and the flush of the queries before it
            if any(("s_getpc" in l) or ("s_swappc" in l) or l.startswith("global_store_dwordx2") for l in b["ins"]):
                continue
            ins += b["ins"]
    ncw = sum(1 for l in ins if l.startswith("ds_write_b32"))
    if ncw and any(l.startswith("v_pk_") for l in ins) and ncw not in loops:
        loops[ncw] = ins


def mix(ins):
    v = [l for l in ins if l.startswith("v_")]
    return {"valu": len(v), "pk": sum(1 for l in v if l.startswith("v_pk_")), "sgpr_valu": sum(1 for l in v if re.match(r"v_(cmp|readlane|writelane|mbcnt|readfirstlane|addc)", l)),
            "salu": sum(1 for l in ins if l.startswith("s_") and not re.match(r"s_(c?branch|waitcnt|nop|getpc|swappc|setpc)", l)),
            "branch": sum(1 for l in ins if re.match(r"s_c?branch", l)), "nop": sum(1 for l in ins if l.startswith("s_nop")),
            "lds": sum(1 for l in ins if l.startswith("ds_")), "vmem": sum(1 for l in ins if l.startswith("buffer_") or l.startswith("global_")),
            "wait": sum(1 for l in ins if l.startswith("s_waitcnt"))}


print("\nstatic instruction mix of the compiled query loop, per query (hot path: without the slab-change and flush blocks and without the loop for records > 64 entries):")
print("  NC  VALU (pk, sgpr-touching)  SALU  branch  nop  LDS  VMEM  waitcnt")
for k in sorted(loops):
    m = mix(loops[k])
    print(f"  {k:2d}  {m['valu']:4d} ({m['pk']:2d}, {m['sgpr_valu']:2d})          {m['salu']:4d}  {m['branch']:4d}  {m['nop']:3d}  {m['lds']:3d}  {m['vmem']:3d}  {m['wait']:3d}")

# ---------------------------------------------------------------- 3. minimal mix
# per chunk of 64 candidates and one query:
#   tests        3 sub + 3 mul + 2 add, packed two chunks per instruction (4 v_pk per chunk), every op individually rounded (strict mode:
#                no fma may replace a mul + add); 1 v_cmp that writes the 64-lane hit mask
#   compaction   rank of every hit lane among the hits: 2 v_mbcnt (lo, hi) -- a wave64 prefix count has no cheaper form;
#                1 exec move (scalar) + 1 write of the hit lanes (LDS here, global before) + 1 v_add of the lane's write address
# per query:     3 v_readlane (x, y, z of the query), 1 v_readlane (record length), 1 address set-up (v_lshl_add), 1 LDS read-back,
#                1 coalesced store of the record, 2 v_writelane (count, position: written for the whole cell at once),
#                scalar: self bit clear, length arithmetic (3), slab check + branch (2), position / left updates (2), loop (3), exec restore (1)
MIN_CHUNK = {"pk": 4, "sgpr_valu": 3, "plain_valu": 1, "salu": 1, "lds": 1}
MIN_QUERY = {"sgpr_valu": 6, "plain_valu": 1, "salu": 12, "lds": 1, "vmem": 1}
COST = {"pk": 4.7, "sgpr_valu": 4.3, "plain_valu": 2.6, "salu": 4.3, "lds": 4.0, "vmem": 4.0}
SIMDS, HZ = 1024, 2.4e9


def time_ms(per_query):
    """(perfect overlap of the scalar unit with the vector / memory pipes, no overlap) in ms for n queries"""
    vec = sum(per_query.get(k, 0.0) * COST[k] for k in ("pk", "sgpr_valu", "plain_valu", "lds", "vmem"))
    sca = per_query.get("salu", 0.0) * COST["salu"]
    return n * max(vec, sca) / SIMDS / HZ * 1e3, n * (vec + sca) / SIMDS / HZ * 1e3


floor = {k: MIN_CHUNK.get(k, 0) * chunks_per_query + MIN_QUERY.get(k, 0) for k in COST}
lo_ms, hi_ms = time_ms(floor)
print(f"\nminimal mix per query at {chunks_per_query:.2f} chunks: " + ", ".join(f"{k} {v:.1f}" for k, v in floor.items())
      + f"  = {sum(floor.values()):.1f} instructions")
print(f"  -> issue time {lo_ms:.3f} ms (scalar unit perfectly overlapped) ... {hi_ms:.3f} ms (no overlap)")

# what the compiled loops issue, weighted with the histogram (loop body only; the per-cell part -- lookups, candidate loads, flush -- comes on top)
tot = {k: 0.0 for k in COST}
for k, (c, q) in hist.items():
    m = mix(loops.get(min(k, max(loops)), loops[max(loops)]))
    tot["pk"] += m["pk"] * q
    tot["sgpr_valu"] += m["sgpr_valu"] * q
    tot["plain_valu"] += (m["valu"] - m["pk"] - m["sgpr_valu"]) * q
    tot["salu"] += (m["salu"] + m["branch"] + m["nop"] + m["wait"]) * q
    tot["lds"] += m["lds"] * q
    tot["vmem"] += m["vmem"] * q
comp = {k: v / n for k, v in tot.items()}
c_lo, c_hi = time_ms(comp)
print(f"compiled loops per query (hot path, weighted with the histogram; the per-cell part comes on top): " + ", ".join(f"{k} {v:.1f}" for k, v in comp.items())
      + f"  = {sum(comp.values()):.1f} instructions")
print(f"  -> issue time {c_lo:.3f} ms ... {c_hi:.3f} ms")

if args.pmc and os.path.exists(args.pmc):
    prof = json.load(open(args.pmc))
    kn = [k for k in prof["kernels"] if k.startswith("k_query_pool_fast<0, false, false, true, false>")]
    if kn:
        p = prof["kernels"][kn[0]]["pmc"]
        tr = prof["kernels"][kn[0]].get("trace") or {}
        print(f"\nmeasured ({args.pmc}): per query VALU {p.get('SQ_INSTS_VALU', 0) / n:.1f}, SALU {p.get('SQ_INSTS_SALU', 0) / n:.1f}, "
              f"VMEM wr {p.get('SQ_INSTS_VMEM_WR', 0) / n:.2f}, rd {p.get('SQ_INSTS_VMEM_RD', 0) / n:.2f}, LDS {p.get('SQ_INSTS_LDS', 0) / n:.2f}; "
              f"kernel {tr.get('avg_us', 0) / 1e3:.3f} ms")

bytes_q = 16 * n + 4 * (59.24 * n + n) + 8 * n
print(f"\nfor comparison: algorithmic bytes of the query {bytes_q / 1e9:.2f} GB -> {bytes_q / 8e12 * 1e3:.3f} ms at 8 TB/s; "
      f"0.30 of the HBM roofline = {bytes_q / 8e12 / 0.30 * 1e3:.3f} ms, 0.50 = {bytes_q / 8e12 / 0.5 * 1e3:.3f} ms")
