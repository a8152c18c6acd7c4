#!/usr/bin/env python3
"""How long does the torch-side halo selection + packing of treensearch_amd.multi take on one GPU (no communication)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from treensearch_amd import datagen as D
from treensearch_amd.multi import slab_halo_masks

n = 10_000_000
pts = torch.from_numpy(D.uniform_cloud(n, 12345)).cuda()
gids = torch.arange(n, dtype=torch.int64, device="cuda")
halo = 0.0113

def pack(mask, cols=4):
    sel = torch.nonzero(mask, as_tuple=False).squeeze(1)
    out = torch.empty((sel.numel(), cols + 1), dtype=torch.float32, device="cuda")
    out[:, 0:3] = pts.index_select(0, sel)
    out[:, cols - 1:cols + 1] = gids.index_select(0, sel).view(-1, 1).view(torch.float32).view(-1, 2)
    return out

def step():
    ml, mr = slab_halo_masks(pts[:, 0], 0.0, 1.0, halo, True, True)
    a, b = pack(ml), pack(mr)
    return a.shape[0] + b.shape[0]

for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): m = step()
torch.cuda.synchronize()
print(f"torch halo select+pack: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per step, {m} halo points")

import treensearch_amd as T
from treensearch_amd.multi import SlabExchange
ex = SlabExchange(0.0, 1.0, halo, packer=T.TreeNSearch(stream=torch.cuda.current_stream().cuda_stream))
for _ in range(3): ex._pack_device(pts, gids, None, True, True, 4)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): a, b = ex._pack_device(pts, gids, None, True, True, 4)
torch.cuda.synchronize()
print(f"tnsx_halo_pack kernel : {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per step, {a.shape[0] + b.shape[0] - 2} halo points")
