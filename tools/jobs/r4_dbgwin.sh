ulimit -c 0
export TMPDIR=/tmp
python - <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import treensearch_amd.api as A
A._lib = None; A.LIB_PATH = os.path.abspath("ab_libs/libtnsx_dbgwin.so")
n = 300000
rng = np.random.default_rng(11)
z = rng.random(n, dtype=np.float32) ** np.float32(3.0)
pts = np.stack([rng.random(n, dtype=np.float32), rng.random(n, dtype=np.float32), z], axis=1).astype(np.float32)
pts[0] = (0, 0, 0); pts[1] = (1, 1, 1)
r = np.float32(0.02)
d = torch.from_numpy(pts).cuda()
ns = A.TreeNSearch(); ns.set_search_radius(r); ns.add_point_set(d); ns.set_active_search(0, 0, True)
for k in range(4):
    if k:
        pts[2:] += (rng.random((n - 2, 3), dtype=np.float32) - np.float32(0.5)) * np.float32(0.1) * r
        np.clip(pts, 0.0, 1.0, out=pts); d.copy_(torch.from_numpy(pts))
    if k < 2:
        st0 = None
        g_o, g_h = np.float32(-0.04), np.float32(0.02000056393444538)
        inv = np.float32(1.0) / g_h
        ijk = np.clip(((pts - g_o) * inv).astype(np.int32), 0, 53)
        key = (ijk[:, 2] * 54 + ijk[:, 1]) * 54 + ijk[:, 0]
        print("host bucket counts 0..23:", np.bincount(key >> 9, minlength=512)[:24], flush=True)
    ns.run(); st = ns.get_stats()
    print(k, {x: st[x] for x in ("one_read_builds", "speculated", "speculation_redos")}, flush=True)
PY
