#!/usr/bin/env python3
"""Prints VGPR/SGPR/scratch/LDS/occupancy per kernel of one source file of treensearch_amd/csrc (gfx950).
env: TNSX_SRC = file name (default tnsx_kernels.hip), TNSX_EXTRA_FLAGS = extra compiler flags (as for treensearch_amd.build)."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math"] + os.environ.get("TNSX_EXTRA_FLAGS", "").split() + ["-x", "hip", "-I" + root + "/include",
       "-c", root + "/treensearch_amd/csrc/" + (os.environ.get("TNSX_SRC", "tnsx_kernels.hip")), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark: .*?:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name)
    if flt and flt not in name: continue
    print(f"{name[:70]:70s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>3} SGPR {r.get('TotalSGPRs', r.get('SGPRs','?')):>4} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?'):>2} LDS {r.get('LDS Size [bytes/block]','?'):>6}")
