#!/usr/bin/env python3
"""Static instruction mix of the innermost loops of one kernel (gfx950 ISA from hipcc -save-temps).
usage: tools/loop_stats.py <mangled-kernel-name-substring> [depth]"""
import re, subprocess, sys, os, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sub = sys.argv[1]; depth = sys.argv[2] if len(sys.argv) > 2 else "3"
tmp = tempfile.mkdtemp()
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-x", "hip", "-I" + root + "/include", "-c",
                root + "/treensearch_amd/csrc/tnsx_query.hip", "-o", tmp + "/q.o", "--cuda-device-only", "-save-temps=obj"], cwd=tmp, capture_output=True)
lines = open(tmp + "/tnsx_query-hip-amdgcn-amd-amdhsa-gfx950.s").read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(sub) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
lines = lines[start:end + 1]
print(lines[0][:90], "lines", len(lines))
hdrs = [i for i, l in enumerate(lines) if "This Inner Loop Header: Depth=" + depth in l]
for h in hdrs:
    j = h
    while not lines[j].startswith(".LBB"): j -= 1
    label = lines[j].split(":")[0]
    e = None
    for k in range(h, min(h + 1200, len(lines))):
        if re.search(r"s_c?branch\w* " + re.escape(label) + r"\b", lines[k]): e = k
    if e is None: continue
    body = [l.strip() for l in lines[j:e + 1] if l.startswith("\t") and not l.strip().startswith(";")]
    cnt = lambda f: sum(1 for l in body if f(l))
    print(f"{label:10s} n {len(body):4d} VALU {cnt(lambda l: l.startswith('v_')):3d} (pk {cnt(lambda l: l.startswith('v_pk_')):2d}) "
          f"SALU {cnt(lambda l: l.startswith('s_') and not re.match(r's_(c?branch|waitcnt|nop)', l)):3d} branch {cnt(lambda l: re.match(r's_c?branch', l) is not None):2d} "
          f"vmem {cnt(lambda l: l.startswith('global_') or l.startswith('buffer_')):2d} nop {cnt(lambda l: l.startswith('s_nop')):2d} wait {cnt(lambda l: l.startswith('s_waitcnt')):2d}")
