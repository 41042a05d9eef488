#!/usr/bin/env python3
"""VGPRs and scratch bytes of every kernel in csrc/vr_backend.s (make -C volume_renderer_amd/csrc asm)."""
import re
import subprocess
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "volume_renderer_amd/csrc/vr_backend.s"
want = sys.argv[2] if len(sys.argv) > 2 else ""
s = open(path).read()
rows = []
for name, body in re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
    if want not in name:
        continue
    v = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
    sc = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
    rows.append((name, v, sc))
dem = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
for (n, v, sc), d in zip(rows, dem):
    d = d.replace("void vr::", "").replace("(vr::TrackParams)", "")
    print(f"{v:4d} vgpr {sc:5d} scratch  {d}")
