#!/usr/bin/env python
"""Register / spill table of every kernel of a .hip source (hipcc -Rpass-analysis=kernel-resource-usage), for profiles/.
usage: python tools/kernel_resources.py clipself_amd/csrc/gemm_stream.hip [name filter] > profiles/rNN_resources.md"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

src = Path(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as tmp:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-munsafe-fp-atomics",
                        "-c", src.name, "-o", f"{tmp}/o.o", "-Rpass-analysis=kernel-resource-usage"], cwd=src.parent, capture_output=True, text=True)
blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
names = [b.split(" [")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
print(f"# {src.name}: kernel resource usage (hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage)\n")
print("| kernel | VGPRs | AGPRs | SGPRs | VGPR spill | SGPR spill | scratch B/lane | waves/SIMD |")
print("|---|---:|---:|---:|---:|---:|---:|---:|")
for b, n in zip(blocks, dem):
    n = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0].replace("void ", "")
    if flt and flt not in n:
        continue
    g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
    vals = [g(k) for k in ("VGPRs", "AGPRs", "TotalSGPRs", "VGPRs Spill", "SGPRs Spill", r"ScratchSize \[bytes/lane\]", r"Occupancy \[waves/SIMD\]")]
    print(f"| `{n}` | " + " | ".join(str(v) for v in vals) + " |")
