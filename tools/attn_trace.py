#!/usr/bin/env python
"""Timeline of the attention forward's workgroups on their CUs (ablation build, env CS_ATTN_TRACE=<file> while running tools/attn_bench.py).
Per workgroup: HW_ID, XCC_ID and the 100 MHz clock at entry / tables in LDS / images built / attended / end.   usage: python tools/attn_trace.py <file>"""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
hw, xcc = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64) & 0xF
cu = (xcc << 8) | ((hw >> 8) & 0xFF)                     # cu_id[11:8], sh_id[12], se_id[15:13] within the XCC
ts = t[:, 2:7].astype(np.int64)
ts = (ts - ts[:, 0].min()) / 100.0                       # us
print(f"{len(t)} workgroups on {len(np.unique(cu))} CUs, launch span {ts[:, 4].max():.1f} us")
names = ["entry->tables", "tables->images", "images->attended", "attended->end", "residence"]
d = np.concatenate([np.diff(ts, axis=1), (ts[:, 4] - ts[:, 0])[:, None]], axis=1)
for i, n in enumerate(names):
    print(f"  {n:18s} mean {d[:, i].mean():6.2f}  p10 {np.percentile(d[:, i], 10):6.2f}  p50 {np.percentile(d[:, i], 50):6.2f}  p90 {np.percentile(d[:, i], 90):6.2f} us")
# per CU: how many residents at a time, and how much of the launch has 0 / 1 / 2 workgroups in the attend phase resp. the load phase
res = {"attend": np.zeros(4), "load": np.zeros(4), "resident": np.zeros(4)}
for c in np.unique(cu):
    w = ts[cu == c]
    for key, (a, b) in {"attend": (2, 3), "load": (0, 2), "resident": (0, 4)}.items():
        ev = sorted([(x, 1) for x in w[:, a]] + [(x, -1) for x in w[:, b]])
        n, last = 0, 0.0
        for x, s in ev:
            res[key][min(n, 3)] += x - last
            n += s; last = x
for key, v in res.items():
    v = v / v.sum()
    print(f"  share of CU time with 0/1/2/3+ workgroups in '{key}': " + " ".join(f"{x:.2f}" for x in v))
c0 = np.unique(cu)[0]
w = ts[cu == c0]
w = w[np.argsort(w[:, 0])]
print(f"first 12 workgroups of CU {c0:#x} (entry, tables, images, attended, end; us):")
for r in w[:12]:
    print("   " + "  ".join(f"{x:7.2f}" for x in r))
