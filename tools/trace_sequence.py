#!/usr/bin/env python
"""Ordered kernel sequence of the LAST full optimizer step in a rocprofv3 --kernel-trace result (rocpd sqlite), with start time, duration and
the idle gap in front of every kernel -- the view that shows launch gaps, stray copies / fills and what sits between the GEMMs.
usage: tools/trace_sequence.py <results.db> [out.txt]     (a step ends with adamw_kernel)"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:90]


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    rows = list(cur.execute("select name, start, end, grid_x, workgroup_x, stream_id from kernels order by start"))
    ends = [i for i, r in enumerate(rows) if "adamw" in r[0]]
    if len(ends) < 2:
        raise SystemExit("need at least two optimizer steps in the trace")
    step = rows[ends[-2] + 1:ends[-1] + 1]
    t0 = step[0][1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    wall, busy = (step[-1][2] - t0) / 1e6, sum(r[2] - r[1] for r in step) / 1e6
    out.write(f"# last full step: {len(step)} kernels, wall {wall:.2f} ms, kernel time {busy:.2f} ms\n")
    # phases by marker kernels: teacher = up to the last attn_cls launch; backward starts at the first backward-only kernel
    cls = max((i for i, r in enumerate(step) if "attn_cls" in r[0]), default=-1)
    bwd = next((i for i, r in enumerate(step) if "cosine_bwd" in r[0] or "l2norm_bwd" in r[0]), len(step))
    teacher_end = next((i for i in range(cls + 1, len(step)) if "im2row" in step[i][0]), cls + 1) if cls >= 0 else 0
    phases = (("teacher", 0, teacher_end), ("student forward + loss", teacher_end, bwd), ("student backward + AdamW", bwd, len(step)))
    for name, a, b in phases:
        if b <= a:
            continue
        seg = step[a:b]
        w = (seg[-1][2] - seg[0][1]) / 1e6
        k = sum(r[2] - r[1] for r in seg) / 1e6
        out.write(f"# {name}: kernels {a}..{b - 1} ({b - a} launches), wall {w:.2f} ms, kernel time {k:.2f} ms, idle {w - k:.2f} ms\n")
        acc = defaultdict(lambda: [0, 0.0])
        for r in seg:
            acc[short(r[0])][0] += 1
            acc[short(r[0])][1] += (r[2] - r[1]) / 1e3
        for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:28]:
            out.write(f"#     {t:9.1f} us  {c:4d} x {t / c:8.1f}  {n}\n")
    prev = None
    for i, r in enumerate(step):
        gap = (r[1] - prev) / 1e3 if prev is not None else 0.0
        out.write(f"{i:4d} {(r[1] - t0) / 1e3:10.1f} us  dur {(r[2] - r[1]) / 1e3:8.1f}  gap {gap:7.1f}  grid {r[3]:>8} x {r[4]:<4} s{r[5]}  {short(r[0])}\n")
        prev = max(prev, r[2]) if prev is not None else r[2]


if __name__ == "__main__":
    main()
