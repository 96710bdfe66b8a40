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
    # phases by marker kernels (inline schedule: CLIPSelf.__call__ runs the student's forward, then the teacher, then the loss): the step's two
    # im2row launches open the student's and the teacher's forward; the backward starts at the first backward-only kernel
    stems = [i for i, r in enumerate(step) if "im2row" in r[0]]
    bwd = next((i for i, r in enumerate(step) if "cosine_bwd" in r[0] or "l2norm_bwd" in r[0]), len(step))
    if len(stems) >= 2 and stems[1] < bwd:
        phases = (("student forward", 0, stems[1]), ("teacher + loss", stems[1], bwd), ("student backward + AdamW", bwd, len(step)))
    else:                                   # overlapped schedule / unknown order: forward part and backward part only
        phases = (("forward (both towers) + loss", 0, bwd), ("student backward + AdamW", bwd, len(step)))
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
