#!/usr/bin/env python
"""Turn a rocprofv3 `--kernel-trace --stats` result (rocpd sqlite .db or *_kernel_stats.csv) into the per-kernel
summary table committed under profiles/.   usage: tools/rocprof_summary.py <db-or-csv> [title] > profiles/<name>.md"""
import csv
import re
import sqlite3
import sys


def _short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:80]


def rows_from_db(path):
    cur = sqlite3.connect(path).cursor()
    return [(n, c, t, a, p) for n, c, t, a, p in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels")]


def rows_from_csv(path):
    out = []
    for r in csv.DictReader(open(path)):
        out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return out


def main():
    path = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else path
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    total = sum(r[2] for r in rows)
    print(f"# {title}\n")
    print(f"Total kernel time {total / 1e3:.2f} ms over {sum(r[1] for r in rows)} dispatches (durations in microseconds).\n")
    print("| kernel | calls | total us | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for n, c, t, a, p in rows:
        if p < 0.05:
            continue
        print(f"| `{_short(n)}` | {c} | {t:.1f} | {a:.1f} | {p:.1f} |")


if __name__ == "__main__":
    main()
