#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc counters from rocpd sqlite results (one counter set per run, as the MI355X guide
prescribes).  usage: tools/rocprof_pmc.py <db> [<db> ...]   -> markdown table: kernel | launches | avg of each counter"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"^void ", "", name).split("(")[0][:70]


def main():
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in sys.argv[1:]:
        cur = sqlite3.connect(path).cursor()
        for kname, cname, value in cur.execute("select kernel_name, counter_name, value from counters_collection"):
            a = acc[short(kname)][cname]
            a[0] += value
            a[1] += 1
    counters = sorted({c for k in acc.values() for c in k})
    print("| kernel | launches | " + " | ".join(f"avg {c}" for c in counters) + " |")
    print("|---|---:|" + "---:|" * len(counters))
    rows = sorted(acc.items(), key=lambda kv: -max(v[0] for v in kv[1].values()))
    for k, cs in rows[:14]:
        n = max(v[1] for v in cs.values())
        print(f"| `{k}` | {n} | " + " | ".join(f"{cs[c][0] / cs[c][1]:.0f}" if c in cs and cs[c][1] else "-" for c in counters) + " |")


if __name__ == "__main__":
    main()
