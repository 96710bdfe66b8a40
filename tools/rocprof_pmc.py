#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc counters from rocpd sqlite results (one counter set per run, as the MI355X guide
prescribes).  usage: tools/rocprof_pmc.py <db> [<db> ...]   -> markdown table: kernel | launches | avg of each counter
       tools/rocprof_pmc.py --dominant <kernel substring> --chunk <crops per launch> --out profiles/pmc_traffic.json <db> [<db> ...]
           additionally writes the FETCH_SIZE / WRITE_SIZE averages (KB) of the largest matching kernel as the tracked summary bench.py
           reads its `roofline.traffic` from"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"^void ", "", name).split("(")[0][:70]


def main():
    argv, opts = sys.argv[1:], {}
    while argv and argv[0].startswith("--"):
        opts[argv[0][2:]] = argv[1]
        argv = argv[2:]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in argv:
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
    if "dominant" in opts:
        import json
        cand = [(k, cs) for k, cs in acc.items() if opts["dominant"] in k and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs]
        k, cs = max(cand, key=lambda kc: kc[1]["FETCH_SIZE"][0] / kc[1]["FETCH_SIZE"][1])
        rec = {"kernel": k, "chunk_crops": int(opts.get("chunk", 2048)), "launches": cs["FETCH_SIZE"][1],
               "fetch_kb": cs["FETCH_SIZE"][0] / cs["FETCH_SIZE"][1], "write_kb": cs["WRITE_SIZE"][0] / cs["WRITE_SIZE"][1],
               "source": opts.get("source", "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), per-launch averages; FETCH_SIZE x 2 on gfx950")}
        with open(opts["out"], "w") as f:
            json.dump({"dominant": rec}, f, indent=1)
            f.write("\n")


if __name__ == "__main__":
    main()
