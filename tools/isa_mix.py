#!/usr/bin/env python
"""Instruction mix of the hottest basic block (most MFMAs) of every kernel matching a name in a hipcc -S listing.
usage: hipcc ... -S --cuda-device-only x.hip -o x.s ; python tools/isa_mix.py x.s <name substring> ..."""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read()
labels = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):", txt, re.M)]
for want in sys.argv[2:]:
    for i, (pos, name) in enumerate(labels):
        if want not in name:
            continue
        end = txt.find("s_endpgm", pos)
        body = txt[pos:end]
        blocks = re.split(r"\n\.LBB\d+_\d+:", body)
        best = max(blocks, key=lambda b: b.count("v_mfma"))
        c = Counter()
        for line in best.split("\n"):
            t = line.strip()
            if not t or t[0] in ";./" or t.endswith(":"):
                continue
            op = t.split()[0]
            if op.startswith("v_mfma"): c["mfma"] += 1
            elif op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")): c["transcendental"] += 1
            elif op.startswith("v_"): c["valu"] += 1
            elif op.startswith("ds_"): c["lds"] += 1
            elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
            elif op.startswith("s_"): c["salu"] += 1
            elif op.startswith(("global_", "buffer_", "scratch_", "flat_")): c["vmem"] += 1
            else: c["other"] += 1
        print(f"{name[:70]}: hottest block {sum(c.values())} instructions: {dict(c)}")
