#!/usr/bin/env python
"""Experiment: the frozen teacher's pass over 2048 crops as TWO half passes on two HIP streams, each with its persistent GEMMs on half the
compute units (reserve = 128), against the usual single pass on the whole chip.  Idea (profiles/r04_u_epilogue_per_cu.md): the K loops are
limited by the power budget's clock and the epilogues / attention by each CU's vector-memory pipe, and nothing overlaps the two kinds of
phase inside one stream -- two de-phased streams on disjoint CUs could.   usage (GPU box): python tools/teacher_two_streams.py [reserve=128] [iters=4]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.open_clip import create_model  # noqa: E402

reserve = int(sys.argv[1]) if len(sys.argv) > 1 else 128
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = "cuda:0"
teacher = create_model("EVA02-CLIP-B-16", "eva", precision="amp_bf16", device=dev, cache_dir=None, trainable=False)
teacher.eval()
eng = teacher.visual.engine
ops = eng.ops
crops = torch.randn(2048, 3, 224, 224, device=dev)
halves = (crops[:1024].contiguous(), crops[1024:].contiguous())


def single():
    return eng.encode_image(crops, chunk=2048)


def dual(stagger_us=0):
    s1, s2 = streams
    cur = torch.cuda.current_stream()
    outs = []
    ops.reserve_compute_units(reserve)
    for k, (s, h) in enumerate(zip(streams, halves)):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            if k == 1 and stagger_us:
                torch.cuda._sleep(int(stagger_us * 1700))          # ~cycles
            outs.append(eng.encode_image(h, chunk=1024))
    for s in streams:
        cur.wait_stream(s)
    ops.reserve_compute_units(0)
    return torch.cat(outs)


streams = (torch.cuda.Stream(), torch.cuda.Stream())
with torch.no_grad():
    ref = single()
    got = dual()
    torch.cuda.synchronize()
    print("max |single - dual| =", float((ref - got).abs().max()), "(identical rows expected: same kernels, other grids)")
    for name, fn in (("single pass, 256 CUs", single), ("two half passes, 2 streams", dual), ("two half passes, second 300 us late", lambda: dual(300)),
                     ("two half passes, second 1500 us late", lambda: dual(1500)), ("single pass, 256 CUs", single)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        print(f"{name:40s} {(time.perf_counter() - t0) / iters * 1e3:8.2f} ms per 2048 crops")
