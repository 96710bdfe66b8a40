#!/usr/bin/env python
"""Timeline of the fp32-residual epilogue of the streaming GEMM (ablation build, env CS_GEMM_TRACE=<file> while running
tools/stream_ablate.py): per workgroup and wave the 100 MHz clock at epilogue entry and, per 32-row block, after (residual loads issued +
slab written) / after every outstanding access has landed / at the block's end.   usage: python tools/epi_trace.py <file>"""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(256, 8, 16).astype(np.int64)
ok = t[:, :, 0] > 0
t = t[ok].reshape(-1, 16) / 100.0                     # us
print(f"{t.shape[0]} (workgroup, wave) records")
issue = np.stack([t[:, 1 + 3 * i] - (t[:, 0] if i == 0 else t[:, 3 * i]) for i in range(4)], axis=1)
wait = np.stack([t[:, 2 + 3 * i] - t[:, 1 + 3 * i] for i in range(4)], axis=1)
work = np.stack([t[:, 3 + 3 * i] - t[:, 2 + 3 * i] for i in range(4)], axis=1)
for name, a in (("loads issued + slab written", issue), ("wait for the rows (vmcnt 0)", wait), ("arithmetic + stores issued", work)):
    print(f"  {name:30s} per block: mean {a.mean():6.2f} us   by block " + " ".join(f"{x:5.2f}" for x in a.mean(axis=0)) +
          f"   p10 {np.percentile(a, 10):5.2f} p90 {np.percentile(a, 90):5.2f}")
tot = t[:, 12] - t[:, 0]
print(f"  epilogue of a wave: mean {tot.mean():.2f} us (p10 {np.percentile(tot, 10):.2f}, p90 {np.percentile(tot, 90):.2f}); "
      f"waiting {wait.sum(axis=1).mean() / tot.mean():.0%}, issue {issue.sum(axis=1).mean() / tot.mean():.0%}, work {work.sum(axis=1).mean() / tot.mean():.0%}")
