#!/usr/bin/env python
"""Headline benchmark: images/sec of the CLIPSelf distillation step (student + teacher), BASELINE.json metric.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" = one full training iteration of the hot path on one synthetic batch already resident in HBM:
teacher EVA02-CLIP-B-16 forward over 64x32 region crops (224^2) -> student dense forward over 64 images (224^2) ->
RoIAlign -> cosine loss -> student backward (12 blocks) -> [bucketed RCCL all-reduce] -> AdamW, through the same
`train_step` the training entrypoint uses.  Workload = BASELINE.json configs[1].

The JSON line also carries
  roofline      MFMA roofline of the dominant kernel (the fused W1|W2 SwiGLU GEMM of the teacher, one shape per
                launch): algorithmic FLOPs per launch / mean launch duration, measured with HIP events recorded on
                the launch stream inside the timed region; peak = 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md).
  cpu_baseline  the fp32 CPU oracle (oracle/eva_ref.py, validated against the reference) timed on the host cores on a
                bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path
from types import SimpleNamespace

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

MODEL = "EVA02-CLIP-B-16"
BATCH, CROPS, SIZE = 64, 32, 224
PEAK_BF16_TFLOPS = 2500.0
DOMINANT_KERNEL = ("gemm_stream_kernel<EPI_SWIGLU_BF16, LN, stats> (teacher W1|W2 GEMM with norm2 folded in + SiLU*mul + ffn_ln partial "
                   "statistics, M=chunk*197,N=4096,K=768)")


def flops_per_image(cfg, k, cls_only=True):
    """2*MACs of the matmuls the step executes."""
    N, C, Hd, E, L, p = cfg.tokens, cfg.width, cfg.hidden, cfg.embed_dim, cfg.layers, cfg.patch_size
    pe = 2 * (N - 1) * (3 * p * p) * C
    blk = 2 * N * C * C * 4 + 4 * N * N * C + 6 * N * C * Hd
    blk_na = 4 * N * C * C + 6 * N * C * Hd
    # teacher: the last block serves the CLS query only (k/v projections over all tokens, everything else on one row)
    blk_cls = 4 * N * C * C + 2 * C * C * 2 + 4 * N * C + 6 * C * Hd
    T = pe + (L - 1) * blk + (blk_cls if cls_only else blk) + 2 * C * E
    Sf = pe + (L - 1) * blk + blk_na + 2 * (N - 1) * C * E
    Sb = 2 * ((L - 1) * blk + blk_na) + 2 * (N - 1) * C * E
    return k * T + Sf + Sb


class KernelTimer:
    """HIP-event timing of one kernel class on its launch stream (our launches go to torch's current stream)."""

    def __init__(self, ops, epi, min_rows=0, max_events=4096):
        self.ops, self.epi, self.min_rows, self.on, self.pairs, self.flops = ops, epi, min_rows, False, [], 0.0
        self._pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max_events)]
        # both entry points of the GEMM (cs_gemm_nt and, with the folded sub-LayerNorm operands, cs_gemm_nt_ln)
        ops.gemm_nt = self._wrap(ops.gemm_nt)
        ops.gemm_nt_ln = self._wrap(ops.gemm_nt_ln)

    def _wrap(self, inner):
        def wrapped(A, B, C, *args, **kw):
            epi = kw.get("epi", args[2] if len(args) > 2 and inner.__name__ == "gemm_nt" else None)
            if self.on and epi == self.epi and A.shape[0] >= self.min_rows and len(self.pairs) < len(self._pool):
                e0, e1 = self._pool[len(self.pairs)]
                e0.record()
                inner(A, B, C, *args, **kw)
                e1.record()
                self.pairs.append((e0, e1))
                self.flops += 2.0 * A.shape[0] * B.shape[0] * A.shape[1]
            else:
                inner(A, B, C, *args, **kw)
        return wrapped

    def result(self):
        if not self.pairs:
            return None
        ms = sum(a.elapsed_time(b) for a, b in self.pairs)
        return dict(launches=len(self.pairs), mean_us=1e3 * ms / len(self.pairs), tflops=self.flops / (ms * 1e-3) / 1e12,
                    flops_per_launch=self.flops / len(self.pairs))


def isolated_swiglu_gemm(ops, M, cfg, launches=5):
    """Mean duration (us) of the dominant kernel's launch shape with nothing else on the GPU (HIP events)."""
    C, Hd = cfg.width, ((cfg.hidden + 63) // 64) * 64
    A = torch.randn(M, C, device="cuda").to(torch.bfloat16)
    W = (torch.randn(2 * Hd, C, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(2 * Hd, device="cuda")
    hid = torch.empty(M, Hd, dtype=torch.bfloat16, device="cuda")
    part = torch.empty(4 * ((Hd + 127) // 128), M, 2, device="cuda")
    # the same operands as in the step: norm2 folded in (row mean / rstd + column sums) and the ffn_ln partial statistics out
    mean, rstd, colsum = torch.zeros(M, device="cuda"), torch.ones(M, device="cuda"), torch.randn(2 * Hd, device="cuda")
    kw = dict(bias=bias, ln_mean=mean, ln_rstd=rstd, ln_colsum=colsum, stats_part=part, epi=3, group=Hd)
    torch.cuda.synchronize()
    ops.gemm_nt_ln(A, W, hid, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        ops.gemm_nt_ln(A, W, hid, **kw)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / launches


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def physical_cores():
    """Physical cores of the host (distinct (package, core) pairs of /proc/cpuinfo; SMT siblings counted once), capped by the cores this
    process may run on."""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if core is not None:
                    seen.add((phys, core))
                phys = core = None
        if core is not None:
            seen.add((phys, core))
        n = len(seen)
    except OSError:
        n = 0
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                                       # a container's CPU quota (cgroup v2 cpu.max = "<quota> <period>" or "max ...")
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            avail = min(avail, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, avail) if n else avail)


def cpu_baseline_worker():
    """fp32 CPU oracle (oracle/eva_ref.py) on the host cores, protocol of SURVEY.md section 8 M5: BASELINE configs[0] exactly (2 images x 8
    boxes, 224^2; 1 warm-up + 3 timed optimizer steps) and the benchmark's own unit scaled down in images only (2 images x 32 crops, 2 timed
    steps; the step is linear in images).  `value` = images/s of the 32-crop steps.  Prints a JSON object."""
    from clipself_amd.config import get_tower_cfg
    from clipself_amd.init import seeded_visual_state, synthetic_batch
    from oracle import eva_ref
    cfg = get_tower_cfg(MODEL)
    student, teacher = seeded_visual_state(cfg, 0), seeded_visual_state(cfg, 0)
    # all physical cores of the host -- unless fewer threads are FASTER on this box (two sockets / a shared host: 128 threads
    # ran the step 6x slower than 32 on the round-3 box): the thread count is calibrated on one untimed step each and the best one is used
    # and reported, so that the baseline is the CPU's best showing
    phys = physical_cores()
    tried = {}
    calib = synthetic_batch(2, 8, SIZE, SIZE, seed=999)
    for n in sorted({phys, max(1, phys // 2), max(1, phys // 4), min(phys, 32)}, reverse=True):
        torch.set_num_threads(n)
        eva_ref.train_steps(student, teacher, cfg, [calib])                  # warm the pools at this width
        t0 = time.time()
        eva_ref.train_steps(student, teacher, cfg, [calib])
        tried[n] = time.time() - t0
    cores = min(tried, key=tried.get)
    torch.set_num_threads(cores)
    student, teacher = seeded_visual_state(cfg, 0), seeded_visual_state(cfg, 0)

    def timed(batches):
        ts = []
        for b in batches:
            t0 = time.time()
            eva_ref.train_steps(student, teacher, cfg, [b])
            ts.append(time.time() - t0)
        return ts

    cfg1 = [synthetic_batch(2, 8, SIZE, SIZE, seed=1234 + i) for i in range(4)]
    timed(cfg1[:1])                                                          # warm-up (allocator, thread pools)
    t1 = timed(cfg1[1:])
    t32 = timed([synthetic_batch(2, CROPS, SIZE, SIZE, seed=4321 + i) for i in range(2)])
    m1, m32 = sum(t1) / len(t1), sum(t32) / len(t32)
    print(json.dumps({"value": 2.0 / m32, "unit": "images/sec", "cores": cores, "kind": "port", "cpu": cpu_model_name(),
                      "cfg1_s_per_step": m1, "cfg1_images_per_sec": 2.0 / m1,
                      "sample": f"{MODEL} fp32 CPU oracle (oracle/eva_ref.py), full optimizer steps (teacher + student fwd/bwd + AdamW): "
                                f"BASELINE configs[0] exactly, 2 images x 8 boxes, {len(t1)} timed steps {m1:.2f} s/step; the benchmark's unit "
                                f"scaled down in images only, 2 images x {CROPS} crops, {len(t32)} timed steps {m32:.2f} s/step = {2.0 / m32:.3f} images/s; "
                                f"{cores} torch threads on {cpu_model_name()} ({phys} physical cores available; one calibration step per "
                                f"thread count: " + ", ".join(f"{n}: {t:.2f} s" for n, t in sorted(tried.items())) + ")"}))


def cpu_baseline(timeout_s=300):
    import subprocess
    try:
        r = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--cpu-baseline-worker"], capture_output=True, text=True,
                           timeout=timeout_s, env={**os.environ, "CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""})
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "error": (r.stderr or "no output")[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "error": f"cpu baseline exceeded {timeout_s}s"}


def launch_ranks(n, argv):
    """Re-execute this script as n ranks of one node (python -m torch.distributed.run, rendezvous on 127.0.0.1, a free port)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    env = {**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
    return subprocess.run(cmd, env=env).returncode


def pmc_traffic_per_launch(chunk_crops):
    """HBM/fabric bytes of one dominant-kernel launch from the tracked PMC summary (profiles/pmc_traffic.json: separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x 2 gfx950 correction; written by tools/rocprof_pmc.py), scaled linearly in the
    chunk; None when the summary does not cover the kernel this build launches."""
    try:
        rec = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())["dominant"]
        return (2.0 * rec["fetch_kb"] + rec["write_kb"]) * 1024.0 * chunk_crops / rec["chunk_crops"], rec
    except (OSError, KeyError, ValueError):
        return None, None


def measure_traffic_live(chunk_crops, timeout_s=150):
    """Fabric bytes of one dominant-kernel launch MEASURED in this run: two rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE: counters
    only, no trace domain, one counter set per pass as MI355X_MICROARCH.md prescribes) over tools/raster_one.py, which launches exactly the
    kernel the step's dominant launch is (same shape, same flags) a few times.  FETCH_SIZE x 2 (gfx950 correction for wide streaming
    reads), both in KB.  Returns (bytes per launch, record) or (None, reason) -- any failure (no rocprofv3, timeout, unreadable result)
    leaves the tracked constant in place."""
    import shutil
    import signal
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="cs_pmc_")
        cmd = [exe, "--pmc", counter, "-d", tmp, "-o", "t", "--", sys.executable, str(ROOT / "tools" / "raster_one.py"), "0", str(chunk_crops), "3"]
        try:
            proc = subprocess.Popen(cmd, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True,
                                    env=dict(os.environ, TMPDIR=tmp))
            try:
                proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)
                return None, f"rocprofv3 --pmc {counter} timed out"
            dbs = [os.path.join(r, f) for r, _, fs in os.walk(tmp) for f in fs if f.endswith(".db")]
            if proc.returncode != 0 or not dbs:
                return None, f"rocprofv3 --pmc {counter}: rc {proc.returncode}, no result database"
            cur = sqlite3.connect(dbs[0]).cursor()
            rows = [v for k, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection")
                    if c == counter and "gemm_stream_kernel<3" in k]
            if not rows:
                return None, f"rocprofv3 --pmc {counter}: the dominant kernel is not in the result"
            vals[counter] = sum(rows) / len(rows)
        except (OSError, sqlite3.Error) as e:
            return None, f"rocprofv3 --pmc {counter}: {e}"
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    rec = {"fetch_kb": vals["FETCH_SIZE"], "write_kb": vals["WRITE_SIZE"], "chunk_crops": chunk_crops,
           "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/raster_one.py, "
                     "per-launch averages; FETCH_SIZE x 2 on gfx950"}
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, rec


ALSO_WORKLOADS = ("recipe_b16", "recipe_l14", "cfg2", "cfg3", "cfg4_bf16", "cfg4_fp8", "openai_b16")
# one-GPU step time the multi-GPU expectation is built on (ms; the driver's BENCH_r05 run of this workload) -- refreshed at the end of a round
ONE_GPU_MS_REFERENCE = 87.3


def also_workloads(steps=8, timeout_s=240):
    """The configurations beside the headline one, driver-observed: the reference's own recipe shape for both towers,
    BASELINE configs[2], configs[3] and configs[4] (bf16 and fp8) at one GPU's share, the headline workload on the OpenAI-CLIP ViT-B/16 towers; a few steps each in a process of their own
    (tools/also_bench.py) after the timed region.  A workload that fails or times out is reported as such; the headline fields are not touched."""
    import subprocess
    res = []
    for w in ALSO_WORKLOADS:
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, str(ROOT / "tools" / "also_bench.py"), w, str(steps)], capture_output=True, text=True, timeout=timeout_s)
            line = next((ln for ln in reversed(r.stdout.strip().splitlines()) if ln.startswith("{")), None)
            rec = json.loads(line) if line else {"id": w, "error": (r.stderr or "no output")[-300:]}
        except subprocess.TimeoutExpired:
            rec = {"id": w, "error": f"exceeded {timeout_s}s"}
        except (OSError, ValueError) as e:
            rec = {"id": w, "error": str(e)[-300:]}
        rec["wall_s"] = round(time.time() - t0, 1)
        res.append(rec)
    return res


def expected_data_parallel(world, cfg_blocks=12, bucket_mb=28.3, one_gpu_ms=ONE_GPU_MS_REFERENCE, backward_ms=10.0, link_gbs_dir=76.0,
                           links=7, reserve_ms=1.3):
    """What a `--gpus N` line should show if the data-parallel path works as designed (DESIGN.md section 5's table as a function of N), printed
    beside the measured `data_parallel` fields so that the first multi-GPU line judges itself.  One node, xGMI fully connected: ~153 GB/s per
    link = ~76 GB/s per direction; ring all-reduce puts 2 (N-1)/N of a bucket on a rank's links; RCCL may spread its rings over min(N-1, 7)
    links (best case) or run one ring (worst case); ring latency ~ 2 (N-1) steps x ~10 us."""
    n = max(world, 1)
    per_bucket_link_mb = 2.0 * (n - 1) / n * bucket_mb
    lat_ms = 2 * (n - 1) * 0.010
    worst = per_bucket_link_mb / link_gbs_dir + lat_ms                 # MB / (GB/s) = ms
    best = per_bucket_link_mb / (link_gbs_dir * max(1, min(n - 1, links))) + lat_ms
    spacing = backward_ms / cfg_blocks
    step_lo, step_hi = one_gpu_ms + best + (reserve_ms if n > 1 else 0.0), one_gpu_ms + worst + (reserve_ms if n > 1 else 0.0)
    return {"allreduce_mb_per_step": round(cfg_blocks * bucket_mb, 1), "buckets_per_step": cfg_blocks,
            "mb_on_a_ranks_links_per_bucket": round(per_bucket_link_mb, 1),
            "bucket_issue_to_done_ms": [round(best, 2), round(worst, 2)], "bucket_issue_spacing_ms": round(spacing, 2),
            "buckets_queue_behind_each_other": bool(worst > spacing),
            "grad_sync_wait_ms": [round(best, 2), round(worst, 2)], "reserve_cost_ms": reserve_ms if n > 1 else 0.0,
            "ms_per_step": [round(step_lo, 1), round(step_hi, 1)],
            "images_per_s": [round(n * BATCH / step_hi * 1e3), round(n * BATCH / step_lo * 1e3)],
            "speedup_over_one_gpu": [round(n * one_gpu_ms / step_hi, 2), round(n * one_gpu_ms / step_lo, 2)],
            "one_gpu_ms_reference": one_gpu_ms,
            "reading": "bucket_issue_to_done_ms >> the range: RCCL's kernels are starved behind the persistent GEMMs -> raise CLIPSELF_RCCL_CUS; "
                       "growing from bucket to bucket: the links are the limit -> --bf16-grad-buckets halves the bytes"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="report roofline.traffic from the tracked PMC summary (profiles/pmc_traffic.json) instead of two rocprofv3 counter passes "
                         "after the timed region (also: CLIPSELF_BENCH_NO_PMC=1)")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the short runs of the other configurations (recipe shape, configs[3], configs[4]) that follow the timed region "
                         "(the `also` key; also: CLIPSELF_BENCH_NO_ALSO=1)")
    ap.add_argument("--also-steps", type=int, default=8)
    ap.add_argument("--teacher-chunk", type=int, default=2048,
                    help="crops per teacher launch; 2048 = the whole batch of configs[1] in one pass (~7 GB of live activations)")
    ap.add_argument("--full-last-block", action="store_true",
                    help="run the teacher's last block over every token instead of the CLS query only (same outputs, more work)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the teacher inline on the main stream instead of one batch ahead on a side stream (A/B switch)")
    ap.add_argument("--no-split-stream", action="store_true",
                    help="A/B: keep the teacher's residual stream in fp32 + bf16 copy (10 B / element / residual GEMM) instead of the two 16-bit planes (8 B)")
    ap.add_argument("--no-block-ln-fold", action="store_true",
                    help="keep norm1 / norm2 of the teacher as LayerNorm kernels (only the two sub-LayerNorms folded; A/B switch)")
    ap.add_argument("--bf16-grad-buckets", action="store_true",
                    help="data parallel: all-reduce the gradient buckets in bf16 (half the xGMI bytes; CLIPSELF_GRAD_BUCKET_DTYPE=bf16)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher check without a GPU: the ranks rendezvous over gloo, all-reduce a one and rank 0 prints the world size")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.cpu_baseline_worker:
        cpu_baseline_worker()
        return
    if a.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # `python bench.py --gpus N` on its own: become the launcher -- one rank per GPU under torch.distributed.run, same arguments
        # (the reference's launch line: scripts/train_clipself_coco_image_patches_eva_vitb16.sh:1, torchrun --nproc_per_node 8)
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", 1)) != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE')} ranks")

    import torch.distributed as dist
    if a.dry_run:
        world, rank = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0))
        total = 1.0
        if world > 1:
            dist.init_process_group(backend="gloo", init_method="env://", world_size=world, rank=rank)
            t = torch.ones(1)
            dist.all_reduce(t)
            total = float(t)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": int(total)}), flush=True)
        return
    from clipself_amd.init import synthetic_batch
    from clipself_amd.open_clip import create_model
    from clipself_amd.training.clipself import CLIPSelf, mark_all_valid
    from clipself_amd.training.distributed import FrozenDataParallel, StudentDataParallel
    from clipself_amd.training.optim import FlatAdamW
    from clipself_amd.training.scheduler import cosine_lr
    from clipself_amd.training.train import train_step

    rank, local, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    # CLIPSELF_FORCE_DIST=1: take the data-parallel path (process group, bucketed all-reduce, CU reservation) even with one rank --
    # the only way to run the RCCL calls of the N-rank path on a one-GPU box
    distributed = world > 1 or os.environ.get("CLIPSELF_FORCE_DIST") == "1"
    # CLIPSELF_DIST_BACKEND=gloo + fewer devices than ranks is the single-GPU rehearsal of the N-rank path used by
    # tests/test_gpu_step.py (RCCL refuses two ranks on one device); the driver's runs use nccl (= RCCL), one rank per GPU.
    if torch.cuda.device_count() < 1:
        sys.exit("bench.py: no ROCm device visible (the step has no CPU path)")
    if world > torch.cuda.device_count() and os.environ.get("CLIPSELF_DIST_BACKEND", "nccl") == "nccl":
        sys.exit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} visible GPUs (RCCL needs one device per rank)")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("CLIPSELF_DIST_BACKEND", "nccl")
        dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank,
                                device_id=torch.device(f"cuda:{local}") if backend == "nccl" else None)
    device = f"cuda:{local}"

    student = create_model(MODEL, "eva", precision="amp_bf16", device=device, cache_dir=None)
    teacher = create_model(MODEL, "eva", precision="amp_bf16", device=device, cache_dir=None, trainable=False)
    teacher.visual.teacher_chunk = a.teacher_chunk
    teacher.visual.engine.cls_only_last_block = not a.full_last_block
    teacher.visual.engine.fold_block_ln = not a.no_block_ln_fold
    teacher.visual.engine.split_stream = not a.no_split_stream
    cfg = student.visual.cfg
    student.lock_image_tower(unlocked_groups=cfg.layers)
    student.train()
    teacher.eval()
    model, dist_model = student, teacher
    if distributed:
        model = StudentDataParallel(student, bucket_dtype=torch.bfloat16 if a.bf16_grad_buckets else None)
        dist_model = FrozenDataParallel(teacher)
    opt = FlatAdamW(student, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1, grad_divisor=float(world))
    sched = cosine_lr(opt, 1e-5, 1000, 100000)
    args = SimpleNamespace(device=device, precision="amp_bf16", distributed=distributed, skip_scheduler=False, grad_clip_norm=None,
                           multiscale=False, extract_type="v2", cosine_weight=1.0)
    # two distinct synthetic batches, alternated: step i trains on batches[i % 2] while the frozen teacher's pass over
    # batches[(i + 1) % 2] runs one step ahead on a side stream (train_step(next_batch=...)); every step -- warm-up and timed --
    # launches exactly one teacher pass and one student pass, and the final synchronize waits for both streams.
    batches = [tuple(t.to(device) for t in synthetic_batch(BATCH, CROPS, SIZE, SIZE, seed=1234 + 977 * j, rank=rank)) for j in range(2)]
    for b in batches:
        mark_all_valid(b[1], True)              # the generator makes every slot valid: no read-back of the validity column per step
    args.teacher_prefetch = not a.no_overlap
    method = CLIPSelf()
    # the fused SwiGLU GEMM is launched by the teacher's engine; the full-token launches (M = chunk*197) are the dominant kernel
    timer = KernelTimer(teacher.visual.engine.ops, epi=3, min_rows=min(a.teacher_chunk, BATCH * CROPS) * cfg.tokens)

    def sync():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    step = 0
    for _ in range(a.warmup):
        train_step(model, method, batches[step % 2], opt, sched, step, dist_model, args, next_batch=batches[(step + 1) % 2])
        step += 1
    sync()
    if distributed:
        model.collect_stats(True)               # per-bucket issue -> completion events and the exposed wait, for the timed steps only
    timer.on = True
    t0 = time.perf_counter()
    last = None
    for _ in range(a.steps):
        last, _, _ = train_step(model, method, batches[step % 2], opt, sched, step, dist_model, args, next_batch=batches[(step + 1) % 2])
        step += 1
    sync()
    elapsed = time.perf_counter() - t0
    timer.on = False
    comm = None
    if distributed:
        # what the step exchanged and how long AdamW stood behind it (max over ranks), and how many ranks the collective really spans
        comm = model.comm_summary()
        t = torch.tensor([elapsed, comm["grad_sync_wait_ms"]], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, comm["grad_sync_wait_ms"] = float(t[0]), float(t[1])
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        comm["ranks_seen"] = int(ones.item())
    loss = last["loss"].detach().item() if last is not None else float("nan")

    if rank == 0:
        ips = world * BATCH * a.steps / elapsed
        F = flops_per_image(cfg, CROPS, cls_only=not a.full_last_block)
        kt = timer.result()
        traffic, pmc = pmc_traffic_per_launch(min(a.teacher_chunk, BATCH * CROPS))
        if world == 1 and not a.no_live_traffic and not a.no_cpu_baseline and os.environ.get("CLIPSELF_BENCH_NO_PMC") != "1":
            # the default bench run re-measures the figure (after the timed region; ~40 s); the tracked constant stays as the fallback
            torch.cuda.synchronize()
            live, rec = measure_traffic_live(min(a.teacher_chunk, BATCH * CROPS))
            if live is not None:
                traffic, pmc = live, rec
            elif pmc is not None:
                pmc = dict(pmc, source=f"tracked constant profiles/pmc_traffic.json ({rec}); " + pmc.get("source", ""))
        out = {
            "metric": "images/sec (student+teacher distill step), ViT-B/16 32 crops/img",
            "value": ips, "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / max(a.steps, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{MODEL} CLIPSelf image-patches step, {BATCH} images x {CROPS} crops per GPU, {SIZE}^2 (BASELINE configs[1])",
                       "global_batch": BATCH * world, "crops_per_image": CROPS, "image_size": SIZE,
                       "parallelism": f"dp{world}", "teacher_chunk": a.teacher_chunk,
                       "teacher_last_block": "full" if a.full_last_block else "cls_query_only",
                       "teacher_schedule": "inline" if a.no_overlap else "one_batch_ahead_on_side_stream", "loss_last_step": loss},
            "step_tflops": F * ips / 1e12, "step_mfma_frac": F * ips / 1e12 / (PEAK_BF16_TFLOPS * world),
        }
        if kt:
            out["roofline"] = {"bound": "mfma", "achieved": kt["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": kt["tflops"] / PEAK_BF16_TFLOPS,
                               # fabric bytes per launch: two rocprofv3 counter passes after the timed region, or the tracked PMC summary (traffic_source says which)
                               "traffic": traffic, "traffic_source": (pmc or {}).get("source"),
                               "kernel": DOMINANT_KERNEL,
                               "launches": kt["launches"], "mean_us": kt["mean_us"], "flops_per_launch": kt["flops_per_launch"]}
            if not a.no_overlap:
                # In the overlapped schedule this kernel shares the CUs with the student's kernels for part of the step, which
                # stretches its launches; the same launch alone on the GPU (after the timed region) is reported beside it.
                iso = isolated_swiglu_gemm(teacher.visual.engine.ops, min(a.teacher_chunk, BATCH * CROPS) * cfg.tokens, cfg)
                out["roofline"]["isolated"] = {"mean_us": iso, "achieved": kt["flops_per_launch"] / iso / 1e6,
                                               "frac": kt["flops_per_launch"] / iso / 1e6 / PEAK_BF16_TFLOPS}
        if comm is not None:
            # data-parallel diagnostics: bytes each rank hands to the all-reduce per step (a ring moves 2 (N-1)/N of them over every
            # xGMI link), buckets, the exposed wait in front of AdamW, CUs the persistent GEMMs leave to RCCL's kernels
            out["data_parallel"] = comm
            out["data_parallel"]["expected"] = expected_data_parallel(world, cfg_blocks=cfg.layers)
        if world == 1 and not a.no_also and not a.no_cpu_baseline and os.environ.get("CLIPSELF_BENCH_NO_ALSO") != "1":      # (quick --no-cpu-baseline runs skip it, like the PMC passes)
            # free this process's towers and activations first: the other workloads get the device to themselves
            del timer, model, dist_model, student, teacher, opt, batches, method, last
            import gc
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            out["also"] = also_workloads(a.also_steps)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
