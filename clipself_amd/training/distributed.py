"""One process per GPU; `torch.distributed` (backend 'nccl' == RCCL over xGMI on ROCm; 'gloo' on CPU tests).

Environment handling follows the reference (src/training/distributed.py:36-114): torchrun-style env://
rendezvous from RANK / LOCAL_RANK / WORLD_SIZE (SLURM variables accepted), `--dist-backend`, `--dist-url`.

Gradient exchange is what the reference *intends* but never executes (its CLIPSelf method unwraps `.module`, so
DistributedDataParallel's reducer never fires -- SURVEY.md D3): the mean over ranks of every student gradient.
Here each transformer block's gradients are one contiguous slice of the flat fp32 grad buffer, so the exchange is
one asynchronous all-reduce per block, issued the moment the block's backward finishes (reverse layer order) and
overlapped with the remaining backward; xGMI is point-to-point, so ~28 MB (B/16) buckets keep every ring step
bandwidth- rather than latency-bound.  SUM on the wire, 1/world folded into the AdamW kernel.
"""
import os

import torch
import torch.distributed as dist


def is_global_master(args):
    return args.rank == 0


def is_local_master(args):
    return args.local_rank == 0


def is_master(args, local=False):
    return is_local_master(args) if local else is_global_master(args)


def world_info_from_env():
    def first(names, default):
        for v in names:
            if v in os.environ:
                return int(os.environ[v])
        return default
    return (first(("LOCAL_RANK", "SLURM_LOCALID"), 0), first(("RANK", "SLURM_PROCID"), 0),
            first(("WORLD_SIZE", "SLURM_NTASKS"), 1))


def is_using_distributed():
    return world_info_from_env()[2] > 1


def init_distributed_device(args):
    args.distributed, args.world_size, args.rank, args.local_rank = False, 1, 0, 0
    if getattr(args, "horovod", False):
        raise NotImplementedError("horovod is not part of the MI355X build; use torchrun (RCCL)")
    if is_using_distributed():
        args.local_rank, args.rank, args.world_size = world_info_from_env()
        if not dist.is_initialized():
            bound = args.dist_backend == "nccl" and torch.cuda.is_available() and not getattr(args, "no_set_device_rank", False)
            dist.init_process_group(backend=args.dist_backend, init_method=args.dist_url, world_size=args.world_size, rank=args.rank,
                                    device_id=torch.device(f"cuda:{args.local_rank}") if bound else None)
        args.distributed = True
    if torch.cuda.is_available():
        device = f"cuda:{args.local_rank}" if args.distributed and not getattr(args, "no_set_device_rank", False) else "cuda:0"
        torch.cuda.set_device(device)
    else:
        device = "cpu"
    args.device = device
    return torch.device(device)


def broadcast_object(args, obj, src=0):
    objects = [obj] if args.rank == src else [None]
    dist.broadcast_object_list(objects, src=src)
    return objects[0]


def all_gather_object(args, obj, dst=0):
    objects = [None for _ in range(args.world_size)]
    dist.all_gather_object(objects, obj)
    return objects


def rccl_reserved_cus() -> int:
    """Compute units the persistent GEMM kernels leave free while gradient buckets are in flight (CLIPSELF_RCCL_CUS, default 16 of 256).
    The GEMMs of the step are persistent kernels that own every CU they are launched on for milliseconds; RCCL's ring kernels (a few
    workgroups per channel) need somewhere to run beside them if the gradient all-reduce is to overlap with backward."""
    return max(0, min(64, int(os.environ.get("CLIPSELF_RCCL_CUS", "16"))))


def rccl_teacher_window() -> int:
    """Leading blocks of a PREFETCHED teacher pass whose GEMMs leave the reserved CUs free (CLIPSELF_RCCL_TEACHER_BLOCKS, default 3).  The
    teacher's pass over the next batch starts on the side stream when the student's backward does, so its first blocks run beside the
    gradient buckets (B/16: 12 buckets over an ~11 ms backward, one teacher block = ~6.5 ms); the later blocks and an inline teacher pass
    run while nothing is in flight and keep all 256 CUs."""
    return max(0, int(os.environ.get("CLIPSELF_RCCL_TEACHER_BLOCKS", "3")))


def _data_parallel_active() -> bool:
    return dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("CLIPSELF_FORCE_DIST") == "1")


def _ops_of(module):
    return getattr(getattr(getattr(module, "visual", None), "engine", None), "ops", None)


def grad_bucket_dtype() -> torch.dtype:
    """Wire format of the gradient buckets (CLIPSELF_GRAD_BUCKET_DTYPE = fp32 | bf16, default fp32).  bf16 halves the bytes every ring step
    moves over its xGMI link (168 MB instead of 336 MB per step for B/16) at the price of one bf16 rounding of each rank's partial sum:
    the all-reduce then returns the SUM of the ranks' bf16-rounded gradients, accumulated by RCCL in bf16."""
    v = os.environ.get("CLIPSELF_GRAD_BUCKET_DTYPE", "fp32").lower()
    if v not in ("fp32", "bf16"):
        raise ValueError(f"CLIPSELF_GRAD_BUCKET_DTYPE={v!r}: fp32 or bf16")
    return torch.float32 if v == "fp32" else torch.bfloat16


class StudentDataParallel(torch.nn.Module):
    """`.module`-carrying wrapper (the reference's methods unwrap it: clipself.py:8-10) that (1) broadcasts rank 0's
    parameters once and (2) arms the engine's per-block grad-ready hook with asynchronous bucket all-reduces.

    The student's persistent GEMMs give up `rccl_reserved_cus()` compute units only between the first bucket of a step and
    `finish_grad_sync` -- the window in which RCCL's kernels have something to do; the student forward keeps the whole chip.

    `collect_stats(True)` (bench.py; off in training runs, where nothing would drain them) records what a step exchanged: buckets, bytes
    on the wire per rank, the time the optimizer stood behind unfinished buckets (`finish_grad_sync`), and per bucket the time from its
    issue (the block's backward done on the compute stream) to its completion on the collective's stream -- device events on a GPU, wall
    clock on CPU -- so that a scaling number can be read against its communication volume without a profiler."""

    def __init__(self, module, process_group=None, bucket_dtype=None):
        super().__init__()
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_dtype = bucket_dtype if bucket_dtype is not None else grad_bucket_dtype()
        eng = module.visual.engine
        dist.broadcast(eng.master, src=0, group=process_group)
        with torch.no_grad():
            dist.broadcast(module.logit_scale.data, src=0, group=process_group)
        eng.sync_shadow()
        self._reserve = rccl_reserved_cus() if _data_parallel_active() else 0
        self._pending = []
        self._wire = None                      # bf16 staging buffer of the flat gradient (bucket_dtype == bf16)
        self._collect = False
        self._observer = None                  # side stream that waits for each bucket's completion (statistics only)
        self.reset_stats()
        if eng.trainable:
            eng.grad_ready_hook = self._on_block_ready

    def collect_stats(self, on: bool = True):
        self._collect = bool(on)
        self.reset_stats()

    def _set_reserve(self, n):
        ops = _ops_of(self.module)
        if ops is not None and hasattr(ops, "reserve_compute_units"):
            ops.reserve_compute_units(n)

    def _on_block_ready(self, block):
        eng = self.module.visual.engine
        lo, hi = eng.bucket_range(block) if hasattr(eng, "bucket_range") else eng.block_ranges[block]     # block index, "head" or "stem"
        g = eng.grad[lo:hi]
        if self.bucket_dtype == torch.float32:
            buf = g
        else:
            if self._wire is None:
                self._wire = torch.empty(eng.grad.numel(), dtype=self.bucket_dtype, device=eng.grad.device)
            buf = self._wire[lo:hi]
            if hasattr(eng.ops, "cast_f32_bf16") and g.is_cuda:
                eng.ops.cast_f32_bf16(g, buf)
            else:
                buf.copy_(g)
        if not self._pending and self._reserve:
            self._set_reserve(self._reserve)               # first bucket of the step: the remaining backward GEMMs leave room for RCCL
        mark = None
        if self._collect:
            if g.is_cuda:
                mark = torch.cuda.Event(enable_timing=True)
                mark.record()
            else:
                import time
                mark = time.perf_counter()
        work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        done = None
        if self._collect and g.is_cuda:
            if self._observer is None:
                self._observer = torch.cuda.Stream(device=g.device)
            with torch.cuda.stream(self._observer):
                work.wait()                                # the OBSERVER stream waits for this bucket; the compute stream does not
                done = torch.cuda.Event(enable_timing=True)
                done.record()
        self._pending.append((work, lo, hi, str(block), mark, done))
        self.stats["buckets"] += 1
        self.stats["bytes"] += buf.numel() * buf.element_size()

    def finish_grad_sync(self):
        """Make the current stream wait for every outstanding bucket (call before the optimizer step)."""
        import time
        eng = self.module.visual.engine
        on_gpu = eng.grad.is_cuda
        ev = None
        if self._collect:
            if on_gpu:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            else:
                t0 = time.perf_counter()
        for work, lo, hi, name, mark, done in self._pending:
            work.wait()
            if self.bucket_dtype != torch.float32:
                eng.grad[lo:hi].copy_(self._wire[lo:hi])          # back to the fp32 accumulator AdamW reads
            if self._collect:
                if on_gpu:
                    self._bucket_events.append((name, mark, done))
                else:
                    self.stats["bucket_ms"].setdefault(name, []).append(1e3 * (time.perf_counter() - mark))
        self._pending.clear()
        if self._reserve:
            self._set_reserve(0)                           # nothing in flight any more: AdamW's successors get the whole chip back
        if self._collect:
            if on_gpu:
                ev[1].record()
                self._wait_events.append(ev)
            else:
                self.stats["wait_ms"] += 1e3 * (time.perf_counter() - t0)
        self.stats["steps"] += 1

    def reset_stats(self):
        self.stats = dict(steps=0, buckets=0, bytes=0, wait_ms=0.0, bucket_ms={})
        self._wait_events = []
        self._bucket_events = []

    def comm_summary(self):
        """Per-step communication figures of this rank since the last reset_stats(): buckets, bytes handed to the all-reduce, exposed wait
        (the time the stream that runs AdamW stood behind unfinished buckets) and -- with collect_stats(True) -- per bucket the mean time from
        issue to completion.  Synchronises the device events."""
        if self._wait_events or self._bucket_events:
            torch.cuda.synchronize()
            self.stats["wait_ms"] += sum(a.elapsed_time(b) for a, b in self._wait_events)
            for name, mark, done in self._bucket_events:
                self.stats["bucket_ms"].setdefault(name, []).append(mark.elapsed_time(done))
            self._wait_events, self._bucket_events = [], []
        n = max(self.stats["steps"], 1)
        out = dict(allreduce_buckets_per_step=self.stats["buckets"] / n, allreduce_bytes_per_step=self.stats["bytes"] / n,
                   grad_sync_wait_ms=self.stats["wait_ms"] / n, grad_bucket_dtype="fp32" if self.bucket_dtype == torch.float32 else "bf16",
                   rccl_reserved_cus=self._reserve, rccl_reserved_window="first bucket .. finish_grad_sync"
                   if self._reserve else "none", stats_collected=self._collect)
        if self.stats["bucket_ms"]:
            # issue order = reverse layer order; value = mean ms from "block's backward done" to "all-reduce of its bucket done"
            out["bucket_issue_to_done_ms"] = {k: round(sum(v) / len(v), 3) for k, v in self.stats["bucket_ms"].items()}
        return out

    def forward(self, *a, **k):
        return self.module(*a, **k)


class FrozenDataParallel(torch.nn.Module):
    """Teacher wrapper: parameters broadcast once, no gradient traffic (main.py:191-192).  A teacher pass that is PREFETCHED beside the
    student's backward leaves the reserved CUs free in its first `rccl_teacher_window()` blocks (engine.rccl_window, applied by
    CLIPSelf.prefetch_teacher); an inline pass runs while no bucket is in flight and reserves nothing."""

    def __init__(self, module, process_group=None):
        super().__init__()
        self.module = module
        dist.broadcast(module.visual.engine.master, src=0, group=process_group)
        module.visual.engine.sync_shadow()
        # the frozen schedule's fold guard is decided HERE, where every rank is (a MAX all-reduce of one scalar), not inside the first forward
        if hasattr(module.visual.engine, "calibrate_block_folds") and module.visual.engine.block_fold_guard and module.visual.engine.fold_block_ln:
            module.visual.engine.calibrate_block_folds()
        self.prefetch_window = (rccl_teacher_window(), rccl_reserved_cus()) if _data_parallel_active() else (0, 0)

    def forward(self, *a, **k):
        return self.module(*a, **k)
