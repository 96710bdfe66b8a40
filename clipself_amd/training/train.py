"""Training loop with the reference's step order (src/training/train.py:62-165):
   scheduler(step) -> optimizer.zero_grad() -> method(batch, ...) -> total_loss.backward() -> [grad clip]
   -> optimizer.step() -> logit_scale.clamp_(0, ln 100) -> meters / log line (loss, samples/s, lr, logit scale).
No GradScaler: the engine's bf16/fp32 mix needs none (precision.py).  In data-parallel runs the per-block gradient
all-reduces launched during backward are awaited right before the optimizer step."""
import logging
import math
import time

import torch

from .distributed import is_master
from .precision import get_autocast


class AverageMeter:
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def unwrap_model(model):
    return model.module if hasattr(model, "module") else model


def backward(total_loss, scaler=None):
    total_loss.backward()


@torch.no_grad()
def student_teacher_ensemble(student, teacher, alpha=0.5):
    """Weight-space ensemble saved at the end of every epoch (train.py:53-59, main.py:280-298)."""
    return {k: v * alpha + teacher[k] * (1.0 - alpha) for k, v in student.items()}


def train_step(model, method, batch, optimizer, scheduler, step, dist_model, args, loss=None, next_batch=None):
    """One iteration of the loop body (train.py:80-119).  Returns (losses, batch_size, logit_scale).
    next_batch (optional): handed to method.prefetch_teacher() once this step's forward is queued, so the frozen teacher's pass
    over the next batch overlaps this step's backward / all-reduce / AdamW on the GPU."""
    device = torch.device(args.device)
    cast_dtype = torch.bfloat16 if args.precision == "bf16" else None
    if scheduler is not None and not getattr(args, "skip_scheduler", False):
        scheduler(step)
    optimizer.zero_grad()
    with get_autocast(args.precision)():
        losses, batch_size, logit_scale = method(batch, model, dist_model, loss, device, cast_dtype, args.distributed, args)
        total_loss = sum(losses.values())
        losses["loss"] = total_loss
    if next_batch is not None and getattr(args, "teacher_prefetch", True) and hasattr(method, "prefetch_teacher"):
        method.prefetch_teacher(next_batch, dist_model, device, cast_dtype, args.distributed)
    backward(total_loss)
    if hasattr(model, "finish_grad_sync"):
        model.finish_grad_sync()
    if getattr(args, "grad_clip_norm", None) is not None:
        # data parallel: the buckets carry the SUM over ranks until AdamW divides by the world size, so the threshold is scaled
        # with it -- the clip coefficient then equals the one of the mean gradient (= a single process on the union batch)
        scale = float(getattr(model, "world", 1))
        torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], args.grad_clip_norm * scale, norm_type=2.0)
    optimizer.step()
    with torch.no_grad():
        # train.py:118-119 clamps every step.  A clamp of an unchanged, in-range scalar is the identity; skipping it then keeps the parameter's
        # version counter still, which RegionCLIP uses to read the temperature back from the device once instead of every step.
        ls = unwrap_model(model).logit_scale
        if ls.grad is not None or not getattr(ls, "_cs_clamped_once", False):
            ls.clamp_(0, math.log(100))
            ls._cs_clamped_once = True
    return losses, batch_size, logit_scale


def train_one_epoch(model, method, data, loss, epoch, optimizer, scaler, scheduler, dist_model, args):
    model.train()
    if dist_model is not None:
        dist_model.eval()
    data["train"].set_epoch(epoch)
    dataloader = data["train"].dataloader
    assert args.accum_freq == 1, "accum freq disabled"
    num_batches_per_epoch = dataloader.num_batches // args.accum_freq
    sample_digits = math.ceil(math.log(dataloader.num_samples + 1, 10))
    losses_m, batch_time_m, data_time_m = {}, AverageMeter(), AverageMeter()
    end = time.time()
    batches = iter(dataloader)
    upcoming = next(batches, None)
    i = -1
    while upcoming is not None:
        batch, upcoming = upcoming, next(batches, None)           # one batch of look-ahead for the teacher prefetch
        i += 1
        step = num_batches_per_epoch * epoch + i
        data_time_m.update(time.time() - end)
        losses, batch_size, logit_scale = train_step(model, method, batch, optimizer, scheduler, step, dist_model, args, loss,
                                                     next_batch=upcoming)
        batch_time_m.update(time.time() - end)
        end = time.time()
        batch_count = i + 1
        if is_master(args) and (i % args.log_every_n_steps == 0 or batch_count == num_batches_per_epoch):
            num_samples = batch_count * batch_size * args.accum_freq * args.world_size
            for key, val in losses.items():                       # .item() = the only device sync, log steps only
                losses_m.setdefault(key, AverageMeter()).update(val.item(), batch_size)
            loss_log = " ".join(f"{k.capitalize()}: {m.val:#.5g} ({m.avg:#.5g})" for k, m in losses_m.items())
            sps = args.accum_freq * args.batch_size * args.world_size / batch_time_m.val
            logging.info(
                f"Train Epoch: {epoch} [{num_samples:>{sample_digits}}/{dataloader.num_samples} "
                f"({100.0 * batch_count / num_batches_per_epoch:.0f}%)] Data (t): {data_time_m.avg:.3f} "
                f"Batch (t): {batch_time_m.avg:.3f}, {sps:#g}/s, {sps / args.world_size:#g}/s/gpu "
                f"LR: {optimizer.param_groups[0]['lr']:5f} Logit Scale: {logit_scale.item():.3f} " + loss_log)
            batch_time_m.reset()
            data_time_m.reset()


def evaluate(model, data, epoch, args):
    """Zero-shot region classification on the validation split + the reference's bookkeeping (train.py:168-195)."""
    import json
    import os
    from .zero_shot import zero_shot_eval
    model.eval()
    metrics = dict(zero_shot_eval(model, data, epoch, args))
    if not is_master(args) or not metrics:
        return {} if not is_master(args) else metrics
    logging.info(f"Eval Epoch: {epoch}. " + ", ".join(f"{k}: {v:.4f}" for k, v in metrics.items()))
    if getattr(args, "save_logs", False) and getattr(args, "checkpoint_path", None):
        with open(os.path.join(args.checkpoint_path, "results.json"), "a+") as f:
            f.write(json.dumps(metrics) + "\n")
    return metrics
