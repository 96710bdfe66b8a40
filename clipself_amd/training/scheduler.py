"""LR schedules with the reference's semantics (src/training/scheduler.py): linear warm-up `lr*(step+1)/warmup`,
then cosine / constant / constant-with-cooldown; the returned callable sets `lr` on every param group."""
import math


def _set(optimizer, lr):
    for group in optimizer.param_groups:
        group["lr"] = lr
    return lr


def _warm(base_lr, warmup, step):
    return base_lr * (step + 1) / warmup


def cosine_lr(optimizer, base_lr, warmup_length, steps):
    def adjust(step):
        if step < warmup_length:
            return _set(optimizer, _warm(base_lr, warmup_length, step))
        done, span = step - warmup_length, steps - warmup_length
        return _set(optimizer, 0.5 * (1 + math.cos(math.pi * done / span)) * base_lr)
    return adjust


def const_lr(optimizer, base_lr, warmup_length, steps):
    def adjust(step):
        return _set(optimizer, _warm(base_lr, warmup_length, step) if step < warmup_length else base_lr)
    return adjust


def const_lr_cooldown(optimizer, base_lr, warmup_length, steps, cooldown_steps, cooldown_power=1.0, cooldown_end_lr=0.0):
    start = steps - cooldown_steps

    def adjust(step):
        if step < warmup_length:
            return _set(optimizer, _warm(base_lr, warmup_length, step))
        if step < start:
            return _set(optimizer, base_lr)
        decay = (1 - (step - start) / (steps - start)) ** cooldown_power
        return _set(optimizer, decay * (base_lr - cooldown_end_lr) + cooldown_end_lr)
    return adjust
