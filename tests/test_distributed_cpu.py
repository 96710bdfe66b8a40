"""CPU, world_size 2 over gloo: the bucketed gradient all-reduce really averages (the failure mode of the reference,
SURVEY.md D3), N ranks x batch B == 1 process x batch N*B, the never-reached tensors do not hang anything, and
both ranks hold identical parameters after the step."""
import os
import socket
import sys
from pathlib import Path
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(cfg):
    from clipself_amd.init import seeded_visual_state
    from clipself_amd.open_clip.model import CustomCLIP
    from oracle.ops_ref import RefOps
    student, teacher = CustomCLIP(cfg, ops=RefOps(), trainable=True), CustomCLIP(cfg, ops=RefOps(), trainable=False)
    return student, teacher, seeded_visual_state


def _args(distributed):
    return SimpleNamespace(device="cpu", precision="amp", distributed=distributed, skip_scheduler=True, grad_clip_norm=None,
                           multiscale=False, extract_type="v2", cosine_weight=1.0)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from clipself_amd.config import tiny_cfg
    from clipself_amd.init import synthetic_batch
    from clipself_amd.training.clipself import CLIPSelf
    from clipself_amd.training.distributed import FrozenDataParallel, StudentDataParallel
    from clipself_amd.training.optim import FlatAdamW
    from clipself_amd.training.train import train_step
    torch.set_num_threads(2)
    cfg = tiny_cfg()
    student, teacher, seeded = _build(cfg)
    # deliberately different initial weights per rank: the wrapper must broadcast rank 0's
    student.visual.engine.load_state(seeded(cfg, 1 + rank))
    teacher.visual.engine.load_state(seeded(cfg, 1 + rank))
    student.lock_image_tower(unlocked_groups=cfg.layers)
    model, dist_model = StudentDataParallel(student), FrozenDataParallel(teacher)
    opt = FlatAdamW(student, lr=1e-3, weight_decay=0.1, grad_divisor=float(world))
    batch = synthetic_batch(2, 3, cfg.image_size, cfg.image_size, seed=40 + rank)
    train_step(model, CLIPSelf(), batch, opt, None, 0, dist_model, _args(True))
    eng = student.visual.engine
    torch.save({"grad": eng.grad.clone(), "master": eng.master.clone()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process_on_the_union_batch(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(2))
    assert torch.equal(r0["master"], r1["master"]), "ranks diverged after the step"
    assert torch.equal(r0["grad"], r1["grad"]), "all-reduced gradients differ between ranks"

    from clipself_amd.config import tiny_cfg
    from clipself_amd.init import synthetic_batch
    from clipself_amd.training.clipself import CLIPSelf
    from clipself_amd.training.optim import FlatAdamW
    from clipself_amd.training.train import train_step
    cfg = tiny_cfg()
    student, teacher, seeded = _build(cfg)
    student.visual.engine.load_state(seeded(cfg, 1))
    teacher.visual.engine.load_state(seeded(cfg, 1))
    student.lock_image_tower(unlocked_groups=cfg.layers)
    parts = [synthetic_batch(2, 3, cfg.image_size, cfg.image_size, seed=40 + r) for r in range(2)]
    union = tuple(torch.cat([p[i] for p in parts]) for i in range(3))
    opt = FlatAdamW(student, lr=1e-3, weight_decay=0.1)
    train_step(student, CLIPSelf(), union, opt, None, 0, teacher, _args(False))
    eng = student.visual.engine
    g_single, g_dist = eng.grad, r0["grad"] / world          # SUM on the wire, 1/world applied inside AdamW
    rel = float((g_single - g_dist).norm() / g_single.norm())
    assert rel < 1e-5, rel
    relp = float((eng.master - r0["master"]).norm() / eng.master.norm())
    assert relp < 1e-5, relp
