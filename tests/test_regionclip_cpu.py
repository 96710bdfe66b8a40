"""RegionCLIP method: call contract, federated BCE loss and gradients against a golden captured from the
reference's own RegionCLIP.__call__ (tests/golden/tiny_regionclip.npz).  CPU: engine + per-kernel references."""
from types import SimpleNamespace

import numpy as np
import torch

from clipself_amd.config import tiny_cfg
from clipself_amd.init import seeded_visual_state, synthetic_batch
from clipself_amd.open_clip.model import CustomCLIP
from clipself_amd.training.region_clip import RegionCLIP, get_fed_loss_inds
from oracle.ops_ref import RefOps


def regionclip_inputs(cfg, n_nouns=150, batch=5, boxes=24, seed=31):
    g = np.random.Generator(np.random.PCG64(seed))
    images, nb, _ = synthetic_batch(batch, boxes, cfg.image_size, cfg.image_size, seed=seed)
    labels = torch.from_numpy(g.permutation(n_nouns)[: batch * boxes].astype(np.float32)).reshape(batch, boxes, 1)
    bx = torch.cat([nb[..., :4], labels, nb[..., 4:5]], dim=-1)
    bx[0, 3, -1] = 0.0
    nouns = torch.from_numpy(g.standard_normal((n_nouns, cfg.embed_dim)).astype(np.float32))
    return images, bx, nouns


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def run_regionclip(ops_factory, device, golden_dir):
    g = np.load(golden_dir / "tiny_regionclip.npz")
    cfg = tiny_cfg()
    student = CustomCLIP(cfg, ops=ops_factory(), trainable=True)
    student.visual.engine.load_state(seeded_visual_state(cfg, 4))
    student.lock_image_tower(unlocked_groups=cfg.layers)
    student.train()
    images, bx, nouns = regionclip_inputs(cfg)
    method = RegionCLIP(SimpleNamespace(), noun_embeddings=nouns)
    args = SimpleNamespace(extract_type="v2", contrast_weight=1.0)
    losses, bs, temp = method((images, bx), student, None, None, device, None, False, args)
    assert set(losses) == {"loss_contrast"} and bs == int(g["bs"])
    assert abs(float(temp) - float(g["temp"])) < 1e-4
    total = sum(losses.values())
    assert abs(float(total.detach()) - float(g["loss"])) / float(g["loss"]) < 2e-3, (float(total.detach()), float(g["loss"]))
    total.backward()
    for k in g.files:
        if k.startswith("grad/"):
            r = rel(dict(student.named_parameters())[k[5:]].grad, g[k])
            assert r < 6e-2, (k, r)


def test_regionclip_matches_reference_golden(golden_dir):
    run_regionclip(RefOps, "cpu", golden_dir)


def test_fed_loss_inds_semantics():
    torch.manual_seed(0)
    labels = torch.tensor([5, 5, 9, 200])
    got = get_fed_loss_inds(labels, 100, 4764)
    assert len(got) == 100 and len(torch.unique(got)) == 100
    assert got[:3].tolist() == [5, 9, 200]
    many = torch.arange(130)
    assert torch.equal(get_fed_loss_inds(many, 100, 4764), many)


def test_oracle_regionclip_loss_is_pinned_on_the_reference_golden(golden_dir):
    """oracle/eva_ref.regionclip_loss (the checker of the full-size `-m gpu` RegionCLIP test) against the loss and gradients captured
    from the reference's own RegionCLIP.__call__."""
    from oracle import eva_ref
    g = np.load(golden_dir / "tiny_regionclip.npz")
    cfg = tiny_cfg()
    names = [k[5:] for k in g.files if k.startswith("grad/")]
    sd = {k: v.clone().requires_grad_(k in names) for k, v in seeded_visual_state(cfg, 4).items()}
    images, bx, nouns = regionclip_inputs(cfg)
    loss = eva_ref.regionclip_loss(sd, cfg, images, bx, nouns)
    assert abs(float(loss.detach()) - float(g["loss"])) / float(g["loss"]) < 2e-6, (float(loss.detach()), float(g["loss"]))
    loss.backward()
    for n in names:
        assert rel(sd[n].grad, g["grad/" + n]) < 2e-5, n
