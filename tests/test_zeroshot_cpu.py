"""CPU: zero-shot region classification (clipself_amd/training/zero_shot.py) against a golden captured from the reference's own
zero_shot.run / macc_with_is_thing on the tiny tower (tests/golden/tiny_zeroshot.npz, oracle/gen_golden.py --zeroshot-only)."""
import json
from types import SimpleNamespace

import numpy as np
import torch

from clipself_amd.config import tiny_cfg
from clipself_amd.init import seeded_visual_state
from clipself_amd.open_clip.model import CustomCLIP
from clipself_amd.training.data import SyntheticPanopticVal, _ValLoader
from clipself_amd.training.zero_shot import macc_with_is_thing, run, zero_shot_eval
from oracle.ops_ref import RefOps


def run_zeroshot(ops_factory, device, golden_dir, log=None):
    g = np.load(golden_dir / "tiny_zeroshot.npz")
    rec = json.loads(str(g["recipe"]))
    cfg = tiny_cfg()
    model = CustomCLIP(cfg, ops=ops_factory(), trainable=False)
    model.visual.engine.load_state(seeded_visual_state(cfg, rec["seed_w"]))
    model.eval()
    val = SyntheticPanopticVal(rec["steps"], rec["batch"], rec["boxes"], cfg.image_size, cfg.image_size, cfg.image_size // cfg.patch_size,
                               cfg.embed_dim, num_classes=rec["num_classes"], seed=rec["seed"])
    args = SimpleNamespace(device=device, precision="fp32", distributed=False, horovod=False, extract_type="v2", image_ave_pool=False,
                           zeroshot_frequency=1, epochs=1, rank=0)
    r = run(model, _ValLoader(val), args)
    r = {k: v.cpu() for k, v in r.items()}
    # bookkeeping is exact
    assert torch.equal(r["label"], torch.as_tensor(g["label"])) and torch.equal(r["thing"], torch.as_tensor(g["thing"]))
    assert torch.allclose(r["size"], torch.as_tensor(g["size"]))
    # similarities of the true class: bf16-operand features against the reference's fp32 ones
    worst = 0.0
    for key in ("rois", "crops", "maskpool"):
        want = torch.as_tensor(g["sim_" + key])
        err = float((r["sim_" + key] - want).abs().max())
        worst = max(worst, err)
        assert err < 2e-2, (key, err)                         # cosine similarities in [-1, 1]
        agree = float((r["hit_" + key] == torch.as_tensor(g["hit_" + key])).float().mean())
        assert agree > 0.97, (key, agree)                     # top-5 membership may flip only on near ties
    if log:
        log(f"zero-shot vs reference golden: worst |d cos| = {worst:.2e}")
    # end to end through zero_shot_eval: metric names as the reference reports them
    data = {"val": SimpleNamespace(dataloader=_ValLoader(val))}
    m = zero_shot_eval(model, data, 1, args)
    assert list(m) == [str(n) for n in g["metric_names"]]
    return m


def test_macc_is_the_reference_metric(golden_dir):
    g = np.load(golden_dir / "tiny_zeroshot.npz")
    got = {}
    for key in ("rois", "crops", "maskpool"):
        got.update(macc_with_is_thing(torch.as_tensor(g["hit_" + key]), torch.as_tensor(g["thing"]), torch.as_tensor(g["label"]), key))
    assert list(got) == [str(n) for n in g["metric_names"]]
    assert np.array_equal(np.array([got[k] for k in got], np.float64), g["metric_values"])


def test_zero_shot_run_matches_reference(golden_dir):
    run_zeroshot(RefOps, "cpu", golden_dir)


def test_zero_shot_run_with_the_openai_family_and_extract_type_v1():
    """`--extract-type v1` on the OpenAI-CLIP family (zero_shot.py:73-76 of the reference): box features through the extra query tokens,
    mask features through encode_masks(mask_attn=True); end to end through run() on the synthetic panoptic loader."""
    from clipself_amd.config import tiny_openai_cfg
    from clipself_amd.open_clip import CLIP
    cfg = tiny_openai_cfg()
    model = CLIP(cfg, ops=RefOps(), trainable=False)
    model.visual.engine.load_state(seeded_visual_state(cfg, 3))
    model.eval()
    g = cfg.image_size // cfg.patch_size
    out = {}
    for et in ("v1", "v2"):
        val = SyntheticPanopticVal(2, 2, 3, cfg.image_size, cfg.image_size, g, cfg.embed_dim, num_classes=7, seed=11)
        args = SimpleNamespace(device="cpu", precision="fp32", distributed=False, horovod=False, extract_type=et, image_ave_pool=False,
                               zeroshot_frequency=1, epochs=1, rank=0)
        out[et] = {k: v.cpu() for k, v in run(model, _ValLoader(val), args).items()}
    for key in ("rois", "maskpool"):
        a, b = out["v1"]["sim_" + key], out["v2"]["sim_" + key]
        assert torch.isfinite(a).all() and a.shape == b.shape
        assert float((a - b).abs().max()) > 1e-4, key       # a different pooling, not the dense-map one under another name
    assert torch.equal(out["v1"]["sim_crops"], out["v2"]["sim_crops"])      # crop features do not depend on the extract type
