"""CPU: `clipself_amd.training.main.main(argv)` end to end with the per-kernel reference ops standing in for the HIP library (injected by
monkeypatching the one place the product constructs its ops) and a tiny tower registered as a model config for the duration of the test:
argument parsing -> models -> data -> epochs {train steps, alpha-ensemble, checkpoint, evaluation} -> resume; the evaluation-only mode of
the reference's scripts/test_*.sh on COCO-panoptic style files; training from COCO-style annotation + image files."""
import json
import logging

import pytest
import torch

from clipself_amd import config as cfgmod
from clipself_amd.config import cfg_dict, tiny_cfg
from oracle.ops_ref import RefOps


@pytest.fixture()
def tiny_model_config(monkeypatch):
    c = tiny_cfg()
    blob = {"embed_dim": c.embed_dim,
            "vision_cfg": {"image_size": c.image_size, "layers": c.layers, "width": c.width, "head_width": c.head_width, "patch_size": c.patch_size,
                           "mlp_ratio": c.mlp_ratio, "rope": True, "pt_hw_seq_len": c.pt_hw_seq_len, "intp_freq": True, "naiveswiglu": True,
                           "subln": True},
            "text_cfg": {"context_length": 8, "vocab_size": 64, "width": c.text_width, "heads": c.text_heads, "layers": c.text_layers}}
    path = cfgmod._CFG_DIR / "EVA02-tiny-entry.json"
    path.write_text(json.dumps(blob))
    levels = {name: logging.getLogger(name).level for name in list(logging.root.manager.loggerDict)}       # setup_logging() rewrites them all
    root_level = logging.root.level
    import clipself_amd.hip as hip
    import clipself_amd.open_clip.model as model
    monkeypatch.setattr(model, "_default_ops", lambda: RefOps())
    monkeypatch.setattr(hip, "HipOps", RefOps)
    yield "EVA02-tiny-entry"
    path.unlink()
    for h in list(logging.root.handlers):                      # main() installs console / file handlers bound to this test's streams
        logging.root.removeHandler(h)
        h.close()
    logging.root.setLevel(root_level)
    for name in list(logging.root.manager.loggerDict):
        logging.getLogger(name).setLevel(levels.get(name, logging.NOTSET))


def _run(argv):
    from clipself_amd.training.main import main
    return main([str(a) for a in argv])


def test_train_checkpoint_evaluate_resume(tiny_model_config, tmp_path, caplog):
    base = ["--model", tiny_model_config, "--pretrained", "eva", "--train-data", "synthetic", "--val-data", "synthetic", "--dataset-type", "grid_distill",
            "--batch-size", 2, "--max-boxes", 3, "--det-image-size", 32, "--synthetic-steps", 2, "--epochs", 1, "--lock-image",
            "--lock-image-unlocked-groups", 2, "--alpha", 0.6, "--lr", 1e-3, "--wd", 0.1, "--warmup", 1, "--log-every-n-steps", 1,
            "--logs", tmp_path, "--cache-dir", "none.pt", "--zeroshot-frequency", 1, "--no-teacher-prefetch"]
    assert _run(base + ["--name", "a"]) == 0
    ck = torch.load(tmp_path / "a" / "checkpoints" / "epoch_1.pt", map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "name", "state_dict", "optimizer"} and ck["epoch"] == 1 and ck["name"] == "a"
    assert "visual.blocks.1.mlp.w3.weight" in ck["state_dict"] and "text.token_embedding.weight" in ck["state_dict"]
    results = (tmp_path / "a" / "checkpoints" / "results.json").read_text().strip().splitlines()
    assert len(results) == 2 and "rois.thing.macc1" in json.loads(results[0])                  # before training and on the saved ensemble
    params = (tmp_path / "a" / "params.txt").read_text()
    assert "beta2: 0.999" in params and "lock_image_unlocked_groups: 2" in params
    # the saved weights are alpha * student + (1 - alpha) * teacher: frozen tensors equal the seeded initialisation
    from clipself_amd.init import seeded_visual_state
    sd0 = seeded_visual_state(tiny_cfg(), 0)
    assert torch.allclose(ck["state_dict"]["visual.head.weight"].cpu(), sd0["visual.head.weight"], rtol=1e-6, atol=0)      # 0.6 w + 0.4 w
    assert not torch.allclose(ck["state_dict"]["visual.blocks.1.mlp.w1.weight"].cpu(), sd0["visual.blocks.1.mlp.w1.weight"], rtol=1e-4, atol=0)
    assert _run(base + ["--name", "b", "--resume", tmp_path / "a" / "checkpoints" / "epoch_1.pt", "--epochs", 2]) == 0
    assert (tmp_path / "b" / "checkpoints" / "epoch_2.pt").exists()
    assert _run(base + ["--name", "a"]) == -1                                                   # an existing experiment is refused (main.py:95-100)


def test_evaluation_only_and_training_from_files(tiny_model_config, tmp_path):
    from test_data_cpu import _write_coco, _write_panoptic
    ann, img_root, seg_root, emb = _write_panoptic(tmp_path)
    # scripts/test_eva_vitb16_macc_boxes_masks.sh: --train-data "" --val-data <panoptic json> ... -> evaluate and stop
    argv = ["--model", tiny_model_config, "--pretrained", "eva", "--train-data", "", "--val-data", ann, "--val-image-root", img_root,
            "--val-segm-root", seg_root, "--embed-path", emb, "--det-image-size", 32, "--downsample-factor", 8, "--batch-size", 1, "--logs", tmp_path,
            "--name", "ev", "--cache-dir", "none.pt", "--extract-type", "v2"]
    assert _run(argv) == 0
    res = json.loads((tmp_path / "ev" / "checkpoints" / "results.json").read_text().strip().splitlines()[-1])
    assert {"rois.thing.macc1", "maskpool.stuff.macc5", "crops.thing.macc1"} <= set(res)
    assert not (tmp_path / "ev" / "checkpoints" / "epoch_1.pt").exists()
    # scripts/train_clipself_coco_region_proposals_*.sh shape: annotation file + image directory, proposals with their boxes
    coco_dir = tmp_path / "coco"
    coco_dir.mkdir()
    tr_ann, tr_root = _write_coco(coco_dir)
    argv = ["--model", tiny_model_config, "--pretrained", "eva", "--train-data", tr_ann, "--train-image-root", tr_root, "--dataset-type", "proposals_distill",
            "--val-data", ann, "--val-image-root", img_root, "--val-segm-root", seg_root, "--embed-path", emb, "--downsample-factor", 8,
            "--batch-size", 2, "--det-image-size", 32, "--epochs", 1, "--lock-image", "--lock-image-unlocked-groups", 1, "--lr", 1e-3, "--warmup", 1,
            "--logs", tmp_path, "--name", "tr", "--cache-dir", "none.pt", "--zeroshot-frequency", 1, "--no-teacher-prefetch"]
    assert _run(argv) == 0
    assert (tmp_path / "tr" / "checkpoints" / "epoch_1.pt").exists()
    assert len((tmp_path / "tr" / "checkpoints" / "results.json").read_text().strip().splitlines()) == 2
    # scripts/train_regionclip_coco_*.sh shape: the same files with category labels, RegionCLIP method, no teacher
    blob = json.loads(tr_ann.read_text())
    blob["categories"] = [{"id": 1, "name": "a"}]
    tr_ann.write_text(json.dumps(blob))
    argv = ["--model", tiny_model_config, "--pretrained", "eva", "--train-data", tr_ann, "--train-image-root", tr_root, "--dataset-type", "region_clip",
            "--val-data", "", "--batch-size", 2, "--det-image-size", 32, "--epochs", 1, "--lock-image", "--lock-image-unlocked-groups", 1, "--lr", 1e-3,
            "--warmup", 1, "--logs", tmp_path, "--name", "rc", "--cache-dir", "none.pt", "--zeroshot-frequency", 0, "--alpha", 0.5]
    assert _run(argv) == 0
    assert (tmp_path / "rc" / "checkpoints" / "epoch_1.pt").exists()
    assert cfg_dict(tiny_cfg())["width"] == 128
