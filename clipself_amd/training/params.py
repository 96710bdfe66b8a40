"""Command-line surface of `training.main` -- the same flags, defaults and post-processing as the reference's
src/training/params.py:26-476, declared from a table.  Flags the reference parses but never reads
 are accepted for script compatibility."""
import argparse
import ast

_S, _I, _F = str, int, float
# (flag, type | 'flag', default [, extra kwargs])
_TABLE = [
    ("max-boxes", _I, 20), ("max-masks", _I, 20), ("downsample-factor", _I, 16), ("alpha", _F, 2.0),
    ("grid-noise", "flag", False), ("shift-range", _F, 0.0), ("scale-range", _F, 0.0), ("crop-scale", _F, 1.0),
    ("box-scale", _F, 1.5), ("multiscale", "flag", False), ("pre-transforms", "flag", False), ("max-size", _I, 1024),
    ("embed-dim", _I, 768), ("fix-logit-scale", "flag", False), ("min-size", _I, 8), ("max-split", _I, 6),
    ("extract-type", _S, "v2", dict(choices=["v1", "v2"])), ("cache-dir", _S, "checkpoints"), ("kl-weight", _F, 1.0),
    ("contrast-weight", _F, 1.0), ("train-ratio", _F, 1.0), ("l1-weight", _F, 0.10), ("smooth-weight", _F, 0.0),
    ("cosine-weight", _F, 1.0), ("det-image-size", _I, 1024), ("train-image-size", _I, 1024),
    ("image-ave-pool", "flag", False), ("roi-teacher", "flag", False), ("mask-thr", _F, 0.7),
    ("train-image-root", _S, "data/coco/val2017"), ("train-ceph-root", _S, ""), ("val-image-root", _S, "data/coco/val2017"),
    ("val-segm-root", _S, "data/coco/annotations/panoptic_val2017"), ("train-segm-root", _S, "data/coco/annotations/panoptic_val2017"),
    ("embed-path", _S, "metadata/coco_clip_hand_craft_RN50.npy"), ("train-embed-path", _S, ""), ("del-dist-model", "flag", False),
    ("train-data", _S, ""), ("val-data", _S, "data/coco/annotations/instances_val2017_100.json"),
    ("dataset-type", None, "grid_distill", dict(choices=["proposals_distill", "region_clip", "grid_distill"])),
    ("test-type", None, "coco_panoptic", dict(choices=["coco_panoptic"])), ("logs", _S, "./logs/"), ("log-local", "flag", False),
    ("name", _S, None), ("workers", _I, 1), ("batch-size", _I, 64), ("epochs", _I, 32), ("lr", _F, 1e-5),
    ("beta1", _F, None), ("beta2", _F, None), ("eps", _F, None), ("wd", _F, 0.2), ("warmup", _I, 10000),
    ("use-bn-sync", "flag", False), ("skip-scheduler", "flag", False), ("lr-scheduler", _S, "cosine"),
    ("lr-cooldown-end", _F, 0.0), ("lr-cooldown-power", _F, 1.0), ("save-frequency", _I, 1), ("save-most-recent", "flag", False),
    ("zeroshot-frequency", _I, 2), ("resume", _S, None),
    ("precision", None, "amp", dict(choices=["amp", "amp_bf16", "amp_bfloat16", "bf16", "fp16", "fp32", "amp_fp8", "amp_fp8_dgrad"])),
    ("model", _S, "RN50"), ("pretrained", _S, ""), ("pretrained-image", "flag", False), ("lock-image", "flag", False),
    ("lock-image-unlocked-groups", _I, 3), ("lock-image-freeze-bn-stats", "flag", True),
    ("image-mean", _F, None, dict(nargs="+", metavar="MEAN")), ("image-std", _F, None, dict(nargs="+", metavar="STD")),
    ("grad-checkpointing", "flag", False), ("gather-with-grad", "flag", False), ("force-image-size", _I, None, dict(nargs="+")),
    ("force-quick-gelu", "flag", False), ("force-patch-dropout", _F, None), ("force-custom-text", "flag", False),
    ("torchscript", "flag", False), ("accum-freq", _I, 1), ("dist-url", _S, "env://"), ("dist-backend", _S, "nccl"),
    ("debug", "flag", False), ("copy-codebase", "flag", False), ("horovod", "flag", False), ("ddp-static-graph", "flag", False),
    ("no-set-device-rank", "flag", False), ("seed", _I, 0), ("grad-clip-norm", _F, None), ("log-every-n-steps", _I, 100),
    ("delete-previous-checkpoint", "flag", False),
]
# additions of this build (absent from the reference): synthetic input pipeline for `--train-data synthetic`
_EXTRA = [("synthetic-steps", _I, 100), ("synthetic-image-size", _I, None), ("teacher-chunk", _I, 2048),
          ("no-teacher-prefetch", "flag", False)]      # run the frozen teacher inline instead of one batch ahead on a side stream


class _KeyValue(argparse.Action):
    def __call__(self, parser, namespace, values, option_string=None):
        out = {}
        for item in values:
            k, v = item.split("=")
            try:
                out[k] = ast.literal_eval(v)
            except ValueError:
                out[k] = str(v)
        setattr(namespace, self.dest, out)


def default_adam_params(model_name: str):
    """CLIP-paper Adam defaults keyed on the model name (params.py:5-11): names containing 'vit' get
    beta2=0.98/eps=1e-6; 'EVA02-CLIP-B-16' does not contain 'vit' (SURVEY.md D6)."""
    if "vit" in model_name.lower():
        return dict(lr=5.0e-4, beta1=0.9, beta2=0.98, eps=1.0e-6)
    return dict(lr=5.0e-4, beta1=0.9, beta2=0.999, eps=1.0e-8)


def parse_args(argv):
    parser = argparse.ArgumentParser()
    for row in _TABLE + _EXTRA:
        flag, kind, default = row[:3]
        kw = dict(row[3]) if len(row) > 3 else {}
        if kind == "flag":
            parser.add_argument("--" + flag, action="store_true", default=default)
        elif kind is None:
            parser.add_argument("--" + flag, default=default, **kw)
        else:
            parser.add_argument("--" + flag, type=kind, default=default, **kw)
    parser.add_argument("--aug-cfg", nargs="*", default={}, action=_KeyValue)
    args = parser.parse_args(argv)
    for name, val in default_adam_params(args.model).items():      # only fills values the user left unset
        if getattr(args, name) is None:
            setattr(args, name, val)
    return args
