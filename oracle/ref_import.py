"""TEST INFRASTRUCTURE -- runs ONLY in the survey/build container, where the
read-only reference checkout exists at /root/reference.  Nothing under tests/
marked gpu, bench.py or smoke() may import this module (the GPU box has no
/root/reference).

Imports the *real* reference implementation (Python) so that golden vectors can
be captured from it (oracle/gen_golden.py).  The reference needs wheels that
are absent here (torchvision, timm, ftfy; xformers/apex are optional), so
``sys.modules`` stand-ins are installed for exactly the names it imports:

  torchvision.ops.roi_align      -> oracle.roi_align_ref (our restatement of the
                                    torchvision algorithm; see that file's header:
                                    parity is unpinned at that boundary)
  timm.{models.layers,layers}    -> drop_path (identity at p=0), to_2tuple,
                                    trunc_normal_ (= torch.nn.init.trunc_normal_)
  timm.loss.LabelSmoothingCrossEntropy, ftfy.fix_text, torchvision.transforms.*
                                 -> names only (never executed on the hot path)

xformers is absent, so the model configs are patched to ``xattn=False`` and the
reference's own math-attention branch (eva_vit_model.py:221-246) runs.
"""
from __future__ import annotations

import importlib.machinery
import os
import sys
import types

REFERENCE_SRC = "/root/reference/src"


def reference_available() -> bool:
    return os.path.isdir(REFERENCE_SRC)


def _mod(name: str, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_standins():
    import torch
    import transformers  # noqa: F401  (must be imported before torchvision is faked)
    from .roi_align_ref import torchvision_roi_align_standin

    class _Named:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    class _Interp:
        BICUBIC = "bicubic"
        BILINEAR = "bilinear"

    tv = _mod("torchvision")
    tv.ops = _mod("torchvision.ops", roi_align=torchvision_roi_align_standin)
    tv.ops.misc = _mod("torchvision.ops.misc", FrozenBatchNorm2d=torch.nn.BatchNorm2d)
    tnames = ["Normalize", "Compose", "RandomResizedCrop", "ToTensor", "Resize", "CenterCrop",
              "RandomCrop", "RandomHorizontalFlip", "ColorJitter", "Pad", "Lambda", "RandomApply",
              "Grayscale", "RandomGrayscale"]
    tv.transforms = _mod("torchvision.transforms", InterpolationMode=_Interp,
                         **{n: type(n, (_Named,), {}) for n in tnames})
    tv.transforms.functional = _mod("torchvision.transforms.functional",
                                    pad=lambda *a, **k: None, resize=lambda *a, **k: None,
                                    InterpolationMode=_Interp)
    tv.datasets = _mod("torchvision.datasets")

    def drop_path(x, drop_prob=0.0, training=False, scale_by_keep=True):
        assert not drop_prob or not training
        return x

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    layers = dict(drop_path=drop_path, to_2tuple=to_2tuple, trunc_normal_=torch.nn.init.trunc_normal_)
    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.layers", **layers)
    _mod("timm.layers", **layers)
    _mod("timm.loss", LabelSmoothingCrossEntropy=type("LabelSmoothingCrossEntropy", (torch.nn.Module,), {}))
    _mod("ftfy", fix_text=lambda s: s)


def import_reference():
    """Returns the reference's ``open_clip`` package with xattn disabled."""
    if not reference_available():
        raise RuntimeError("/root/reference is not present: the reference import only works in the build container")
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    install_standins()
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import open_clip  # the reference's package
    from open_clip.eva_clip import factory as eva_factory
    for cfg in eva_factory._MODEL_CONFIGS.values():
        cfg.get("vision_cfg", {})["xattn"] = False
        cfg.get("text_cfg", {})["xattn"] = False
    return open_clip
