"""Import the reference (TJUMMG/OCPG @ /root/reference) on CPU in the BUILD CONTAINER ONLY.

Used solely by ``make_fixtures.py`` to generate the golden vectors committed next to this file.
Nothing here runs on the GPU box (``/root/reference`` does not exist there) and nothing here is
imported by the product or by the pytest suites.

What is substituted, and why:
  * torchvision / timm / skimage / pycocotools / ftfy are not installed in this image: tiny
    stand-in modules provide the handful of names the reference imports.  The only arithmetic
    among them is ``torchvision.models.resnet50/101`` + ``IntermediateLayerGetter`` -> our oracle
    restatement ``oracle/resnet.py`` (parity for the ResNet body is therefore *unpinned*), and
    ``box_area`` (4 flops).
  * ``MultiScaleDeformableAttention`` (the CUDA extension, cannot be built without CUDA): the
    reference's own pure-PyTorch ``ms_deform_attn_core_pytorch`` (ms_deform_attn_func.py:41-61) is
    plugged into ``MSDeformAttnFunction.apply`` -- the very function the reference's test.py uses as
    ground truth for its kernel.
  * ``models.ocpg.TextEncoder`` needs ``checkpoints/roberta-base`` (absent): replaced by a stand-in
    that returns caller-provided text features (random, stored in the fixture).
"""
import importlib.machinery
import os
import sys
import types

REF = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    """Register stand-ins and put the reference on sys.path. Idempotent."""
    if getattr(install, "_done", False):
        return
    if not os.path.isdir(REF):
        raise RuntimeError("reference checkout not present; fixtures can only be regenerated in the build container")
    sys.dont_write_bytecode = True
    import torch
    import torch.nn.functional as F
    from torch import nn
    import transformers  # noqa: F401  (must be imported before the stubs below)
    from transformers import RobertaModel, RobertaTokenizerFast  # noqa: F401

    repo = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from oracle import resnet as oresnet

    def box_area(b):
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    tv = _mod("torchvision", __version__="0.15.0")
    tv.ops = _mod("torchvision.ops")
    tv.ops.boxes = _mod("torchvision.ops.boxes", box_area=box_area)
    tv.ops.misc = _mod("torchvision.ops.misc", interpolate=F.interpolate)
    tv.models = _mod("torchvision.models", resnet50=oresnet.resnet50, resnet101=oresnet.resnet101)
    tv.models._utils = _mod("torchvision.models._utils", IntermediateLayerGetter=oresnet.IntermediateLayerGetter)
    tv.utils = _mod("torchvision.utils", save_image=lambda *a, **k: None)
    tv.transforms = _mod("torchvision.transforms")

    class DropPath(nn.Module):  # stochastic depth: identity in the deterministic fixtures (drop prob forced to 0)
        def __init__(self, drop_prob=None):
            super().__init__()

        def forward(self, x):
            return x

    timm = _mod("timm")
    timm.models = _mod("timm.models")
    timm.models.layers = _mod("timm.models.layers", DropPath=DropPath, trunc_normal_=nn.init.trunc_normal_,
                              to_2tuple=lambda x: (x, x))
    sk = _mod("skimage")
    sk.color = _mod("skimage.color")
    pc = _mod("pycocotools")
    pc.mask = _mod("pycocotools.mask")
    _mod("ftfy", fix_text=lambda s: s)
    _mod("MultiScaleDeformableAttention")
    if REF not in sys.path:
        sys.path.insert(0, REF)

    # bypass models/__init__.py's eager import so sub-modules can be patched one by one
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF, "models")]
    pkg.__spec__ = importlib.machinery.ModuleSpec("models", None, is_package=True)
    sys.modules["models"] = pkg

    import models.ops.functions.ms_deform_attn_func as fn
    import models.ops.modules.ms_deform_attn as mda

    class _CoreShim:
        """MSDeformAttnFunction.apply -> the reference's own pure-PyTorch core (autograd gives backward)."""
        @staticmethod
        def apply(value, shapes, level_start, loc, attn, im2col_step):
            return fn.ms_deform_attn_core_pytorch(value, shapes, loc, attn)

    mda.MSDeformAttnFunction = _CoreShim
    fn.MSDeformAttnFunction = _CoreShim
    install._done = True


class StandInTextEncoder:
    """Factory for the TextEncoder stand-in: returns the features it was primed with."""

    @staticmethod
    def make(nn, torch):
        class TextEncoder(nn.Module):
            def __init__(self, args):
                super().__init__()
                self.feat_dim = 768
                self.primed = None

            def forward(self, texts, device):
                f, s, m = self.primed
                return f.to(device), s.to(device), m.to(device)
        return TextEncoder


def build_reference_model(args, primed_text):
    """models.ocpg.build(args) with the TextEncoder stand-in primed with (features, sentence, pad_mask)."""
    install()
    import torch
    from torch import nn
    import models.ocpg as ocpg
    ocpg.TextEncoder = StandInTextEncoder.make(nn, torch)
    model, criterion, post = ocpg.build(args)
    model.text_encoder.primed = primed_text
    return model, criterion, post


def reference_args(**over):
    install()
    import opts
    args = opts.get_args_parser().parse_args([])
    args.masks = True
    args.binary = True
    args.with_box_refine = True
    args.freeze_text_encoder = True
    args.device = "cpu"
    args.dataset_file = "ytvos"
    for k, v in over.items():
        setattr(args, k, v)
    return args
