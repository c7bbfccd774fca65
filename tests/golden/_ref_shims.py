"""Import shims that let the *reference* python modules be imported in the build container.

Used ONLY by tests/golden/make_golden.py (never at test time, never on the GPU box).
The reference needs detectron2 / apex / timm / fvcore / yacs / torchvision / cv2 / ... none
of which are installed.  Everything here is a stand-in written for this repo:
  * attribute-tolerant stub modules for libraries whose code never runs on the paths
    we call;
  * functional stand-ins where reference code does call through: yacs CfgNode (dict with
    attribute access), detectron2 `Boxes`, detectron2 `ROIPooler` (delegates to
    oracle.roi_align -- detectron2 is un-vendored third-party, see oracle/roi_align.py),
    timm `DropPath/to_2tuple/trunc_normal_/Mlp`, `torch._six`, and an `nvidia-smi`
    no-op executable for mega_core/utils/distributed.py:64-76.
"""
import importlib.abc
import importlib.machinery
import os
import stat
import sys
import tempfile
import types

import torch

REFERENCE_ROOT = "/root/reference"

_STUB_PREFIXES = ("apex", "detectron2", "timm", "fvcore", "yacs", "torchvision", "cv2", "albumentations",
                  "pycocotools", "mega_core._C", "tensorboardX", "imgaug", "cityscapesscripts", "lvis",
                  "matplotlib", "skimage", "seaborn")


class _AnyMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()


class _Anything(metaclass=_AnyMeta):
    """Callable/decorator/base-class tolerant dummy."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]           # used as a decorator
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __iter__(self):
        return iter(())

    @staticmethod
    def register(*a, **k):
        return lambda f: f


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        v = type(name, (_Anything,), {})
        setattr(self, name, v)
        return v


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if any(fullname == p or fullname.startswith(p + ".") for p in _STUB_PREFIXES):
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


class CfgNode(dict):
    """Minimal yacs.config.CfgNode stand-in: nested dict with attribute access."""

    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        super().__init__()
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def clone(self):
        import copy
        return copy.deepcopy(self)

    def freeze(self):
        pass

    def defrost(self):
        pass

    def merge_from_file(self, path):
        """yacs merge_from_file: the YAML's values over the existing keys (tuples stay tuples where the default is one)"""
        import yaml

        def merge(dst, src):
            for k, v in src.items():
                if isinstance(v, dict):
                    merge(dst[k], v)
                else:
                    if k not in dst:
                        raise KeyError(k)
                    if isinstance(v, str):          # yacs _decode_cfg_value: strings that are Python literals become those
                        import ast
                        try:
                            v = ast.literal_eval(v)
                        except (ValueError, SyntaxError):
                            pass
                    dst[k] = tuple(v) if isinstance(dst[k], tuple) and isinstance(v, list) else v

        with open(path) as f:
            merge(self, yaml.safe_load(f))

    def merge_from_list(self, lst):
        for k, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = v


class Boxes:
    """detectron2.structures.Boxes stand-in (tensor [n,4] xyxy)."""

    def __init__(self, tensor):
        self.tensor = tensor

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def __len__(self):
        return self.tensor.shape[0]


class ROIPooler(torch.nn.Module):
    """detectron2.modeling.poolers.ROIPooler stand-in -> oracle.roi_align.roi_pooler."""

    def __init__(self, output_size, scales, sampling_ratio, pooler_type, canonical_box_size=224, canonical_level=4):
        super().__init__()
        assert pooler_type == "ROIAlignV2"
        self.output_size, self.scales, self.sampling_ratio = output_size, tuple(scales), sampling_ratio

    def forward(self, x, box_lists):
        from oracle.roi_align import roi_pooler
        boxes = torch.stack([b.tensor for b in box_lists])
        return roi_pooler(list(x), boxes, self.output_size, self.scales, self.sampling_ratio)


class _DropPath(torch.nn.Module):
    def __init__(self, p=0.0):
        super().__init__()

    def forward(self, x):
        return x


class _Mlp(torch.nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=torch.nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = torch.nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = torch.nn.Linear(hidden_features, out_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


def install():
    if getattr(install, "_done", False):
        return
    install._done = True
    # nvidia-smi no-op
    d = tempfile.mkdtemp(prefix="dvid_shim_")
    p = os.path.join(d, "nvidia-smi")
    with open(p, "w") as f:
        f.write("#!/bin/sh\nexit 0\n")
    os.chmod(p, os.stat(p).st_mode | stat.S_IEXEC)
    os.environ["PATH"] = d + os.pathsep + os.environ.get("PATH", "")
    # torch._six
    six = types.ModuleType("torch._six")
    six.PY3 = True
    six.string_classes = (str,)
    six.int_classes = (int,)
    torch._six = six
    sys.modules["torch._six"] = six
    sys.meta_path.insert(0, _StubFinder())
    # functional stand-ins
    import yacs.config
    yacs.config.CfgNode = CfgNode
    import detectron2.structures
    detectron2.structures.Boxes = Boxes
    import detectron2.modeling.poolers
    detectron2.modeling.poolers.ROIPooler = ROIPooler
    import detectron2.modeling
    detectron2.modeling.Backbone = torch.nn.Module
    import detectron2.modeling.backbone
    detectron2.modeling.backbone.Backbone = torch.nn.Module
    import detectron2.modeling.backbone.backbone
    detectron2.modeling.backbone.backbone.Backbone = torch.nn.Module
    import timm.models.layers
    timm.models.layers.DropPath = _DropPath
    timm.models.layers.Mlp = _Mlp
    timm.models.layers.to_2tuple = lambda x: (x, x)
    timm.models.layers.trunc_normal_ = torch.nn.init.trunc_normal_
    import apex
    apex.amp = _StubModule("apex.amp")
    apex.amp.float_function = lambda f: f
    apex.amp.half_function = lambda f: f
    sys.modules["apex.amp"] = apex.amp
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if repo not in sys.path:
        sys.path.insert(0, repo)


def head_cfg(hidden=256, nheads=8, dim_ff=2048, dim_dynamic=64, num_classes=30, num_proposals=300,
             num_heads=3, num_heads_local=1, sample_step=1, infer_batch=8):
    """A cfg object exposing exactly the keys DynamicHead/RCNNHead/DynamicConv read
    (box_head.py:158-237, :440-493, :668-685)."""
    C = CfgNode
    return C({"MODEL": {
        "DiffusionDet": {"NUM_CLASSES": num_classes, "HIDDEN_DIM": hidden, "DIM_FEEDFORWARD": dim_ff, "NHEADS": nheads,
                         "DROPOUT": 0.0, "ACTIVATION": "relu", "NUM_HEADS": num_heads, "NUM_HEADS_LOCAL": num_heads_local,
                         "DEEP_SUPERVISION": True, "USE_FOCAL": True, "USE_FED_LOSS": False, "PRIOR_PROB": 0.01,
                         "NUM_PROPOSALS": num_proposals, "SAMPLE_STEP": sample_step, "NUM_CLS": 1, "NUM_REG": 3,
                         "DIM_DYNAMIC": dim_dynamic, "NUM_DYNAMIC": 2},
        "ROI_HEADS": {"IN_FEATURES": ["p3", "p4", "p5"]},
        "ROI_BOX_HEAD": {"POOLER_RESOLUTION": 7, "POOLER_SAMPLING_RATIO": 2, "POOLER_TYPE": "ROIAlignV2"},
        "VID": {"MEGA": {"ALL_FRAME_INTERVAL": 8, "KEY_FRAME_LOCATION": 0,
                         "GLOBAL": {"ENABLE": True, "RES_STAGE": 1}},
                "ROI_BOX_HEAD": {"ATTENTION": {"ENABLE": False, "STAGE": 1}}}},
        "INPUT": {"INFER_BATCH": infer_batch}})
