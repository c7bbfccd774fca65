"""Meta-architecture registry (mirror of mega_core/modeling/detector/detectors.py:11-22)."""
from .diffusion_det import DiffusionDet

_DETECTION_META_ARCHITECTURES = {"DiffusionDet": DiffusionDet}


def build_detection_model(cfg):
    meta_arch = _DETECTION_META_ARCHITECTURES[cfg.MODEL.META_ARCHITECTURE]
    return meta_arch(cfg)
