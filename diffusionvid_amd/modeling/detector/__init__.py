from .detectors import build_detection_model

__all__ = ["build_detection_model"]
