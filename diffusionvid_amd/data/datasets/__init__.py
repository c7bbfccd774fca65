from .vid import VIDFrameList, VIDMEGATestDataset  # noqa: F401
