from .defaults import _C as cfg, add_diffusiondet_config
from .node import CfgNode


def get_cfg(config_file=None, opts=None, base_file=None):
    """Fresh config in the reference's merge order (tools/test_net.py:77-83):
    BASE_RCNN_{n}gpu.yaml -> add_diffusiondet_config -> model yaml -> CLI opts."""
    c = cfg.clone()
    if base_file:
        c.merge_from_file(base_file)
    add_diffusiondet_config(c)
    if config_file:
        c.merge_from_file(config_file)
    if opts:
        c.merge_from_list(list(opts))
    return c


__all__ = ["cfg", "CfgNode", "add_diffusiondet_config", "get_cfg"]
