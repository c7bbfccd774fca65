"""A small yacs-compatible config node (yacs is not installed in the target image).

Supports what the reference's entry point does with its global `cfg`
(tools/test_net.py:77-83): attribute access, `merge_from_file(yaml)`, `merge_from_list([...])`
with literal parsing, `freeze()/defrost()`, `clone()`; unknown keys raise KeyError as in yacs.
"""
import ast
import copy

import yaml


class CfgNode(dict):
    _FROZEN = "__frozen__"

    def __init__(self, init_dict=None):
        super().__init__()
        object.__setattr__(self, CfgNode._FROZEN, False)
        for k, v in (init_dict or {}).items():
            dict.__setitem__(self, k, CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if object.__getattribute__(self, CfgNode._FROZEN):
            raise AttributeError("Attempted to set {} to {}, but CfgNode is immutable".format(name, value))
        self[name] = value

    def _set_frozen(self, flag):
        object.__setattr__(self, CfgNode._FROZEN, flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def is_frozen(self):
        return object.__getattribute__(self, CfgNode._FROZEN)

    def clone(self):
        c = copy.deepcopy(self)
        return c

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        return out

    @staticmethod
    def _coerce(new, old, key):
        if isinstance(old, CfgNode) or old is None or new is None:
            return new
        if isinstance(old, tuple) and isinstance(new, list):
            return tuple(new)
        if isinstance(old, list) and isinstance(new, tuple):
            return list(new)
        if isinstance(old, float) and isinstance(new, int):
            return float(new)
        if type(old) is not type(new) and not (isinstance(old, (list, tuple)) and isinstance(new, (list, tuple))):
            raise ValueError("Type mismatch ({} vs. {}) for config key: {}".format(type(old), type(new), key))
        return new

    def _merge(self, other, path):
        for k, v in other.items():
            full = ".".join(path + [k])
            if k not in self:
                raise KeyError("Non-existent config key: {}".format(full))
            if isinstance(v, dict):
                if not isinstance(self[k], CfgNode):
                    raise ValueError("Type mismatch for config key: {}".format(full))
                self[k]._merge(v, path + [k])
            else:
                if isinstance(v, str):
                    v = CfgNode._parse(v)
                dict.__setitem__(self, k, CfgNode._coerce(v, self[k], full))

    @staticmethod
    def _parse(s):
        try:
            return ast.literal_eval(s)
        except (ValueError, SyntaxError):
            return s

    def merge_from_file(self, path):
        with open(path, "r") as f:
            data = yaml.safe_load(f) or {}
        self._merge(data, [])

    def merge_from_other_cfg(self, other):
        self._merge(other, [])

    def merge_from_list(self, lst):
        if len(lst) % 2:
            raise AssertionError("Override list has odd length: {}".format(lst))
        for full, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = full.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError("Non-existent key: {}".format(full))
                node = node[p]
            if parts[-1] not in node:
                raise KeyError("Non-existent key: {}".format(full))
            if isinstance(v, str):
                v = CfgNode._parse(v)
            dict.__setitem__(node, parts[-1], CfgNode._coerce(v, node[parts[-1]], full))
