"""Oracle precision policy (TEST INFRASTRUCTURE ONLY).

"fp32" (default): the plain restatement of the reference in fp32.

"fp16": the same algorithm under the storage policy of the MI355X path -- the one apex O1 gives the reference
(convolutions / linear layers / matmuls take fp16 operands, LayerNorm / softmax / box math
stay fp32): every conv / linear WEIGHT is rounded to fp16 (FrozenBN folded in first, as the runtime's repack does),
every tensor the HIP path keeps in fp16 between kernels (backbone activations, RoI tiles, q/k/v, attention
probabilities and outputs, dynamic parameters, DynamicConv intermediates, FFN hidden layer, modulated tower inputs)
is rounded to fp16 where that path stores it, and all accumulation stays fp32.  Products of two fp16 numbers are
exact in fp32, so against this oracle the HIP kernels differ only by fp32 summation order (~1e-6 relative) and by the
rare last-bit flip that such a difference causes at the next fp16 store: what is left of an end-to-end comparison is
kernel error, not precision policy.
"""
import contextlib

import torch

_POLICY = "fp32"
_ONLY = None          # fp16 policy restricted to the stages whose tag starts with one of these prefixes (None: everywhere)
_STAGE = ""           # tag of the stage the oracle is in (backbone_r101 / head set it: "backbone", "head.<series index>.<part>")


def policy():
    return _POLICY


def _active():
    return _POLICY == "fp16" and (_ONLY is None or any(_STAGE.startswith(p) for p in _ONLY))


def is_fp16():
    return _active()


@contextlib.contextmanager
def use(name, only=None):
    """`only`: stage-tag prefixes -- the storage policy applies inside those stages and nowhere else (tools/diag_logit_error_stages.py:
    which stage's fp16 stores account for the path's logit differences)"""
    global _POLICY, _ONLY
    assert name in ("fp32", "fp16")
    old, _POLICY, old_only, _ONLY = _POLICY, name, _ONLY, (tuple(only) if only is not None else None)
    try:
        yield
    finally:
        _POLICY, _ONLY = old, old_only


@contextlib.contextmanager
def stage(tag):
    global _STAGE
    old, _STAGE = _STAGE, tag
    try:
        yield
    finally:
        _STAGE = old


def r16(t):
    """round to fp16 storage and back (identity under the fp32 policy, and outside the selected stages)"""
    if not _active():
        return t
    return t.to(torch.float16).to(torch.float32)


w16 = r16       # weights
a16 = r16       # stored activations
