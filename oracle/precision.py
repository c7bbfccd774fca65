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


def policy():
    return _POLICY


def is_fp16():
    return _POLICY == "fp16"


@contextlib.contextmanager
def use(name):
    global _POLICY
    assert name in ("fp32", "fp16")
    old, _POLICY = _POLICY, name
    try:
        yield
    finally:
        _POLICY = old


def r16(t):
    """round to fp16 storage and back (identity under the fp32 policy)"""
    if _POLICY != "fp16":
        return t
    return t.to(torch.float16).to(torch.float32)


w16 = r16       # weights
a16 = r16       # stored activations
