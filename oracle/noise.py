"""CPU restatement of the device-side counter-based N(0, 1) draws (csrc/boxes.hip: counter_normal_kernel; C ABI
`dvid_counter_normal`) -- TEST INFRASTRUCTURE, like everything under oracle/: imported by tests/ and bench.py's cpu_baseline leg only.

The reference draws its noise with `torch.randn(shape, device=self.device)` on the device (diffusion_det.py:449, :542, :587,
:595), i.e. from generator state nothing else can reproduce; parity needs injected draws.  Here a draw is a pure function of
(key, element index):
    Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; multipliers 0xD2511F53 /
    0xCD9E8D57, Weyl key increments 0x9E3779B9 / 0xBB67AE85) on counter (q, 0, 0, 0) with the 64-bit key of the image gives four
    32-bit words x0..x3 for elements 4 q .. 4 q + 3;  u = (x + 0.5) / 2^32 in fp64 (exact; never 0 or 1);
    elements 4 q, 4 q + 1 = sqrt(-2 ln u(x0)) * (cos, sin)(2 pi u(x1)), elements 4 q + 2, 4 q + 3 the same of (x2, x3); computed in
    fp64 and rounded once to fp32.
Integer part: bit-exact by construction (pinned against the Random123 known-answer vectors in tests/test_oracle_golden.py).  The
fp64 log / cos / sin of the device library and of numpy agree to an ulp or two of fp64, so after the rounding to fp32 the two sides
produce the same fp32 value except when the fp64 result lies within ~1e-16 relative of a rounding boundary (probability ~4e-9 per
value); the GPU test compares a million values and allows a handful of 1-ulp differences.

Keys follow diffusionvid_amd.utils.synthetic.noise_fn: (video, call frame, kind, step, image) -> one integer; consecutive images
have consecutive keys, so one kernel launch draws a whole batch.
"""
import numpy as np
import torch

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_KINDS = {"box_init": 0, "img": 1, "ddim": 2, "renew": 3}


def philox4x32_10(counter, key):
    """counter: uint32 [..., 4]; key: (k0, k1) python ints -> uint32 [..., 4]"""
    c = [counter[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c[0], _M1 * c[2]
        n0 = (p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)
        c = [n0 & _MASK, p1 & _MASK, n2 & _MASK, p0 & _MASK]
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return np.stack(c, axis=-1).astype(np.uint32)


def counter_normal(key, n):
    """the first n elements of the stream keyed `key` (64-bit) -> float32 [n]"""
    quads = (n + 3) // 4
    ctr = np.zeros((quads, 4), dtype=np.uint32)
    q = np.arange(quads, dtype=np.uint64)
    ctr[:, 0] = (q & _MASK).astype(np.uint32)
    ctr[:, 1] = (q >> np.uint64(32)).astype(np.uint32)
    key = int(key) & 0xFFFFFFFFFFFFFFFF
    x = philox4x32_10(ctr, (key & 0xFFFFFFFF, key >> 32)).astype(np.float64)
    u = (x + 0.5) * (1.0 / 4294967296.0)
    out = np.empty((quads, 4), dtype=np.float64)
    for h in (0, 1):
        r = np.sqrt(-2.0 * np.log(u[:, 2 * h]))
        th = 6.283185307179586476925286766559 * u[:, 2 * h + 1]
        out[:, 2 * h] = r * np.cos(th)
        out[:, 2 * h + 1] = r * np.sin(th)
    return out.reshape(-1)[:n].astype(np.float32)


def draw_key(kind, frame_id, step, image, video=0):
    """the integer diffusionvid_amd.utils.synthetic.noise_fn seeds its generator with; here the Philox key"""
    return 2000 + ((((video * 100003 + frame_id) * 4 + _KINDS[kind]) * 64 + step) * 64 + image)


def noise_fn(kind, frame_id, step, image, shape, video=0):
    """drop-in for synthetic.noise_fn on the oracle side when the GPU path draws on the device (`synthetic.DeviceNoise`)"""
    n = int(np.prod(shape))
    return torch.from_numpy(counter_normal(draw_key(kind, frame_id, step, image, video), n)).reshape(tuple(shape))
