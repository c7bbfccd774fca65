"""Oracle: global-memory pruning by greedy farthest-point sampling (CPU).

Follows mega_core/modeling/detector/diffusion_det.py:841-896 (update_erase_memory,
select_farthest_k_greedy_cuda) and mega_core/csrc/cuda/fps.cu:11-15 (block size),
:25-142 (kernel).  `getGreedyPerm` (mega_core/modeling/roi_heads/box_head/
roi_box_feature_extractors.py:538-570) is the reference's own CPU statement of the
same greedy rule and is what the golden vectors pin this file against.

Tie rule.  fps.cu resolves equal `temp` values by its thread mapping: thread `tid`
scans k = tid, tid+bs, ... keeping the FIRST maximum (strict `>`, :66-67), then a
shared-memory tree reduction keeps the LOWER slot on ties (`v2 > v1 ? i2 : i1`, :17-22)
with strides bs/2 ... 1, i.e. the surviving thread is the one whose bit-reversed tid is
smallest.  `fps_kernel_order` reproduces that total order so results are bit-identical
to the CUDA kernel on any input, including exact ties (duplicate features).
"""
import numpy as np
import torch


def opt_n_threads(work_size, total_threads=1024):
    """fps.cu:11-15."""
    pow_2 = int(np.log(float(work_size)) / np.log(2.0))
    return max(min(1 << pow_2, total_threads), 1)


def _bitrev(x, bits):
    r = np.zeros_like(x)
    for b in range(bits):
        r |= ((x >> b) & 1) << (bits - 1 - b)
    return r


def fps_tie_priority(n, bs):
    """priority[k]: smaller wins among equal values (see module docstring)."""
    k = np.arange(n, dtype=np.int64)
    bits = int(np.log2(bs)) if bs > 1 else 0
    return _bitrev(k % bs, bits) * (n // bs + 2) + k // bs


def fps_kernel_order(D, m, bs=None):
    """furthest_point_sampling_kernel for b=1.  D [n,n] float32 -> int32 [m]."""
    D = np.asarray(D, dtype=np.float32)
    n = D.shape[0]
    if bs is None:
        bs = opt_n_threads(n)
    prio = fps_tie_priority(n, bs)
    temp = np.full(n, 1e10, dtype=np.float32)
    idx = np.zeros(m, dtype=np.int32)
    old = 0
    for j in range(1, m):
        temp = np.minimum(D[old], temp)
        best = temp.max()
        cand = np.nonzero(temp == best)[0]
        old = int(cand[np.argmin(prio[cand])]) if best > -1 else 0
        idx[j] = old
    return idx


def get_greedy_perm(D, N, start=0):
    """roi_box_feature_extractors.py:538-570 (argmax = first maximum)."""
    perm = torch.zeros(N, dtype=torch.int64)
    perm[0] = start
    ds = D[start, :]
    for i in range(1, N):
        idx = torch.argmax(ds)
        perm[i] = idx
        ds = torch.min(ds, D[idx, :])
    return perm


def select_farthest_k_greedy(merged_feat, k):
    """diffusion_det.py:869-896."""
    distance = torch.cdist(merged_feat, merged_feat, p=2.0)
    idx = fps_kernel_order(distance.numpy(), k)
    return torch.from_numpy(idx.astype(np.int64))


def update_erase_memory(feats_new, feats_mem, target_size):
    """diffusion_det.py:841-867 (greedy branch, no rois)."""
    merged = [f for f in (feats_mem, feats_new) if f is not None]
    merged_feat = torch.cat(merged, dim=0)
    if len(merged_feat) <= target_size:
        return merged_feat, torch.arange(len(merged_feat))
    idx = select_farthest_k_greedy(merged_feat.contiguous(), target_size)
    return merged_feat[idx], idx
