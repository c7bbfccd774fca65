"""Oracle: Swin-Transformer + FPN backbone (CPU fp32), functional.

Follows mega_core/modeling/backbone/swintransformer.py: window_partition/reverse :68-96,
WindowAttention.forward :135-176 (relative-position bias :158-161, shift mask :163-167),
SwinTransformerBlock.forward :216-276 (pad to window multiples :236-239, cyclic shift :243-263),
PatchMerging.forward :296-321, BasicLayer.forward :383-419 (shift mask construction :387-406),
PatchEmbed.forward :441-458, SwinTransformer.forward :626-648 (per-output LayerNorm :640-646),
size2config :655-712 and the detectron2 FPN wrapper :735-751 (restated in backbone_r101.fpn).
Parameter names: `backbone.bottom_up.{patch_embed,layers.i.blocks.j,layers.i.downsample,norm{i}}.*`.
"""
import math

import torch
import torch.nn.functional as F

from .backbone_r101 import fpn

SWIN_B = dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window=7)


def _ln(x, sd, name):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5)


def relative_position_index(ws):
    """[ws*ws, ws*ws] index into the (2ws-1)^2 bias table (swintransformer.py:122-131)."""
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def shift_attn_mask(H, W, ws, shift):
    """[nW, ws*ws, ws*ws] with 0 / -100 (swintransformer.py:387-406)."""
    Hp, Wp = math.ceil(H / ws) * ws, math.ceil(W / ws) * ws
    img = torch.zeros((Hp, Wp))
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[hs, wsl] = cnt
            cnt += 1
    mw = img.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    d = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.where(d != 0, torch.full_like(d, -100.0), torch.zeros_like(d))


def window_attention(sd, pfx, xw, num_heads, ws, mask):
    """xw [nW*B, ws*ws, C] -> same (swintransformer.py:135-176)."""
    B_, N, C = xw.shape
    hd = C // num_heads
    qkv = F.linear(xw, sd[pfx + ".qkv.weight"], sd[pfx + ".qkv.bias"]).reshape(B_, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    bias = sd[pfx + ".relative_position_bias_table"][relative_position_index(ws).view(-1)].view(N, N, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.view(B_ // nW, nW, num_heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, num_heads, N, N)
    attn = torch.softmax(attn, dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    return F.linear(out, sd[pfx + ".proj.weight"], sd[pfx + ".proj.bias"])


def swin_block(sd, pfx, x, H, W, num_heads, ws, shift, mask):
    """x [B, H*W, C] -> [B, H*W, C] (swintransformer.py:216-276)."""
    B, L, C = x.shape
    shortcut = x
    y = _ln(x, sd, pfx + ".norm1").view(B, H, W, C)
    pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
    y = F.pad(y, (0, 0, 0, pad_r, 0, pad_b))
    Hp, Wp = H + pad_b, W + pad_r
    if shift > 0:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    yw = y.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    aw = window_attention(sd, pfx + ".attn", yw, num_heads, ws, mask if shift > 0 else None)
    y = aw.view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    if shift > 0:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    y = y[:, :H, :W, :].reshape(B, H * W, C)
    x = shortcut + y
    h = F.gelu(F.linear(_ln(x, sd, pfx + ".norm2"), sd[pfx + ".mlp.fc1.weight"], sd[pfx + ".mlp.fc1.bias"]))
    return x + F.linear(h, sd[pfx + ".mlp.fc2.weight"], sd[pfx + ".mlp.fc2.bias"])


def patch_merging(sd, pfx, x, H, W):
    """swintransformer.py:296-321."""
    B, L, C = x.shape
    x = x.view(B, H, W, C)
    if H % 2 or W % 2:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    x = x.view(B, -1, 4 * C)
    return F.linear(_ln(x, sd, pfx + ".norm"), sd[pfx + ".reduction.weight"])


def swin_body(images_norm, sd, pfx="backbone.bottom_up.", embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32),
              window=7, out_indices=(1, 2, 3), patch=4):
    """images_norm NCHW -> {swin1, swin2, swin3} NCHW (swintransformer.py:626-648)."""
    x = images_norm
    _, _, H, W = x.shape
    if W % patch:
        x = F.pad(x, (0, patch - W % patch))
    if H % patch:
        x = F.pad(x, (0, 0, 0, patch - H % patch))
    x = F.conv2d(x, sd[pfx + "patch_embed.proj.weight"], sd[pfx + "patch_embed.proj.bias"], stride=patch)
    Wh, Ww = x.shape[2], x.shape[3]
    x = _ln(x.flatten(2).transpose(1, 2), sd, pfx + "patch_embed.norm")
    outs = {}
    for i, depth in enumerate(depths):
        C = embed_dim * 2 ** i
        shift = window // 2
        mask = shift_attn_mask(Wh, Ww, window, shift)
        for j in range(depth):
            x = swin_block(sd, f"{pfx}layers.{i}.blocks.{j}", x, Wh, Ww, num_heads[i], window, 0 if j % 2 == 0 else shift, mask)
        if i in out_indices:
            o = _ln(x, sd, f"{pfx}norm{i}")
            outs[f"swin{i}"] = o.view(-1, Wh, Ww, C).permute(0, 3, 1, 2).contiguous()
        if i < len(depths) - 1:
            x = patch_merging(sd, f"{pfx}layers.{i}.downsample", x, Wh, Ww)
            Wh, Ww = (Wh + 1) // 2, (Ww + 1) // 2
    return outs


def backbone_swin_fpn(images_norm, sd, pfx="backbone.", **kw):
    """Swin body + detectron2 FPN over swin1..3 -> {p3, p4, p5, p6}."""
    body = swin_body(images_norm, sd, pfx + "bottom_up.", **kw)
    feats = {"res3": body["swin1"], "res4": body["swin2"], "res5": body["swin3"]}
    return fpn(feats, sd, pfx)
