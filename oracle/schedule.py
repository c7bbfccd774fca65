"""Oracle: diffusion schedule tables, time embedding, box<->noise transforms.

Follows mega_core/modeling/detector/diffusion_det.py:50-61 (cosine_beta_schedule),
:223-267 (buffers), :649-677 (predict_noise_from_start / model_predictions) and
mega_core/modeling/roi_heads/box_head/box_head.py:729-741 (SinusoidalPositionEmbeddings),
:216-223 (time_mlp), mega_core/modeling/roi_heads/box_head/loss.py:201-212 (box converters).
"""
import math

import torch
import torch.nn.functional as F


def cosine_beta_schedule(timesteps, s=0.008):
    # diffusion_det.py:50-61
    steps = timesteps + 1
    x = torch.linspace(0, timesteps, steps, dtype=torch.float64)
    alphas_cumprod = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    alphas_cumprod = alphas_cumprod / alphas_cumprod[0]
    betas = 1 - (alphas_cumprod[1:] / alphas_cumprod[:-1])
    return torch.clip(betas, 0, 0.999)


def schedule_buffers(timesteps=1000):
    """diffusion_det.py:223-267 -- the registered buffers (fp32 unless noted)."""
    betas = cosine_beta_schedule(timesteps)              # fp64
    alphas = 1.0 - betas
    alphas_cumprod = torch.cumprod(alphas, dim=0).to(torch.float32)
    alphas_cumprod_prev = F.pad(alphas_cumprod[:-1], (1, 0), value=1.0)
    buf = {
        "betas": betas,
        "alphas_cumprod": alphas_cumprod,
        "alphas_cumprod_prev": alphas_cumprod_prev,
        "sqrt_alphas_cumprod": torch.sqrt(alphas_cumprod),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - alphas_cumprod),
        "log_one_minus_alphas_cumprod": torch.log(1.0 - alphas_cumprod),
        "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / alphas_cumprod),
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / alphas_cumprod - 1),
    }
    posterior_variance = betas * (1.0 - alphas_cumprod_prev) / (1.0 - alphas_cumprod)
    buf["posterior_variance"] = posterior_variance
    buf["posterior_log_variance_clipped"] = torch.log(posterior_variance.clamp(min=1e-20))
    buf["posterior_mean_coef1"] = betas * torch.sqrt(alphas_cumprod_prev) / (1.0 - alphas_cumprod)
    buf["posterior_mean_coef2"] = (1.0 - alphas_cumprod_prev) * torch.sqrt(alphas) / (1.0 - alphas_cumprod)
    return buf


def time_pairs(num_timesteps, sampling_timesteps):
    """diffusion_det.py:536-539."""
    times = torch.linspace(-1, num_timesteps - 1, steps=sampling_timesteps + 1)
    times = list(reversed(times.int().tolist()))
    return list(zip(times[:-1], times[1:]))


def ddim_coefficients(alphas_cumprod, time, time_next, eta=1.0):
    """diffusion_det.py:577-584: returns (sqrt(alpha_next) fp32, c fp32, sigma fp32)."""
    alpha = alphas_cumprod[time].to(torch.float64)
    alpha_next = alphas_cumprod[time_next].to(torch.float64)
    sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
    c = (1 - alpha_next - sigma ** 2).sqrt()
    return alphas_cumprod[time_next].sqrt(), c.to(torch.float32), sigma.to(torch.float32)


def sinusoidal_embedding(time, dim):
    # box_head.py:734-741
    half_dim = dim // 2
    e = math.log(10000) / (half_dim - 1)
    e = torch.exp(torch.arange(half_dim) * -e)
    e = time[:, None] * e[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def time_mlp(sd, pfx, t, d_model):
    """box_head.py:218-223: Sinusoidal -> Linear -> GELU -> Linear.  t: int64 [B]."""
    x = sinusoidal_embedding(t, d_model)
    x = F.linear(x, sd[pfx + "time_mlp.1.weight"], sd[pfx + "time_mlp.1.bias"])
    x = F.gelu(x)
    return F.linear(x, sd[pfx + "time_mlp.3.weight"], sd[pfx + "time_mlp.3.bias"])


def box_cxcywh_to_xyxy(x):
    # loss.py:201-205
    x_c, y_c, w, h = x.unbind(-1)
    return torch.stack([(x_c - 0.5 * w), (y_c - 0.5 * h), (x_c + 0.5 * w), (y_c + 0.5 * h)], dim=-1)


def box_xyxy_to_cxcywh(x):
    # loss.py:208-212
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, (x1 - x0), (y1 - y0)], dim=-1)


def noise_to_boxes(x, images_whwh, scale):
    """diffusion_det.py:657-660: noisy cxcywh in [-scale,scale] -> absolute xyxy."""
    x_boxes = torch.clamp(x, min=-1 * scale, max=scale)
    x_boxes = ((x_boxes / scale) + 1) / 2
    x_boxes = box_cxcywh_to_xyxy(x_boxes)
    return x_boxes * images_whwh[:, None, :]


def boxes_to_x_start(pred_boxes, images_whwh, scale):
    """diffusion_det.py:666-672 (NB: divides by images_whwh[0] for all frames)."""
    x_start = pred_boxes / images_whwh[0, None, :]
    x_start = box_xyxy_to_cxcywh(x_start)
    x_start = (x_start * 2 - 1.0) * scale
    return torch.clamp(x_start, min=-1 * scale, max=scale)


def predict_noise_from_start(buf, x_t, t, x0):
    """diffusion_det.py:649-653."""
    a = buf["sqrt_recip_alphas_cumprod"].gather(-1, t).reshape(-1, 1, 1)
    b = buf["sqrt_recipm1_alphas_cumprod"].gather(-1, t).reshape(-1, 1, 1)
    return (a * x_t - x0) / b
