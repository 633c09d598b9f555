"""Oracle (TEST INFRASTRUCTURE) — functional restatement of the reference RawNet2 forward
(``front_proc='sinc'``, ``aggregate='asp'``: the variant the fusion models build,
``src/models/Raw_ECAPA_sinc_asp.py:26-28``).

Follows ``src/models/RawNet2_custom.py:161-227`` and ``src/models/RawNet_baseline.py:13-24``
(LayerNorm), ``:62-68`` (AFMS), ``:221-232`` (RawNetBasicBlock), ``:265-361`` (SincConv_fast).
PINNED against the imported reference by ``oracle/make_golden.py`` -> ``tests/golden/rawnet2_*.npz``.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

LAYERS = (1, 1, 1, 2, 1, 2)       # RawNet2_custom.py:231


def bn(x, sd, p, eps=1e-5):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, eps)


def layer_norm(x, sd, eps=1e-6):
    """RawNet_baseline.py:21-24: gamma*(x-mean)/(std_unbiased + eps) + beta over the sample axis."""
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)
    return sd["ln.gamma"] * (x - mean) / (std + eps) + sd["ln.beta"]


def sinc_filters(low_hz_, band_hz_, kernel_size=251, sample_rate=16000, min_low_hz=50, min_band_hz=50):
    """RawNet_baseline.py:313-318 (window_, n_) and :339-357 (band-pass construction).  (F, K)."""
    dt = low_hz_.dtype
    n_lin = torch.linspace(0, (kernel_size / 2) - 1, steps=int(kernel_size / 2))        # float32 in the reference
    window_ = (0.54 - 0.46 * torch.cos(2 * math.pi * n_lin / kernel_size)).to(dt)
    n = (kernel_size - 1) / 2.0
    n_ = (2 * math.pi * torch.arange(-n, 0).view(1, -1) / sample_rate).to(dt)
    low = min_low_hz + torch.abs(low_hz_)
    high = torch.clamp(low + min_band_hz + torch.abs(band_hz_), min_low_hz, sample_rate / 2)
    band = (high - low)[:, 0]
    f_low = torch.matmul(low, n_)
    f_high = torch.matmul(high, n_)
    left = ((torch.sin(f_high) - torch.sin(f_low)) / (n_ / 2)) * window_
    center = 2 * band.view(-1, 1)
    right = torch.flip(left, dims=[1])
    bp = torch.cat([left, center, right], dim=1)
    return bp / (2 * band[:, None])


def afms(x, sd, p):
    """RawNet_baseline.py:62-68."""
    y = F.adaptive_avg_pool1d(x, 1).view(x.size(0), -1)
    y = torch.sigmoid(F.linear(y, sd[p + ".fc.weight"], sd[p + ".fc.bias"])).view(x.size(0), x.size(1), -1)
    return (x + sd[p + ".alpha"]) * y


def basic_block(x, sd, p, downsample):
    """RawNetBasicBlock.forward RawNet_baseline.py:221-232 (pre-activation; identity shortcut
    takes the *pre-BN* x, the conv shortcut takes lrelu(bn1(x)))."""
    out = F.leaky_relu(bn(x, sd, p + ".bn1"), 0.3)
    shortcut = F.conv1d(out, sd[p + ".shortcut.0.weight"]) if (p + ".shortcut.0.weight") in sd else x
    out = F.conv1d(out, sd[p + ".conv1.weight"], padding=1)
    out = F.conv1d(F.leaky_relu(bn(out, sd, p + ".bn2"), 0.3), sd[p + ".conv2.weight"], padding=1)
    out = out + shortcut
    if downsample:
        out = F.max_pool1d(out, 3)
    return afms(out, sd, p + ".afms")


def rawnet2_forward(x, sd, stages=None):
    """RawNet2.forward RawNet2_custom.py:161-227.  x: (B, 32000) waveform -> (B, nOut)."""
    x = layer_norm(x, sd)                                                          # :171
    filt = sinc_filters(sd["first_conv.low_hz_"], sd["first_conv.band_hz_"])       # RawNet_baseline.py:320-357
    if stages is not None:
        stages["sinc_filters"] = filt
    x = F.conv1d(x.unsqueeze(1), filt.unsqueeze(1))                                # :359-361 (valid)
    x = F.max_pool1d(torch.abs(x), 3)                                              # RawNet2_custom.py:174
    x = F.leaky_relu(bn(x, sd, "first_bn"), 0.3)                                   # :175-176
    if stages is not None:
        stages["front"] = x
    for li, nblk in enumerate(LAYERS, start=1):
        for b in range(nblk):
            x = basic_block(x, sd, f"layer{li}.{b}", downsample=(b == nblk - 1))   # :149-159
        if stages is not None:
            stages[f"layer{li}"] = x
    x = F.leaky_relu(bn(x, sd, "bn_before_agg"), 0.3)                              # :215-216
    a = F.conv1d(x, sd["attention.0.weight"], sd["attention.0.bias"])              # :105-111
    a = bn(F.leaky_relu(a, 0.01), sd, "attention.2")
    a = F.conv1d(a, sd["attention.3.weight"], sd["attention.3.bias"])
    w = F.softmax(a, dim=-1)
    m = torch.sum(x * w, dim=-1)                                                   # :218-221
    s = torch.sqrt((torch.sum((x ** 2) * w, dim=-1) - m ** 2).clamp(min=1e-5))
    x = torch.cat([m, s], dim=1)
    x = F.linear(x, sd["fc.weight"], sd["fc.bias"])                                # :224
    return x.squeeze()
