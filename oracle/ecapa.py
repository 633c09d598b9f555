"""Oracle (TEST INFRASTRUCTURE) — functional restatement of the reference ECAPA-TDNN forward.

Follows reference ``src/models/ECAPA_TDNN.py`` (SpeechBrain-derived) with its SpeechBrain
wrappers ``src/models/layers/cnn.py:91-159,787-805`` (Conv1d: reflect "same" padding) and
``src/models/layers/normalization.py:70-97`` (BatchNorm1d, eval mode).  PINNED against the
imported reference by ``oracle/make_golden.py`` -> ``tests/golden/ecapa_*.npz``.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

KERNEL_SIZES = (5, 3, 3, 3, 1)   # ECAPA_TDNN.py:379
DILATIONS = (1, 2, 3, 4, 1)      # ECAPA_TDNN.py:380
SCALE = 8                        # ECAPA_TDNN.py:382


def to_torch_sd(sd, dtype=torch.float32):
    out = {}
    for k, v in sd.items():
        t = torch.as_tensor(v)
        out[k] = t.to(dtype) if t.is_floating_point() else t
    return out


def bn(x, sd, p, eps=1e-5):
    """nn.BatchNorm1d eval: normalization.py:70-97 -> torch batch_norm with running stats."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, eps)


def conv_same(x, sd, p, dilation=1):
    """cnn.py:105-108,133-159,787-805: reflect-pad d*(k-1)/2 both sides then conv1d(padding=0)."""
    w = sd[p + ".weight"]
    k = w.shape[-1]
    pad = dilation * (k - 1) // 2
    if pad:
        x = F.pad(x, (pad, pad), mode="reflect")
    return F.conv1d(x, w, sd[p + ".bias"], dilation=dilation)


def tdnn(x, sd, p, dilation, act):
    """TDNNBlock.forward ECAPA_TDNN.py:68-69: norm(activation(conv(x)))."""
    return bn(act(conv_same(x, sd, p + ".conv.conv", dilation)), sd, p + ".norm.norm")


def gelu(x):
    return F.gelu(x)              # nn.GELU() default = exact erf form (ECAPA_TDNN.py:377)


def res2net(x, sd, p, dilation):
    """Res2NetBlock.forward ECAPA_TDNN.py:118-129 (inner TDNNBlocks use the default ReLU, :107-112)."""
    ys = []
    y = None
    for i, xi in enumerate(torch.chunk(x, SCALE, dim=1)):
        if i == 0:
            y = xi
        elif i == 1:
            y = tdnn(xi, sd, f"{p}.blocks.{i - 1}", dilation, F.relu)
        else:
            y = tdnn(xi + y, sd, f"{p}.blocks.{i - 1}", dilation, F.relu)
        ys.append(y)
    return torch.cat(ys, dim=1)


def se_block(x, sd, p):
    """SEBlock.forward ECAPA_TDNN.py:164-177 with lengths=None."""
    s = x.mean(dim=2, keepdim=True)
    s = F.relu(conv_same(s, sd, p + ".conv1.conv"))
    s = torch.sigmoid(conv_same(s, sd, p + ".conv2.conv"))
    return s * x


def se_res2net_block(x, sd, p, dilation):
    """SERes2NetBlock.forward ECAPA_TDNN.py:326-336 (no shortcut conv: Cin == Cout)."""
    residual = x
    x = tdnn(x, sd, p + ".tdnn1", 1, gelu)
    x = res2net(x, sd, p + ".res2net_block", dilation)
    x = tdnn(x, sd, p + ".tdnn2", 1, gelu)
    x = se_block(x, sd, p + ".se_block")
    return x + residual


def asp(x, sd, p="asp", eps=1e-12):
    """AttentiveStatisticsPooling.forward ECAPA_TDNN.py:213-260, lengths=None, global_context=True."""
    L = x.shape[-1]

    def stats(x, m):
        mean = (m * x).sum(2)
        std = torch.sqrt((m * (x - mean.unsqueeze(2)).pow(2)).sum(2).clamp(eps))
        return mean, std

    mask = torch.ones(x.shape[0], 1, L, dtype=x.dtype)
    total = mask.sum(dim=2, keepdim=True)
    mean, std = stats(x, mask / total)
    attn = torch.cat([x, mean.unsqueeze(2).repeat(1, 1, L), std.unsqueeze(2).repeat(1, 1, L)], dim=1)
    attn = conv_same(torch.tanh(tdnn(attn, sd, p + ".tdnn", 1, F.relu)), sd, p + ".conv.conv")
    attn = F.softmax(attn, dim=2)
    mean, std = stats(x, attn)
    return torch.cat((mean, std), dim=1).unsqueeze(2)


def ecapa_forward(mel, sd, features="melspectrogram", input_norm=False, stages=None):
    """ECAPA_TDNN.forward ECAPA_TDNN.py:460-502.  mel: (B, n_mels, T) mel power (or any feature when
    features != 'melspectrogram').  Returns (B, nOut) — squeeze() as the reference (B==1 -> (nOut,)).
    ``stages``: optional dict filled with intermediate tensors (B, C, T)."""
    x = mel
    if features.strip() == "melspectrogram":          # :473-476
        x = (x + 1e-6).log()
        x = x - torch.mean(x, dim=-1, keepdim=True)
    if input_norm:                                    # :406-409,477-478
        x = F.instance_norm(x, weight=sd["instance_norm.weight"], bias=sd["instance_norm.bias"], eps=1e-5)
    if stages is not None:
        stages["input"] = x
    xl = []
    x = tdnn(x, sd, "blocks.0", DILATIONS[0], gelu)
    xl.append(x)
    for i in (1, 2, 3):
        x = se_res2net_block(x, sd, f"blocks.{i}", DILATIONS[i])
        xl.append(x)
    if stages is not None:
        for i, t in enumerate(xl):
            stages[f"blocks.{i}"] = t
    x = torch.cat(xl[1:], dim=1)
    x = tdnn(x, sd, "mfa", 1, gelu)
    if stages is not None:
        stages["mfa"] = x
    x = asp(x, sd)
    if stages is not None:
        stages["asp"] = x
    x = bn(x, sd, "asp_bn.norm")
    if stages is not None:
        stages["asp_bn"] = x
    x = conv_same(x, sd, "fc.conv")
    return x.squeeze()
