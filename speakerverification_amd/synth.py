"""Parameter specs and deterministic synthetic weights for the hot-path models.

The reference ships no checkpoints and there is no network, so tests, smoke() and bench.py
run on random-initialised weights of the reference architectures.  This module is the single
source of truth for the *names and shapes* of the reference state dicts

  * ECAPA-TDNN  — reference ``src/models/ECAPA_TDNN.py:339-458`` (231 tensors),
  * RawNet2     — reference ``src/models/RawNet2_custom.py:18-135`` + ``RawNet_baseline.py``
                  (147 tensors, ``front_proc='sinc'``, ``aggregate='asp'``),

and generates values from a ``numpy`` PCG64 stream in state-dict order, so the CPU oracle and the
HIP path see bit-identical weights on any machine.  ``tests/test_oracle_golden.py`` checks the
specs against key/shape lists captured from the imported reference (``tests/golden/*.json``).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

ECAPA_KERNEL_SIZES = (5, 3, 3, 3, 1)      # ECAPA_TDNN.py:379
ECAPA_DILATIONS = (1, 2, 3, 4, 1)         # ECAPA_TDNN.py:380
RES2NET_SCALE = 8                         # ECAPA_TDNN.py:382
SE_CHANNELS = 128                         # ECAPA_TDNN.py:383
ATT_CHANNELS = 128                        # ECAPA_TDNN.py:381

RAWNET2_LAYERS = (1, 1, 1, 2, 1, 2)                    # RawNet2_custom.py:231
RAWNET2_FILTERS = (128, 128, 256, 256, 512, 512)       # RawNet2_custom.py:232
RAWNET2_SINC_K = 251                                   # RawNet2_custom.py:36


def _bn(prefix, c):
    return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,)),
            (prefix + ".running_mean", (c,)), (prefix + ".running_var", (c,)),
            (prefix + ".num_batches_tracked", ())]


def _tdnn(prefix, cin, cout, k):
    return ([(prefix + ".conv.conv.weight", (cout, cin, k)), (prefix + ".conv.conv.bias", (cout,))]
            + _bn(prefix + ".norm.norm", cout))


def ecapa_param_spec(C=1024, n_mels=80, nOut=192, input_norm=False):
    """Ordered (name, shape) list == ``ECAPA_TDNN(...).state_dict()`` of the reference."""
    C3 = 3 * C
    spec = []
    spec += _tdnn("blocks.0", n_mels, C, ECAPA_KERNEL_SIZES[0])
    for i in (1, 2, 3):
        p = f"blocks.{i}"
        spec += _tdnn(p + ".tdnn1", C, C, 1)
        for j in range(RES2NET_SCALE - 1):
            spec += _tdnn(p + f".res2net_block.blocks.{j}", C // RES2NET_SCALE, C // RES2NET_SCALE,
                          ECAPA_KERNEL_SIZES[i])
        spec += _tdnn(p + ".tdnn2", C, C, 1)
        spec += [(p + ".se_block.conv1.conv.weight", (SE_CHANNELS, C, 1)),
                 (p + ".se_block.conv1.conv.bias", (SE_CHANNELS,)),
                 (p + ".se_block.conv2.conv.weight", (C, SE_CHANNELS, 1)),
                 (p + ".se_block.conv2.conv.bias", (C,))]
    if input_norm:  # nn.InstanceNorm1d(affine=True, track_running_stats=False), ECAPA_TDNN.py:406-409;
        # registered after the (still empty) blocks ModuleList, so it follows blocks.* in state_dict order
        spec += [("instance_norm.weight", (n_mels,)), ("instance_norm.bias", (n_mels,))]
    spec += _tdnn("mfa", C3, C3, 1)
    spec += _tdnn("asp.tdnn", 3 * C3, ATT_CHANNELS, 1)
    spec += [("asp.conv.conv.weight", (C3, ATT_CHANNELS, 1)), ("asp.conv.conv.bias", (C3,))]
    spec += _bn("asp_bn.norm", 2 * C3)
    spec += [("fc.conv.weight", (nOut, 2 * C3, 1)), ("fc.conv.bias", (nOut,))]
    return spec


def rawnet2_param_spec(nOut=320, nb_samp=32000, att_dim=128):
    """Ordered (name, shape) list == ``RawNet2_custom.MainModel(front_proc='sinc',
    aggregate='asp').state_dict()`` of the reference."""
    f = RAWNET2_FILTERS
    spec = [("ln.gamma", (nb_samp,)), ("ln.beta", (nb_samp,)),
            ("first_conv.low_hz_", (f[0], 1)), ("first_conv.band_hz_", (f[0], 1))]
    spec += _bn("first_bn", f[0])
    inpl = f[0]
    for li, (nblk, planes) in enumerate(zip(RAWNET2_LAYERS, f), start=1):
        for b in range(nblk):
            p = f"layer{li}.{b}"
            spec += _bn(p + ".bn1", inpl)
            spec += [(p + ".conv1.weight", (planes, inpl, 3))]
            spec += _bn(p + ".bn2", planes)
            spec += [(p + ".conv2.weight", (planes, planes, 3)),
                     (p + ".afms.alpha", (planes, 1)),
                     (p + ".afms.fc.weight", (planes, planes)), (p + ".afms.fc.bias", (planes,))]
            if inpl != planes:
                spec += [(p + ".shortcut.0.weight", (planes, inpl, 1))]
            inpl = planes
    spec += _bn("bn_before_agg", f[5])
    spec += [("attention.0.weight", (att_dim, f[5], 1)), ("attention.0.bias", (att_dim,))]
    spec += _bn("attention.2", att_dim)
    spec += [("attention.3.weight", (f[5], att_dim, 1)), ("attention.3.bias", (f[5],)),
             ("fc.weight", (nOut, 2 * f[5])), ("fc.bias", (nOut,))]
    return spec


def _sinc_init(n_filt, sample_rate=16000, min_low_hz=50, min_band_hz=50):
    # same initial values as RawNet_baseline.py:296-310 (mel-spaced band edges)
    to_mel = lambda hz: 2595 * np.log10(1 + hz / 700)
    to_hz = lambda mel: 700 * (10 ** (mel / 2595) - 1)
    high_hz = sample_rate / 2 - (min_low_hz + min_band_hz)
    hz = to_hz(np.linspace(to_mel(10), to_mel(high_hz), n_filt + 1))
    return hz[:-1].astype(np.float32), np.diff(hz).astype(np.float32)


def synth_state_dict(spec, seed=1):
    """Deterministic non-trivial weights (numpy float32 / int64) in ``spec`` order.

    Conv / linear weights are He-scaled so activations stay O(1) through the stack; BatchNorm
    running statistics and affine terms are randomised so that BN cannot be mistaken for the
    identity; sinc band edges start from the reference initialisation plus jitter.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    for name, shape in spec:
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            sd[name] = np.array(1000, dtype=np.int64)
        elif leaf == "running_var":
            sd[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif leaf == "running_mean":
            sd[name] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif leaf == "gamma":                                  # RawNet2 LayerNorm
            sd[name] = (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif leaf == "beta":
            sd[name] = (0.01 * rng.standard_normal(shape)).astype(np.float32)
        elif leaf == "alpha":                                  # AFMS
            sd[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif leaf in ("low_hz_", "band_hz_"):
            lo, band = _sinc_init(shape[0])
            base = lo if leaf == "low_hz_" else band
            jitter = rng.uniform(-3.0, 3.0, shape[0]).astype(np.float32)
            sd[name] = (base + jitter).reshape(shape).astype(np.float32)
        elif leaf == "weight" and len(shape) == 1:             # BN / InstanceNorm affine weight
            sd[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif leaf == "bias":
            sd[name] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif leaf == "weight":
            fan_in = int(np.prod(shape[1:]))
            std = math.sqrt(2.0 / fan_in)
            # RawNet2's residual stack has no normalisation on the skip path: with full He gain the
            # activations grow to O(500) and the reference's own fp32 arithmetic is only good to ~1e-3
            # (fp32 vs fp64 oracle).  Gain 0.7 keeps outputs at the O(20) the reference shows with its
            # default init and the fp32 noise floor at ~1e-5, so that a 1e-4 parity bar means something.
            if name.startswith("layer") and (".conv" in name or "shortcut" in name):
                std *= 0.7
            sd[name] = (std * rng.standard_normal(shape)).astype(np.float32)
        else:  # pragma: no cover
            raise KeyError(name)
    return sd


def synth_waveforms(batch, length=32000, seed=20220829):
    """``0.1·N(0,1)`` clipped to [-1,1]; seed = yaml/configuration-voxceleb.yaml:15."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = 0.1 * rng.standard_normal((batch, length), dtype=np.float32)
    return np.clip(x, -1.0, 1.0).astype(np.float32)


def synth_speechlike(batch, length=32000, seed=7):
    """Coloured, amplitude-modulated noise: a less flat spectrum than white noise, used by the
    fbank parity tests so that mel powers span several decades as they do on speech."""
    rng = np.random.Generator(np.random.PCG64(seed))
    from scipy.signal import lfilter
    x = rng.standard_normal((batch, length)).astype(np.float64)
    y = lfilter([1.0], [1.0, -0.95], x, axis=1)
    env = 0.3 + 0.7 * np.abs(np.sin(2 * np.pi * np.arange(length) / 4000.0 + rng.uniform(0, 6, (batch, 1))))
    y = y * env
    y = 0.5 * y / np.abs(y).max(axis=1, keepdims=True)
    return y.astype(np.float32)


def synth_mel(batch, n_mels=80, frames=401, seed=11):
    """Positive mel-power-like features spanning several decades, (B, n_mels, T) float32."""
    rng = np.random.Generator(np.random.PCG64(seed))
    base = rng.normal(-4.0, 1.5, (batch, n_mels, 1))
    x = np.exp(base + 1.2 * rng.standard_normal((batch, n_mels, frames)))
    return x.astype(np.float32)


def synth_speaker_embeddings(n, n_speakers=5994, dim=192, seed=2, same_cos=(0.5, 0.8), n_groups=1, group_cos=0.35):
    """Embeddings with the structure trained speaker embeddings have: `n_speakers` unit centroids, embedding i =
    a c_s(i) + sqrt(1 - a^2) v_i with v_i a random unit direction and a^2 ~ U(same_cos), so that two embeddings of one speaker have cosine
    a a' in same_cos (the regime of tests/golden/e2e_speakers.npz) and an embedding scores ~a against its own centroid.  n_groups > 1 puts
    the centroids themselves into groups (gender / language): c_s = sqrt(group_cos) g_k(s) + sqrt(1 - group_cos) u_s, so centroids of one group
    have cosine ~group_cos and the cohort scores of an embedding are BIMODAL — what the normal-quantile threshold of the fused AS-norm
    kernel does not fit.  Returns (embeddings (n, dim), centroids (n_speakers, dim), speaker of each embedding), all fp32 / int32."""
    rng = np.random.Generator(np.random.PCG64(seed))
    def unit(shape):
        x = rng.standard_normal(shape, dtype=np.float32)
        return x / np.linalg.norm(x, axis=-1, keepdims=True)
    cent = unit((n_speakers, dim))
    if n_groups > 1:
        g = unit((n_groups, dim))
        grp = np.arange(n_speakers) % n_groups
        cent = np.float32(np.sqrt(group_cos)) * g[grp] + np.float32(np.sqrt(1.0 - group_cos)) * cent
        cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    spk = rng.integers(0, n_speakers, n).astype(np.int32)
    a = np.sqrt(rng.uniform(same_cos[0], same_cos[1], n)).astype(np.float32)[:, None]
    e = a * cent[spk] + np.sqrt(1.0 - a * a) * unit((n, dim))
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    return e.astype(np.float32), cent.astype(np.float32), spk


def synth_embeddings(n, dim=192, seed=2, normalize=True):
    """l2norm(standard_normal((n, dim)))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    e = rng.standard_normal((n, dim), dtype=np.float32)
    if normalize:
        e /= np.linalg.norm(e, axis=1, keepdims=True)
    return e.astype(np.float32)
