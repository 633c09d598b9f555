"""TEST INFRASTRUCTURE — CPU restatement (numpy) of the counter-based synthetic-waveform generator
``svhip_synth_waveforms`` (speakerverification_amd/csrc/synth.hip).

Nothing in the reference corresponds to this: its evaluation reads audio files (src/model.py:363-394).  SURVEY.md §8d
(config 5) asks for 1 M utterances "generated on-device per shard from a counter-based RNG (Philox)".  The integer stream
below is standard Philox4x32-10 (Salmon et al., SC'11; known-answer vectors of the Random123 distribution are checked in
tests/test_oracle_golden.py), so it is pinned independently of the HIP kernel.
"""
from __future__ import annotations

import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter: np.ndarray, key) -> np.ndarray:
    """counter: (..., 4) uint32, key: (k0, k1) -> (..., 4) uint32."""
    c = np.asarray(counter, dtype=np.uint64)
    c0, c1, c2, c3 = c[..., 0], c[..., 1], c[..., 2], c[..., 3]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & _MASK
        n1 = p1 & _MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & _MASK
        n3 = p0 & _MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def synth_waveforms(seed: int, first_utt: int, B: int, L: int) -> np.ndarray:
    """(B, L) float32: utterances [first_utt, first_utt + B) of stream `seed` (float32 arithmetic, as the kernel)."""
    assert L % 4 == 0
    q = np.arange(L // 4, dtype=np.uint64)
    out = np.empty((B, L), np.float32)
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    for b in range(B):
        u = first_utt + b
        ctr = np.stack([q, np.full_like(q, u & 0xFFFFFFFF), np.full_like(q, (u >> 32) & 0xFFFFFFFF), np.zeros_like(q)], axis=-1)
        r = philox4x32_10(ctr, key)
        scale = np.float32(2.3283064365386963e-10)
        u1 = (r[:, 0::2].astype(np.float32) + np.float32(0.5)) * scale
        u2 = (r[:, 1::2].astype(np.float32) + np.float32(0.5)) * scale
        rad = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
        ang = (np.float32(6.283185307179586) * u2).astype(np.float32)
        v = np.empty((L // 4, 4), np.float32)
        v[:, 0::2] = rad * np.cos(ang)
        v[:, 1::2] = rad * np.sin(ang)
        out[b] = np.clip(np.float32(0.1) * v, -1.0, 1.0).reshape(-1)
    return out
