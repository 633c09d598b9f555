"""Freeze the mel front-end's definition of record (TEST INFRASTRUCTURE; build container, run once):

    python -m oracle.make_fbank_fixture          # writes tests/golden/fbank.npz

F2-F3 are PARITY UNPINNED (nnAudio is absent, see oracle/__init__.py): no reference-held vector exists for them, and
this script does not change that.  What it does is stop the definition from drifting silently: the outputs of
``oracle/fbank.py`` (float64 and float32 arithmetic) on seeded inputs are committed, and both the oracle
(tests/test_oracle_golden.py) and the HIP kernel (tests/test_gpu_fbank.py) are checked against the committed arrays
rather than against whatever ``oracle/fbank.py`` says today.  Inputs are regenerated from seeds on both sides
(speakerverification_amd/synth.py); the border-impulse inputs are described by (row, index, value) triples.
"""
import os

import numpy as np
import torch

from oracle import fbank as o_fbank
from speakerverification_amd import synth

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fbank.npz")
IMPULSES = [(0, 0, 1.0), (1, 31999, 1.0), (2, 1, -0.5)]


def impulse_wave():
    w = np.zeros((3, 32000), np.float32)
    for r, i, v in IMPULSES:
        w[r, i] = v
    return w


def main():
    white = synth.synth_waveforms(1)
    speech = synth.synth_speechlike(1)
    imp = impulse_wave()
    out = {"impulses": np.array(IMPULSES, np.float64)}
    for name, wav in (("white", white), ("speech", speech), ("impulse", imp)):
        t = torch.from_numpy(wav)
        out[name + "_f64"] = o_fbank.melspectrogram(t.double()).numpy()
        out[name + "_f32"] = o_fbank.melspectrogram(t).numpy()
        out[name + "_logmel_f64"] = o_fbank.log_mean_norm(torch.from_numpy(out[name + "_f64"])).numpy().astype(np.float32)
    np.savez_compressed(OUT, **out)
    print(OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
