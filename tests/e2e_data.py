"""Synthetic inputs of the config-1 end-to-end plumbing case (BASELINE configs[0]):
8 synthetic 16 kHz WAV files of different lengths (so that wrap-padding, exact-length and multi-crop
paths are all exercised) and the 28-trial list of all pairs.  Regenerated from seeds on both sides
(oracle/make_golden.py with the reference, tests with the HIP path); never committed."""
import os

import numpy as np
import scipy.io.wavfile as wavfile

E2E_SEED_W = 1
E2E_LENGTHS = [32000, 48000, 20000, 64000, 32001, 40000, 31999, 56000]


def make_e2e_files(folder):
    rng = np.random.Generator(np.random.PCG64(20220829))
    files = []
    for i, n in enumerate(E2E_LENGTHS):
        x = 0.1 * rng.standard_normal(n, dtype=np.float32)
        x += 0.05 * np.sin(2 * np.pi * (200 + 50 * i) * np.arange(n) / 16000.0).astype(np.float32)
        pcm = np.clip(np.round(x * 32767.0), -32768, 32767).astype(np.int16)
        path = os.path.join(folder, f"utt{i}.wav")
        wavfile.write(path, 16000, pcm)
        files.append(path)
    lines = []
    k = 0
    for i in range(len(files)):
        for j in range(i + 1, len(files)):
            lines.append(f"{k % 2} {files[i]} {files[j]}\n")
            k += 1
    trial_path = os.path.join(folder, "trials.txt")
    with open(trial_path, "w") as fh:
        fh.writelines(lines)
    return files, trial_path, lines
