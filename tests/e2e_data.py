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


# ---- a second end-to-end case whose scores are SENSITIVE ---------------------------------------------------------
# The 28 golden scores of the case above span 0.9970 - 0.9996 (random weights map every noise utterance onto almost the same
# direction), so a plumbing error that moves a score by less than 1e-3 would pass.  Here: four synthetic "speakers" (harmonic source at
# a speaker f0 with vibrato, two formant resonances, amplitude modulation, a little noise), two utterances each, of different lengths;
# and the statistics of `asp_bn` are CALIBRATED to the pooled statistics of these files (stored in the fixture: a trained model's
# BatchNorm matches its data, the random one does not), which removes the common direction: the golden cosine scores span -0.7 .. +0.8,
# same-speaker pairs on top.
E2E2_SEED_W = 2
E2E2_LENGTHS = [32000, 41000, 36000, 32000, 52000, 33000, 40000, 47000]      # file i: speaker i // 2


def speaker_wave(spk, utt, n):
    from scipy.signal import lfilter
    rng = np.random.Generator(np.random.PCG64(1000 * spk + utt))
    f0 = [95.0, 140.0, 210.0, 120.0][spk] * (1.0 + 0.03 * rng.standard_normal())
    t = np.arange(n) / 16000.0
    vib = 1.0 + 0.02 * np.sin(2 * np.pi * (4.0 + spk) * t + rng.uniform(0, 6))
    phase = 2 * np.pi * np.cumsum(f0 * vib) / 16000.0
    src = sum(np.sin(k * phase) / k for k in range(1, 30))
    src = src * (0.6 + 0.4 * np.sin(2 * np.pi * (2.0 + 0.5 * utt) * t + spk)) + 0.02 * rng.standard_normal(n)
    y = src
    for fc, bw in [((500, 80), (1500, 120)), ((300, 60), (2300, 150)), ((700, 90), (1100, 100)), ((400, 70), (2000, 140))][spk]:
        r = np.exp(-np.pi * bw / 16000.0)
        th = 2 * np.pi * fc / 16000.0
        y = lfilter([1.0], [1.0, -2 * r * np.cos(th), r * r], y)
    return (0.3 * y / np.abs(y).max()).astype(np.float32)


def make_e2e_speaker_files(folder):
    files = []
    for i, n in enumerate(E2E2_LENGTHS):
        x = speaker_wave(i // 2, i % 2, n)
        pcm = np.clip(np.round(x * 32767.0), -32768, 32767).astype(np.int16)
        path = os.path.join(folder, f"spk{i // 2}_utt{i % 2}.wav")
        wavfile.write(path, 16000, pcm)
        files.append(path)
    lines = []
    for i in range(len(files)):
        for j in range(i + 1, len(files)):
            lines.append(f"{1 if i // 2 == j // 2 else 0} {files[i]} {files[j]}\n")
    trial_path = os.path.join(folder, "trials_speakers.txt")
    with open(trial_path, "w") as fh:
        fh.writelines(lines)
    return files, trial_path, lines
