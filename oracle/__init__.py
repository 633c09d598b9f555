"""CPU oracle for the speaker-embedding hot path — TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-CPU / numpy, fp32 or fp64) of the reference's algorithm
for the path named by BASELINE.json: log-mel fbank -> ECAPA-TDNN / RawNet2 forward -> cosine /
AS-norm scoring.  Every function cites the reference file:line it follows.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` — as the checker or the timed CPU baseline, never as the thing shipped.  Nothing under
``speakerverification_amd/`` imports ``oracle``; the product path fails loudly when the HIP library
is missing instead of falling back to this code.

Parity pinning status
---------------------
* ECAPA-TDNN body, RawNet2 body, ``PreEmphasis``, cosine / AS-norm / p-norm scoring and eval-mode
  cropping are PINNED: ``oracle/make_golden.py`` imports the reference (``/root/reference/src``, build
  container only) and commits its outputs as fixtures under ``tests/golden``; ``tests/`` check this
  package against those fixtures.
* The mel front-end F2–F3 (STFT-by-conv power spectrum + Slaney mel bank) is **PARITY UNPINNED**:
  the reference delegates it to the third-party package ``nnAudio`` (``src/requirements.txt:22``,
  no version pin, not vendored, not installable here), and holds no tests or golden vectors for it.
  ``oracle/fbank.py`` restates nnAudio's published formulation (0.3.x ``features.mel.MelSpectrogram``
  -> ``features.stft.STFT`` + librosa-style ``get_mel``) and is the definition of record.
"""
