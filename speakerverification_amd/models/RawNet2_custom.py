"""Drop-in for the reference plug-in ``models/RawNet2_custom.py`` (MainModel :230-243) in the variant
the fusion models instantiate (``front_proc='sinc'``, ``aggregate='asp'``, Raw_ECAPA_sinc_asp.py:26-28).

    model = MainModel(nOut=320, front_proc='sinc', aggregate='asp', att_dim=128, audio_spec={...})
    emb = model(wav)          # (B, 32000) waveform -> (B, nOut); (nOut,) for B == 1

State-dict keys are the reference's (147 tensors); the sinc band-pass filters are rebuilt from
``first_conv.low_hz_`` / ``band_hz_`` once per weight load instead of once per forward.
"""
from __future__ import annotations

from .. import synth
from ._base import HipModule


class RawNet2(HipModule):
    model_kind = "rawnet2"

    def __init__(self, nOut=512, front_proc="sinc", aggregate="gru", att_dim=128, audio_spec=None, device=None,
                 compute=None, max_batch=None, **kwargs):
        if front_proc != "sinc" or aggregate != "asp" or att_dim != 128:
            raise NotImplementedError("only front_proc='sinc', aggregate='asp', att_dim=128 is built "
                                      "(the variant of Raw_ECAPA_sinc_asp.py:26-28)")
        audio_spec = audio_spec or {"sample_rate": 16000, "sentence_len": 2.0}
        if int(audio_spec["sample_rate"]) != 16000:
            raise NotImplementedError("the sinc front-end is built for sample_rate 16000 (RawNet2_custom.py:55-63)")
        self.nb_samp = int(audio_spec["sentence_len"] * audio_spec["sample_rate"])      # LayerNorm(nb_samp), :58-60
        # hip_compute: "f32" (exact fp32 MFMA) | "f32x3" | "f16" (fp16 storage + fp16 MFMA: RawNet2's fast mode) | "bf16" (the same
        # kernels on bf16: range-safe, but RawNet2 loses two digits to bf16 weight rounding) | "half" = this model's 16-bit mode (f16)
        compute = compute or kwargs.get("hip_compute", "f32")
        compute = {"half": "f16", "fp16": "f16"}.get(compute, compute)
        # range_fallback (round 6): what an fp16 handle that reports an overflow (SVHIP_ERR_NONFINITE: this checkpoint's activations pass
        # 65504) is replaced by — a NEW handle in that mode, with a warning; None re-raises.  "f32" is the exact mode; "bf16" is range-safe and
        # fast but loses accuracy on RawNet2 (bf16 weight rounding: cosine 0.995 to fp32 on a well-scaled checkpoint, 0.7 - 0.97 on the
        # ill-scaled one of tests/test_gpu_rawnet2.py), so it is not the default
        self._range_fallback = kwargs.get("range_fallback", "f32")
        max_batch = int(max_batch or kwargs.get("embed_batch", 256))
        super().__init__(synth.rawnet2_param_spec(nOut=nOut, nb_samp=self.nb_samp, att_dim=att_dim),
                         dict(embed_dim=nOut), device=device if device is not None else kwargs.get("device"),
                         compute=compute, max_batch=max_batch, primary_samples=self.nb_samp)

    def forward(self, x):
        if x.ndim != 2 or x.shape[1] != self.nb_samp:
            raise ValueError(f"RawNet2 was built for (batch, {self.nb_samp}) waveforms, got {tuple(x.shape)} "
                             "(LayerNorm gamma/beta fix the length, RawNet_baseline.py:16-18)")
        from .._lib import SvhipNumericError, ERR_NONFINITE
        eng = self._get_engine(self.nb_samp)
        try:
            return self._squeeze(self._batched(eng.embed_wave, x))
        except SvhipNumericError as e:
            if e.code != ERR_NONFINITE or self._compute != "f16" or not self._range_fallback:
                raise
            import warnings
            warnings.warn(f"RawNet2 fp16 handle: {e}; rebuilding this module's handle with compute = {self._range_fallback!r} "
                          "(every later forward runs in that mode)", RuntimeWarning, stacklevel=2)
            self._compute = self._range_fallback
            self._drop_engine()
            eng = self._get_engine(self.nb_samp)
            return self._squeeze(self._batched(eng.embed_wave, x))


def MainModel(nOut=512, **kwargs):
    return RawNet2(nOut=nOut, **kwargs)
