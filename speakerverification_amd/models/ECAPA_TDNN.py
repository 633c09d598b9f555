"""Drop-in for the reference plug-in ``models/ECAPA_TDNN.py`` (MainModel :505-507, ECAPA_TDNN :339-502).

    model = MainModel(nOut=192, channels=[1024]*4+[3072], n_mels=80, features='melspectrogram', ...)
    emb = model(mel)          # (B, 80, T) mel power -> (B, nOut); (nOut,) for B == 1

The forward runs in libsvhip (HIP, gfx950).  State-dict keys are the reference's (231 tensors).
"""
from __future__ import annotations

from .. import synth
from ._base import HipModule


def _crop_samples(audio_spec):
    """evaluation crops are sentence_len * sample_rate samples long (audio_loader.py:100-110)"""
    try:
        return int(audio_spec["sentence_len"] * audio_spec["sample_rate"])
    except Exception:
        return None


class ECAPA_TDNN(HipModule):
    model_kind = "ecapa"

    def __init__(self, input_size=80, lin_neurons=192, activation=None, channels=(1024, 1024, 1024, 1024, 3072),
                 kernel_sizes=(5, 3, 3, 3, 1), dilations=(1, 2, 3, 4, 1), attention_channels=128, res2net_scale=8,
                 se_channels=128, input_norm=False, global_context=True, device=None, compute=None, max_batch=None,
                 **kwargs):
        channels = list(channels)
        C = channels[0]
        if channels != [C] * 4 + [3 * C]:
            raise NotImplementedError("channels must be [C, C, C, C, 3C] (ECAPA_TDNN.py:378)")
        if list(kernel_sizes) != [5, 3, 3, 3, 1] or list(dilations) != [1, 2, 3, 4, 1]:
            raise NotImplementedError("only the reference kernel_sizes / dilations are built (ECAPA_TDNN.py:379-380)")
        if attention_channels != 128 or res2net_scale != 8 or se_channels != 128 or not global_context:
            raise NotImplementedError("only the reference attention / Res2Net / SE geometry is built")
        if activation is not None and getattr(activation, "__name__", "GELU") != "GELU":
            raise NotImplementedError("activation must be GELU (ECAPA_TDNN.py:377)")
        n_mels = kwargs.get("n_mels", input_size)
        assert input_size == n_mels, "inappropriate input size, should equal feature_dim"     # ECAPA_TDNN.py:394
        self.channels = channels
        self.input_norm = bool(input_norm)
        self.kwargs = kwargs
        features = str(kwargs.get("features", "melspectrogram")).strip()
        self.log_input = features == "melspectrogram"                                          # ECAPA_TDNN.py:473
        compute = compute or kwargs.get("hip_compute", "f32")
        compute = {"half": "bf16"}.get(compute, compute)        # "half" = each model's own 16-bit mode (ECAPA: bf16, RawNet2: f16)
        hop = kwargs.get("hop_length", 80)
        self._hop = hop
        # the mel front-end of the fused waveform path takes the same keywords the reference's feature factory reads
        # from the config (models/FeatureExtraction/feature.py:66-71), so an overriding YAML changes both consistently
        fe = {k: kwargs[k] for k in ("sr", "n_fft", "win_length", "fmin", "fmax", "pre_emphasis") if k in kwargs}
        if kwargs.get("window", "hamming") != "hamming":
            raise NotImplementedError("only the hamming window of feature.py:68 is built")
        max_batch = int(max_batch or kwargs.get("embed_batch", 256))
        super().__init__(synth.ecapa_param_spec(C=C, n_mels=n_mels, nOut=lin_neurons, input_norm=self.input_norm),
                         dict(channels=C, n_mels=n_mels, embed_dim=lin_neurons, log_input=self.log_input,
                              input_norm=self.input_norm, hop_length=hop, **fe),
                         device=device if device is not None else kwargs.get("device"), compute=compute,
                         max_batch=max_batch, primary_samples=_crop_samples(kwargs.get("audio_spec")))

    def forward(self, x, lengths=None):
        """x: (B, n_mels, T) features, torch tensor (CPU / CUDA) or numpy.  lengths is ignored exactly as
        in the reference's call path (src/model.py never passes it)."""
        if x.ndim != 3:
            raise ValueError(f"expected (batch, n_mels, frames), got {tuple(x.shape)}")
        T = x.shape[2]
        eng = self._get_engine((T - 1) * self._hop, batch=x.shape[0])
        return self._squeeze(self._batched(eng.embed_features, x, eng.max_batch))

    def embed_wave(self, wav):
        """fused waveform -> embedding (fbank + forward in one library call)"""
        eng = self._get_engine(wav.shape[1], batch=wav.shape[0])
        return self._squeeze(self._batched(eng.embed_wave, wav, eng.max_batch))


def MainModel(nOut=512, **kwargs):
    return ECAPA_TDNN(lin_neurons=nOut, **kwargs)
