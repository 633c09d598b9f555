"""speakerverification_amd — MI355X-native speaker-embedding inference + scoring hot path.

Drop-in counterparts of the reference's plug-in seams (hiimmuc/SpeakerVerification):

  reference                                              here
  -----------------------------------------------------  ------------------------------------------------
  models/<Name>.py::MainModel            (src/model.py:69-73)   speakerverification_amd.models.<Name>.MainModel
  models/FeatureExtraction/feature.py    (src/model.py:61-65)   speakerverification_amd.models.FeatureExtraction.feature
  utils.similarity_measure               (src/utils.py:126)     speakerverification_amd.scoring.similarity_measure
  SpeakerEncoder / WrappedModel / ModelHandling (src/model.py)  speakerverification_amd.model

Everything computes through libsvhip.so (hand-written HIP for gfx950) via the C ABI in
include/svhip.h; there is no CPU fallback.
"""
__version__ = "0.1.0"
