# library used WITHOUT torch in the process: numpy host pointers only (C-ABI standalone)
import sys, types
sys.modules['torch'] = None   # make `import torch` fail
sys.path.insert(0, '.')
import numpy as np
from speakerverification_amd.engine import Engine
from speakerverification_amd import synth
e = Engine(model="none", max_batch=2)
m = e.fbank(synth.synth_waveforms(2))
print("no-torch fbank ok", m.shape, float(m.max()))
