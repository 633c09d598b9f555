"""developer probe (round 4): per-kernel table of the fusion model's ECAPA branch (C = 512, mel power without log) at B = 256."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from speakerverification_amd import synth
from speakerverification_amd.engine import Engine

dev = torch.device("cuda:0")
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng0 = bench.make_engine("ecapa", "bf16", 256, 0)
wavs = bench.synth_batches(eng0, 4, 256, 0, dev)
eng = Engine(model="ecapa", compute="bf16", channels=512, embed_dim=192, max_batch=256, samples=bench.SAMPLES, log_input=False, device=0, stream=st.cuda_stream)
eng.load_state_dict(synth.synth_state_dict(synth.ecapa_param_spec(C=512), seed=1))
eng.finalize()
out = torch.empty((B, 192), device=dev, dtype=torch.float32)
for _ in range(5): eng.embed_wave(wavs[0][:B], out=out, async_=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 40
for k in range(n): eng.embed_wave(wavs[k % 4][:B], out=out, async_=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("ECAPA C=512 bf16 B=%d: %.3f ms/step, %.0f utt/s" % (B, dt * 1e3, B / dt))
eng.profile(True)
for k in range(5): eng.embed_wave(wavs[k % 4][:B], out=out, async_=True)
torch.cuda.synchronize()
prof = eng.profile_results()
tot = sum(v["ms"] for v in prof.values())
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    print("  %-18s %7.1f us x %4.1f = %7.1f us  %s" % (k, v["ms"] / max(1, v["launches"]) * 1e3, v["launches"] / 5, v["ms"] / 5 * 1e3,
                                                     ("%.0f TF" % (v["flops"] / (v["ms"] * 1e-3) / 1e12)) if v["flops"] > 0 else ""))
print("  sum %.3f ms" % (tot / 5))
