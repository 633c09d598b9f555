"""developer probe: AS-norm statistics at BASELINE configs[3] size (1.2 M x 192, cohort 5994, top 200) under handle options
   python tools/asnorm_probe.py [opt=val ...]   (alternates the given option set with the defaults, 4 rounds)"""
import sys, time
import torch
sys.path.insert(0, ".")
from speakerverification_amd.engine import Engine
opts = dict(a.split("=") for a in sys.argv[1:])
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
with torch.cuda.stream(st):
    eng = Engine(model="none", device=0, stream=st.cuda_stream)
    N, K, top, D = 1_200_000, 5994, 200, 192
    g = torch.Generator(device=dev).manual_seed(2)
    E = torch.randn((N, D), generator=g, device=dev); eng.l2norm_(E)
    C = torch.randn((K, D), generator=g, device=dev); eng.l2norm_(C)
    ref = None
    for rnd in range(4):
        for on in (0, 1):
            for k, v in opts.items():
                eng.set_option(k, int(v) if on else 0)
            eng.asnorm_stats(E, C, top)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mu, sd = eng.asnorm_stats(E, C, top)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rnd == 3:
                eng.profile(True)
                eng.asnorm_stats(E, C, top)
                torch.cuda.synchronize()
                print({k: (round(v["ms"], 3), v["launches"]) for k, v in eng.profile_results().items()})
                eng.profile(False)
            if ref is None:
                ref = (mu.clone(), sd.clone())
            print(f"opts {'on ' if on else 'off'} {dt*1e3:8.3f} ms  {N/dt/1e6:7.1f} M rows/s  max|dmu| {float((mu-ref[0]).abs().max()):.2e} max|dsd| {float((sd-ref[1]).abs().max()):.2e}", flush=True)
