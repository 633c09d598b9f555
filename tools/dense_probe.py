"""developer probe: dense score matrix 16 384 x 16 384 x 192 (svhip_score_matrix), time and error against float64 on a corner"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from speakerverification_amd.engine import Engine
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
with torch.cuda.stream(st):
    eng = Engine(model="none", device=0, stream=st.cuda_stream)
    g = torch.Generator(device=dev).manual_seed(2)
    E = torch.randn((32768, 192), generator=g, device=dev); eng.l2norm_(E)
    A, B = E[:16384], E[16384:]
    out = torch.empty((16384, 16384), device=dev)
    for rnd in range(4):
        eng.score_matrix(A, B, out); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): eng.score_matrix(A, B, out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"dense 16384^2: {dt*1e3:.3f} ms  {2*16384*16384*192/dt/1e12:.1f} TFLOP/s  {16384*16384*4/dt/1e12:.2f} TB/s of output", flush=True)
    want = A[:256].double() @ B[:300].double().T
    print("max err vs f64 (256 x 300 corner):", float((out[:256, :300].double() - want).abs().max()))
