# developer helper (round 4): the latency record alone (all modes)
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, torch, bench
dev = torch.device("cuda:0")
st = torch.cuda.Stream(device=dev)
with torch.cuda.stream(st):
    eng = bench.make_engine("ecapa", "bf16", 256, 0)
    wavs = bench.synth_batches(eng, 2, 256, 0, dev)
    r = bench.latency_bench(0, dev, wavs)
for mode, rows in r.items():
    if not isinstance(rows, dict): continue
    for b, x in rows.items():
        print(mode, b, round(x["ms_per_call"], 3), round(x["ms_per_call_pipelined"], 3), x["launches_per_call"], x["kernels"])
PY
