#!/usr/bin/env python
"""bench.py — throughput of the speaker-embedding hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]            # N > 1: starts its own N ranks (torch.distributed.run child)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --config shard [--gpus N] [--utts-per-gpu 125000]   # BASELINE configs[4]

Default workload (BASELINE.json configs[1]): ECAPA-TDNN C=1024, bf16 MFMA / fp32 accumulate, batch = 256 utterances of
2 s @ 16 kHz per step, waveforms resident in HBM (8 distinct batches = 262 MB are rotated, so the timed loop does not
re-read one Infinity-Cache-resident buffer), one step = fbank -> ECAPA forward -> (256, 192) embeddings.  With N > 1
every rank embeds its own 256 utterances per step (weak scaling, no data-path collective) and the shard embeddings are
assembled with ONE RCCL all-gather at the end of the timed region (reference: all_gather_object, src/model.py:400-404) —
issued through the C ABI (svhip_allgather_rows) on the library's stream; torch.distributed (gloo) only carries the
128-byte RCCL id, the barriers and the max-over-ranks of the wall time.

stdout carries exactly ONE JSON line on rank 0, compact (<= 4 KB: `compact_headline`; the driver keeps a bounded tail of stdout and
round 4's 20 KB line did not parse): the contract's fields; `roofline` = the dominant kernel (the pointwise-conv MFMA GEMM) timed with HIP
events on the launch stream inside the timed region; `cpu_baseline` = the CPU oracle timed on this box's host cores on a bounded sample
(rank 0, N = 1 only); `check` (the last timed step's embeddings verified against the fp32-parity path and a bitwise re-run); `sustained`;
`sub` = {record name: embeddings/s} and `scoring` = the BASELINE configs[3] summary (trial pairs/s).  The sub-records themselves — `kernels`
(per-kernel table), `scoring` (+ CPU baselines), `rawnet2` (configs[2]), `ecapa_f32` / `ecapa_f32x3` / `rawnet2_f32x3` (the 1e-4-parity
paths), `latency` (B = 1 .. 32 per call), `fusion`, `pcie`, `headline_full` — are written one JSON line each, tagged {"record": name}, to
stderr as they are measured and to --record-file (default gpurun_out/bench_records.jsonl).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 256
SAMPLES = 32000
CHANNELS = 1024
EMBED = 192
NBATCH = 8                     # distinct waveform batches rotated in the timed loop (8 x 32.8 MB > the 256 MB Infinity Cache)
PEAK_BF16_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
SEED_STREAM = 20220829         # synthetic utterance stream of the embedding benches (yaml/configuration-voxceleb.yaml:15)
SEED_SHARD = 5


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--compute", default="bf16", choices=["bf16", "f16", "f32", "f32x3"],
                    help="bf16: ECAPA's 16-bit mode (configs[1]); f16: RawNet2's (fp16 storage + fp16 MFMA); f32 / f32x3: the 1e-4-parity paths")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--model", default="ecapa", choices=["ecapa", "rawnet2"], help="ecapa = headline (configs[1]); rawnet2 = configs[2]")
    ap.add_argument("--config", default="batch", choices=["batch", "shard"],
                    help="batch = K steps of one batch per GPU (configs[1]/[2]); shard = configs[4]: a sharded utterance list, "
                         "one all-gather, row-sharded scoring")
    ap.add_argument("--utts-per-gpu", type=int, default=125000, help="--config shard: utterances per GPU (1 M / 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scoring", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the rawnet2 / ecapa_f32 / pcie / latency sub-records")
    ap.add_argument("--allow-gloo", action="store_true",
                    help="N > 1: if the RCCL communicator cannot be created, carry the all-gather over gloo (host round trip) instead of "
                         "failing; the record then says so.  Without this flag a multi-GPU run that is not on RCCL exits non-zero.")
    ap.add_argument("--sustain-seconds", type=float, default=2.0, help="length of the extra back-to-back run reported as `sustained` (0: skip)")
    ap.add_argument("--record-file", default=None,
                    help="JSON-lines file that receives every sub-record and the full (verbose) headline record; default "
                         "gpurun_out/bench_records.jsonl when gpurun_out/ exists, otherwise none.  stdout carries ONE compact line.")
    return ap.parse_args()


HEADLINE_MAX_BYTES = 4096


class Records:
    """Where the detail goes.  stdout carries exactly ONE JSON line, the compact headline (<= HEADLINE_MAX_BYTES: the driver keeps
    a bounded tail of stdout, and round 4's 20 KB line did not parse).  Every sub-record (`rawnet2`, `latency`, `scoring`, per-kernel
    tables ...) is written as its own JSON line, tagged {"record": name}, to stderr as it is measured and to --record-file."""

    def __init__(self, path=None):
        if path is None and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
            path = os.path.join(ROOT, "gpurun_out", "bench_records.jsonl")
        self.path = path
        self.f = None
        if path:
            try:
                self.f = open(path, "w")
            except OSError:
                self.f = None

    def emit(self, name, rec):
        text = json.dumps({"record": name, **(rec if isinstance(rec, dict) else {"value": rec})})
        print(text, file=sys.stderr, flush=True)
        if self.f:
            self.f.write(text + "\n")
            self.f.flush()

    def close(self):
        if self.f:
            self.f.close()
            self.f = None


def _num(x, nd=4):
    """floats of the compact line: 6 significant digits are more than any of these measurements carry"""
    if isinstance(x, float):
        return float(f"{x:.6g}")
    return x


def compact_headline(full, sub=None, scoring=None):
    """The ONE stdout line: the contract's fields, the roofline and cpu_baseline objects, the verification verdict, the sustained
    figure, a flat `sub` map {name: embeddings/s} and a scoring summary.  Everything else lives in the records (see Records)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    line = {k: _num(full[k]) for k in keep if k in full}
    cfg = full.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "batch_per_gpu", "utterances_per_gpu", "batch", "collective") if k in cfg}
    r = full.get("roofline")
    if r:
        line["roofline"] = {k: _num(r.get(k)) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                                          "flops_per_launch", "algorithmic_bytes_per_launch", "avg_launch_ms", "launches")
                            if k in r}
    c = full.get("cpu_baseline")
    line["cpu_baseline"] = ({k: _num(c.get(k)) for k in ("value", "unit", "cores", "kind", "cpu", "sample")} if c else None)
    chk = full.get("check")
    if chk:
        line["check"] = {k: _num(chk.get(k)) for k in ("ok", "bitwise_rerun", "finite", "min_cosine_vs_f32_path", "max_err_over_scale")
                         if k in chk}
    if full.get("sustained"):
        line["sustained"] = {k: _num(full["sustained"][k]) for k in ("value", "ms_per_step", "steps", "seconds")}
    if "whole_path_TFLOPs" in full:
        line["whole_path_TFLOPs"] = _num(full["whole_path_TFLOPs"])
    sh = full.get("shard")
    if sh:
        line["shard"] = {k: _num(sh.get(k)) for k in ("allgather_ms", "allgather_carrier", "rccl_world", "gathered_rows",
                                                       "blocks_bitwise_ok", "cross_rank_ok", "own_block_intact",
                                                       "cosine_trials_per_s", "asnorm_trials_per_s") if k in sh}
    for k in ("embed_seconds", "end_to_end_seconds"):
        if k in full:
            line[k] = _num(full[k])
    if sub:
        line["sub"] = {k: _num(v) for k, v in sub.items()}
    if scoring:
        line["scoring"] = {k: _num(v) for k, v in scoring.items()}
    text = json.dumps(line)
    if len(text) > HEADLINE_MAX_BYTES:            # never let the headline outgrow the driver's tail: drop the optional parts first
        for k in ("sub", "shard", "sustained", "scoring", "check"):
            line.pop(k, None)
            text = json.dumps(line)
            if len(text) <= HEADLINE_MAX_BYTES:
                break
    return text


class quiet_stdout:
    """fd-level stdout -> stderr while communicators come up: RCCL and gloo print banners on stdout, and stdout must carry
    exactly ONE JSON line"""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` outside a launcher: start the N ranks as a child torch.distributed.run BEFORE anything in this
    process touches the GPU, and hand its exit code back (never measures a single GPU under an N-GPU label)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def host_cores():
    """CPU cores this process may actually use (affinity mask and cgroup quota, not the machine total)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, n)


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(budget_s=28.0):
    """CPU oracle (oracle/: torch-CPU restatement pinned against the reference) on bounded samples of the same workload
    (SURVEY §8d): fbank + ECAPA C=1024 fp32 on all usable host cores at B = 8 (BASELINE configs[0]'s batch) and at B = 256
    (configs[1]'s batch) — median over timed passes: >= 5 at B = 8, as many as the time budget allows at B = 256 (a pass is ~8 s).
    `value` is the B = 256 median (the configuration the GPU number is quoted on)."""
    import statistics
    import torch
    from oracle import ecapa as o_ecapa, fbank as o_fbank
    from speakerverification_amd import synth
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = o_ecapa.to_torch_sd(synth.synth_state_dict(synth.ecapa_param_spec(C=CHANNELS), seed=1))
    wav = torch.from_numpy(synth.synth_waveforms(256, SAMPLES))
    t_all = time.perf_counter()
    rec = {}
    with torch.no_grad():
        o_ecapa.ecapa_forward(o_fbank.melspectrogram(wav[:2]), sd)       # warm-up
        for B, min_passes, share in ((8, 5, 0.2), (256, 2, 1.0)):
            times = []
            t0 = time.perf_counter()
            while len(times) < min_passes or (time.perf_counter() - t0 < share * budget_s and len(times) < 9):
                t1 = time.perf_counter()
                o_ecapa.ecapa_forward(o_fbank.melspectrogram(wav[:B]), sd)
                times.append(time.perf_counter() - t1)
                if time.perf_counter() - t_all > budget_s + 12.0:
                    break
            med = statistics.median(times)
            rec[f"B{B}"] = {"embeddings_per_s": B / med, "median_pass_s": med, "passes": len(times),
                            "pass_s": [round(t, 4) for t in times]}
    dt = time.perf_counter() - t_all
    return {"value": rec["B256"]["embeddings_per_s"], "unit": "embeddings/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "B8": rec["B8"], "B256": rec["B256"],
            "sample": f"fbank + ECAPA-TDNN C=1024 fp32, torch-CPU oracle, {cores} threads: {rec['B8']['passes']} passes of 8 utterances and "
                      f"{rec['B256']['passes']} passes of 256 utterances (2 s @ 16 kHz), medians; {dt:.1f} s in all"}


def scoring_cpu_baseline(E, cohort, ia, ib, top):
    """the reference-style per-trial Python loop (utils.py:126-169 through the oracle) and a numpy batched GEMM,
    on a bounded sample of the same trial list, on this box's host cores."""
    import numpy as np
    import torch
    from oracle import scoring as o_scoring
    cores = host_cores()
    torch.set_num_threads(cores)
    res = {"cores": cores, "kind": "port"}
    Et = torch.from_numpy(E)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 2.0 and n < len(ia):
        for p in range(n, min(n + 2000, len(ia))):
            o_scoring.cosine_similarity(Et[ia[p]][None], Et[ib[p]][None])
        n = min(n + 2000, len(ia))
    res["cosine_per_trial_loop_trials_per_s"] = n / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 3.0 and n < len(ia):
        for p in range(n, min(n + 200, len(ia))):
            o_scoring.zt_norm_similarity(E[ia[p]][None], E[ib[p]][None], cohort, top=top)
        n = min(n + 200, len(ia))
    res["asnorm_per_trial_loop_trials_per_s"] = n / (time.perf_counter() - t0)
    m = min(len(ia), 200_000)
    t0 = time.perf_counter()
    np.abs(np.einsum("pd,pd->p", E[ia[:m]], E[ib[:m]]))
    res["cosine_numpy_batched_trials_per_s"] = m / (time.perf_counter() - t0)
    rows = min(E.shape[0], 16384)
    t0 = time.perf_counter()
    S = E[:rows] @ cohort.T
    S = -np.partition(-S, top - 1, axis=1)[:, :top]
    S.mean(axis=1), S.std(axis=1)
    res["asnorm_stats_numpy_rows_per_s"] = rows / (time.perf_counter() - t0)
    res["sample"] = "per-trial loops: 2-3 s each; numpy batched: 200 k trials / 16 k rows of the same 1.2 M-trial list"
    return res


def scoring_bench(dev, with_cpu=True, compute="f32"):
    """BASELINE config 4: 1.2 M synthetic 192-d embeddings, 1.2 M-trial list, cohort 5994, top 200."""
    import numpy as np
    import torch
    from speakerverification_amd.engine import Engine
    assert torch.cuda.current_stream().cuda_stream != 0
    eng = Engine(model="none", device=dev.index, stream=torch.cuda.current_stream().cuda_stream, compute=compute)
    N, P, K, top, D = 1_200_000, 1_200_000, 5994, 200, 192
    g = torch.Generator(device=dev).manual_seed(2)
    E = torch.randn((N, D), generator=g, device=dev, dtype=torch.float32)
    eng.l2norm_(E)
    cohort = torch.randn((K, D), generator=g, device=dev, dtype=torch.float32)
    eng.l2norm_(cohort)
    ia = torch.arange(P, device=dev, dtype=torch.int32)
    ib = torch.randperm(N, generator=g, device=dev)[:P].to(torch.int32)
    out = torch.empty(P, device=dev, dtype=torch.float32)

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    res = {}
    t = timed(lambda: eng.score_pairs(E, ia, ib, out))
    res["cosine_pairs_per_s"] = P / t
    res["cosine_pairs_GBps"] = P * 1540 / t / 1e9
    res["cosine_pairs_frac_of_hbm_peak"] = P * 1540 / t / 1e9 / PEAK_HBM_GBS
    mu_sd = {}

    def stats():
        mu_sd["v"] = eng.asnorm_stats(E, cohort, top)
    t_stats = timed(stats, reps=1)
    mu, sd = mu_sd["v"]
    t_pairs = timed(lambda: eng.asnorm_pairs(E, mu, sd, ia, ib, out))
    res["asnorm_pairs_per_s"] = P / (t_stats + t_pairs)
    res["asnorm_stats_s"] = t_stats
    res["asnorm_cohort_gemm_TFLOPs"] = 2.0 * N * K * D / t_stats / 1e12
    A, Bm = E[:16384], E[16384:32768]
    dense = torch.empty((16384, 16384), device=dev, dtype=torch.float32)
    t = timed(lambda: eng.score_matrix(A, Bm, dense))
    res["dense_pairs_per_s"] = 16384 * 16384 / t
    res["dense_TFLOPs"] = 2.0 * 16384 * 16384 * D / t / 1e12
    # (the dense score GEMM runs as three fp16 MFMAs per product on half-plane operands since round 4: its ceiling is the 16-bit peak / 3)
    res["dense_frac_of_split_ceiling"] = res["dense_TFLOPs"] / (PEAK_BF16_TFLOPS / 3.0)
    res["asnorm_frac_of_split_ceiling"] = res["asnorm_cohort_gemm_TFLOPs"] / (PEAK_BF16_TFLOPS / 3.0)
    res["asnorm_fallback_rows"] = int(eng.asnorm_last_fallback)
    # the same AS-norm on embeddings with speaker structure — 5 994 centroids as the cohort, embeddings = centroid +
    # within-speaker noise (same-speaker cosine 0.5 - 0.8), and with the centroids in two groups (cosine 0.35 inside a group: bimodal cohort
    # scores, which the normal-quantile threshold of the fused kernel does not fit: those rows take the refit passes).  Sampled rows are
    # checked against the float64 statement on the host.
    def structured(n_groups):
        gg = torch.Generator(device=dev).manual_seed(40 + n_groups)
        def unit(*shape):
            x = torch.randn(shape, generator=gg, device=dev, dtype=torch.float32)
            return x / x.norm(dim=-1, keepdim=True)
        cent = unit(K, D)
        if n_groups > 1:
            grp = torch.arange(K, device=dev) % n_groups
            cent = (0.35 ** 0.5) * unit(n_groups, D)[grp] + (0.65 ** 0.5) * cent
            cent = cent / cent.norm(dim=1, keepdim=True)
        spk = torch.randint(0, K, (N,), generator=gg, device=dev)
        a = (0.5 + 0.3 * torch.rand((N, 1), generator=gg, device=dev)).sqrt()
        Es = a * cent[spk] + (1.0 - a * a).sqrt() * unit(N, D)
        Es = Es / Es.norm(dim=1, keepdim=True)
        box = {}
        def run():
            box["v"] = eng.asnorm_stats(Es, cent.contiguous(), top)
        t = timed(run, reps=1)
        slab, (refit, passes) = int(eng.asnorm_last_fallback), eng.asnorm_last_refit
        m_, s_ = box["v"]
        rows = torch.arange(0, N, N // 64, device=dev)[:64]
        S = (Es[rows].double() @ cent.double().T).sort(dim=1, descending=True).values[:, :top]
        em = float((m_[rows].double() - S.mean(1)).abs().max())
        es = float(((s_[rows].double() - S.std(1, unbiased=False)).abs() / S.std(1, unbiased=False)).max())
        rec = {"asnorm_stats_s": t, "vs_gaussian": t / t_stats, "refit_rows": int(refit), "refit_passes": int(passes), "slab_rows": slab,
               "slab_fraction": slab / N, "sampled_mu_err": em, "sampled_sigma_rel_err": es, "ok": bool(em <= 2e-7 and es <= 1e-5)}
        if refit > N // 100:          # what round 5 did with such rows: the slab path (N x K scores through HBM) for every one of them
            eng.set_option("asnorm_norefit", 1)
            rec["slab_route_s"] = timed(run, reps=1)
            rec["slab_route_rows"] = int(eng.asnorm_last_fallback)
            eng.set_option("asnorm_norefit", 0)
        return rec
    res["asnorm_speakers"] = structured(1)
    res["asnorm_speaker_groups"] = structured(2)
    # verification metrics over the same 1.2 M-trial list (EER / minDCF inputs; host arrays in, PCIe included)
    sc_host = out.cpu().numpy()
    lab_host = (np.arange(P) % 2).astype(np.int32)
    t = timed(lambda: eng.min_dcf(sc_host, lab_host, 0.05, 1, 1))
    res["min_dcf_trials_per_s"] = P / t
    t = timed(lambda: eng.roc_points(sc_host, lab_host))
    res["roc_points_trials_per_s"] = P / t
    res["config"] = {"embeddings": N, "trials": P, "cohort": K, "top": top, "dim": D, "gemm": compute}
    if with_cpu:
        res["cpu_baseline"] = scoring_cpu_baseline(E[:200_000].cpu().numpy(), cohort.cpu().numpy(),
                                                   np.arange(200_000), np.random.default_rng(4).permutation(200_000), top)
    eng.close()
    return res


def make_engine(model, compute, B, local, embed=None):
    import torch
    from speakerverification_amd import synth
    from speakerverification_amd.engine import Engine
    st = torch.cuda.current_stream().cuda_stream
    if model == "rawnet2":
        eng = Engine(model="rawnet2", compute=compute, embed_dim=embed or 320, max_batch=B, samples=SAMPLES, device=local, stream=st)
        eng.load_state_dict(synth.synth_state_dict(synth.rawnet2_param_spec(nOut=embed or 320), seed=1))
    else:
        eng = Engine(model="ecapa", compute=compute, channels=CHANNELS, embed_dim=embed or EMBED, max_batch=B,
                     samples=SAMPLES, device=local, stream=st)
        eng.load_state_dict(synth.synth_state_dict(synth.ecapa_param_spec(C=CHANNELS), seed=1))
    eng.finalize()
    return eng


def dominant_label(eng, wav):
    """The dominant kernel of a step = the profile label that carries the most algorithmic FLOPs, read from one profiled (untimed)
    step of THIS engine on THIS batch — whatever route the library takes for the shape, batch size and device at hand (the label
    used to be re-derived here from a copy of the C++ routing and went stale: ADVICE r3)."""
    import torch
    scratch = torch.empty((wav.shape[0], eng.embed_dim), device=wav.device, dtype=torch.float32)
    eng.profile(True)
    eng.embed_wave(wav, out=scratch, async_=True)
    torch.cuda.synchronize()
    prof = eng.profile_results()
    eng.profile(False)
    cands = {k: v["flops"] for k, v in prof.items() if v["flops"] > 0 and v["launches"] > 0}
    if not cands:
        raise RuntimeError("no kernel of the step reports FLOPs: cannot name the roofline kernel")
    return max(cands, key=cands.get)


def roofline_of(prof, label, compute):
    # f32x3: three bf16 MFMAs per product -> the bf16 peak / 3 is what an x3 kernel can deliver in reference-graph FLOPs
    # (f16: the fp16 MFMA forms take the cycles of the bf16 ones, MI355X_MICROARCH.md)
    peak = {"bf16": PEAK_BF16_TFLOPS, "f16": PEAK_BF16_TFLOPS, "f32": PEAK_F32_TFLOPS, "f32x3": PEAK_BF16_TFLOPS / 3.0}[compute]
    if label not in prof or prof[label]["launches"] <= 0:
        raise RuntimeError(f"roofline kernel {label!r} was not launched inside the timed region (profile labels: {sorted(prof)})")
    dom = prof[label]
    avg_ms = dom["ms"] / max(1, dom["launches"])
    achieved = (dom["flops"] / max(1, dom["launches"])) / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    traffic, src = None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            traffic = json.load(open(pmc_path)).get(label, {}).get("hbm_bytes_per_launch")
            src = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, gfx950 x2 fetch correction; not re-measured in this run)"
        except Exception:
            traffic = None
    return {"kernel": label, "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "traffic": traffic, "traffic_source": src, "avg_launch_ms": avg_ms, "launches": dom["launches"],
            "flops_per_launch": dom["flops"] / max(1, dom["launches"])}


def embed_loop(eng, wavs, K, W, B, shard, label, barrier=lambda: None):
    """W warm-up + K timed steps over the rotating waveform batches; the dominant kernel's launches are bracketed by HIP
    events on the launch stream inside the timed region (resolved after it)."""
    import torch
    scratch = torch.empty((B, eng.embed_dim), device=shard.device, dtype=torch.float32)
    for w in range(W):
        eng.embed_wave(wavs[w % len(wavs)], out=scratch, async_=True)
    torch.cuda.synchronize()
    eng.profile(True, only=label)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        eng.embed_wave(wavs[k % len(wavs)], out=shard[k * B:(k + 1) * B], async_=True)
    return t0


def kernel_table(eng, wavs, B, n_steps, dev):
    import torch
    scratch = torch.empty((B, eng.embed_dim), device=dev, dtype=torch.float32)
    eng.profile(True)            # untimed pass: every kernel bracketed, for the per-kernel table
    for k in range(n_steps):
        eng.embed_wave(wavs[k % len(wavs)], out=scratch, async_=True)
    torch.cuda.synchronize()
    prof_all = eng.profile_results()
    eng.profile(False)
    return {k: {"avg_ms": v["ms"] / max(1, v["launches"]), "launches_per_step": v["launches"] / n_steps,
                "ms_per_step": v["ms"] / n_steps,
                "TFLOPs": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and v["flops"] > 0 else None}
            for k, v in prof_all.items()}


def sub_bench(model, compute, B, local, dev, wavs, steps=10, warmup=2):
    """a smaller record of the same shape for the other configurations (rawnet2 = configs[2]; ecapa f32 = the 1e-4-parity path)"""
    import torch
    eng = make_engine(model, compute, B, local)
    label = dominant_label(eng, wavs[0])
    shard = torch.empty((steps * B, eng.embed_dim), device=dev, dtype=torch.float32)
    t0 = embed_loop(eng, wavs, steps, warmup, B, shard, label)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = eng.profile_results()
    rec = {"value": steps * B / dt, "unit": "embeddings/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "batch": B,
           "dtype": compute, "finite": bool(torch.isfinite(shard).all().item()),
           "whole_path_TFLOPs": eng.flops_per_utterance * steps * B / dt / 1e12,
           "roofline": roofline_of(prof, label, compute)}
    rec["check"] = verify_last_step(eng, wavs[(steps - 1) % len(wavs)], shard[(steps - 1) * B:steps * B], local, dev)
    kt = kernel_table(eng, wavs, B, min(steps, 5), dev)
    rec["launches_per_step"] = sum(v["launches_per_step"] for v in kt.values())
    rec["kernels"] = {k: {"ms_per_step": v["ms_per_step"], "TFLOPs": v["TFLOPs"]} for k, v in kt.items()}
    eng.close()
    return rec


def latency_bench(local, dev, wavs, batches=(1, 10, 20, 32), modes=("bf16", "f32x3", "rawnet2_f16"), full_rate=None):
    """The reference API's native operating point (row a14): `embed_utterance` / per-file enrolment embeds B = num_eval = 10 - 20 crops
    of ONE file per call (src/model.py:675-704, yaml/configuration-voxceleb.yaml:156).  One engine per mode, created for the Python
    wrapper's default max_batch = 256, called with B rows: `ms_per_call` = a synchronous call (host launch overhead + the GPU work of
    one call, what a per-file loop sees); `utt_per_s` = the same calls enqueued back to back (a loop over files without a host sync);
    launches per call and the per-kernel table of one call."""
    import torch
    res = {}
    for mode in modes:
        eng = make_engine("rawnet2", "f16", BATCH, local) if mode == "rawnet2_f16" else make_engine("ecapa", mode, BATCH, local)
        rows = {}
        for b in batches:
            w = wavs[0][:b].contiguous()
            out = torch.empty((b, eng.embed_dim), device=dev, dtype=torch.float32)
            for _ in range(3):
                eng.embed_wave(w, out=out, async_=True)
            torch.cuda.synchronize()
            n = 30
            t0 = time.perf_counter()
            for _ in range(n):
                eng.embed_wave(w, out=out, async_=True)
                torch.cuda.synchronize()
            t_sync = (time.perf_counter() - t0) / n
            t0 = time.perf_counter()
            for _ in range(n):
                eng.embed_wave(w, out=out, async_=True)
            torch.cuda.synchronize()
            t_pipe = (time.perf_counter() - t0) / n
            eng.profile(True)
            eng.embed_wave(w, out=out, async_=True)
            torch.cuda.synchronize()
            prof = eng.profile_results()
            eng.profile(False)
            rec = {"ms_per_call": t_sync * 1e3, "utt_per_s_sync": b / t_sync, "ms_per_call_pipelined": t_pipe * 1e3, "utt_per_s": b / t_pipe,
                   "launches_per_call": int(sum(v["launches"] for v in prof.values())),
                   "gpu_ms_in_kernels": sum(v["ms"] for v in prof.values()),
                   "kernels": {k: round(v["ms"] * 1e3, 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:8]}}
            if full_rate and mode == "bf16":
                rec["fraction_of_B256_rate"] = rec["utt_per_s"] / full_rate
            rows[str(b)] = rec
        res[mode] = rows
        eng.close()
    res["note"] = ("kernels: us per call of the eight longest labels (every launch bracketed by events: the bracketed call is slower than "
                   "ms_per_call); utt_per_s = pipelined calls, no host sync between them")
    return res


def multi_stream_bench(model, compute, B, local, dev, wavs, n_streams=3, steps=30):
    """serving pattern: `n_streams` engines on their own streams take alternate batches, so the small, under-filled kernels of one
    batch run beside the big ones of another (RawNet2's late blocks are grids of 86 - 400 workgroups).  Reported next to the
    single-stream number, never instead of it."""
    import torch
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    engs, outs = [], []
    for st in streams:
        with torch.cuda.stream(st):
            engs.append(make_engine(model, compute, B, local))
            outs.append(torch.empty((B, engs[-1].embed_dim), device=dev, dtype=torch.float32))
    torch.cuda.synchronize()
    dt = None
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            i = k % n_streams
            with torch.cuda.stream(streams[i]):
                engs[i].embed_wave(wavs[k % len(wavs)], out=outs[i], async_=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    ok = all(bool(torch.isfinite(o).all().item()) for o in outs)
    for e in engs:
        e.close()
    return {"value": steps * B / dt, "unit": "embeddings/s", "ms_per_step": dt / steps * 1e3, "streams": n_streams, "steps": steps, "finite": ok}


def fusion_bench(B, local, dev, wavs, steps=20):
    """the reference's production model Raw_ECAPA_sinc_asp — ECAPA-TDNN C = 512 (192-d, mel power without log,
    `features: raw`) and RawNet2 (320-d) on the SAME device-resident waveform batch, concatenated to 512-d.  `serial`: both
    engines on one stream; `two_streams`: one stream per branch (RawNet2's small late kernels run beside ECAPA's GEMMs)."""
    import torch
    from speakerverification_amd import synth
    from speakerverification_amd.engine import Engine
    res = {}
    for name, nst in (("serial", 1), ("two_streams", 2)):
        streams = [torch.cuda.Stream(device=dev) for _ in range(nst)]
        with torch.cuda.stream(streams[0]):
            ee = Engine(model="ecapa", compute="bf16", channels=512, embed_dim=192, max_batch=B, samples=SAMPLES, log_input=False,
                        device=local, stream=streams[0].cuda_stream)
            ee.load_state_dict(synth.synth_state_dict(synth.ecapa_param_spec(C=512), seed=1))
            ee.finalize()
        with torch.cuda.stream(streams[-1]):
            er = Engine(model="rawnet2", compute="f16", embed_dim=320, max_batch=B, samples=SAMPLES, device=local, stream=streams[-1].cuda_stream)
            er.load_state_dict(synth.synth_state_dict(synth.rawnet2_param_spec(nOut=320), seed=1))
            er.finalize()
        out = torch.empty((B, 512), device=dev, dtype=torch.float32)
        oe, orn = torch.empty((B, 192), device=dev, dtype=torch.float32), torch.empty((B, 320), device=dev, dtype=torch.float32)
        dt = None
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(steps):
                w = wavs[k % len(wavs)]
                with torch.cuda.stream(streams[0]):
                    ee.embed_wave(w, out=oe, async_=True)
                with torch.cuda.stream(streams[-1]):
                    er.embed_wave(w, out=orn, async_=True)
                if nst == 2:
                    streams[0].wait_stream(streams[1])
                with torch.cuda.stream(streams[0]):
                    out[:, :192].copy_(oe, non_blocking=True)          # torch.cat([out1, out2], dim=-1)  (Raw_ECAPA_sinc_asp.py:50)
                    out[:, 192:].copy_(orn, non_blocking=True)
                if nst == 2:
                    streams[1].wait_stream(streams[0])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        res[name] = {"value": steps * B / dt, "unit": "embeddings/s", "ms_per_step": dt / steps * 1e3, "finite": bool(torch.isfinite(out).all().item())}
        ee.close()
        er.close()
    res["config"] = {"model": "Raw_ECAPA_sinc_asp (ECAPA-TDNN C=512 -> 192-d on mel power, RawNet2 sinc/asp -> 320-d)", "batch": B, "dtype": "bf16 (ECAPA branch) + f16 (RawNet2 branch): hip_compute='half'"}
    return res


def pcie_bench(eng, dev, B):
    """PCIe-inclusive rate: decoded 16-bit PCM on the host -> int16 over PCIe -> device crop (svhip_crop_pcm16) -> embed.
    (Never `value`: BASELINE's metric is quoted on HBM-resident waveforms.)  Two forms: `serial` = one synchronous crop call per
    step from pageable memory on the crop engine's own stream (the embed of the previous batch, enqueued asynchronously, runs
    under it); `overlapped` = pinned PCM and SVHIP_ASYNC, the copy + crop of batch k+1 one batch ahead of the embed stream."""
    import numpy as np
    import torch
    from speakerverification_amd.engine import Engine, TRANSFER_STATS
    rng = np.random.Generator(np.random.PCG64(11))
    files = [np.clip(rng.standard_normal(48000) * 3276.8, -32768, 32767).astype(np.int16) for _ in range(B)]   # 3 s files, 1 crop each
    lens = np.full(B, 48000, np.int32)
    offs = np.arange(B, dtype=np.int64) * 48000
    packed = np.concatenate(files)
    s_embed = torch.cuda.current_stream()
    s_crop = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s_crop):
        crop_eng = Engine(model="none", device=dev.index, stream=s_crop.cuda_stream)
    crops = [torch.empty((B, SAMPLES), device=dev, dtype=torch.float32) for _ in range(2)]
    out = torch.empty((B, eng.embed_dim), device=dev, dtype=torch.float32)
    res = {"unit": "embeddings/s", "batch": B,
           "note": "per step: 256 x 3 s int16 files host->device, device crop (one 2 s crop per file), embed"}
    reps = 8
    # serial: synchronous crop call from pageable memory, then embed
    h0 = TRANSFER_STATS["h2d_bytes"]
    for it in range(reps + 1):
        if it == 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        with torch.cuda.stream(s_crop):
            crop_eng.crop_pcm16_packed(packed, offs, lens, 1, SAMPLES, out=crops[0])
        eng.embed_wave(crops[0], out=out, async_=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    res["serial"] = {"value": B / dt, "ms_per_step": dt * 1e3}
    res["h2d_bytes_per_step"] = (TRANSFER_STATS["h2d_bytes"] - h0) / (reps + 1)
    # overlapped: pinned PCM, crop stream one batch ahead of the embed stream
    pinned = torch.from_numpy(packed).pin_memory()
    ev_crop = [torch.cuda.Event() for _ in range(2)]
    ev_emb = [torch.cuda.Event() for _ in range(2)]
    for it in range(reps + 2):
        if it == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        i = it & 1
        with torch.cuda.stream(s_crop):
            if it >= 2:
                s_crop.wait_event(ev_emb[i])                  # the embed that read crops[i] two steps ago is done
            crop_eng.crop_pcm16_packed(pinned, offs, lens, 1, SAMPLES, out=crops[i], async_=True)
            ev_crop[i].record(s_crop)
        s_embed.wait_event(ev_crop[i])
        eng.embed_wave(crops[i], out=out, async_=True)
        ev_emb[i].record(s_embed)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    res["overlapped"] = {"value": B / dt, "ms_per_step": dt * 1e3}
    best = max(("serial", "overlapped"), key=lambda k: res[k]["value"])
    res["value"], res["ms_per_step"], res["best"] = res[best]["value"], res[best]["ms_per_step"], best
    res["finite"] = bool(torch.isfinite(out).all().item())
    crop_eng.close()
    return res


def verify_last_step(eng, wav_last, emb_last, local, dev):
    """the timed loop's LAST step: (a) bitwise equal to a re-run of the same batch (no race / stale buffer); (b) 8 of its rows
    against the exact-fp32-MFMA engine (the 1e-4-parity path): cosine and max error relative to the embedding scale."""
    import torch
    rerun = torch.empty_like(emb_last)
    eng.embed_wave(wav_last, out=rerun, async_=True)
    torch.cuda.synchronize()
    rec = {"bitwise_rerun": bool(torch.equal(rerun, emb_last)), "finite": bool(torch.isfinite(emb_last).all().item())}
    if eng.compute in ("bf16", "f16", "f32x3"):
        f32 = make_engine(eng.model, "f32", 8, local, embed=eng.embed_dim)
        ref = torch.empty((8, eng.embed_dim), device=dev, dtype=torch.float32)
        f32.embed_wave(wav_last[:8].contiguous(), out=ref, async_=True)
        torch.cuda.synchronize()
        f32.close()
        got = emb_last[:8]
        cos = torch.nn.functional.cosine_similarity(got, ref, dim=1)
        rec["rows_checked"] = 8
        rec["min_cosine_vs_f32_path"] = float(cos.min().item())
        rec["max_err_over_scale"] = float(((got - ref).abs().max() / ref.abs().max()).item())
        # 16-bit modes: cosine >= 0.999 and <= 3 % of the scale; f32x3: the 1e-4 parity bar (of the scale)
        bar_cos, bar_err = (0.999, 0.03) if eng.compute in ("bf16", "f16") else (0.999999, 1e-4)
        if eng.compute == "bf16" and eng.model == "rawnet2":
            bar_cos, bar_err = 0.99, 0.15           # RawNet2's bf16 mode is the range-safe fallback with loose, documented bars (its fast mode is f16)
        rec["bars"] = {"min_cosine": bar_cos, "max_err_over_scale": bar_err}
        rec["ok"] = rec["bitwise_rerun"] and rec["finite"] and rec["min_cosine_vs_f32_path"] >= bar_cos and rec["max_err_over_scale"] <= bar_err
    else:
        rec["ok"] = rec["bitwise_rerun"] and rec["finite"]
    return rec


def make_comm(eng, ranks, allow_gloo=False):
    """RCCL communicator under the C ABI.  If RCCL cannot come up the run FAILS (every rank exits non-zero) — a number measured over a
    gloo host round trip must not carry an N-GPU label by accident — unless --allow-gloo asks for the fallback, which the record then
    names in `allgather_carrier`."""
    if not ranks.launched:
        return None, "none (1 GPU, no launcher)"
    from speakerverification_amd import distributed as sv_dist
    try:
        if os.environ.get("SVHIP_BENCH_NO_RCCL"):       # developer hook: rehearse the fallback (e.g. 2 ranks on ONE GPU, which RCCL refuses)
            raise RuntimeError("SVHIP_BENCH_NO_RCCL is set")
        with quiet_stdout():
            comm = sv_dist.LibComm(eng, ranks.rank, ranks.world)
        ok = 1
    except Exception as e:  # noqa: BLE001 - any failure to bring RCCL up
        comm, ok = None, 0
        print(f"[bench] rank {ranks.rank}: RCCL communicator failed ({e!r}); falling back to gloo", file=sys.stderr)
    if ranks.min(ok) == 0:          # all ranks must agree on the carrier
        if comm is not None:
            eng.lib.svhip_comm_destroy(eng.h)
        if not allow_gloo:
            ranks.close()
            raise SystemExit("bench.py: the RCCL communicator could not be created on every rank and --allow-gloo was not given: "
                             "refusing to report a multi-GPU number over a host round trip")
        return None, "torch.distributed gloo all_gather (RCCL communicator could not be created on this node; --allow-gloo)"
    rw = eng.comm_rank_world()
    if rw != (ranks.rank, ranks.world):
        raise SystemExit(f"bench.py: the library's communicator reports rank/world {rw}, the launcher {(ranks.rank, ranks.world)}")
    return comm, "svhip_allgather_rows (RCCL under the C ABI)"


def gather_rows(eng, comm, ranks, local, out):
    """the path's exchange step: RCCL on the engine's stream, or (fallback) gloo through host memory"""
    import torch
    if comm is not None:
        eng.allgather_rows(local, out=out, async_=True)
    elif ranks.dist is not None and ranks.world > 1:
        torch.cuda.synchronize()
        host = torch.empty((ranks.world * local.shape[0], local.shape[1]), dtype=torch.float32)
        ranks.dist.all_gather_into_tensor(host, local.cpu())
        out.copy_(host)
    else:
        out.copy_(local)


def block_checksum(block):
    """order-independent, exact checksum of an fp32 block: the int64 sum of its words read as int32"""
    import torch
    return int(block.contiguous().view(torch.int32).to(torch.int64).sum().item())


def verify_gather(ranks, shard, gathered, n_local, regen_embed, rows=8, min_cos=0.9999):
    """Does the gathered matrix hold EVERY rank's block, and the right utterances in it?  (reference: all_gather_object,
    src/model.py:400-411.)  (a) every rank checksums its own block before the exchange; the checksums travel over the control plane
    (gloo) and are compared on every rank with the checksums of the blocks as they arrived: bit-exact, every byte of every block;
    (b) rank 0 regenerates the first `rows` utterances of every OTHER rank's block (the synthetic stream is a pure function of the
    utterance index), embeds them itself and compares with the gathered rows: cosine >= `min_cos` (a row of a 16-bit engine moves by
    round-off with its position in the batch).  Device-agnostic (tests drive it on CPU tensors over gloo at W = 3)."""
    import torch
    world, rank = ranks.world, ranks.rank
    mine = block_checksum(shard[:n_local])
    reported = ranks.gather_int64(mine)
    arrived = [block_checksum(gathered[r * n_local:(r + 1) * n_local]) for r in range(world)]
    rec = {"block_checksums": reported, "blocks_bitwise_ok": bool(arrived == reported)}
    worst, checked = 1.0, 0
    if rank == 0:
        for r in range(1, world):
            n = min(rows, n_local)
            ref = regen_embed(r, n)
            got = gathered[r * n_local:r * n_local + n]
            cos = torch.nn.functional.cosine_similarity(got.float(), ref.float().to(got.device), dim=1)
            worst = min(worst, float(cos.min().item()))
            checked += n
    rec["cross_rank_rows_checked"] = checked
    rec["cross_rank_min_cosine"] = worst if checked else None
    ok_here = rec["blocks_bitwise_ok"] and worst >= min_cos
    rec["cross_rank_ok"] = bool(ranks.min(1.0 if ok_here else 0.0) == 1.0)       # every rank's byte check and rank 0's content check
    return rec


class Ranks:
    """control plane of a multi-rank run: torch.distributed (gloo) for the RCCL id, barriers and the max over ranks;
    the data-path collective is RCCL under the C ABI (LibComm on an Engine)."""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("SVHIP_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # (override: developer rehearsals on one GPU)
        self.dist = None
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        self.launched = "WORLD_SIZE" in os.environ        # under a launcher the multi-rank code path runs even at world size 1
        if self.launched:
            import torch.distributed as dist
            with quiet_stdout():
                dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def min(self, x: float) -> float:
        if self.dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return float(t.item())

    def max(self, x: float) -> float:
        if self.dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_int64(self, x: int):
        """one integer per rank, on every rank (control plane)"""
        if self.dist is None:
            return [int(x)]
        import torch
        out = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
        self.dist.all_gather(out, torch.tensor([int(x)], dtype=torch.int64))
        return [int(t.item()) for t in out]

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))                       # nothing above touched the GPU
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible); the product path has no CPU fallback")
    ranks = Ranks(args)
    torch.cuda.set_device(ranks.local)
    dev = torch.device("cuda", ranks.local)
    # one torch stream carries everything (library kernels, RCCL all-gather, HIP events)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        if args.config == "shard":
            run_shard(args, ranks, dev)
        else:
            run_batch(args, ranks, dev)
    ranks.close()


def synth_batches(eng, n_batches, B, first_utt, dev, seed=SEED_STREAM):
    """n_batches distinct (B, L) waveform batches of the synthetic utterance stream, generated in HBM (svhip_synth_waveforms)"""
    import torch
    wavs = []
    for j in range(n_batches):
        w = torch.empty((B, SAMPLES), device=dev, dtype=torch.float32)
        eng.synth_waveforms(seed, first_utt + j * B, B, SAMPLES, out=w, async_=True)
        wavs.append(w)
    torch.cuda.synchronize()
    return wavs


def shard_tail(eng, comm, ranks, shard, dev, n_local, do_scoring=True, carrier="", regen_embed=None):
    """after embedding: ONE all-gather of the (n_local, D) block per rank, then config-4-style scoring of the gathered matrix,
    row-sharded by enrol index: rank r scores the trials (i, pi(i)) with i in its block, and computes the AS-norm cohort
    statistics of its own rows (gathered with a second, small all-gather of (n_local, 2))."""
    import torch
    from speakerverification_amd import distributed as sv_dist
    world, rank = ranks.world, ranks.rank
    D = shard.shape[1]
    rec = {}
    gathered = torch.empty((world * n_local, D), device=dev, dtype=torch.float32)
    ranks.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gather_rows(eng, comm, ranks, shard, gathered)
    torch.cuda.synchronize()
    rec["allgather_ms"] = ranks.max((time.perf_counter() - t0) * 1e3)
    rec["allgather_bytes_per_rank"] = n_local * D * 4
    rec["allgather_carrier"] = carrier
    rec["gathered_rows"] = world * n_local
    lo = rank * n_local
    own_block_ok = bool(torch.equal(gathered[lo:lo + n_local], shard))
    rec["own_block_intact"] = own_block_ok
    rec["rccl_world"] = eng.comm_rank_world()[1] if comm is not None else 0
    if regen_embed is not None:
        rec.update(verify_gather(ranks, shard, gathered, n_local, regen_embed))
    if not do_scoring:
        return rec, gathered
    N = world * n_local
    K, top = 5994, 200
    g = torch.Generator(device=dev).manual_seed(3)
    cohort = torch.randn((K, D), generator=g, device=dev, dtype=torch.float32)
    eng.l2norm_(cohort)
    gp = torch.Generator(device=dev).manual_seed(4)
    perm = torch.randperm(N, generator=gp, device=dev).to(torch.int32)             # the same permutation on every rank
    ia = torch.arange(lo, lo + n_local, device=dev, dtype=torch.int32)              # this rank's trials: enrol index in its block
    ib = perm[lo:lo + n_local].contiguous()
    out = torch.empty(n_local, device=dev, dtype=torch.float32)
    ranks.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.l2norm_(gathered)
    eng.score_pairs(gathered, ia, ib, out)
    torch.cuda.synchronize()
    t_cos = ranks.max(time.perf_counter() - t0)
    ranks.barrier()
    t0 = time.perf_counter()
    mu, sd = eng.asnorm_stats(gathered[lo:lo + n_local], cohort, top)               # row-sharded cohort GEMM + top-k
    stats_loc = torch.stack([mu, sd], dim=1).contiguous()
    stats_all = torch.empty((N, 2), device=dev, dtype=torch.float32)
    gather_rows(eng, comm, ranks, stats_loc, stats_all)
    mu_all, sd_all = stats_all[:, 0].contiguous(), stats_all[:, 1].contiguous()
    eng.asnorm_pairs(gathered, mu_all, sd_all, ia, ib, out)
    torch.cuda.synchronize()
    t_as = ranks.max(time.perf_counter() - t0)
    rec.update({"trials": N, "cohort": K, "top": top, "cosine_s": t_cos, "cosine_trials_per_s": N / t_cos,
                "asnorm_s": t_as, "asnorm_trials_per_s": N / t_as, "scores_finite": bool(torch.isfinite(out).all().item()),
                "scoring": "row-sharded by enrol index; stats of own rows + one (n_local, 2) all-gather"})
    return rec, gathered


def run_batch(args, ranks, dev):
    import torch
    from speakerverification_amd import distributed as sv_dist
    B, K, W = args.batch, args.steps, args.warmup
    rank, world, local = ranks.rank, ranks.world, ranks.local
    eng = make_engine(args.model, args.compute, B, local)
    embed = eng.embed_dim
    comm, carrier = make_comm(eng, ranks, args.allow_gloo)

    # synthetic waveforms, resident in HBM before the timed region: NBATCH distinct batches per rank, rotated
    wavs = synth_batches(eng, NBATCH, B, rank * NBATCH * B, dev)
    label = dominant_label(eng, wavs[0])
    shard = torch.empty((K * B, embed), device=dev, dtype=torch.float32)   # this rank's embeddings
    gathered = torch.empty((world * K * B, embed), device=dev, dtype=torch.float32) if ranks.launched else shard

    t0 = embed_loop(eng, wavs, K, W, B, shard, label, ranks.barrier)
    if ranks.launched:
        gather_rows(eng, comm, ranks, shard, gathered)             # the path's single exchange step (RCCL over xGMI)
    torch.cuda.synchronize()
    ranks.barrier()
    torch.cuda.synchronize()
    dt = ranks.max(time.perf_counter() - t0)
    prof = eng.profile_results()
    eng.profile(False)
    check = verify_last_step(eng, wavs[(K - 1) % NBATCH], shard[(K - 1) * B:K * B], local, dev)
    shard_rec = None
    if ranks.launched:
        def regen(r, n):          # the first n rows of rank r's block = its step 0 = utterances [r * NBATCH * B, ...) of the stream
            w = torch.empty((n, SAMPLES), device=dev, dtype=torch.float32)
            eng.synth_waveforms(SEED_STREAM, r * NBATCH * B, n, SAMPLES, out=w, async_=True)
            o = torch.empty((n, embed), device=dev, dtype=torch.float32)
            eng.embed_wave(w, out=o, async_=True)
            torch.cuda.synchronize()
            return o
        shard_rec, _ = shard_tail(eng, comm, ranks, shard, dev, K * B, carrier=carrier, regen_embed=regen)
    sustained = None
    if args.sustain_seconds > 0 and rank == 0 and world == 1:
        # the K timed steps above are ~0.1 s on a clock-limited chip: the same loop back to back for >= 2 s, reported next to it
        n_s = max(K, int(args.sustain_seconds / max(dt / K, 1e-6)) + 1)
        scratch = torch.empty((B, embed), device=dev, dtype=torch.float32)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for k in range(n_s):
            eng.embed_wave(wavs[k % NBATCH], out=scratch, async_=True)
        torch.cuda.synchronize()
        d_s = time.perf_counter() - ts
        sustained = {"seconds": d_s, "steps": n_s, "value": n_s * B / d_s, "unit": "embeddings/s", "ms_per_step": d_s / n_s * 1e3}
    n_all = min(K, 10)
    kern = kernel_table(eng, wavs, B, n_all, dev) if rank == 0 else {}

    if rank == 0:
        recs = Records(getattr(args, "record_file", None))
        total_utts = world * K * B
        line = {
            "metric": "embeddings/sec (2 s @16 kHz)", "value": total_utts / dt, "unit": "embeddings/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.compute, "data": "synthetic",
            "config": {"workload": ("ECAPA-TDNN C=1024 fbank+conv+ASP, batch=256 x 2 s @ 16 kHz per GPU per step "
                                    "(BASELINE configs[1]), HBM-resident waveforms -> 192-d embeddings") if args.model == "ecapa" else
                                   ("RawNet2 sinc front-end + 8 residual blocks + ASP, batch=256 x 2 s @ 16 kHz per GPU per step "
                                    "(BASELINE configs[2]), HBM-resident waveforms -> 320-d embeddings"),
                       "batch_per_gpu": B, "samples": SAMPLES, "frames": eng.frames, "embed_dim": embed,
                       "waveform_batches_rotated": NBATCH, "waveform_bytes_resident": NBATCH * B * SAMPLES * 4,
                       "collective": ("one all-gather of the shard embeddings inside the timed region: " + carrier) if ranks.launched else "none"},
            "finite": check["finite"], "check": check,
            "whole_path_TFLOPs": eng.flops_per_utterance * total_utts / dt / 1e12,
            "flops_per_utterance": eng.flops_per_utterance,
            "roofline": roofline_of(prof, label, args.compute),
        }
        recs.emit("kernels", {"kernels": kern,
                              "note": "per-kernel table from a separate profiled pass of %d steps (every launch bracketed); "
                                      "the roofline kernel is timed live inside the timed region" % n_all})
        if shard_rec is not None:
            line["shard"] = shard_rec
        line["sustained"] = sustained
        line["cpu_baseline"] = cpu_baseline() if (world == 1 and not args.no_cpu_baseline and args.model == "ecapa") else None
        sub, scoring = {}, None
        if world == 1 and not args.no_scoring and args.model == "ecapa":
            try:
                sc = scoring_bench(dev, with_cpu=not args.no_cpu_baseline)
                recs.emit("scoring", sc)
                scoring = {"cosine_pairs_per_s": sc["cosine_pairs_per_s"], "cosine_frac_of_hbm_peak": sc["cosine_pairs_frac_of_hbm_peak"],
                           "asnorm_pairs_per_s": sc["asnorm_pairs_per_s"], "asnorm_stats_ms": sc["asnorm_stats_s"] * 1e3,
                           "asnorm_frac": sc["asnorm_frac_of_split_ceiling"], "dense_pairs_per_s": sc["dense_pairs_per_s"],
                           "dense_frac": sc["dense_frac_of_split_ceiling"], "workload": "configs[3]: 1.2 M x 192, cohort 5994, top 200",
                           "asnorm_fallback_rows": sc.get("asnorm_fallback_rows")}
                for key, short in (("asnorm_speakers", "spk"), ("asnorm_speaker_groups", "grp")):      # speaker-structured embeddings
                    r = sc.get(key) or {}
                    scoring[f"asnorm_{short}_ms"] = r.get("asnorm_stats_s", 0.0) * 1e3
                    scoring[f"asnorm_{short}_slab_frac"] = r.get("slab_fraction")
                    scoring[f"asnorm_{short}_refit_rows"] = r.get("refit_rows")
                    scoring[f"asnorm_{short}_ok"] = r.get("ok")
                if sc.get("cpu_baseline"):
                    scoring["cpu_asnorm_loop_trials_per_s"] = sc["cpu_baseline"].get("asnorm_per_trial_loop_trials_per_s")
                    scoring["cpu_cosine_numpy_trials_per_s"] = sc["cpu_baseline"].get("cosine_numpy_batched_trials_per_s")
                if not args.no_extras:   # same workload on an f32x3 handle
                    x3 = scoring_bench(dev, with_cpu=False, compute="f32x3")
                    recs.emit("scoring_f32x3", {k: x3[k] for k in ("asnorm_pairs_per_s", "asnorm_stats_s", "asnorm_cohort_gemm_TFLOPs",
                                                                     "dense_pairs_per_s", "dense_TFLOPs", "config")})
            except Exception as e:  # scoring is reported next to, not inside, the headline
                recs.emit("scoring", {"error": repr(e)})
                scoring = {"error": repr(e)[:200]}
        if world == 1 and not args.no_extras and args.model == "ecapa" and args.compute == "bf16":
            for name, fn in (("rawnet2", lambda: sub_bench("rawnet2", "f16", B, local, dev, wavs)),
                             ("rawnet2_3_streams", lambda: multi_stream_bench("rawnet2", "f16", B, local, dev, wavs)),
                             ("rawnet2_bf16", lambda: sub_bench("rawnet2", "bf16", B, local, dev, wavs, steps=5, warmup=1)),
                             ("ecapa_f32", lambda: sub_bench("ecapa", "f32", B, local, dev, wavs, steps=3, warmup=1)),
                             ("ecapa_f32x3", lambda: sub_bench("ecapa", "f32x3", B, local, dev, wavs, steps=4, warmup=1)),
                             ("rawnet2_f32x3", lambda: sub_bench("rawnet2", "f32x3", B, local, dev, wavs, steps=4, warmup=1)),
                             ("latency", lambda: latency_bench(local, dev, wavs, full_rate=total_utts / dt)),
                             ("fusion", lambda: fusion_bench(B, local, dev, wavs)),
                             ("pcie", lambda: pcie_bench(eng, dev, B))):
                try:
                    rec = fn()
                except Exception as e:
                    rec = {"error": repr(e)}
                recs.emit(name, rec)
                if "error" in rec:
                    sub[name] = "error"
                elif name == "latency":
                    for mode, rows in rec.items():
                        if isinstance(rows, dict) and "20" in rows:
                            sub[f"latency_{mode}_B20_ms"] = rows["20"]["ms_per_call_pipelined"]
                            if "10" in rows:
                                sub[f"latency_{mode}_B10_ms"] = rows["10"]["ms_per_call_pipelined"]
                    if "bf16" in rec and "20" in rec["bf16"]:
                        sub["latency_bf16_B20_frac_of_B256_rate"] = rec["bf16"]["20"].get("fraction_of_B256_rate")
                elif name == "fusion":
                    sub["fusion_serial"], sub["fusion_two_streams"] = rec["serial"]["value"], rec["two_streams"]["value"]
                else:
                    sub[name] = rec["value"]
                    if isinstance(rec.get("check"), dict) and not rec["check"].get("ok", True):
                        sub[name + "_check"] = "FAILED"
        recs.emit("headline_full", line)
        recs.close()
        sys.stderr.flush()
        print(compact_headline(line, sub=sub, scoring=scoring), flush=True)
    eng.close()


def run_shard(args, ranks, dev, keep=False):
    """BASELINE configs[4] / SURVEY §8d config 5: a synthetic utterance list sharded in contiguous blocks (n_local per GPU, generated
    on the device from the counter-based stream — no waveform crosses PCIe), embedded in batches of 256, ONE all-gather of the
    (n_local, 192) fp32 block per rank, then config-4-style scoring of the gathered matrix, row-sharded by enrol index."""
    import torch
    from speakerverification_amd import distributed as sv_dist
    B = args.batch
    rank, world, local = ranks.rank, ranks.world, ranks.local
    n_local = args.utts_per_gpu
    eng = make_engine(args.model, args.compute, B, local)
    comm, carrier = make_comm(eng, ranks, args.allow_gloo)
    shard = torch.empty((n_local, eng.embed_dim), device=dev, dtype=torch.float32)
    wav = [torch.empty((B, SAMPLES), device=dev, dtype=torch.float32) for _ in range(2)]
    first = rank * n_local
    eng.synth_waveforms(SEED_SHARD, first, B, SAMPLES, out=wav[0], async_=True)
    label = dominant_label(eng, wav[0])
    for w in range(args.warmup):
        eng.synth_waveforms(SEED_SHARD, first, B, SAMPLES, out=wav[0], async_=True)
        eng.embed_wave(wav[0], out=shard[:B], async_=True)
    torch.cuda.synchronize()
    eng.profile(True, only=label)
    ranks.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 0
    for u0 in range(0, n_local, B):
        n = min(B, n_local - u0)
        buf = wav[steps & 1]
        eng.synth_waveforms(SEED_SHARD, first + u0, n, SAMPLES, out=buf[:n], async_=True)     # generated in HBM, same stream
        eng.embed_wave(buf[:n], out=shard[u0:u0 + n], async_=True)
        steps += 1
    torch.cuda.synchronize()
    ranks.barrier()
    torch.cuda.synchronize()
    dt = ranks.max(time.perf_counter() - t0)
    prof = eng.profile_results()
    eng.profile(False)
    def regen(r, n):              # the first n utterances of rank r's block of the SEED_SHARD stream
        w = torch.empty((n, SAMPLES), device=dev, dtype=torch.float32)
        eng.synth_waveforms(SEED_SHARD, r * n_local, n, SAMPLES, out=w, async_=True)
        o = torch.empty((n, eng.embed_dim), device=dev, dtype=torch.float32)
        eng.embed_wave(w, out=o, async_=True)
        torch.cuda.synchronize()
        return o
    rec, _ = shard_tail(eng, comm, ranks, shard, dev, n_local, carrier=carrier, regen_embed=regen if ranks.launched else None)
    line = None
    if rank == 0:
        total = world * n_local
        line = {"metric": "embeddings/sec (2 s @16 kHz)", "value": total / dt, "unit": "embeddings/s", "n_gpus": world,
                "steps": steps, "warmup": args.warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": args.compute, "data": "synthetic",
                "config": {"workload": "ECAPA-TDNN C=1024 bf16, synthetic utterance list sharded across the GPUs (BASELINE configs[4]: "
                                       "125 000 x 2 s @ 16 kHz per GPU = 1 M on 8), waveforms generated on the device (Philox stream, inside "
                                       "the timed region), one RCCL all-gather of embeddings, then row-sharded cosine + AS-norm scoring",
                           "utterances_per_gpu": n_local, "batch": B, "embed_dim": eng.embed_dim},
                "embed_seconds": dt, "finite": bool(torch.isfinite(shard).all().item()),
                "whole_path_TFLOPs": eng.flops_per_utterance * total / dt / 1e12,
                "roofline": roofline_of(prof, label, args.compute), "shard": rec,
                "end_to_end_seconds": dt + rec["allgather_ms"] * 1e-3 + rec.get("cosine_s", 0.0) + rec.get("asnorm_s", 0.0),
                "cpu_baseline": None}
        recs = Records(getattr(args, "record_file", None))
        recs.emit("headline_full", line)
        recs.close()
        sys.stderr.flush()
        print(compact_headline(line), flush=True)
    eng.close()
    return (line, shard) if keep else None           # (tests/test_gpu_fullsize.py checks the record and the embeddings themselves)


if __name__ == "__main__":
    main()
