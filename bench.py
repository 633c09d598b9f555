#!/usr/bin/env python
"""bench.py — throughput of the speaker-embedding hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): ECAPA-TDNN C=1024, bf16 MFMA / fp32 accumulate, batch = 256
utterances of 2 s @ 16 kHz per step, waveforms resident in HBM, one step = fbank -> ECAPA forward
-> (256, 192) embeddings.  With N > 1 every rank embeds its own 256 utterances per step (weak
scaling, no data-path collective) and the shard embeddings are assembled with ONE RCCL all-gather
at the end of the timed region (reference: all_gather_object, src/model.py:400-404).

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the field contract): `value` is whole-job
embeddings/s; `roofline` is the dominant kernel (the pointwise-conv MFMA GEMM) timed with HIP
events on the launch stream inside the timed region; `cpu_baseline` is the CPU oracle timed on
this box's host cores on a bounded sample (rank 0, N = 1 only); `scoring` reports trial-pairs/s.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from speakerverification_amd import synth  # noqa: E402
from speakerverification_amd.engine import Engine  # noqa: E402

BATCH = 256
SAMPLES = 32000
CHANNELS = 1024
EMBED = 192
PEAK_BF16_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
PEAK_F32_TFLOPS = 157.3
DOMINANT = "gemm_pw2"          # gemm_pw2_kernel<EPI_GELU>: tdnn1/tdnn2 x3 + mfa (the 256 x 256 bf16 pointwise-conv GEMM)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--compute", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--model", default="ecapa", choices=["ecapa", "rawnet2"], help="ecapa = headline (configs[1]); rawnet2 = configs[2]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scoring", action="store_true")
    return ap.parse_args()


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


def cpu_baseline(sample_utts=8, reps=3):
    """CPU oracle (oracle/: torch-CPU restatement pinned against the reference) on a bounded sample of
    the same workload: `sample_utts` waveforms through fbank + ECAPA C=1024 fp32, all host cores."""
    from oracle import ecapa as o_ecapa, fbank as o_fbank
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = o_ecapa.to_torch_sd(synth.synth_state_dict(synth.ecapa_param_spec(C=CHANNELS), seed=1))
    wav = torch.from_numpy(synth.synth_waveforms(sample_utts, SAMPLES))
    with torch.no_grad():
        o_ecapa.ecapa_forward(o_fbank.melspectrogram(wav[:2]), sd)       # warm-up
        t0 = time.perf_counter()
        n = 0
        while n < reps or time.perf_counter() - t0 < 10.0:
            o_ecapa.ecapa_forward(o_fbank.melspectrogram(wav), sd)
            n += 1
            if time.perf_counter() - t0 > 30.0:
                break
        dt = time.perf_counter() - t0
    return {"value": n * sample_utts / dt, "unit": "embeddings/s", "cores": cores, "kind": "port",
            "sample": f"{n} x {sample_utts} utterances (2 s @ 16 kHz), fbank + ECAPA-TDNN C=1024 fp32, torch-CPU oracle, {dt:.1f} s"}


def scoring_bench(dev):
    """BASELINE config 4: 1.2 M synthetic 192-d embeddings, 1.2 M-trial list, cohort 5994, top 200."""
    assert torch.cuda.current_stream().cuda_stream != 0
    eng = Engine(model="none", device=dev.index, stream=torch.cuda.current_stream().cuda_stream)
    N, P, K, top = 1_200_000, 1_200_000, 5994, 200
    g = torch.Generator(device=dev).manual_seed(2)
    E = torch.randn((N, EMBED), generator=g, device=dev, dtype=torch.float32)
    eng.l2norm_(E)
    cohort = torch.randn((K, EMBED), generator=g, device=dev, dtype=torch.float32)
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
    mu_sd = {}

    def stats():
        mu_sd["v"] = eng.asnorm_stats(E, cohort, top)
    t_stats = timed(stats, reps=1)
    mu, sd = mu_sd["v"]
    t_pairs = timed(lambda: eng.asnorm_pairs(E, mu, sd, ia, ib, out))
    res["asnorm_pairs_per_s"] = P / (t_stats + t_pairs)
    res["asnorm_stats_s"] = t_stats
    res["asnorm_cohort_gemm_TFLOPs"] = 2.0 * N * K * EMBED / t_stats / 1e12
    A, Bm = E[:16384], E[16384:32768]
    dense = torch.empty((16384, 16384), device=dev, dtype=torch.float32)
    t = timed(lambda: eng.score_matrix(A, Bm, dense))
    res["dense_pairs_per_s"] = 16384 * 16384 / t
    res["dense_TFLOPs"] = 2.0 * 16384 * 16384 * EMBED / t / 1e12
    # verification metrics over the same 1.2 M-trial list (EER / minDCF inputs; host arrays in, PCIe included)
    sc_host = out.cpu().numpy()
    lab_host = (np.arange(P) % 2).astype(np.int32)
    t = timed(lambda: eng.min_dcf(sc_host, lab_host, 0.05, 1, 1))
    res["min_dcf_trials_per_s"] = P / t
    t = timed(lambda: eng.roc_points(sc_host, lab_host))
    res["roc_points_trials_per_s"] = P / t
    res["config"] = {"embeddings": N, "trials": P, "cohort": K, "top": top, "dim": EMBED}
    eng.close()
    return res


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible); the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    # one torch stream carries everything (library kernels, RCCL all-gather, HIP events)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        run(args, rank, world, local, dev, dist)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run(args, rank, world, local, dev, dist):
    B, K, W = args.batch, args.steps, args.warmup
    global EMBED, DOMINANT
    if args.compute == "f32" and args.model == "ecapa":
        DOMINANT = "gemm_pw"
    if args.model == "rawnet2":
        EMBED, DOMINANT = 320, "gemm_conv"
        eng = Engine(model="rawnet2", compute=args.compute, embed_dim=EMBED, max_batch=B, samples=SAMPLES, device=local,
                     stream=torch.cuda.current_stream().cuda_stream)
        eng.load_state_dict(synth.synth_state_dict(synth.rawnet2_param_spec(nOut=EMBED), seed=1))
    else:
        eng = Engine(model="ecapa", compute=args.compute, channels=CHANNELS, embed_dim=EMBED, max_batch=B,
                     samples=SAMPLES, device=local, stream=torch.cuda.current_stream().cuda_stream)
        eng.load_state_dict(synth.synth_state_dict(synth.ecapa_param_spec(C=CHANNELS), seed=1))
    eng.finalize()

    # synthetic waveforms, resident in HBM before the timed region (rank-dependent seed)
    wav = torch.from_numpy(synth.synth_waveforms(B, SAMPLES, seed=20220829 + rank)).to(dev)
    shard = torch.empty((K * B, EMBED), device=dev, dtype=torch.float32)   # this rank's embeddings
    gathered = torch.empty((world * K * B, EMBED), device=dev, dtype=torch.float32) if world > 1 else shard
    scratch = torch.empty((B, EMBED), device=dev, dtype=torch.float32)

    for _ in range(W):
        eng.embed_wave(wav, out=scratch, async_=True)
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    # HIP events bracket the dominant kernel's launches on the launch stream inside the timed region (resolved after it);
    # the other kernels are timed in a separate profiled pass below: two events per launch on ~40 launches per step
    # cost ~2 % of the step in queue bubbles
    eng.profile(True, only=DOMINANT)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        eng.embed_wave(wav, out=shard[k * B:(k + 1) * B], async_=True)
    if dist is not None:
        dist.all_gather_into_tensor(gathered, shard)       # the path's single exchange step (RCCL over xGMI)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = eng.profile_results()
    eng.profile(True)            # untimed pass: every kernel bracketed, for the per-kernel table
    for k in range(min(K, 10)):
        eng.embed_wave(wav, out=scratch, async_=True)
    torch.cuda.synchronize()
    prof_all = eng.profile_results()
    n_all = min(K, 10)
    eng.profile(False)

    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        ok = bool(torch.isfinite(shard).all().item())
        peak = PEAK_BF16_TFLOPS if args.compute == "bf16" else PEAK_F32_TFLOPS
        dom = prof.get(DOMINANT, {"ms": 0.0, "launches": 0, "flops": 0.0})
        avg_ms = dom["ms"] / max(1, dom["launches"])
        achieved = (dom["flops"] / max(1, dom["launches"])) / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):
            try:
                traffic = json.load(open(pmc_path)).get(DOMINANT, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        kern = {k: {"avg_ms": v["ms"] / max(1, v["launches"]), "launches_per_step": v["launches"] / n_all,
                    "ms_per_step": v["ms"] / n_all,
                    "TFLOPs": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and v["flops"] > 0 else None}
                for k, v in prof_all.items()}
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
                       "batch_per_gpu": B, "samples": SAMPLES, "frames": eng.frames, "embed_dim": EMBED,
                       "collective": "one all_gather_into_tensor of the shard embeddings" if world > 1 else "none"},
            "finite": ok,
            "whole_path_TFLOPs": eng.flops_per_utterance * total_utts / dt / 1e12,
            "flops_per_utterance": eng.flops_per_utterance,
            "roofline": {"kernel": DOMINANT, "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "avg_launch_ms": avg_ms,
                         "launches": dom["launches"]},
            "kernels": kern,
            "kernels_note": "per-kernel table from a separate profiled pass of %d steps (every launch bracketed); "
                            "the roofline kernel is timed live inside the timed region" % n_all,
        }
        if world == 1 and not args.no_cpu_baseline and args.model == "ecapa":
            line["cpu_baseline"] = cpu_baseline()
        else:
            line["cpu_baseline"] = None
        if world == 1 and not args.no_scoring and args.model == "ecapa":
            try:
                line["scoring"] = scoring_bench(dev)
            except Exception as e:  # scoring is reported next to, not inside, the headline
                line["scoring"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
