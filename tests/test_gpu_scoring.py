"""GPU parity: scoring kernels through the C ABI vs reference-pinned golden values and the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import scoring as o_scoring
from speakerverification_amd import synth
from speakerverification_amd.engine import Engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    return Engine(model="none", max_batch=1)


def test_l2norm_and_cosine_pairs(eng):
    rng = np.random.Generator(np.random.PCG64(9))
    E = rng.standard_normal((1000, 192)).astype(np.float32) * 3.0
    E[7] = 0.0                                        # zero row: eps clamps, no NaN
    ia = rng.integers(0, 1000, 5000).astype(np.int32)
    ib = rng.integers(0, 1000, 5000).astype(np.int32)
    got = eng.score_pairs(E, ia, ib)
    want = o_scoring.cosine_pairs(E, ia, ib)
    assert np.all(np.isfinite(got))
    assert float(np.abs(got - want).max()) <= 1e-5
    En = E.copy()
    eng.l2norm_(En)
    ref = torch.nn.functional.normalize(torch.from_numpy(E), p=2, dim=1).numpy()
    assert float(np.abs(En - ref).max()) <= 1e-6
    # empty pair list
    assert eng.score_pairs(E, ia[:0], ib[:0]).shape == (0,)


def test_golden_cosine_one_crop(eng, golden_dir):
    """reference utils.cosine_similarity on single-crop inputs == |cos| of the pair kernel."""
    g = np.load(os.path.join(golden_dir, "scoring.npz"))
    R, Cm = g["R"][:, 0, :], g["C"][:, 0, :]
    E = np.concatenate([R, Cm]).astype(np.float32)
    n = R.shape[0]
    ia, ib = np.arange(n, dtype=np.int32), np.arange(n, 2 * n, dtype=np.int32)
    got = eng.score_pairs(E, ia, ib)
    want = np.array([o_scoring.cosine_similarity(torch.from_numpy(R[i:i + 1]), torch.from_numpy(Cm[i:i + 1])) for i in range(n)])
    assert float(np.abs(got - want).max()) <= 1e-5


def test_golden_asnorm(eng, golden_dir):
    """reference utils.ZT_norm_similarity (multi-crop) == GEMM form on crop means."""
    g = np.load(os.path.join(golden_dir, "scoring.npz"))
    top = int(g["top"])
    cohort = g["cohort"]
    Rn = torch.nn.functional.normalize(torch.from_numpy(g["R"]), p=2, dim=2).numpy()
    Cn = torch.nn.functional.normalize(torch.from_numpy(g["C"]), p=2, dim=2).numpy()
    n = Rn.shape[0]
    E = np.concatenate([Rn.mean(axis=1), Cn.mean(axis=1)]).astype(np.float32)     # crop means
    ia, ib = np.arange(n, dtype=np.int32), np.arange(n, 2 * n, dtype=np.int32)
    mu, sd = eng.asnorm_stats(E, cohort, top)
    got = eng.asnorm_pairs(E, mu, sd, ia, ib)
    assert float(np.abs(got - g["zt_norm"]).max()) <= 1e-4, np.abs(got - g["zt_norm"]).max()
    # default top=-1 of the reference (drops the smallest cohort score)
    mu, sd = eng.asnorm_stats(E, cohort, -1)
    got = eng.asnorm_pairs(E, mu, sd, ia, ib)
    assert float(np.abs(got - g["zt_norm_default_top"]).max()) <= 1e-4


def test_asnorm_and_dense_scores_on_a_split_bf16_handle(golden_dir):
    """compute="f32x3" handles run the cohort / dense score GEMMs as three bf16 MFMAs per product on hi / lo-split fp32 operands
    (csrc/gemm_pw.hip X3, ~2^-17 per product).  The cohort moments hold 2e-6 / 1e-4 relative and raw cosine scores 2e-6, but
    AS-norm divides by sd ~ 0.05, so normalised scores of O(10) carry ~1e-5 RELATIVE error: the stated bar for this opt-in mode is
    2e-4 absolute on the golden AS-norm scores (measured 1.4e-4); the exact-fp32 handle (the default) keeps 1e-4."""
    x3 = Engine(model="none", max_batch=1, compute="f32x3")
    g = np.load(os.path.join(golden_dir, "scoring.npz"))
    top = int(g["top"])
    Rn = torch.nn.functional.normalize(torch.from_numpy(g["R"]), p=2, dim=2).numpy()
    Cn = torch.nn.functional.normalize(torch.from_numpy(g["C"]), p=2, dim=2).numpy()
    n = Rn.shape[0]
    E = np.concatenate([Rn.mean(axis=1), Cn.mean(axis=1)]).astype(np.float32)
    ia, ib = np.arange(n, dtype=np.int32), np.arange(n, 2 * n, dtype=np.int32)
    mu, sd = x3.asnorm_stats(E, g["cohort"], top)
    got = x3.asnorm_pairs(E, mu, sd, ia, ib)
    assert float(np.abs(got - g["zt_norm"]).max()) <= 2e-4
    assert float((np.abs(got - g["zt_norm"]) / np.abs(g["zt_norm"])).max()) <= 2e-5
    E = synth.synth_embeddings(300, seed=21)
    cohort = synth.synth_embeddings(5994, seed=22)
    mu, sd = x3.asnorm_stats(E, cohort, 200)
    rmu, rsd = o_scoring.asnorm_stats(E, cohort, 200)
    print("x3 mu err", float(np.abs(mu - rmu).max()), "sd rel", float(np.abs(sd - rsd).max() / rsd.min()))
    assert float(np.abs(mu - rmu).max()) <= 2e-6
    assert float(np.abs(sd - rsd).max() / rsd.min()) <= 1e-4
    A, B = synth.synth_embeddings(70, seed=1), synth.synth_embeddings(130, seed=2)
    assert float(np.abs(x3.score_matrix(A, B) - A.astype(np.float64) @ B.astype(np.float64).T).max()) <= 2e-6
    x3.close()


@pytest.mark.parametrize("K,top", [(5994, 200), (257, 200), (64, 64), (1000, 1)])
def test_asnorm_stats_vs_oracle(eng, K, top):
    E = synth.synth_embeddings(300, seed=21)
    cohort = synth.synth_embeddings(K, seed=22)
    cohort[3] = cohort[5]                              # ties in the cohort scores
    mu, sd = eng.asnorm_stats(E, cohort, top)
    rmu, rsd = o_scoring.asnorm_stats(E, cohort, top)
    assert float(np.abs(mu - rmu).max()) <= 1e-6
    if top > 1:
        assert float(np.abs(sd - rsd).max() / rsd.min()) <= 1e-4
    else:
        assert float(np.abs(sd).max()) <= 1e-6


def test_score_matrix(eng):
    A = synth.synth_embeddings(300, seed=31)
    B = synth.synth_embeddings(517, seed=32)
    got = eng.score_matrix(A, B)
    want = A.astype(np.float64) @ B.astype(np.float64).T
    assert got.shape == (300, 517)
    assert float(np.abs(got - want).max()) <= 1e-6


@pytest.mark.parametrize("Na,Nb,D", [(300, 517, 192), (1, 5, 192), (129, 64, 256), (1000, 4096, 192), (260, 33, 256), (70, 130, 64)])
def test_score_matrix_row_streaming_kernel_against_the_tiled_one(Na, Nb, D):
    """Round 4: D = 192 / 256 score matrices run on score_h3w (rows of A in registers as half hi | lo parts, B streamed past them as half
    planes, three fp16 16x16x32 MFMAs per product block, scores stored from the accumulators); option score_tiled keeps the tiled split
    GEMM.  Both against the float64 product: ragged sizes (single rows, Nb not a multiple of 4 or 32, several column slices), host and
    device operands; D = 64 takes the tiled kernel either way."""
    import torch
    eng = Engine(model="none", max_batch=1)
    rng = np.random.Generator(np.random.PCG64(Na * 7 + Nb))
    A = rng.standard_normal((Na, D)).astype(np.float32); A /= np.linalg.norm(A, axis=1, keepdims=True)
    B = rng.standard_normal((Nb, D)).astype(np.float32); B /= np.linalg.norm(B, axis=1, keepdims=True)
    want = A.astype(np.float64) @ B.astype(np.float64).T
    got = eng.score_matrix(A, B)
    eng.set_option("score_tiled", 1)
    tiled = eng.score_matrix(A, B)
    eng.set_option("score_tiled", 0)
    assert got.shape == (Na, Nb)
    print(f"score_matrix {Na} x {Nb} x {D}: row-streaming vs f64 {np.abs(got - want).max():.2e}, tiled vs f64 {np.abs(tiled - want).max():.2e}")
    assert float(np.abs(got - want).max()) <= 2e-7 and float(np.abs(tiled - want).max()) <= 1e-6
    Ad, Bd = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    out = torch.full((Na, Nb), float("nan"), device="cuda")
    eng.score_matrix(Ad, Bd, out)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), got)
    eng.close()


def test_large_pair_list_properties():
    """full-size (BASELINE config 4 scale) invariants: symmetry, self-pairs == 1, device pointers."""
    eng = Engine(model="none", max_batch=1)
    dev = torch.device("cuda", 0)
    N = 1_200_000
    g = torch.Generator(device=dev).manual_seed(2)
    E = torch.randn((N, 192), generator=g, device=dev)
    eng.l2norm_(E)
    ia = torch.arange(N, device=dev, dtype=torch.int32)
    ib = torch.randperm(N, generator=g, device=dev).to(torch.int32)
    ab = eng.score_pairs(E, ia, ib)
    ba = eng.score_pairs(E, ib, ia)
    aa = eng.score_pairs(E, ia, ia)
    assert torch.equal(ab, ba)
    assert float((aa - 1).abs().max()) <= 1e-6
    ref = (E[ia.long()[:4096]] * E[ib.long()[:4096]]).sum(1).abs()
    assert float((ab[:4096] - ref).abs().max()) <= 1e-6
    eng.close()


def test_device_cropping_matches_reference_semantics(eng):
    """svhip_crop_pcm16 == loadWAV's eval-mode crops of 16-bit files (oracle crop_eval on x/32768), bit for bit."""
    rng = np.random.Generator(np.random.PCG64(99))
    files = [rng.integers(-20000, 20000, n).astype(np.int16) for n in (32000, 48000, 20000, 64000, 32001, 31999, 90011)]
    for ne in (1, 2, 3, 10):
        got = eng.crop_pcm16(files, ne, 32000).reshape(len(files), ne, 32000)
        for f, a in enumerate(files):
            want = o_scoring.crop_eval(a.astype(np.float32) / 32768.0, 32000, ne, peak_normalize=False)
            assert np.array_equal(got[f], want), (f, ne)


@pytest.mark.parametrize("N,D,P,K,top", [(1, 192, 1, 3, 2), (7, 64, 5, 64, 64), (1000, 256, 999, 300, 200), (33, 192, 64, 5, 200),
                                         (4097, 96, 1, 201, -1)])
def test_scoring_size_sweep(eng, N, D, P, K, top):
    """Odd sizes through every scoring entry point (one embedding, one trial, cohorts smaller than `top`, top = -1, embedding
    widths other than 192): results against the numpy statements, and above all no out-of-bounds access."""
    rng = np.random.Generator(np.random.PCG64(N + D + P + K))
    E = rng.standard_normal((N, D)).astype(np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    cohort = rng.standard_normal((K, D)).astype(np.float32)
    cohort /= np.linalg.norm(cohort, axis=1, keepdims=True)
    ia = rng.integers(0, N, P).astype(np.int32)
    ib = rng.integers(0, N, P).astype(np.int32)
    got = eng.score_pairs(E, ia, ib)
    assert float(np.abs(got - np.abs((E[ia] * E[ib]).sum(1))).max()) <= 1e-5
    M = eng.score_matrix(E[: min(N, 50)], cohort)
    assert float(np.abs(M - E[: min(N, 50)] @ cohort.T).max()) <= 1e-4
    mu, sd = eng.asnorm_stats(E, cohort, top)
    S = np.sort(E @ cohort.T, axis=1)[:, ::-1]
    kk = K + top if top < 0 else min(top, K)
    S = S[:, :kk]
    assert float(np.abs(mu - S.mean(1)).max()) <= 1e-4 and float(np.abs(sd - S.std(1)).max()) <= 1e-4
    sc = eng.asnorm_pairs(E, mu, sd, ia, ib)
    raw = (E[ia] * E[ib]).sum(1)
    want = 0.5 * ((raw - mu[ia]) / sd[ia] + (raw - mu[ib]) / sd[ib])
    assert float(np.abs(sc - want).max()) <= 2e-3 * max(1.0, float(np.abs(want).max()))


@pytest.mark.parametrize("N,D,K,top", [(517, 192, 5994, 200), (130, 256, 1000, 50), (1000, 256, 2000, 256), (64, 192, 801, 200),
                                       (3, 192, 64, 1), (129, 192, 5995, 200)])
def test_fused_asnorm_kernel_shapes(eng, N, D, K, top):
    """csrc/asnorm_fused.hip (scores selected in the MFMA accumulators, never stored): every embedding width it is built for,
    ragged N / K (masked rows of the last workgroup, masked cohort rows of the last block), top at both ends of its range."""
    rng = np.random.Generator(np.random.PCG64(N * 7 + D + K))
    E = rng.standard_normal((N, D)).astype(np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    cohort = rng.standard_normal((K, D)).astype(np.float32)
    cohort /= np.linalg.norm(cohort, axis=1, keepdims=True)
    cohort[3] = cohort[5]                              # ties
    mu, sd = eng.asnorm_stats(E, cohort, top)
    assert eng.asnorm_last_fallback >= 0               # the fused path ran
    rmu, rsd = o_scoring.asnorm_stats(E, cohort, top)
    assert float(np.abs(mu - rmu).max()) <= 1e-6
    if top > 1:
        assert float(np.abs(sd - rsd).max() / rsd.min()) <= 1e-4


def test_fused_asnorm_hands_odd_embeddings_to_the_slab_path(eng):
    """The fused kernel's threshold assumes roughly normal cohort scores.  Embeddings for which it keeps fewer than `top`
    candidates (a cohort of near-duplicates: no spread) or overflows a candidate list (a heavy upper tail) must be flagged and
    answered by the slab path — same numbers as the oracle either way."""
    rng = np.random.Generator(np.random.PCG64(77))
    D, K, top = 192, 2000, 200
    cohort = rng.standard_normal((K, D)).astype(np.float32)
    cohort /= np.linalg.norm(cohort, axis=1, keepdims=True)
    E = rng.standard_normal((300, D)).astype(np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    # heavy upper tail: 700 cohort rows are copies of one direction (plus noise); embeddings near that direction see 700 high scores
    hot = cohort[0].copy()
    noise = rng.standard_normal((700, D)).astype(np.float32) * 0.02
    cohort[100:800] = hot + noise
    cohort /= np.linalg.norm(cohort, axis=1, keepdims=True)
    E[10] = hot
    E[11] = hot + 0.05 * E[11]
    E[11] /= np.linalg.norm(E[11])
    E[12] = 0.0                                         # every score equal (zero): no candidate is above the threshold
    mu, sd = eng.asnorm_stats(E, cohort, top)
    assert eng.asnorm_last_fallback >= 1, eng.asnorm_last_fallback
    rmu, rsd = o_scoring.asnorm_stats(E, cohort, top)
    assert float(np.abs(mu - rmu).max()) <= 2e-6
    ok = rsd > 1e-6
    assert float((np.abs(sd - rsd)[ok] / rsd[ok]).max()) <= 1e-4
    assert float(np.abs(sd[~ok]).max(initial=0.0)) <= 1e-6


def test_fused_asnorm_at_scale():
    """BASELINE config 4's shape on the device: 262 144 embeddings (two 131 072-row launches: the chunking of the candidate buffer)
    against 5 994 cohort speakers, top 200; sampled rows against the float64 oracle, and the slab path as a second opinion."""
    eng = Engine(model="none", max_batch=1)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(5)
    N, K, D, top = 262_144 + 77, 5994, 192, 200
    E = torch.randn((N, D), generator=g, device=dev)
    eng.l2norm_(E)
    cohort = torch.randn((K, D), generator=g, device=dev)
    eng.l2norm_(cohort)
    mu, sd = eng.asnorm_stats(E, cohort, top)
    assert eng.asnorm_last_fallback == 0
    assert bool(torch.isfinite(mu).all()) and bool(torch.isfinite(sd).all())
    idx = np.r_[0:64, 131_000:131_200, N - 100:N]
    rmu, rsd = o_scoring.asnorm_stats(E[idx].cpu().numpy(), cohort.cpu().numpy(), top)
    assert float(np.abs(mu[idx].cpu().numpy() - rmu).max()) <= 1e-6
    assert float(np.abs(sd[idx].cpu().numpy() - rsd).max() / rsd.min()) <= 1e-4
    # the TWO-STREAM form of the fused path (candidate statistics of chunk c on a second stream under the MFMA kernel of
    # chunk c + 1: aux stream, four events, two candidate buffers) runs only for N > 131 072 with the fp32-MFMA form — three chunks here
    eng.set_option("asnorm_f32mfma", 1)
    mu3, sd3 = eng.asnorm_stats(E, cohort, top)
    eng.set_option("asnorm_f32mfma", 0)
    assert eng.asnorm_last_fallback == 0
    assert float((mu3 - mu).abs().max()) <= 5e-7 and float((sd3 - sd).abs().max()) <= 5e-7
    assert float(np.abs(mu3[idx].cpu().numpy() - rmu).max()) <= 1e-6
    eng.set_option("asnorm_slab", 1)            # (the environment is only read when a handle is created)
    mu2, sd2 = eng.asnorm_stats(E[:50_000], cohort, top)
    eng.set_option("asnorm_slab", 0)
    assert eng.asnorm_last_fallback == -1
    assert float((mu2 - mu[:50_000]).abs().max()) <= 1e-6 and float((sd2 - sd[:50_000]).abs().max()) <= 1e-6
    eng.close()


@pytest.mark.parametrize("n_files,n_crops,D,P", [(6, 3, 192, 40), (40, 10, 192, 500), (3, 1, 64, 7), (5, 2, 100, 16)])
def test_whole_trial_scoring_modes(eng, n_files, n_crops, D, P):
    """svhip_score_trials / svhip_mean_crops against the reference's per-trial expressions (torch on the CPU): cosine and pnorm
    over aligned crops (utils.py:163-169), the cohorts_path=None pairwise distance over the (n, D, n) broadcast
    (model.py:425-431), crop means; host arrays and device tensors."""
    import torch.nn.functional as TF
    rng = np.random.Generator(np.random.PCG64(n_files * 100 + n_crops))
    F = rng.standard_normal((n_files, n_crops, D)).astype(np.float32)
    F /= np.linalg.norm(F, axis=2, keepdims=True)
    ia = rng.integers(0, n_files, P).astype(np.int32)
    ib = rng.integers(0, n_files, P).astype(np.int32)
    ib[0] = ia[0]
    t = torch.from_numpy(F)
    want = {
        "cosine": [o_scoring.cosine_similarity(t[a], t[b]) for a, b in zip(ia, ib)],
        "pnorm": [o_scoring.pnorm_similarity(t[a], t[b]) for a, b in zip(ia, ib)],
        "pdist": [-float(torch.mean(TF.pairwise_distance(t[a].unsqueeze(-1), t[b].unsqueeze(-1).transpose(0, 2)))) for a, b in zip(ia, ib)],
    }
    Fd, iad, ibd = torch.from_numpy(F).cuda(), torch.from_numpy(ia).cuda(), torch.from_numpy(ib).cuda()
    for mode, w in want.items():
        got = eng.score_trials(F, ia, ib, mode)
        assert got.shape == (P,) and float(np.abs(got - np.array(w, np.float32)).max()) <= 2e-6, mode
        gd = eng.score_trials(Fd, iad, ibd, mode)
        assert gd.is_cuda and np.array_equal(gd.cpu().numpy(), got), mode
    m = eng.mean_crops(F)
    assert m.shape == (n_files, D) and float(np.abs(m - F.mean(axis=1)).max()) <= 1e-6
    assert np.array_equal(eng.mean_crops(Fd).cpu().numpy(), m)
    with pytest.raises(Exception):
        eng.score_trials(F, np.array([n_files], np.int32), np.array([0], np.int32), "cosine")      # index out of range


@pytest.mark.parametrize("sa,sb,D", [(1.0, 1.0, 192), (3.0e5, 1.0, 192), (1.0e-6, 1.0e-6, 192), (7.0e6, 2.0e-7, 192), (40.0, 1.0e5, 192),
                                      (3.0e5, 1.0e-3, 256), (1.0e-6, 1.0, 256)])
def test_half_plane_score_kernels_take_operands_of_any_magnitude(sa, sb, D):
    """since round 4 the dense score GEMMs and the AS-norm kernel carry their operands as IEEE-half hi | lo planes — on
    every handle, the 'exact' scoring handle included — and the reference scores RAW embeddings when `normalize` is off.  Round 4 clamped
    |x| > 65504 silently and lost the low bits of components below 2^-3.  Round 5 scales every operand by an exact power of two first
    (a row by its own max |x|, the streamed matrix by its global max |x|; asnorm_fused.hip "operand scaling"): embeddings of norm 3e5,
    1e-6, 7e6 x 2e-7 ... against the float64 statement, relative to |a| |b| — the bar of the unit-vector test, 2e-7.  One row holds a
    component 2^-20 of its largest (it must not vanish), one is all zeros."""
    eng = Engine(model="none", max_batch=1)
    rng = np.random.Generator(np.random.PCG64(77))
    Na, Nb, K, top = 300, 517, 5994, 200      # (K = 5994: the fused AS-norm kernel; smaller cohorts take the slab path)
    A = rng.standard_normal((Na, D)).astype(np.float32); A /= np.linalg.norm(A, axis=1, keepdims=True)
    B = rng.standard_normal((Nb, D)).astype(np.float32); B /= np.linalg.norm(B, axis=1, keepdims=True)
    A[7, 1:] *= np.float32(2.0 ** -20)                                  # one dominant component, the rest 2^-20 of it
    A[9] = 0.0
    A, B = (A * np.float32(sa)).astype(np.float32), (B * np.float32(sb)).astype(np.float32)
    want = A.astype(np.float64) @ B.astype(np.float64).T
    got = eng.score_matrix(A, B)
    na, nb = np.linalg.norm(A.astype(np.float64), axis=1), np.linalg.norm(B.astype(np.float64), axis=1)
    rel = np.abs(got - want) / np.maximum(np.outer(na, nb), 1e-300)
    rel[9] = np.abs(got[9])                                              # the zero row: exactly zero
    print(f"score_matrix, |a| = {sa:g}, |b| = {sb:g}: max error {rel.max():.2e} of |a||b|")
    assert np.isfinite(got).all() and float(rel.max()) <= 2e-7
    # AS-norm statistics of un-normalised embeddings against an un-normalised cohort: mu and sigma scale with |e| |c|
    E = A[:256].copy()
    E[9] = E[10]
    C = rng.standard_normal((K, D)).astype(np.float32); C /= np.linalg.norm(C, axis=1, keepdims=True)
    C = (C * np.float32(sb)).astype(np.float32)
    mu, sd = eng.asnorm_stats(E, C, top)
    rmu, rsd = o_scoring.asnorm_stats(E.astype(np.float64), C.astype(np.float64), top)
    scale = np.linalg.norm(E.astype(np.float64), axis=1) * float(sb)
    emu = float(np.max(np.abs(mu - rmu) / scale)); esd = float(np.max(np.abs(sd - rsd) / scale))
    print(f"asnorm_stats: mu within {emu:.2e}, sigma within {esd:.2e} of |e| |c|; fallback rows {eng.asnorm_last_fallback}")
    assert emu <= 2e-7 and esd <= 2e-6 and eng.asnorm_last_fallback >= 0       # (>= 0: the fused kernel ran; some rows may have been handed over)
    eng.set_option("asnorm_slab", 1)                                      # the slab path: score_h3w + the top-k kernel
    mu2, sd2 = eng.asnorm_stats(E, C, top)
    eng.set_option("asnorm_slab", 0)
    assert float(np.max(np.abs(mu2 - rmu) / scale)) <= 2e-7 and float(np.max(np.abs(sd2 - rsd) / scale)) <= 2e-6
    # a NaN component stays a NaN in its row of the score matrix, and nowhere else
    A2 = A.copy(); A2[3, 5] = np.nan
    g2 = eng.score_matrix(A2, B)
    assert np.isnan(g2[3]).all() and np.array_equal(np.delete(g2, 3, axis=0), np.delete(got, 3, axis=0))
    # a non-finite element of the STREAMED operand (B, the cohort) must not set the scale of the finite elements
    # around it (exponent 255 gave 2^-122: every finite score came back 0).  np.inner gives NaN / inf in that column only.
    for bad in (np.nan, np.inf):
        B2 = B.copy(); B2[11, 2] = bad
        g3 = eng.score_matrix(A, B2)
        assert not np.isfinite(np.delete(g3[:, 11], 9)).any(), bad              # (row 9 of A is all zeros: 0 * inf = NaN, 0 * NaN = NaN either way)
        assert np.array_equal(np.delete(g3, 11, axis=1), np.delete(got, 11, axis=1)), bad
    C2 = C.copy(); C2[123, 7] = np.nan
    mu3, sd3 = eng.asnorm_stats(E, C2, top)
    # a NaN cohort score poisons the statistics of every embedding, as sorting NaNs does in the reference; it must NOT come back as finite zeros
    assert not ((mu3 == 0).all() and (sd3 == 0).all())
    C3 = C.copy(); C3[123] = np.float32(0.0); C3[123, 0] = np.float32(sb) * np.float32(1e-3)      # a tiny but finite row: statistics as before
    mu4, sd4 = eng.asnorm_stats(E, C3, top)
    rmu4, rsd4 = o_scoring.asnorm_stats(E.astype(np.float64), C3.astype(np.float64), top)
    assert float(np.max(np.abs(mu4 - rmu4) / scale)) <= 2e-7 and float(np.max(np.abs(sd4 - rsd4) / scale)) <= 2e-6
    eng.close()


@pytest.mark.parametrize("n_groups", [1, 2, 3])
def test_asnorm_on_speaker_structured_embeddings(n_groups):
    """the fused AS-norm kernel's threshold is a NORMAL quantile of the row's exact cohort-score moments, and every test
    and bench fed it isotropic Gaussian embeddings.  Here: 5 994 speaker centroids as the cohort, embeddings = centroid + within-speaker
    noise (same-speaker cosine 0.5 - 0.8: each embedding has one cohort score far in the tail), and with n_groups > 1 the centroids cluster
    (cosine 0.35 inside a group): BIMODAL cohort scores, for which the normal quantile passes too few or too many candidates.  Round 6:
    such rows are decided by the same fused kernel with a threshold re-derived from its own counts (svhip_asnorm_last_refit) instead of
    the slab path.  Every row against the float64 oracle at the bars of the Gaussian test; the slab route (option asnorm_norefit) agrees."""
    eng = Engine(model="none", max_batch=1)
    N, K, top, D = 3000, 5994, 200, 192
    E, C, _ = synth.synth_speaker_embeddings(N, n_speakers=K, dim=D, seed=40 + n_groups, n_groups=n_groups)
    mu, sd = eng.asnorm_stats(E, C, top)
    slab, (refit, passes) = eng.asnorm_last_fallback, eng.asnorm_last_refit
    rmu, rsd = o_scoring.asnorm_stats(E.astype(np.float64), C.astype(np.float64), top)
    emu, esd = float(np.abs(mu - rmu).max()), float(np.abs(sd - rsd).max() / rsd.min())
    print(f"groups {n_groups}: refit rows {refit} in {passes} passes, slab rows {slab} of {N}; mu err {emu:.2e}, sigma rel err {esd:.2e}")
    assert emu <= 2e-7 and esd <= 1e-5
    assert slab >= 0 and slab <= N // 100, "more than 1 % of the rows still take the slab path"
    if n_groups == 1:
        assert refit <= N // 20               # one tail score per row does not disturb the normal quantile
    else:
        assert refit > 0 and 1 <= passes <= 3
    eng.set_option("asnorm_norefit", 1)
    mu2, sd2 = eng.asnorm_stats(E, C, top)
    eng.set_option("asnorm_norefit", 0)
    assert eng.asnorm_last_refit == (0, 0)
    assert float(np.abs(mu2 - rmu).max()) <= 2e-7 and float(np.abs(sd2 - rsd).max() / rsd.min()) <= 1e-5
    # device operands, asynchronous caller stream: the same values
    Ed, Cd = torch.from_numpy(E).cuda(), torch.from_numpy(C).cuda()
    mud, sdd = eng.asnorm_stats(Ed, Cd, top)
    assert np.array_equal(mud.cpu().numpy(), mu) and np.array_equal(sdd.cpu().numpy(), sd)
    eng.close()


def test_pnorm_similarity_for_any_p_matches_the_reference(eng, golden_dir):
    """VERDICT r4 item 7b / ADVICE r3: pnorm_similarity(ref, com, p) (src/utils.py:167-169 -> F.pairwise_distance(p=p, eps=1e-6)) raised
    for p != 2.  svhip_score_trials_pnorm serves every p torch does: golden values generated by the reference itself
    (tests/golden/pnorm_p.npz: p = 1, 3, 0.5, 1.5, +-inf, 0, -2; one crop holds 17 differences that cancel the eps exactly), through the
    trial-list kernel (host arrays and device tensors) and through the reference-named per-trial API."""
    from speakerverification_amd import scoring
    g = np.load(os.path.join(golden_dir, "pnorm_p.npz"))
    R, Cm = g["R"], g["C"]
    n = R.shape[0]
    F = np.ascontiguousarray(np.concatenate([R, Cm]))                       # (2 n files, crops, D)
    ia, ib = np.arange(n, dtype=np.int32), np.arange(n, 2 * n, dtype=np.int32)
    Fd, iad, ibd = torch.from_numpy(F).cuda(), torch.from_numpy(ia).cuda(), torch.from_numpy(ib).cuda()
    for k, pv in enumerate(g["p"]):
        want = g["pnorm"][k]
        got = eng.score_trials(F, ia, ib, "pnorm", p=float(pv))
        err = float(np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))))
        print(f"p = {pv}: max err {err:.2e} (relative above 1)")
        assert err <= 1e-5, pv
        assert np.array_equal(eng.score_trials(Fd, iad, ibd, "pnorm", p=float(pv)).cpu().numpy(), got)
        one = scoring.pnorm_similarity(R[3], Cm[3], p=float(pv))            # the reference's per-trial signature
        assert abs(one - want[3]) <= 1e-5 * max(1.0, abs(want[3]))
    # p = 2 through the general entry point == the dedicated kernel
    assert np.allclose(eng.score_trials(F, ia, ib, "pnorm", p=2.0), eng.score_trials(F, ia, ib, "pnorm"), rtol=0, atol=0)
    with pytest.raises(Exception):
        eng.score_trials(F, ia, ib, "pnorm", p=float("nan"))


def test_asnorm_six_bf16_mfma_form_agrees_with_the_fp32_mfma_form(monkeypatch):
    """D = 192 runs the fused AS-norm kernel on SIX bf16 MFMAs per product block (every fp32 value split exactly into three bf16
    parts; the three smallest of the nine partial products, <= 2^-26 relative, dropped): scores to fp32 rounding.  Against the
    exact-fp32-MFMA form of the same kernel (option asnorm_f32mfma) and against the float64 oracle, on a cohort with ties and a
    ragged last block; same bars as the fp32 form."""
    eng = Engine(model="none", max_batch=1)
    rng = np.random.Generator(np.random.PCG64(606))
    N, D, K, top = 1333, 192, 5994, 200
    E = rng.standard_normal((N, D)).astype(np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    cohort = rng.standard_normal((K, D)).astype(np.float32)
    cohort /= np.linalg.norm(cohort, axis=1, keepdims=True)
    cohort[7] = cohort[9]
    # round 4: the default split form is TWO half planes / three fp16 MFMAs ("h3"); option asnorm_x6 selects round 3's three bf16 planes
    mu3, sd3 = eng.asnorm_stats(E, cohort, top)
    assert eng.asnorm_last_fallback == 0
    eng.set_option("asnorm_x6", 1)
    mu6, sd6 = eng.asnorm_stats(E, cohort, top)
    eng.set_option("asnorm_x6", 0)
    assert eng.asnorm_last_fallback == 0
    eng.set_option("asnorm_f32mfma", 1)
    mu1, sd1 = eng.asnorm_stats(E, cohort, top)
    eng.set_option("asnorm_f32mfma", 0)
    assert eng.asnorm_last_fallback == 0
    rmu, rsd = o_scoring.asnorm_stats(E, cohort, top)
    print("x6 vs f64 oracle: mu", float(np.abs(mu6 - rmu).max()), "sd rel", float((np.abs(sd6 - rsd) / rsd).max()),
          "| fp32 MFMA vs oracle: mu", float(np.abs(mu1 - rmu).max()), "| x6 vs fp32 MFMA: mu", float(np.abs(mu6 - mu1).max()))
    assert float(np.abs(mu6 - rmu).max()) <= 1e-6 and float(np.abs(mu1 - rmu).max()) <= 1e-6
    assert float((np.abs(sd6 - rsd) / rsd).max()) <= 1e-4
    assert float(np.abs(mu6 - mu1).max()) <= 5e-7 and float(np.abs(sd6 - sd1).max()) <= 5e-7
    print("h3 (two half planes, three fp16 MFMAs) vs f64 oracle: mu", float(np.abs(mu3 - rmu).max()), "sd rel", float((np.abs(sd3 - rsd) / rsd).max()),
          "| h3 vs fp32 MFMA: mu", float(np.abs(mu3 - mu1).max()))
    assert float(np.abs(mu3 - rmu).max()) <= 1e-6 and float((np.abs(sd3 - rsd) / rsd).max()) <= 1e-4
    assert float(np.abs(mu3 - mu1).max()) <= 5e-7 and float(np.abs(sd3 - sd1).max()) <= 5e-7
    # round 4 (late): the default runs on v_mfma_f32_16x16x32_f16 (four candidate lists per embedding); option asnorm_w32 keeps the 32-wide
    # form (two lists).  Same three products per block in the same order within a k step; the k steps are 32 wide instead of 16
    eng.set_option("asnorm_w32", 1)
    muw, sdw = eng.asnorm_stats(E, cohort, top)
    eng.set_option("asnorm_w32", 0)
    assert eng.asnorm_last_fallback == 0
    print("16-wide vs 32-wide half-plane kernel: mu", float(np.abs(mu3 - muw).max()), "sd", float(np.abs(sd3 - sdw).max()))
    assert float(np.abs(mu3 - muw).max()) <= 2e-7 and float(np.abs(sd3 - sdw).max()) <= 2e-7
    assert float(np.abs(muw - rmu).max()) <= 1e-6
    eng.close()
