"""CPU: host-side logic of the drop-in layer (audio cropping, trial-list plumbing, batched scoring
statements, multi-process sharding over gloo) — no GPU, the Engine is replaced by tests/fakes.py."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import scoring as o_scoring
from speakerverification_amd import audio, distributed as sv_dist, scoring
from speakerverification_amd import model as sv_model
from tests import fakes
from tests.e2e_data import make_e2e_files


def test_loadwav_matches_reference_cropping(golden_dir):
    g = np.load(os.path.join(golden_dir, "crop.npz"))
    spec = {"sample_rate": 16000, "channels": 1, "sentence_len": 2.0, "win_len": 0.025, "hop_len": 0.01}
    rng = np.random.Generator(np.random.PCG64(77))
    for name in ("long", "short", "exact", "long10"):
        n, ne = int(g[name + "_len"]), int(g[name + "_num_eval"])
        a = (0.3 * rng.standard_normal(n)).astype(np.float32)
        got = audio.loadWAV(a, spec, evalmode=True, num_eval=ne)
        assert got.shape == (ne, 32000) and got.dtype == np.float32
        assert np.array_equal(got[:, :4], g[name + "_first"]) and np.array_equal(got[:, -4:], g[name + "_last"])
    whole = audio.loadWAV((0.3 * rng.standard_normal(40000)).astype(np.float32), spec, num_eval=0)
    assert whole.shape == (1, 40000)
    with pytest.raises(NotImplementedError):
        audio.loadWAV(np.zeros(10, np.float32), spec, evalmode=False)


def test_wav_reader_scaling(tmp_path):
    files, _, _ = make_e2e_files(str(tmp_path))
    a, sr = audio.read_wav(files[0])
    assert sr == 16000 and a.dtype == np.float32 and a.shape == (32000,) and np.abs(a).max() <= 1.0


@pytest.fixture
def fake_engine(monkeypatch):
    eng = fakes.FakeScoringEngine()
    monkeypatch.setattr(scoring, "scoring_engine", lambda device=0: eng)
    return eng


def test_score_trials_equals_reference_per_trial_loop(fake_engine, golden_dir):
    g = np.load(os.path.join(golden_dir, "scoring.npz"))
    Rn = torch.nn.functional.normalize(torch.from_numpy(g["R"]), p=2, dim=2).numpy()
    Cn = torch.nn.functional.normalize(torch.from_numpy(g["C"]), p=2, dim=2).numpy()
    n = Rn.shape[0]
    feats = np.concatenate([Rn, Cn])                     # (2n files, 3 crops, D)
    ia, ib = np.arange(n), np.arange(n, 2 * n)
    assert np.abs(scoring.score_trials(feats, ia, ib, "cosine") - g["cosine"]).max() < 1e-5
    assert np.abs(scoring.score_trials(feats, ia, ib, "norm", cohorts=g["cohort"], top=int(g["top"])) - g["zt_norm"]).max() < 1e-4
    assert np.abs(scoring.score_trials(feats, ia, ib, "pnorm") - g["pnorm"]).max() < 1e-5
    assert scoring.score_trials(feats, ia[:0], ib[:0], "cosine").shape == (0,)
    # per-trial compatibility signature (utils.py:126-132)
    assert abs(scoring.similarity_measure("cosine", torch.from_numpy(Rn[0]), torch.from_numpy(Cn[0])) - g["cosine"][0]) < 1e-5
    assert abs(scoring.similarity_measure("zt_norm", Rn[0], Cn[0], cohorts=g["cohort"], top=int(g["top"])) - g["zt_norm"][0]) < 1e-4


class _FakeS:
    def eval(self): return self
    def state_dict(self): return {}


def _make_handler(tmp_path, dim=16):
    enc = sv_model.SpeakerEncoder.__new__(sv_model.SpeakerEncoder)
    enc.model = {"name": "ECAPA_TDNN", "nOut": dim}
    enc.criterion = {"name": "AAmSoftmaxAP"}
    enc.test_normalize = True
    enc.features = "melspectrogram"
    enc.__S__ = _FakeS()
    emb = fakes.fake_embedder(dim)
    enc.forward = lambda data, label=None: emb(data.reshape(-1, data.shape[-1]))
    net = sv_model.WrappedModel(enc)
    net.forward = lambda x, label=None: enc.forward(x)
    spec = {"sample_rate": 16000, "channels": 1, "sentence_len": 2.0, "win_len": 0.025, "hop_len": 0.01}
    mh = sv_model.ModelHandling(net, audio_spec=spec, save_folder=str(tmp_path), embed_batch=5)
    return mh, emb, spec


def _expected_scores(files, lines, emb, spec, num_eval):
    feats = {}
    for f in files:
        a, _ = audio.read_wav(f)
        feats[f] = torch.nn.functional.normalize(torch.from_numpy(emb(o_scoring.crop_eval(a, 32000, num_eval, peak_normalize=False))), p=2, dim=1)
    return [o_scoring.cosine_similarity(feats[l.split()[1]], feats[l.split()[2]]) for l in lines]


def test_evaluate_from_list_plumbing(fake_engine, tmp_path):
    files, trial_path, lines = make_e2e_files(str(tmp_path))
    mh, emb, spec = _make_handler(tmp_path)
    for ne in (1, 3):
        sc, lab, tr = mh.evaluateFromList(listfilename=trial_path, distributed=False, dataloader_options={"num_workers": 0},
                                          cohorts_path="unused", num_eval=ne, scoring_mode="cosine")
        assert len(sc) == len(lab) == len(tr) == 28 and all(isinstance(s, float) for s in sc)
        assert lab == [int(l.split()[0]) for l in lines]
        assert tr == [l.split()[1] + " " + l.split()[2] for l in lines]
        assert np.abs(np.array(sc) - np.array(_expected_scores(files, lines, emb, spec, ne))).max() < 1e-5
    e = mh.embed_utterance(files[1], num_eval=4, normalize=True)
    assert tuple(e.shape) == (4, 16) and np.allclose(np.linalg.norm(e.numpy(), axis=1), 1.0, atol=1e-5)


def test_test_from_list_and_prepare(fake_engine, tmp_path):
    files, _, _ = make_e2e_files(str(tmp_path))
    mh, emb, spec = _make_handler(tmp_path)
    csv_path = tmp_path / "pairs.txt"
    csv_path.write_text("audio_1,audio_2\n" + "".join(f"{files[i]},{files[i + 1]}\n" for i in range(4)))
    res = mh.testFromList(test_list=str(csv_path), thresh_score=0.5, cohorts_path=None, num_eval=2, scoring_mode="cosine",
                          output_file=str(tmp_path / "out.txt"))
    assert len(res) == 4 and res[0].startswith("utt0.wav,utt1.wav,")
    rows = (tmp_path / "out.txt").read_text().splitlines()
    assert rows[0] == "audio_1,audio_2,pred_label,score" and len(rows) == 5
    meta = tmp_path / "train.txt"
    meta.write_text("".join(f"spk{i // 4} {f}\n" for i, f in enumerate(files)))
    out = tmp_path / "cohort.npy"
    assert mh.prepare(save_path=str(out), prepare_type="cohorts", num_eval=2, source=str(meta)) is True
    cohort = np.load(out)
    assert cohort.shape == (2, 16)           # 2 speakers, first 3 files each, mean of L2-normalised crop embeddings
    exp0 = np.concatenate([torch.nn.functional.normalize(torch.from_numpy(emb(o_scoring.crop_eval(audio.read_wav(f)[0], 32000, 2, peak_normalize=False))), dim=1).numpy()
                           for f in files[:3]]).mean(0)
    assert np.abs(cohort[0] - exp0).max() < 1e-5


def test_pairwise_distance_scoring_goes_through_the_trial_kernel(fake_engine, tmp_path):
    """cohorts_path=None scoring (model.py:425-431, F.pairwise_distance over the broadcast (n, D, n) difference): ModelHandling
    normalises, then hands the whole trial list to ONE scoring call in mode 'pdist' (svhip_score_trials on a GPU; here the
    stand-in engine, which evaluates the reference's own per-trial expression) — no (P, n, D, n) temporary on the host."""
    import torch.nn.functional as F
    mh, emb, spec = _make_handler(tmp_path)
    rng = np.random.Generator(np.random.PCG64(5))
    n_files, n, D = 40, 10, 192
    feats = rng.standard_normal((n_files, n, D)).astype(np.float32)
    ia = rng.integers(0, n_files, 64).astype(np.int32)
    ib = rng.integers(0, n_files, 64).astype(np.int32)
    ib[:4] = ia[:4]                                   # a file against itself
    calls = []
    orig = fake_engine.score_trials
    fake_engine.score_trials = lambda Fm, a, b, mode="cosine", out=None: (calls.append((mode, len(a))), orig(Fm, a, b, mode))[1]
    got = mh._score(feats.copy(), ia, ib, "cosine", None, None)
    assert calls == [("pdist", 64)]
    fn = F.normalize(torch.from_numpy(feats).reshape(-1, D), p=2, dim=1).reshape(n_files, n, D)
    want = [float(-torch.mean(F.pairwise_distance(fn[a].unsqueeze(-1), fn[b].unsqueeze(-1).transpose(0, 2)))) for a, b in zip(ia, ib)]
    assert got.dtype == np.float32 and np.abs(got - np.array(want, np.float32)).max() < 2e-6


def test_trial_rows_partition_the_list():
    rng = np.random.Generator(np.random.PCG64(9))
    for n, w in ((10, 3), (1000, 8), (7, 8)):
        ia = rng.integers(0, n, 500)
        rows = [sv_dist.trial_rows_of_rank(ia, n, r, w) for r in range(w)]
        assert sorted(np.concatenate(rows).tolist()) == list(range(500))
        for r in range(w):
            lo, hi, _ = sv_dist.shard_bounds(n, r, w)
            assert ((ia[rows[r]] >= lo) & (ia[rows[r]] < hi)).all()


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 9, 1000):
        for w in (1, 2, 3, 8):
            got = []
            for r in range(w):
                lo, hi, per = sv_dist.shard_bounds(n, r, w)
                assert 0 <= hi - lo <= per
                got += list(range(lo, hi))
            assert got == list(range(n))


def _gloo_worker(rank, world, port, tmp, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from speakerverification_amd import scoring as sc_mod
        eng = fakes.FakeScoringEngine()
        sc_mod.scoring_engine = lambda device=0: eng
        # 1. dense all-gather of ragged shards
        n = 7
        lo, hi, _ = sv_dist.shard_bounds(n, rank, world)
        local = torch.arange(lo, hi, dtype=torch.float32)[:, None] * torch.ones(1, 3)
        full = sv_dist.all_gather_rows(local, n)
        ok1 = bool(torch.equal(full, torch.arange(n, dtype=torch.float32)[:, None] * torch.ones(1, 3)))
        # 2. evaluateFromList shards files over ranks and scores on rank 0
        files, trial_path, lines = make_e2e_files(os.path.join(tmp, f"r{rank}"))
        mh, emb, spec = _make_handler(tmp)
        sc, lab, tr = mh.evaluateFromList(listfilename=trial_path, distributed=True, dataloader_options={}, cohorts_path="x",
                                          num_eval=2, scoring_mode="cosine")
        # 3. row-sharded scoring: every rank scores the trials of its enrol block, rank 0 assembles them
        sc2, _, _ = mh.evaluateFromList(listfilename=trial_path, distributed=True, dataloader_options={}, cohorts_path="x",
                                        num_eval=2, scoring_mode="cosine", shard_scoring=True)
        q.put((rank, ok1, sc, lab, sc2))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharded_evaluation(tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    for r in range(2):
        os.makedirs(tmp_path / f"r{r}", exist_ok=True)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, ok1, sc, lab, sc2 = q.get(timeout=120)
        res[rank] = (ok1, sc, lab, sc2)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][0] and res[1][0]
    assert len(res[0][1]) == 28 and res[1][1] == []             # rank 0 scores, other ranks return empty lists
    assert res[1][3] == [] and np.abs(np.array(res[0][3]) - np.array(res[0][1])).max() < 1e-6   # row-sharded == rank-0 scoring
    # single-process answer must match
    files, trial_path, lines = make_e2e_files(str(tmp_path / "single"))  if os.makedirs(tmp_path / "single", exist_ok=True) is None else None
    eng = fakes.FakeScoringEngine()
    scoring.scoring_engine = lambda device=0: eng
    mh, emb, spec = _make_handler(tmp_path)
    sc, _, _ = mh.evaluateFromList(listfilename=trial_path, distributed=False, dataloader_options={}, cohorts_path="x",
                                   num_eval=2, scoring_mode="cosine")
    assert np.abs(np.array(sc) - np.array(res[0][1])).max() < 1e-6


@pytest.mark.parametrize("case", ["small", "ties", "distinct", "skewed"])
def test_metrics_host_selection_matches_sklearn(case):
    """The host half of speakerverification_amd.metrics (drop_intermediate, (0, 0) start point, precision / recall, auc) fed
    with CPU-computed curve points must reproduce scikit-learn's roc_curve / precision_recall_curve / auc exactly — the device
    half only supplies (fps, tps, thresholds), which tests/test_gpu_metrics.py pins separately."""
    from sklearn import metrics as skm
    from oracle import metrics as o_metrics
    from speakerverification_amd import metrics as m
    from tests.metrics_data import metrics_case
    sc, lab = metrics_case(case)
    fps, tps, thr = o_metrics.binary_clf_curve(lab, sc)
    fpr, tpr, th = m._roc_curve(fps, tps, thr.astype(np.float32))
    rf, rt, rth = skm.roc_curve(lab, sc.astype(np.float64), pos_label=1)
    assert np.array_equal(fpr, rf) and np.array_equal(tpr, rt) and np.array_equal(th, rth)
    pr, rc, pth = m._precision_recall_curve(fps, tps, thr.astype(np.float32))
    sp, sr, sth = skm.precision_recall_curve(lab, sc.astype(np.float64), pos_label=1)
    assert np.array_equal(pr, sp) and np.array_equal(rc, sr) and np.array_equal(pth, sth)
    assert m._auc(fpr * 100, tpr) == skm.auc(rf * 100, rt)


def test_front_end_keywords_reach_the_fused_waveform_path():
    """a config that overrides the mel front-end (feature.py:66-71 keywords) must change the fused
    ECAPA waveform path too — the keywords are forwarded to the engine — and an unsupported window is refused."""
    from speakerverification_amd.models import ECAPA_TDNN
    m = ECAPA_TDNN.MainModel(nOut=192, channels=[64] * 4 + [192], n_mels=80, features="melspectrogram",
                             sr=16000, n_fft=400, win_length=400, hop_length=160, fmin=20.0, fmax=7600.0, pre_emphasis=False)
    kw = m._engine_kwargs
    assert (kw["sr"], kw["n_fft"], kw["win_length"], kw["hop_length"], kw["fmin"], kw["fmax"], kw["pre_emphasis"]) == \
           (16000, 400, 400, 160, 20.0, 7600.0, False)
    assert m._max_batch == 256
    assert ECAPA_TDNN.MainModel(nOut=192, channels=[64] * 4 + [192], embed_batch=32)._max_batch == 32
    with pytest.raises(NotImplementedError):
        ECAPA_TDNN.MainModel(nOut=192, channels=[64] * 4 + [192], window="hann")


class _FakeCommWorld:
    """W simulated ranks in one process: every rank's engine.allgather_rows blocks on a barrier, then returns the concatenation of
    all ranks' blocks in rank order — the contract of svhip_allgather_rows (equal row counts per rank, block r at out + r*rows*D)."""

    def __init__(self, world):
        import threading
        self.world = world
        self.blocks = [None] * world
        self.barrier = threading.Barrier(world)
        self.inits = []

    def engine(self, rank):
        outer = self

        class _Eng:
            def comm_init(self, id_bytes, r, w):
                outer.inits.append((bytes(id_bytes), r, w))

            def allgather_rows(self, padded, out=None):
                is_t = isinstance(padded, torch.Tensor)
                a = padded.numpy() if is_t else np.asarray(padded)
                assert a.ndim == 2 and a.dtype == np.float32
                outer.blocks[rank] = a.copy()
                outer.barrier.wait(timeout=30)
                rows = {b.shape for b in outer.blocks}
                assert len(rows) == 1, rows                      # the collective needs equal blocks: padding is LibComm's job
                full = np.concatenate(outer.blocks, 0)
                outer.barrier.wait(timeout=30)
                return torch.from_numpy(full) if is_t else full
        return _Eng()


@pytest.mark.parametrize("n,world,tail,as_torch", [(7, 2, (5,), False), (7, 3, (2, 3), False), (7, 8, (4,), False), (0, 2, (3,), False),
                                                   (1000, 8, (192,), True), (9, 4, (), False)])
def test_libcomm_all_gather_rows_at_world_sizes_above_one(n, world, tail, as_torch):
    """LibComm.all_gather_rows (the RCCL carrier's host logic) had only ever run at world size 1.  W simulated ranks with
    a stand-in engine: short last shard (zero padding up to ceil(N / W) rows), EMPTY shards (W > N), block placement in rank
    order, multi-dimensional rows — the result must be the full matrix on every rank, and equal to what the torch.distributed
    carrier defines (shard_bounds order)."""
    import threading
    rng = np.random.Generator(np.random.PCG64(n * 31 + world))
    full = rng.standard_normal((n,) + tail).astype(np.float32)
    w = _FakeCommWorld(world)
    res, err = [None] * world, []

    def run(r):
        try:
            comm = sv_dist.LibComm(w.engine(r), rank=r, world=world, id_bytes=b"\x07" * 128)
            lo, hi, per = sv_dist.shard_bounds(n, r, world)
            local = full[lo:hi]
            got = comm.all_gather_rows(torch.from_numpy(local) if as_torch else local, n)
            res[r] = got.numpy() if as_torch else got
        except Exception as e:          # pragma: no cover
            err.append((r, repr(e)))
            try:
                w.barrier.abort()
            except Exception:
                pass

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=60)
    assert not err, err
    assert sorted(w.inits) == sorted((b"\x07" * 128, r, world) for r in range(world))
    for r in range(world):
        assert res[r].shape == full.shape and np.array_equal(res[r], full), r


def test_primary_engine_geometry_comes_from_the_configured_crop_length():
    """the handle that gets the full max_batch workspace is the configured evaluation crop (audio_spec), not whichever
    length happens to arrive first (a whole-file call would otherwise claim a max_batch x 10^4-frame workspace)."""
    from speakerverification_amd.models import ECAPA_TDNN, RawNet2_custom
    spec = {"sample_rate": 16000, "sentence_len": 2.0}
    m = ECAPA_TDNN.MainModel(nOut=192, channels=[64] * 4 + [192], audio_spec=spec)
    assert m._primary == 32000
    m._drop_engine()
    assert m._primary == 32000
    assert ECAPA_TDNN.MainModel(nOut=192, channels=[64] * 4 + [192])._primary is None          # unknown until the first call
    assert RawNet2_custom.MainModel(nOut=320, front_proc="sinc", aggregate="asp", audio_spec=spec)._primary == 32000


def _verify_gather_worker(rank, world, port, corrupt, q):
    """bench.verify_gather on CPU tensors over gloo: the check the N-GPU bench runs on its gathered embedding matrix"""
    import argparse
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    ranks = bench.Ranks(argparse.Namespace(gpus=world))
    try:
        n_local, D = 37, 24

        def embed(utt):                      # a pure function of the utterance index, like the synthetic stream + the engine
            g = torch.Generator().manual_seed(1000 + int(utt))
            return torch.randn(D, generator=g)

        shard = torch.stack([embed(rank * n_local + i) for i in range(n_local)])
        blocks = [torch.zeros_like(shard) for _ in range(world)]
        ranks.dist.all_gather(blocks, shard)
        gathered = torch.cat(blocks)
        if corrupt == "bytes" and rank == 1:
            gathered[2 * n_local + 5, 3] += 1e-3           # one word of rank 2's block arrives wrong on rank 1 only
        if corrupt == "swap":                               # every rank sees ranks 1 and 2 exchanged: checksums of the blocks disagree
            gathered = torch.cat([blocks[0], blocks[2], blocks[1]])
        if corrupt == "content":                            # bytes arrive intact, but rank 2 embedded the WRONG utterances
            pass
        regen = lambda r, n: torch.stack([embed(r * n_local + i + (7 if (corrupt == "content" and r == 2) else 0)) for i in range(n)])
        rec = bench.verify_gather(ranks, shard, gathered, n_local, regen)
        q.put((rank, rec))
    finally:
        ranks.close()


@pytest.mark.parametrize("corrupt", ["none", "bytes", "swap", "content"])
def test_bench_cross_rank_verification_at_world_three(corrupt):
    """at N > 1 the bench must prove that the OTHER ranks' blocks arrived (it only checked its own).  Three gloo
    ranks on the CPU: the clean exchange passes; a flipped word on one rank, two blocks exchanged, and a rank that embedded the wrong
    utterances each make `cross_rank_ok` false on EVERY rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + {"none": 0, "bytes": 1, "swap": 2, "content": 3}[corrupt]
    procs = [ctx.Process(target=_verify_gather_worker, args=(r, 3, port, corrupt, q)) for r in range(3)]
    for p in procs:
        p.start()
    recs = dict(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(3):
        assert recs[r]["cross_rank_ok"] == (corrupt == "none"), (corrupt, recs)
        assert len(recs[r]["block_checksums"]) == 3
    assert recs[0]["cross_rank_rows_checked"] == 16 and recs[1]["cross_rank_rows_checked"] == 0
    if corrupt == "none":
        assert all(recs[r]["blocks_bitwise_ok"] for r in range(3)) and recs[0]["cross_rank_min_cosine"] > 0.99999
    if corrupt == "bytes":
        assert [recs[r]["blocks_bitwise_ok"] for r in range(3)] == [True, False, True]
    if corrupt == "content":
        assert all(recs[r]["blocks_bitwise_ok"] for r in range(3)) and recs[0]["cross_rank_min_cosine"] < 0.9


def test_bench_headline_is_one_compact_parsable_line(tmp_path, capsys):
    """round 4's ONE 20 KB stdout line outgrew the driver's stdout tail and `BENCH_r04.parsed` was null.  The
    headline is now built by bench.compact_headline: <= 4 KB whatever the sub-records hold, json-parsable, carrying the contract's
    fields + roofline + cpu_baseline; the detail goes to bench.Records (stderr + a JSON-lines file), never to stdout."""
    import json
    import bench
    big = {"kernels": {f"kernel_{i}": {"avg_ms": 0.1 * i, "launches_per_step": 3.0, "ms_per_step": 0.3 * i, "TFLOPs": None} for i in range(60)}}
    full = {"metric": "embeddings/sec (2 s @16 kHz)", "value": 58023.21234567, "unit": "embeddings/s", "n_gpus": 1, "steps": 20, "warmup": 5,
            "ms_per_step": 4.41203123, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "ECAPA-TDNN C=1024 ... (BASELINE configs[1])", "batch_per_gpu": 256, "samples": 32000, "collective": "none"},
            "check": {"ok": True, "bitwise_rerun": True, "finite": True, "min_cosine_vs_f32_path": 0.9999, "max_err_over_scale": 0.011,
                      "bars": {"min_cosine": 0.999}},
            "roofline": {"kernel": "gemm_pw3", "bound": "mfma", "achieved": 1098.4, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.43936,
                         "traffic": 1.0256e9, "traffic_source": "x" * 300, "avg_launch_ms": 0.42, "launches": 140, "flops_per_launch": 4.6e11},
            "cpu_baseline": {"value": 25.9, "unit": "embeddings/s", "cores": 16, "kind": "port", "cpu": "EPYC", "sample": "s" * 200,
                             "B8": {"pass_s": [1.0] * 9}, "B256": {"pass_s": [9.0] * 3}},
            "sustained": {"value": 58903.5, "ms_per_step": 4.346, "steps": 454, "seconds": 1.97, "unit": "embeddings/s"},
            "whole_path_TFLOPs": 872.4, **big}
    sub = {f"mode_{i}": 1234.5678 * i for i in range(12)}
    text = bench.compact_headline(full, sub=sub, scoring={"cosine_pairs_per_s": 4.0e9, "asnorm_pairs_per_s": 1.1e8, "asnorm_frac": 0.31})
    assert "\n" not in text and len(text) <= bench.HEADLINE_MAX_BYTES
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in line
    assert line["steps"] == 20 and line["warmup"] == 5 and line["config"]["workload"].startswith("ECAPA")
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "launches")) <= set(line["roofline"])
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-4
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"])
    assert "kernels" not in line and "traffic_source" not in line["roofline"] and "B8" not in line["cpu_baseline"]
    assert line["sub"]["mode_3"] == pytest.approx(3703.7, rel=1e-4) and line["scoring"]["asnorm_frac"] == 0.31
    # a pathological sub map cannot push the line past the bound: optional parts are dropped, the contract's fields stay
    text = bench.compact_headline(full, sub={f"k{i}": "x" * 50 for i in range(200)})
    assert len(text) <= bench.HEADLINE_MAX_BYTES and "roofline" in json.loads(text) and "sub" not in json.loads(text)
    # Records: tagged JSON lines to stderr and to the file, nothing on stdout
    recs = bench.Records(str(tmp_path / "r.jsonl"))
    recs.emit("kernels", big)
    recs.emit("rawnet2", {"value": 1.0})
    recs.close()
    cap = capsys.readouterr()
    assert cap.out == ""
    lines = [json.loads(ln) for ln in open(tmp_path / "r.jsonl")]
    assert [ln["record"] for ln in lines] == ["kernels", "rawnet2"] and len(lines[0]["kernels"]) == 60
    assert [json.loads(ln)["record"] for ln in cap.err.strip().splitlines()] == ["kernels", "rawnet2"]


def test_persistent_gemm_item_walk_covers_every_tile_once():
    """The work list of the persistent GEMM (csrc/gemm_pw3.hip: `item_tile`, `pw3_grid`, round 5) restated in Python: for any tile count and
    grid cap, the items that the G workgroups walk (item w, w + G, w + 2 G ...) cover every tile exactly once — as one whole item, or as its
    two column halves when the last partial round holds at most G / 2 tiles (and for grids of at most cap / 2 tiles: halves only) — and no
    workgroup walks more than ceil-many items.  A host-side guard for index arithmetic that otherwise only GPU parity tests exercise."""
    def grid_of(ntiles, cap, tail_split=True):
        if ntiles >= cap:
            return cap
        return 2 * ntiles if (tail_split and 2 * ntiles <= cap) else ntiles

    def walk(ntiles, G, tail_split=True):
        qfull = (ntiles // G) * G
        rtail = ntiles - qfull
        split = tail_split and rtail > 0 and 2 * rtail <= G
        nitems = qfull + 2 * rtail if split else ntiles
        seen = {}
        per_wg = []
        for wg in range(G):
            n = 0
            w = wg
            while w < nitems:
                if not split or w < qfull:
                    tile, hsel = w, 0
                else:
                    tile, hsel = qfull + ((w - qfull) >> 1), 1 + ((w - qfull) & 1)
                seen.setdefault(tile, []).append(hsel)
                n += 1
                w += G
            per_wg.append(n)
        return seen, per_wg, split

    rng = np.random.Generator(np.random.PCG64(5))
    cases = [(1604, 256), (4812, 256), (384, 256), (128, 256), (26, 5), (16, 3), (40, 8), (1, 256), (2, 3), (129, 256), (255, 256)]
    cases += [(int(rng.integers(1, 6000)), int(rng.integers(1, 300))) for _ in range(300)]
    for ntiles, cap in cases:
        for ts in (True, False):
            G = grid_of(ntiles, cap, ts)
            assert 1 <= G <= max(cap, 1) and G <= 2 * ntiles
            seen, per_wg, split = walk(ntiles, G, ts)
            assert sorted(seen) == list(range(ntiles)), (ntiles, cap)
            for tile, hs in seen.items():
                assert sorted(hs) in ([0], [1, 2]), (ntiles, cap, tile, hs)
            assert max(per_wg) - min(per_wg) <= 1
            if not ts:
                assert not split and all(hs == [0] for hs in seen.values())
    # the shapes DESIGN.md quotes: 1 604 tiles on 256 CUs = 6 whole rounds + 136 halves; 128 tiles = 256 halves; mfa's 4 812 tiles are not split
    seen, per_wg, split = walk(1604, 256)
    assert split and sum(1 for hs in seen.values() if hs != [0]) == 68 and max(per_wg) == 7 and per_wg.count(7) == 136
    assert grid_of(128, 256) == 256 and walk(128, 256)[2] and not walk(4812, 256)[2]
