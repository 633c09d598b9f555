"""GPU: the multi-GPU code path at world size 1 (one MI355X per gpurun box): RCCL under the C ABI, torch.distributed
backend "nccl" (= RCCL), the sharded evaluateFromList, the on-device synthetic utterance stream, and what crosses PCIe."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import synthwave as o_synth
from speakerverification_amd import _lib, distributed as sv_dist, engine as sv_engine, scoring, synth
from speakerverification_amd.engine import Engine
from speakerverification_amd.model import ModelHandling, SpeakerEncoder, WrappedModel
from tests.e2e_data import E2E_SEED_W, make_e2e_files
from tests.test_gpu_e2e import ARGS

pytestmark = pytest.mark.gpu


def test_synth_waveforms_match_the_counter_rng_oracle():
    eng = Engine(model="none", max_batch=1)
    got = eng.synth_waveforms(5, 1000, 3, 32000)
    want = o_synth.synth_waveforms(5, 1000, 3, 32000)
    assert got.shape == want.shape == (3, 32000)
    assert float(np.abs(got - want).max()) <= 1e-6           # same Philox words; logf / sincosf differ by an ulp or two
    dev = torch.empty((2, 32000), device="cuda", dtype=torch.float32)
    eng.synth_waveforms(5, 1001, 2, 32000, out=dev)
    assert np.array_equal(dev.cpu().numpy(), got[1:3])       # a block is a pure function of (seed, utterance)
    with pytest.raises(_lib.SvhipError):
        eng.synth_waveforms(5, 0, 1, 31999)                  # L % 4 != 0
    eng.close()


def test_c_abi_allgather_world1_device_and_host_pointers():
    eng = Engine(model="none", max_batch=1)
    lib = eng.lib
    x = np.arange(12, dtype=np.float32).reshape(3, 4)
    out = np.empty_like(x)
    rc = lib.svhip_allgather_rows(eng.h, x.ctypes.data, 3, 4, out.ctypes.data, 0)
    assert rc == -3 and b"communicator" in lib.svhip_last_error(eng.h)          # SVHIP_ERR_STATE before comm_init
    comm = sv_dist.LibComm(eng, rank=0, world=1)                                  # ncclCommInitRank: real RCCL, one rank
    r, w = C.c_int32(-1), C.c_int32(-1)
    assert lib.svhip_comm_rank(eng.h, C.byref(r), C.byref(w)) == 0 and (r.value, w.value) == (0, 1)
    assert np.array_equal(eng.allgather_rows(x), x)                               # host pointers (staged inside the library)
    xd = torch.randn(1000, 192, device="cuda")
    od = eng.allgather_rows(xd)
    assert od.is_cuda and torch.equal(od, xd) and od.data_ptr() != xd.data_ptr()
    full = comm.all_gather_rows(x[:3], 3)
    assert np.array_equal(full, x)
    with pytest.raises(_lib.SvhipError):
        eng.comm_init(Engine.comm_unique_id(), 0, 1)                              # one communicator per handle
    assert lib.svhip_comm_destroy(eng.h) == 0
    eng.close()


@pytest.fixture(scope="module")
def nccl_group():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 1000))
    import torch.distributed as dist
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
    yield dist
    dist.destroy_process_group()


def test_torch_nccl_all_gather_rows_world1(nccl_group):
    local = torch.arange(15, dtype=torch.float32, device="cuda").reshape(5, 3)
    full = sv_dist.all_gather_rows(local, 5)                 # runs the all_gather_into_tensor on RCCL (not the W == 1 shortcut)
    assert full.is_cuda and torch.equal(full, local) and full.data_ptr() != local.data_ptr()


def test_evaluate_from_list_distributed_matches_single_process(nccl_group, tmp_path, golden_dir):
    """evaluateFromList(distributed=True) under an initialised nccl group (world 1): shard -> device crop -> embed ->
    svhip_allgather_rows -> score; must equal the non-distributed result and the reference's golden scores."""
    tmp = str(tmp_path)
    net = WrappedModel(SpeakerEncoder(**ARGS))
    mh = ModelHandling(net, **dict(ARGS, save_folder=tmp))
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=512), seed=E2E_SEED_W)
    net.module.load_state_dict({"__S__." + k: v for k, v in sd.items()})
    files, trial_path, lines = make_e2e_files(tmp)
    kw = dict(listfilename=trial_path, dataloader_options={}, cohorts_path="unused", num_eval=2, scoring_mode="cosine")
    single = mh.evaluateFromList(distributed=False, **kw)
    multi = mh.evaluateFromList(distributed=True, **kw)
    comm = scoring._comms.get((0, 0, 1))
    assert comm is not None and comm.world == 1                        # the RCCL communicator was really used (cached next to the engine)
    # a second ModelHandling in the same process reuses it (svhip_comm_init refuses a second communicator on a handle)
    mh2 = ModelHandling(net, **dict(ARGS, save_folder=tmp))
    again = mh2.evaluateFromList(distributed=True, **kw)
    assert np.array_equal(np.array(again[0]), np.array(single[0])) and scoring._comms[(0, 0, 1)] is comm
    assert multi[1] == single[1] and multi[2] == single[2]
    assert np.array_equal(np.array(multi[0]), np.array(single[0]))
    g = np.load(os.path.join(golden_dir, "e2e_config1.npz"))
    assert float(np.abs(np.array(multi[0]) - g["scores_ne2"]).max()) <= 1e-4


def test_only_int16_pcm_crosses_pcie(tmp_path):
    """16-bit WAV files are cropped on the device; no fp32 crop crosses PCIe on the way to the embedder."""
    tmp = str(tmp_path)
    net = WrappedModel(SpeakerEncoder(**ARGS))
    mh = ModelHandling(net, **dict(ARGS, save_folder=tmp))
    files, trial_path, lines = make_e2e_files(tmp)
    pcm_bytes = sum(os.path.getsize(f) - 44 for f in files)
    num_eval = 10
    mh._embed_files(files[:1], num_eval)                                  # builds the engines
    before = dict(sv_engine.TRANSFER_STATS)
    feats = mh._embed_files(files, num_eval)
    h2d = sv_engine.TRANSFER_STATS["h2d_bytes"] - before["h2d_bytes"]
    d2h = sv_engine.TRANSFER_STATS["d2h_bytes"] - before["d2h_bytes"]
    crop_bytes = len(files) * num_eval * 32000 * 4
    assert feats.shape == (len(files), num_eval, 192) and np.isfinite(feats).all()
    assert pcm_bytes <= h2d <= pcm_bytes + 64 * len(files), (h2d, pcm_bytes)      # the int16 samples + offsets / lengths
    assert h2d < crop_bytes / 10                                          # the host path would have shipped 10.2 MB of fp32 crops
    assert d2h == feats.nbytes                                            # host result requested: the embeddings come back, crops never
    # the device path and the host path (device_crop=False) agree to fp32 round-off (same crops, same kernels)
    mh.device_crop = False
    host = mh._embed_files(files, num_eval)
    mh.device_crop = True
    assert float(np.abs(host - feats).max()) <= 1e-5 * max(1.0, float(np.abs(host).max()))


def test_per_trial_scoring_api_with_the_real_engine(golden_dir):
    """VERDICT r1 #6a (row a19): scoring.similarity_measure / ZT_norm_similarity / pnorm_similarity through the REAL engine
    against the reference's own per-trial outputs (tests/golden/scoring.npz)."""
    g = np.load(os.path.join(golden_dir, "scoring.npz"))
    Rn = torch.nn.functional.normalize(torch.from_numpy(g["R"]), p=2, dim=2)
    Cn = torch.nn.functional.normalize(torch.from_numpy(g["C"]), p=2, dim=2)
    n, top = Rn.shape[0], int(g["top"])
    cos = [scoring.similarity_measure("cosine", Rn[i], Cn[i]) for i in range(n)]
    assert float(np.abs(np.array(cos) - g["cosine"]).max()) <= 1e-5
    zt = [scoring.similarity_measure("zt_norm", Rn[i].numpy(), Cn[i].numpy(), cohorts=g["cohort"], top=top) for i in range(n)]
    assert float(np.abs(np.array(zt) - g["zt_norm"]).max()) <= 1e-4
    zt_d = [scoring.ZT_norm_similarity(Rn[i], Cn[i], g["cohort"]) for i in range(n)]          # default top=-1 drops the smallest
    assert float(np.abs(np.array(zt_d) - g["zt_norm_default_top"]).max()) <= 1e-4
    pn = [scoring.similarity_measure("pnorm", Rn[i], Cn[i]) for i in range(n)]                # host numpy by design
    assert float(np.abs(np.array(pn) - g["pnorm"]).max()) <= 1e-5
    # batched statement == per-trial statement on the real engine
    feats = np.concatenate([Rn.numpy(), Cn.numpy()])
    ia, ib = np.arange(n), np.arange(n, 2 * n)
    assert float(np.abs(scoring.score_trials(feats, ia, ib, "cosine") - g["cosine"]).max()) <= 1e-5
    assert float(np.abs(scoring.score_trials(feats, ia, ib, "norm", cohorts=g["cohort"], top=top) - g["zt_norm"]).max()) <= 1e-4
    assert float(np.abs(scoring.score_trials(feats, ia, ib, "pnorm") - g["pnorm"]).max()) <= 1e-5


def test_handles_on_two_devices_in_one_process():
    """launch attributes are kept per device, so one process may own handles on several GPUs.  (gpurun boxes show one
    GPU: skipped there; the bookkeeping itself is covered on the CPU by svhip_selftest.)"""
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=512), seed=1)
    wav = synth.synth_waveforms(4)
    outs = []
    for dev in (0, 1):
        eng = Engine(model="ecapa", compute="bf16", channels=512, max_batch=4, device=dev)
        eng.load_state_dict(sd)
        eng.finalize()
        outs.append(eng.embed_wave(wav))
        eng.close()
    assert np.array_equal(outs[0], outs[1])


def test_load_tensor_rejects_null_data():
    """a 0-d tensor still holds one element; NULL data is refused for every rank."""
    eng = Engine(model="ecapa", channels=64, max_batch=1)
    shape = (C.c_int64 * 1)(0)
    rc = eng.lib.svhip_load_tensor(eng.h, b"blocks.0.norm.norm.num_batches_tracked", None, shape, 0, _lib.I64)
    assert rc == -1 and b"null data" in eng.lib.svhip_last_error(eng.h)
    eng.close()


def test_enrol_to_score_stays_on_the_device(tmp_path, golden_dir):
    """evaluateFromList keeps the (n_files, num_eval, 192) block in HBM from the embed calls through
    normalisation to the scoring kernel.  PCIe traffic of a whole evaluation: the int16 PCM and the trial indices up, the P
    scores down — nothing else (src/model.py:386-448 moves every embedding to the host and back per trial)."""
    tmp = str(tmp_path)
    net = WrappedModel(SpeakerEncoder(**ARGS))
    mh = ModelHandling(net, **dict(ARGS, save_folder=tmp))
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=512), seed=E2E_SEED_W)
    net.module.load_state_dict({"__S__." + k: v for k, v in sd.items()})
    files, trial_path, lines = make_e2e_files(tmp)
    P = len(lines)
    pcm_bytes = sum(os.path.getsize(f) - 44 for f in files)
    g = np.load(os.path.join(golden_dir, "e2e_config1.npz"))
    for mode, key, extra_up in (("cosine", "scores_ne2", 0),):
        kw = dict(listfilename=trial_path, distributed=False, dataloader_options={}, cohorts_path="unused", num_eval=2, scoring_mode=mode)
        mh.evaluateFromList(**kw)                                          # warm: engines, scratch
        before = dict(sv_engine.TRANSFER_STATS)
        sc, lab, tr = mh.evaluateFromList(**kw)
        h2d = sv_engine.TRANSFER_STATS["h2d_bytes"] - before["h2d_bytes"]
        d2h = sv_engine.TRANSFER_STATS["d2h_bytes"] - before["d2h_bytes"]
        assert d2h == 4 * P, (d2h, P)                                      # the scores, nothing else
        assert pcm_bytes + 8 * P <= h2d <= pcm_bytes + 8 * P + 64 * len(files) + extra_up, (h2d, pcm_bytes, P)
        assert float(np.abs(np.array(sc) - g[key]).max()) <= 1e-4
    # 'norm': the cohort goes up as well, the cohort statistics never come down
    cohort = synth.synth_embeddings(300, seed=5)
    cpath = os.path.join(tmp, "cohort.npy")
    np.save(cpath, cohort)
    kw = dict(listfilename=trial_path, distributed=False, dataloader_options={}, cohorts_path=cpath, num_eval=2, scoring_mode="norm")
    host = ModelHandling(net, **dict(ARGS, save_folder=tmp, device_feats=False)).evaluateFromList(**kw)
    mh.evaluateFromList(**kw)
    before = dict(sv_engine.TRANSFER_STATS)
    sc, _, _ = mh.evaluateFromList(**kw)
    d2h = sv_engine.TRANSFER_STATS["d2h_bytes"] - before["d2h_bytes"]
    h2d = sv_engine.TRANSFER_STATS["h2d_bytes"] - before["h2d_bytes"]
    assert d2h == 4 * P
    assert h2d <= pcm_bytes + 8 * P + 64 * len(files) + cohort.nbytes
    assert float(np.abs(np.array(sc) - np.array(host[0])).max()) <= 1e-4 * max(1.0, float(np.abs(np.array(host[0])).max()))
    # cohorts_path=None (pairwise-distance scoring) and pnorm through the same device-resident path against the host-array path
    for kw2 in (dict(cohorts_path=None, scoring_mode="cosine"), dict(cohorts_path="unused", scoring_mode="pnorm")):
        kw = dict(listfilename=trial_path, distributed=False, dataloader_options={}, num_eval=2, **kw2)
        a = mh.evaluateFromList(**kw)[0]
        b = ModelHandling(net, **dict(ARGS, save_folder=tmp, device_feats=False)).evaluateFromList(**kw)[0]
        assert float(np.abs(np.array(a) - np.array(b)).max()) <= 1e-6


def test_prepare_embed_matches_embed_utterance(tmp_path):
    """VERDICT r2 #6 / src/model.py:610-670: prepare('embed') for a directory tree of speakers (embeds.pt (num_eval, nOut,
    n_class) + classes.npy) and for a list of sources (mean embedding); rows must equal the per-file embed_utterance means."""
    import shutil
    tmp = str(tmp_path)
    net = WrappedModel(SpeakerEncoder(**ARGS))
    mh = ModelHandling(net, **dict(ARGS, save_folder=tmp))
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=512), seed=E2E_SEED_W)
    net.module.load_state_dict({"__S__." + k: v for k, v in sd.items()})
    files, _, _ = make_e2e_files(tmp)
    root = os.path.join(tmp, "spk")
    for s in range(2):
        os.makedirs(os.path.join(root, f"speaker{s}"))
        for f in files[3 * s:3 * s + 3]:
            shutil.copy(f, os.path.join(root, f"speaker{s}", os.path.basename(f)))
    out = os.path.join(tmp, "prep")
    os.makedirs(out)
    assert mh.prepare(save_path=out, prepare_type="embed", num_eval=4, source=root) is True
    embeds = torch.load(os.path.join(out, "embeds.pt"))
    classes = np.load(os.path.join(out, "classes.npy"), allow_pickle=True).item()
    assert tuple(embeds.shape) == (4, 192, 2) and sorted(classes.values()) == ["speaker0", "speaker1"]
    for idx, name in classes.items():
        s = int(name[-1])
        want = np.stack([mh.embed_utterance(f, num_eval=4, normalize=True).numpy() for f in sorted(files[3 * s:3 * s + 3])], 0).mean(0)
        got = embeds[:, :, idx].numpy()
        assert float(np.abs(got - want).max()) <= 1e-6
        assert np.allclose(np.linalg.norm(want, axis=1) <= 1.0 + 1e-6, True)
    m = mh.prepare(save_path=None, prepare_type="embed", num_eval=4, source=[files[0], files[1]])
    want = np.stack([mh.embed_utterance(f, num_eval=4, normalize=True).numpy() for f in files[:2]], 0).mean(0)
    assert tuple(m.shape) == (4, 192) and float(np.abs(m.numpy() - want).max()) <= 1e-6
    with pytest.raises(ValueError):
        mh.prepare(save_path=None, prepare_type="embed", num_eval=4, source=None)
