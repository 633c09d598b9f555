"""GPU: BASELINE config 1 — the drop-in SpeakerEncoder / ModelHandling on the HIP path reproduce the
(scores, labels, trials) triple and embed_utterance outputs captured from the REFERENCE'S OWN
ModelHandling on CPU (tests/golden/e2e_config1.npz, oracle/make_golden.py::golden_e2e)."""
import os

import numpy as np
import pytest

from speakerverification_amd import synth
from speakerverification_amd.model import ModelHandling, SpeakerEncoder, WrappedModel
from tests.e2e_data import E2E_SEED_W, make_e2e_files

pytestmark = pytest.mark.gpu

ARGS = dict(
    device="cuda", gpu=0, model={"name": "ECAPA_TDNN", "nOut": 192},
    criterion={"name": "AAmSoftmaxAP", "margin": 0.25, "scale": 30},
    classifier={"input_size": 192, "out_neurons": 10},
    optimizer={"name": "adam", "weight_decay": 2e-5, "lr_decay": 0.95},
    callbacks={"name": "steplr"}, features="melspectrogram", include_top=False, n_mels=80, nClasses=10,
    lr=0.001, step_size=10, channels=[512] * 4 + [1536],
    dataloader_options={"nPerSpeaker": 2, "num_workers": 0, "batch_size": 2},
    audio_spec={"sample_rate": 16000, "channels": 1, "sentence_len": 2.0, "win_len": 0.025, "hop_len": 0.01},
    augment=False, augment_options={"augment_chain": []},
)


@pytest.fixture(scope="module")
def handler(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("e2e"))
    net = WrappedModel(SpeakerEncoder(**ARGS))
    mh = ModelHandling(net, **dict(ARGS, save_folder=tmp))
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=512), seed=E2E_SEED_W)
    # checkpoint round trip through the reference's layout (keys '__S__.*', torch.save / loadParameters)
    import torch
    ck = os.path.join(tmp, "best_state.pt")
    torch.save({"__S__." + k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()} | {"__L__.w": torch.zeros(1)}, ck)
    mh.loadParameters(ck, show_error=False)
    return mh, tmp


def test_evaluate_from_list_matches_reference(handler, golden_dir):
    mh, tmp = handler
    g = np.load(os.path.join(golden_dir, "e2e_config1.npz"))
    files, trial_path, lines = make_e2e_files(tmp)
    for ne in (2, 3):
        sc, lab, tr = mh.evaluateFromList(listfilename=trial_path, distributed=False,
                                          dataloader_options=ARGS["dataloader_options"], cohorts_path="unused",
                                          num_eval=ne, scoring_mode="cosine")
        assert lab == list(g[f"labels_ne{ne}"])
        assert tr == [ln.split()[1] + " " + ln.split()[2] for ln in lines]
        err = float(np.abs(np.array(sc) - g[f"scores_ne{ne}"]).max())
        assert err <= 1e-4, err            # north_star: cosine scores within 1e-4 of the reference CPU path


def test_embed_utterance_matches_reference(handler, golden_dir):
    mh, tmp = handler
    g = np.load(os.path.join(golden_dir, "e2e_config1.npz"))
    files, _, _ = make_e2e_files(tmp)
    e = mh.embed_utterance(files[0], num_eval=3, normalize=True).numpy()
    assert e.shape == g["embed_utt0_ne3"].shape == (3, 192)
    assert float(np.abs(e - g["embed_utt0_ne3"]).max()) <= 1e-4
    a = (0.25 * np.sin(np.arange(40000) / 7.0)).astype(np.float32)
    e2 = mh.embed_utterance(a, num_eval=2, normalize=False).numpy()
    ref = g["embed_array_ne2"]
    assert float(np.abs(e2 - ref).max()) <= 1e-4 * max(1.0, float(np.abs(ref).max()))


def test_single_crop_returns_vector_like_the_reference():
    """SpeakerEncoder.forward on one row -> (nOut,) (stack(dim=1).squeeze(), src/model.py:125)."""
    enc = SpeakerEncoder(**ARGS)
    out = enc.forward(synth.synth_waveforms(1))
    assert tuple(out.shape) == (192,)
    out = enc.forward(synth.synth_waveforms(3))
    assert tuple(out.shape) == (3, 192)


def test_prepare_cohorts_and_test_from_list_match_reference(handler, golden_dir):
    """ModelHandling.prepare('cohorts') (src/model.py:578-609) and testFromList (:455-554) vs the reference's outputs."""
    mh, tmp = handler
    g = np.load(os.path.join(golden_dir, "e2e_config1.npz"))
    files, _, _ = make_e2e_files(tmp)
    meta = os.path.join(tmp, "train_meta.txt")
    with open(meta, "w") as fh:
        fh.writelines(f"spk{i // 4} {f}\n" for i, f in enumerate(files))
    out = os.path.join(tmp, "cohort.npy")
    assert mh.prepare(save_path=out, prepare_type="cohorts", num_eval=2, source=meta) is True
    cohort = np.load(out)
    assert cohort.shape == g["cohort_ne2"].shape
    assert float(np.abs(cohort - g["cohort_ne2"]).max()) <= 1e-4
    pairs = os.path.join(tmp, "pairs.txt")
    with open(pairs, "w") as fh:
        fh.write("audio_1,audio_2\n")
        fh.writelines(f"{files[i]},{files[i + 1]}\n" for i in range(5))
    res = mh.testFromList(test_list=pairs, thresh_score=0.5, cohorts_path=None, num_eval=2, scoring_mode="cosine",
                          output_file=os.path.join(tmp, "pairs_out.txt"))
    got = np.array([float(r.split(",")[2]) for r in res])
    assert float(np.abs(got - g["test_scores_ne2"]).max()) <= 1e-4
    # AS-norm mode end to end on the device embeddings: counterpart == the reference's ZT_norm_similarity per trial
    from oracle import scoring as o_scoring
    import torch
    rng = np.random.Generator(np.random.PCG64(3))
    coh = rng.standard_normal((300, 192)).astype(np.float32)
    coh /= np.linalg.norm(coh, axis=1, keepdims=True)
    cpath = os.path.join(tmp, "big_cohort.npy")
    np.save(cpath, coh)
    trial_path = os.path.join(tmp, "trials.txt")
    sc, lab, tr = mh.evaluateFromList(listfilename=trial_path, distributed=False, dataloader_options={}, cohorts_path=cpath,
                                      num_eval=2, scoring_mode="norm")
    embs = {f: torch.nn.functional.normalize(mh.embed_utterance(f, num_eval=2, normalize=False), p=2, dim=1).numpy() for f in files}
    want = [o_scoring.zt_norm_similarity(embs[t.split()[0]], embs[t.split()[1]], coh, 200) for t in tr]
    assert float(np.abs(np.array(sc) - np.array(want)).max()) <= 2e-3      # (s - mu) / sigma amplifies 1e-6 score noise by 1/sigma ~ 1e2


def test_evaluate_from_list_f32x3_matches_reference(tmp_path, golden_dir):
    """the split-bf16 compute mode (hip_compute='f32x3') end to end: the reference's golden cosine scores within the same 1e-4"""
    import torch
    tmp = str(tmp_path)
    args = dict(ARGS, hip_compute="f32x3", save_folder=tmp)
    net = WrappedModel(SpeakerEncoder(**args))
    mh = ModelHandling(net, **args)
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=512), seed=E2E_SEED_W)
    net.module.load_state_dict({"__S__." + k: v for k, v in sd.items()})
    g = np.load(os.path.join(golden_dir, "e2e_config1.npz"))
    files, trial_path, lines = make_e2e_files(tmp)
    sc, lab, tr = mh.evaluateFromList(listfilename=trial_path, distributed=False, dataloader_options={}, cohorts_path="unused",
                                      num_eval=2, scoring_mode="cosine")
    err = float(np.abs(np.array(sc) - g["scores_ne2"]).max())
    print("f32x3 e2e cosine score error", err)
    assert err <= 1e-4, err


@pytest.mark.parametrize("compute", ["f32", "f32x3"])
def test_speaker_fixture_with_sensitive_scores_matches_reference(tmp_path, golden_dir, compute):
    """the 28 golden scores of the config-1 case span 0.9970 - 0.9996, so a plumbing error worth less than 1e-3
    of cosine would pass it.  This case (tests/e2e_data.py: four synthetic speakers x two files, asp_bn statistics calibrated to
    the files as a trained BatchNorm's would be; generated by the REFERENCE'S ModelHandling in oracle/make_golden.py) has cosine
    scores from 0.05 to 0.78, same-speaker pairs on top: swapped files, a wrong crop or a wrong crop mean move a score by tenths.
    The calibrated BatchNorm divides the pooled statistics by their small spread over the files (~50 x amplification), so this is
    also the hardest fixture for the arithmetic: measured, exact fp32 is within 3.3e-6 of the reference's scores and the split-bf16
    mode (2^-17 per product) within 2.6e-5 — both are held to north_star's 1e-4."""
    import torch
    from speakerverification_amd import scoring
    from tests.e2e_data import E2E2_SEED_W, make_e2e_speaker_files
    tmp = str(tmp_path)
    g = np.load(os.path.join(golden_dir, "e2e_speakers.npz"))
    args = dict(ARGS, hip_compute=compute, save_folder=tmp)
    net = WrappedModel(SpeakerEncoder(**args))
    mh = ModelHandling(net, **args)
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=512), seed=E2E2_SEED_W)
    sd["asp_bn.norm.running_mean"] = g["asp_bn_running_mean"]
    sd["asp_bn.norm.running_var"] = g["asp_bn_running_var"]
    net.module.load_state_dict({"__S__." + k: v for k, v in sd.items()})
    files, trial_path, lines = make_e2e_speaker_files(tmp)
    sc, lab, tr = mh.evaluateFromList(listfilename=trial_path, distributed=False, dataloader_options={}, cohorts_path="unused",
                                      num_eval=2, scoring_mode="cosine")
    ref = g["scores_ne2"]
    assert lab == list(g["labels_ne2"]) and len(sc) == 28
    assert ref.max() - ref.min() >= 0.3 and ref[g["labels_ne2"] == 1].min() > ref[g["labels_ne2"] == 0].max() - 0.3      # the fixture IS sensitive
    err = float(np.abs(np.array(sc) - ref).max())
    emb = np.stack([mh.embed_utterance(f, num_eval=2, normalize=False).numpy() for f in files])
    e_err = float(np.abs(emb - g["embeddings_ne2"]).max() / np.abs(g["embeddings_ne2"]).max())
    # a mixed-up file order or crop would show here as O(0.1): the check that the scores are attached to the right trials
    shuffled = np.abs(np.array(sc) - ref[::-1]).max()
    print(f"{compute}: speaker fixture: cosine score error {err:.2e}, embedding error / scale {e_err:.2e} (reversed trial list would give {shuffled:.2f})")
    bar = 1e-4
    assert err <= bar and e_err <= bar and shuffled > 0.1
    # cohort preparation (speaker means) and AS-norm over this cohort, top 3 of 4: the reference's ZT_norm_similarity per trial
    meta = os.path.join(tmp, "train_meta.txt")
    with open(meta, "w") as fh:
        fh.writelines(f"spk{i // 2} {f}\n" for i, f in enumerate(files))
    cpath = os.path.join(tmp, "cohort.npy")
    assert mh.prepare(save_path=cpath, prepare_type="cohorts", num_eval=2, source=meta) is True
    cohort = np.load(cpath)
    assert float(np.abs(cohort - g["cohort_ne2"]).max()) <= bar
    feats = torch.nn.functional.normalize(torch.from_numpy(emb), p=2, dim=2).numpy()
    index = {f: i for i, f in enumerate(files)}
    ia = [index[ln.split()[1]] for ln in lines]
    ib = [index[ln.split()[2]] for ln in lines]
    zt = scoring.score_trials(feats, ia, ib, "norm", cohorts=g["cohort_ne2"], top=3)
    z_err = float(np.abs(np.asarray(zt) - g["zt_norm_top3"]).max())
    print(f"{compute}: AS-norm (top 3 of 4) error {z_err:.2e} on scores spanning {g['zt_norm_top3'].min():.2f} .. {g['zt_norm_top3'].max():.2f}")
    assert z_err <= 50 * bar           # (s - mu) / sigma with sigma ~ 0.1 - 0.3 over three cohort scores
