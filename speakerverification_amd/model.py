"""Counterparts of the reference's ``src/model.py`` for the inference hot path.

Same class / method names, arguments and return shapes as the reference so that
``inference.py``-style drivers keep working:

    net = WrappedModel(SpeakerEncoder(**vars(args)))
    speaker_model = ModelHandling(net, **vars(args))
    speaker_model.loadParameters(path)
    scores, labels, trials = speaker_model.evaluateFromList(listfilename=..., distributed=False,
                                                           dataloader_options=..., cohorts_path=...,
                                                           num_eval=10, scoring_mode='cosine')

What changed underneath: crops of MANY files are batched into one device call
(reference: batch = one file), embeddings stay in one dense matrix, and the whole trial list is
scored by one kernel launch per mode (reference: three host<->device crossings per trial).
Training (``fit``), ONNX export and report writing are out of scope and raise NotImplementedError.
"""
from __future__ import annotations

import csv
import importlib
import itertools
import os
from pathlib import Path

import numpy as np

from . import distributed as sv_dist
from . import scoring
from .audio import loadWAV, read_pcm16
from .engine import _is_torch, to_device, to_host

try:
    import torch
    import torch.nn.functional as F
except Exception:  # pragma: no cover
    torch = None

# losses whose test_normalize is False in the reference (src/losses/Prototypical.py:21); every other
# loss sets it True (src/losses/*.py), which is what reaches the hot path (src/model.py:90-94,421-423)
_NO_TEST_NORMALIZE = {"Prototypical"}


def _host(x):
    return x.detach().cpu().numpy() if _is_torch(x) else np.asarray(x)


class WrappedModel:
    """src/model.py:24-33 — the DataParallel-compatibility wrapper: forward delegates to .module"""

    def __init__(self, model):
        self.module = model

    def forward(self, x, label=None):
        return self.module(x, label)

    __call__ = forward

    def eval(self):
        self.module.eval()
        return self

    def train(self, mode=True):
        self.module.train(mode)
        return self

    def state_dict(self):
        return self.module.state_dict()

    def parameters(self):
        return self.module.parameters()


class SpeakerEncoder:
    """src/model.py:36-135: feature extractor + __S__ (+ the loss's test_normalize flag)."""

    def __init__(self, model, criterion=None, classifier=None, optimizer=None, features="melspectrogram",
                 device="cuda", gpu=0, include_top=False, **kwargs):
        self.model = model
        self.criterion = criterion or {"name": "AAmSoftmaxAP"}
        self.classifier = classifier
        self.optimizer = optimizer
        self.gpu = gpu
        self.device = f"{device}:{gpu}"
        self.n_mels = kwargs.get("n_mels", 80)
        self.features = features.lower()
        if include_top:
            raise NotImplementedError("include_top (classification head) is outside the inference hot path")
        kw = dict(kwargs)
        kw.setdefault("n_mels", self.n_mels)
        kw.setdefault("augment", False)
        kw.setdefault("augment_options", {"augment_chain": []})
        if self.features in ("mfcc", "melspectrogram"):
            fx = importlib.import_module("speakerverification_amd.models.FeatureExtraction.feature")
            self.compute_features = getattr(fx, self.features)(**kw).to(self.device)
        else:
            self.compute_features = None
        if not isinstance(self.model["name"], str):
            raise NotImplementedError("Mixed_model lists are outside the hot path")
        mod = importlib.import_module("speakerverification_amd.models." + self.model["name"])
        self.__S__ = mod.MainModel(nOut=self.model["nOut"], features=self.features, device=self.device, **kw).to(self.device)
        self.test_normalize = self.criterion.get("name") not in _NO_TEST_NORMALIZE
        self.training = False

    # nn.Module look-alikes ------------------------------------------------------------------------------
    def eval(self):
        self.__S__.eval()
        return self

    def train(self, mode=True):
        self.__S__.train(mode)
        return self

    def parameters(self):
        return self.__S__.parameters()

    def state_dict(self):
        """reference checkpoint layout: keys prefixed with __S__. (src/model.py:710-716)"""
        return {"__S__." + k: v for k, v in self.__S__.state_dict().items()}

    def load_state_dict(self, sd, strict=True):
        sub = {k[len("__S__."):]: v for k, v in sd.items() if k.startswith("__S__.")}
        return self.__S__.load_state_dict(sub, strict=strict)

    def forward(self, data, label=None):
        """src/model.py:104-125 with label=None: (..., L) waveforms -> (N, nOut) embeddings
        ((nOut,) for a single row, as the reference's stack(dim=1).squeeze())."""
        if label is not None:
            raise NotImplementedError("loss evaluation (training) is outside the inference hot path")
        L = data.shape[-1]
        data = data.reshape(-1, L)
        if self.features != "raw" and hasattr(self.__S__, "embed_wave") and self._fusable():
            out = self.__S__.embed_wave(data)                   # fbank + forward in one library call
        else:
            inp = self.compute_features(data) if self.features != "raw" else data
            out = self.__S__.forward(inp)
        return out

    __call__ = forward

    def _fusable(self):
        """the fused path bakes the front-end defaults into the model handle; only use it when the
        feature extractor has exactly those defaults"""
        p = getattr(self.compute_features, "p", None)
        return p == dict(sr=8000, n_fft=512, win_length=200, n_mels=self.n_mels, hop_length=80,
                         fmin=0.0, fmax=None, pre_emphasis=True)


class ModelHandling:
    """src/model.py:138-792, inference methods only."""

    def __init__(self, encoder_model, optimizer=None, callbacks=None, device="cuda", gpu=0, mixedprec=False, **kwargs):
        self.kwargs = kwargs
        self.save_path = kwargs.get("save_folder", ".")
        self.audio_spec = kwargs["audio_spec"]
        self.__model__ = encoder_model
        self.model_name = self.__model__.module.model["name"]
        self.criterion = self.__model__.module.criterion["name"]
        self.gpu = gpu
        self.device = f"{device}:{gpu}"
        self.embed_batch = int(kwargs.get("embed_batch", 256))     # crops per device call (cross-file batching)
        self.device_crop = bool(kwargs.get("device_crop", True))   # 16-bit PCM files: crop on the device (svhip_crop_pcm16)
        self.device_feats = bool(kwargs.get("device_feats", True)) # keep enrol -> score on the device when a GPU is there

    # ---- training-only surface ---------------------------------------------------------------------------
    def fit(self, *a, **k):
        raise NotImplementedError("training is outside the scope of the MI355X inference path")

    def export_onnx(self, *a, **k):
        raise NotImplementedError("ONNX export is outside the scope of the MI355X inference path")

    # ---- embedding -------------------------------------------------------------------------------------------
    def _embed_crops(self, crops: np.ndarray) -> np.ndarray:
        """(n, L) float32 crops -> (n, nOut) float32, in device batches of embed_batch rows."""
        outs = []
        for i in range(0, crops.shape[0], self.embed_batch):
            o = self.__model__.forward(np.ascontiguousarray(crops[i:i + self.embed_batch], dtype=np.float32))
            o = o.detach().cpu().numpy() if _is_torch(o) else np.asarray(o)
            outs.append(o.reshape(-1, o.shape[-1]))
        return np.concatenate(outs, 0)

    def _device_path_ok(self, num_eval):
        """the device crop -> embed path needs fixed-length crops, a GPU, and a model that takes raw waveforms on the device"""
        if not self.device_crop or num_eval <= 0 or torch is None or not torch.cuda.is_available():
            return False
        enc = self.__model__.module
        S = getattr(enc, "__S__", None)
        if S is None:
            return False
        if enc.features == "raw":
            return callable(getattr(S, "forward", None)) and getattr(S, "accepts_device_wave", False)
        return hasattr(S, "embed_wave") and enc._fusable()

    def _feats_on_device(self):
        """evaluateFromList / testFromList keep the (n_files, num_eval, nOut) block in HBM from the embed calls to the scoring
        kernels when there is a GPU to hold it (only the P scores come back); CPU test doubles keep numpy."""
        return self.device_feats and torch is not None and torch.cuda.is_available()

    def _embed_files(self, files, num_eval, on_device=False):
        """crops of several files travel in one batch (the reference runs one file per forward,
        src/model.py:386-394).  Returns a dense (n_files, num_eval, nOut) float32 block: a numpy array, or with
        on_device=True a CUDA tensor that never visits the host.

        16-bit PCM files take the device path: the int16 samples are uploaded once, `svhip_crop_pcm16` cuts the eval-mode
        crops (audio_loader.py:110-150) straight into an HBM buffer, and the embed call consumes that buffer as a device
        pointer in batches of `embed_batch` crops across files — no fp32 crop crosses PCIe.
        Other sources (float / 24-bit files, ndarrays, num_eval == 0) keep the host path (loadWAV)."""
        feats = None
        max_audio = int(self.audio_spec["sentence_len"] * self.audio_spec["sample_rate"])
        use_dev = self._device_path_ok(num_eval)
        pending, owners = [], []            # host path: fp32 crops
        pcm_pending, pcm_owners = [], []    # device path: int16 files
        ncrop = max(1, num_eval)

        def store(idx_n, emb):
            nonlocal feats
            D = emb.shape[-1]
            if feats is None:
                feats = (torch.zeros((len(files), ncrop, D), dtype=torch.float32, device=f"cuda:{self.gpu}") if on_device
                         else np.zeros((len(files), ncrop, D), np.float32))
            if on_device and not (_is_torch(emb) and emb.is_cuda):
                emb = to_device(np.ascontiguousarray(_host(emb), np.float32), self.gpu)      # host-path files: only their embeddings go up
            elif not on_device:
                emb = to_host(emb) if _is_torch(emb) else np.asarray(emb)
            # files of one flush are consecutive and (almost always) full: one block copy; otherwise file by file
            idx0 = idx_n[0][0]
            if all(n == ncrop for _, n in idx_n) and [i for i, _ in idx_n] == list(range(idx0, idx0 + len(idx_n))):
                feats[idx0:idx0 + len(idx_n)] = emb.reshape(len(idx_n), ncrop, D)
                return
            pos = 0
            for idx, n in idx_n:
                feats[idx, :n] = emb[pos:pos + n]
                pos += n

        def flush():
            if pending:
                store(owners, self._embed_crops(np.concatenate(pending, 0)))
                pending.clear()
                owners.clear()

        def flush_pcm():
            if not pcm_pending:
                return
            eng = scoring.scoring_engine(self.gpu)
            n = len(pcm_pending) * num_eval
            crops = torch.empty((n, max_audio), dtype=torch.float32, device=f"cuda:{self.gpu}")
            eng.crop_pcm16(pcm_pending, num_eval, max_audio, out=crops)
            outs = []
            for i in range(0, n, self.embed_batch):
                o = self.__model__.forward(crops[i:i + self.embed_batch])
                outs.append(o.reshape(-1, o.shape[-1]))
            store(pcm_owners, torch.cat(outs, 0) if len(outs) > 1 else outs[0])
            pcm_pending.clear()
            pcm_owners.clear()

        for idx, f in enumerate(files):
            pcm = read_pcm16(f, self.audio_spec["sample_rate"]) if (use_dev and isinstance(f, (str, Path))) else None
            if pcm is not None:
                pcm_pending.append(pcm)
                pcm_owners.append((idx, num_eval))
                if len(pcm_pending) * num_eval >= self.embed_batch:
                    flush_pcm()
                continue
            audio = loadWAV(f, self.audio_spec, evalmode=True, augment=False, augment_options=[], num_eval=num_eval,
                            random_chunk=False)
            if num_eval == 0:                       # whole file: variable length, one forward per file
                flush()
                store([(idx, 1)], self._embed_crops(audio))
                continue
            pending.append(audio)
            owners.append((idx, audio.shape[0]))
            if sum(n for _, n in owners) >= self.embed_batch:
                flush()
        flush()
        flush_pcm()
        return feats

    def embed_utterance(self, source, num_eval=20, normalize=False):
        """src/model.py:675-704: path or ndarray -> (num_eval, nOut) CPU tensor (L2-normalised on request)."""
        audio = loadWAV(source, self.audio_spec, evalmode=True, augment=False, augment_options=[], num_eval=num_eval,
                        random_chunk=False)
        emb = self._embed_crops(audio)
        if normalize:
            eng = scoring.scoring_engine(self.gpu)
            emb = np.ascontiguousarray(emb)
            eng.l2norm_(emb)
        return torch.from_numpy(emb) if torch is not None else emb

    # ---- evaluation ----------------------------------------------------------------------------------------------
    def _score(self, feats, ia, ib, scoring_mode, cohorts, cohorts_path):
        """(n_files, n_crops, D) embeddings (numpy, or a CUDA tensor that stays where it is) + trial indices -> P float32 scores
        on the host.  Normalisation and every scoring mode are library kernels; with device embeddings the PCIe traffic is the
        index lists up, the P scores down (and the cohort up, for 'norm')."""
        on_dev = _is_torch(feats) and feats.is_cuda
        if self.__model__.module.test_normalize:                            # F.normalize(p=2, dim=1), model.py:421-423
            if on_dev:
                feats = feats.contiguous()
                scoring.scoring_engine(self.gpu).l2norm_(feats.view(-1, feats.shape[-1]))
            else:
                flat = np.ascontiguousarray(np.asarray(feats, np.float32).reshape(-1, feats.shape[-1]))
                scoring.scoring_engine(self.gpu).l2norm_(flat)
                feats = flat.reshape(feats.shape)
        if cohorts_path is None:
            # model.py:425-431: F.pairwise_distance(ref (n,D,1), com (1,D,n)) takes the 2-norm over the LAST axis of the
            # broadcast (n, D, n) difference, i.e. over the crops j of `com`; score = -mean.  One kernel over the trial list
            # (svhip_score_trials, SVHIP_TRIAL_PDIST): O(P n^2 D) work, no (P, n, D, n) temporary.
            mode = "pdist"
        elif scoring_mode in ("norm", "cosine", "pnorm"):
            mode = scoring_mode
        else:
            raise ValueError(f"unknown scoring_mode {scoring_mode}")
        s = scoring.score_trials(feats, ia, ib, mode, cohorts=cohorts, top=200, device=self.gpu)
        return np.asarray(to_host(s), np.float32)

    def evaluateFromList(self, listfilename, distributed=False, dataloader_options=None,
                         cohorts_path="checkpoint/dump_cohorts.npy", num_eval=10, scoring_mode="cosine", **kwargs):
        """src/model.py:306-450.  Trial lines ``label file1 file2`` (or CSV with a header line).
        Returns (all_scores, all_labels, all_trials); empty lists on ranks != 0 when distributed."""
        self.__model__.eval()
        cohorts = np.load(cohorts_path) if (cohorts_path is not None and scoring_mode == "norm") else None
        with open(listfilename) as fh:
            lines = fh.readlines()
        determinator = "," if len(lines[0].split(",")) > 1 else " "
        start_index = 1 if determinator == "," else 0
        files = list(itertools.chain(*[x.strip().split(determinator)[-2:] for x in lines[start_index:]]))
        setfiles = sorted(set(files))
        rank, world = sv_dist.rank_world() if distributed else (0, 1)
        lo, hi, _ = sv_dist.shard_bounds(len(setfiles), rank, world)
        on_dev = self._feats_on_device()
        local = self._embed_files(setfiles[lo:hi], num_eval, on_device=on_dev)
        if distributed and sv_dist.is_distributed():
            nOut = self.__model__.module.model["nOut"]
            if local is None:
                local = (torch.zeros((0, max(1, num_eval), nOut), dtype=torch.float32, device=f"cuda:{self.gpu}") if on_dev
                         else np.zeros((0, max(1, num_eval), nOut), np.float32))
            feats = self._gather_rows(local, len(setfiles))            # ONE collective (reference: all_gather_object)
        else:
            feats = local
        all_scores, all_labels, all_trials = [], [], []
        index = {str(Path(f)): i for i, f in enumerate(setfiles)}
        index.update({f: i for i, f in enumerate(setfiles)})
        ia, ib = [], []
        for line in lines[start_index:]:
            data = line.strip().split(determinator) if determinator == "," else line.split()
            if len(data) < 3:
                continue
            data = data[-3:]
            ia.append(index[data[1]])
            ib.append(index[data[2]])
            all_labels.append(int(data[0]))
            all_trials.append(data[1] + " " + data[2])
        ia, ib = np.asarray(ia, np.int32), np.asarray(ib, np.int32)
        if kwargs.get("shard_scoring") and world > 1:
            # row-sharded scoring: every rank scores the trials whose enrol file sits in its block; the
            # P floats are then assembled with a second, small all-gather
            mine = sv_dist.trial_rows_of_rank(ia, len(setfiles), rank, world)
            s_loc = self._score(feats, ia[mine], ib[mine], scoring_mode, cohorts, cohorts_path)
            counts = [len(sv_dist.trial_rows_of_rank(ia, len(setfiles), r, world)) for r in range(world)]
            per = max(1, max(counts))
            pad = np.zeros((per, 1), np.float32)
            pad[:len(s_loc), 0] = s_loc
            allp = np.asarray(to_host(self._gather_rows(pad, per * world, exact=True))).reshape(world, per)
            if rank == 0:
                s = np.empty(len(ia), np.float32)
                for r in range(world):
                    s[sv_dist.trial_rows_of_rank(ia, len(setfiles), r, world)] = allp[r, :counts[r]]
                all_scores = [float(v) for v in s]
        elif rank == 0:
            s = self._score(feats, ia, ib, scoring_mode, cohorts, cohorts_path)
            all_scores = [float(v) for v in s]
        if rank != 0:
            return [], [], []
        return all_scores, all_labels, all_trials

    def _gather_rows(self, local, n_total: int, exact=False):
        """the path's exchange step.  With an Engine that owns RCCL (a real one on a GPU): svhip_allgather_rows on the scoring
        engine's stream, host or device rows alike; torch.distributed only ships the RCCL id once, and the communicator is cached
        next to the engine (scoring.lib_comm), not per ModelHandling.  Otherwise (gloo CPU tests with a stand-in engine):
        torch.distributed all_gather_into_tensor.  exact=True: every rank passes exactly n_total / world rows."""
        rank, world = sv_dist.rank_world()
        eng = scoring.scoring_engine(self.gpu)
        # (an engine without comm_init is a CPU stand-in; more ranks than GPUs means ranks share a device, which RCCL refuses)
        if hasattr(eng, "comm_init") and torch is not None and torch.cuda.is_available() and world <= torch.cuda.device_count():
            comm = scoring.lib_comm(self.gpu, rank, world)
            if not _is_torch(local):
                local = np.ascontiguousarray(local, np.float32)
            return comm.all_gather_rows(local, n_total)
        return sv_dist.all_gather_rows(torch.from_numpy(np.ascontiguousarray(_host(local))), n_total).numpy()

    def testFromList(self, test_list="evaluation_test.txt", thresh_score=0.5, distributed=False, dataloader_options=None,
                     cohorts_path=None, num_eval=10, scoring_mode="norm", output_file=None):
        """src/model.py:455-554: CSV ``audio_1,audio_2`` -> writes audio_1,audio_2,pred_label,score."""
        self.__model__.eval()
        cohorts = np.load(cohorts_path) if (cohorts_path is not None and scoring_mode == "norm") else None
        save_root = os.path.join(self.save_path, f"{self.model_name}/{self.criterion}/result")
        if output_file is None:
            output_file = test_list.replace(".txt", "_result.txt")
        write_file = Path(save_root, output_file) if os.path.split(output_file)[0] == "" else output_file
        files, lines = [], []
        with open(Path(test_list), newline="") as rf:
            reader = csv.reader(rf, delimiter=",")
            next(reader, None)
            for row in reader:
                files += [row[0], row[1]]
                lines.append(row)
        setfiles = sorted(set(files))
        feats = self._embed_files(setfiles, num_eval, on_device=self._feats_on_device())
        index = {f: i for i, f in enumerate(setfiles)}
        ia = np.asarray([index[r[0]] for r in lines], np.int32)
        ib = np.asarray([index[r[1]] for r in lines], np.int32)
        # the reference routes every mode through similarity_measure here (cohorts_path may be None for cosine)
        s = self._score(feats, ia, ib, scoring_mode, cohorts, cohorts_path if scoring_mode == "norm" else "")
        results = []
        os.makedirs(os.path.dirname(str(write_file)) or ".", exist_ok=True)
        with open(write_file, "w", newline="") as wf:
            w = csv.writer(wf, delimiter=",")
            w.writerow(["audio_1", "audio_2", "pred_label", "score"])
            for row, score in zip(lines, s):
                score = float(score)
                w.writerow([row[0], row[1], "1" if score >= thresh_score else "0", score])
                results.append(f"{Path(row[0]).name},{Path(row[1]).name},{score}")
        return results

    # ---- cohort / enrolment preparation ------------------------------------------------------------------------------
    def prepare(self, save_path=None, prepare_type="cohorts", num_eval=10, source=None, **kwargs):
        """src/model.py:559-670."""
        self.__model__.eval()
        if not source:
            raise ValueError("Please provide appropriate source!")
        if prepare_type == "cohorts":
            n_emb_per_spk = 3
            assert isinstance(source, str), "Please provide path to train metadata files"
            spk_files = {}
            with open(Path(source)) as fh:
                for line in fh.read().splitlines():
                    data = line.split()
                    if len(data) >= 2:
                        spk_files.setdefault(data[0], []).append(data[1])
            files, owner = [], []
            for si, (spk, paths) in enumerate(spk_files.items()):
                for pth in paths[:n_emb_per_spk]:
                    files.append(pth)
                    owner.append(si)
            feats = self._embed_files(files, num_eval)                              # (n_files, n_crops, D)
            flat = np.ascontiguousarray(feats.reshape(-1, feats.shape[-1]))
            scoring.scoring_engine(self.gpu).l2norm_(flat)                          # embed_utterance(normalize=True)
            feats = flat.reshape(feats.shape)
            owner = np.asarray(owner)
            cohort = np.vstack([feats[owner == si].reshape(-1, feats.shape[-1]).mean(axis=0, keepdims=True)
                                for si in range(len(spk_files))])
            if save_path:
                np.save(save_path, np.array(cohort))
            return True
        if prepare_type == "embed":
            norm = self.__model__.module.test_normalize
            if isinstance(source, str):
                speaker_dirs = [x for x in Path(source).iterdir() if x.is_dir()]
                embeds, classes = [], {}
                for idx, d in enumerate(speaker_dirs):
                    classes[idx] = d.stem
                    fl = [str(f) for f in d.glob("*.wav")]
                    e = np.stack([self.embed_utterance(f, num_eval=num_eval, normalize=norm).numpy() for f in fl], 0)
                    embeds.append(e.mean(axis=0))
                embeds = torch.from_numpy(np.stack(embeds, -1))                     # (num_eval, nOut, n_class)
                if save_path:
                    torch.save(embeds, Path(save_path, "embeds.pt"))
                    np.save(str(Path(save_path, "classes.npy")), classes)
                return True
            if isinstance(source, list):
                e = np.stack([self.embed_utterance(a, num_eval=num_eval, normalize=norm).numpy() for a in source], 0)
                mean_embed = torch.from_numpy(e.mean(axis=0))
                if save_path:
                    torch.save(mean_embed, Path(save_path, "embeds.pt"))
                return mean_embed
        raise NotImplementedError

    # ---- checkpoints ------------------------------------------------------------------------------------------------------
    def saveParameters(self, path):
        torch.save(self.__model__.module.state_dict(), path)

    def loadParameters(self, path, show_error=True):
        """src/model.py:718-746: name-matched copy; 'module.' prefixes stripped; unknown or mis-shaped
        tensors are reported and skipped.  (map_location is 'cpu': weights are repacked for the device.)"""
        if not os.path.exists(path):
            raise FileNotFoundError("Model's path is not exists")
        from . import checkpoint
        if checkpoint.is_blob(path):                     # packed blob (checkpoint.convert_checkpoint): bare __S__ names
            _, sd = checkpoint.read_blob(path)
            loaded = {"__S__." + k: torch.from_numpy(v) for k, v in sd.items()}
        else:
            loaded = torch.load(path, map_location="cpu")
        own = self.__model__.module.state_dict()
        keep = {}
        for name, param in loaded.items():
            orig = name
            if name not in own:
                name = name.replace("module.", "")
                if name not in own:
                    if show_error and name.startswith("__S__."):
                        print("{} is not in the model.".format(orig))
                    continue
            if tuple(own[name].shape) != tuple(param.shape):
                if show_error:
                    print("Wrong parameter length: {}, model: {}, loaded: {}".format(orig, tuple(own[name].shape), tuple(param.shape)))
                continue
            keep[name] = param
        self.__model__.module.load_state_dict(keep, strict=False)
