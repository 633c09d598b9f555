"""CPU stand-ins used ONLY by the host-logic tests (-m "not gpu"): they answer the Engine's scoring /
embedding calls with the oracle so that list parsing, batching, sharding and return shapes of the
ModelHandling counterpart can be checked without a GPU.  Never imported by the product."""
import numpy as np
import torch

from oracle import scoring as o_scoring


class FakeScoringEngine:
    def l2norm_(self, E):
        n = np.maximum(np.linalg.norm(E, axis=1, keepdims=True), 1e-12)
        E /= n
        return E

    def score_pairs(self, E, ia, ib, out=None):
        return o_scoring.cosine_pairs(np.asarray(E), np.asarray(ia), np.asarray(ib)).astype(np.float32)

    def asnorm_stats(self, E, cohort, top=200):
        K = cohort.shape[0]
        top = K + top if top < 0 else min(top, K)
        mu, sd = o_scoring.asnorm_stats(np.asarray(E), np.asarray(cohort), top)
        return mu.astype(np.float32), sd.astype(np.float32)

    def score_trials(self, F, ia, ib, mode="cosine", out=None):
        """svhip_score_trials by the reference's own per-trial statements (oracle / torch)"""
        import torch.nn.functional as TF
        F = np.asarray(F, np.float32)
        t = torch.from_numpy(F)
        res = np.empty(len(ia), np.float32)
        for p, (a, b) in enumerate(zip(np.asarray(ia), np.asarray(ib))):
            if mode == "cosine":
                res[p] = o_scoring.cosine_similarity(t[a], t[b])
            elif mode == "pnorm":
                res[p] = o_scoring.pnorm_similarity(t[a], t[b])
            else:           # pdist: model.py:425-431
                res[p] = -float(torch.mean(TF.pairwise_distance(t[a].unsqueeze(-1), t[b].unsqueeze(-1).transpose(0, 2))))
        return res

    def mean_crops(self, F, out=None):
        return np.asarray(F, np.float32).mean(axis=1)

    def asnorm_pairs(self, E, mu, sd, ia, ib, out=None):
        E = np.asarray(E, np.float64)
        s = np.sum(E[ia] * E[ib], axis=1)
        return (0.5 * ((s - mu[ia]) / sd[ia] + (s - mu[ib]) / sd[ib])).astype(np.float32)


def fake_embedder(dim=16):
    """deterministic 'embedding' of a crop: fixed random projection of simple waveform statistics"""
    rng = np.random.Generator(np.random.PCG64(123))
    P = rng.standard_normal((8, dim)).astype(np.float32)

    def embed(crops):
        c = np.asarray(crops, np.float32)
        f = np.stack([c.mean(1), c.std(1), np.abs(c).max(1), c[:, 0], c[:, -1], c[:, ::7].mean(1),
                      (c[:, 1:] * c[:, :-1]).mean(1), np.ones(len(c), np.float32)], 1)
        return f @ P
    return embed
