"""Seeded trial lists of the metrics fixtures: regenerated on both sides (oracle/make_golden.py with the reference, tests with
the HIP path); never committed."""
import numpy as np


def metrics_case(name):
    """Seeded trial lists shared by the generator and the tests (scores are float32 values, as they come out of the scorer)."""
    seed, P, quant, pos_rate = {"small": (5, 40, 0.25, 0.5), "ties": (6, 3000, 0.01, 0.3), "distinct": (7, 5000, 0.0, 0.5),
                                "skewed": (8, 4000, 0.002, 0.05)}[name]
    rng = np.random.Generator(np.random.PCG64(seed))
    lab = (rng.random(P) < pos_rate).astype(np.int64)
    lab[0], lab[1] = 1, 0
    sc = (rng.standard_normal(P) + 1.5 * lab).astype(np.float32) * np.float32(0.25)
    if quant:
        sc = (np.round(sc / np.float32(quant)) * np.float32(quant)).astype(np.float32)      # many tied scores
    return sc, lab
