"""Seeded logits vectors for the sampler parity tests (shared by tools/make_golden.py, the tests and bench probes).

Only the generator parameters are stored in the golden fixture; the vectors (Qwen3's 151 936-entry vocabulary) are
regenerated from the seed.  Shapes of distribution: near-uniform (random-init models), peaked (trained models), one
dominant token, numerators that underflow to denormals / zero, many exactly equal probabilities (sort-order ties).
"""
import numpy as np

V_QWEN3 = 151936
COINS = (0.0, 0.37, 0.93, 0.99999994)

# (seed, sigma, mode, repetition_penalty, temperature, top_p, n_history)
CASES = [
    (1, 3.0, "plain", 1.0, 1.0, 0.9, 0),
    (2, 3.0, "plain", 1.1, 0.8, 0.9, 40),
    (3, 10.0, "plain", 1.3, 1.0, 0.95, 300),
    (4, 10.0, "holes", 1.0, 0.7, 0.9, 0),
    (5, 2.0, "peak", 1.2, 1.0, 0.5, 17),
    (6, 40.0, "plain", 1.0, 1.5, 0.9, 0),
    (7, 4.0, "tail_peak", 1.0, 1.0, 0.8, 0),
    (8, 3.0, "ties", 1.1, 1.0, 0.9, 64),
    (9, 0.6, "plain", 1.0, 1.0, 0.9, 0),          # near-uniform: every token passes the cutoff -> host fall-back
    (10, 3.0, "plain", 1.3, 0.0, 0.9, 200),       # temperature 0: penalised arg-max
    (11, 6.0, "ties", 1.0, 2.0, 0.99, 0),
    (12, 5.0, "plain", 0.8, 0.9, 0.3, 120),       # penalty < 1 (the reference divides regardless of sign)
]


def logits_of(seed, sigma, mode, V=V_QWEN3):
    rng = np.random.default_rng(seed)
    l = (sigma * rng.standard_normal(V)).astype(np.float32)
    if mode == "peak":
        l[int(rng.integers(V))] += np.float32(30.0)
    elif mode == "holes":
        l[rng.random(V) < 1.0 / 7.0] -= np.float32(95.0)
    elif mode == "tail_peak":
        l[V - 1] += np.float32(25.0)
        l[: V // 2] -= np.float32(101.0)
    elif mode == "ties":
        l = (np.round(l * 4.0) / 4.0).astype(np.float32)
    return np.ascontiguousarray(l, np.float32)


def history_of(seed, n, V=V_QWEN3):
    return np.random.default_rng(1000 + seed).integers(0, V, size=n).astype(np.uint32)
