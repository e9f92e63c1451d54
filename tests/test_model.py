"""The data-parallel algorithm (tests/model.py, numpy) against the oracle: proves
that LMS pieces with cut points + prefix doubling + bucket-at-a-time induction
yield the reference's suffix array.  CPU only."""
import itertools

import numpy as np

import model
import oracle


def _check(x, sigma, W=None):
    assert (model.suffix_array(x, sigma, W) == oracle.sa_is_strict(x, sigma)).all()


def test_exhaustive_small():
    for n in range(0, 7):
        for tup in itertools.product((1, 2, 3), repeat=n):
            for W in (1, 2, None):
                _check(np.array(tup, dtype=np.uint8), 4, W)


def test_random():
    rng = np.random.default_rng(7)
    for sigma in (2, 3, 5, 21, 256):
        for n in (1, 10, 1000, 20000):
            for W in (1, 3, None):
                _check(rng.integers(1, sigma, size=n, dtype=np.uint8), sigma, W)


def test_golden(golden):
    for name, c in golden.items():
        if c["sym"].size <= 20000 and not (c["sigma"] == c["sym"].size + 1):
            assert (model.suffix_array(c["sym"], c["sigma"]) == c["sa"]).all(), name
