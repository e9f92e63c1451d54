"""Synthetic benchmark inputs: the splitmix64 stream shared with oracle_synth and sx_synth_dev.

symbol i = 1 + (splitmix64(seed + (i+1)*golden) >> 33) % (sigma - 1)
(fixed seeds, no libc rand()).
"""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def synth(n, sigma, seed, start=0):
    """uint8 array of n symbols in [1, sigma), positions start .. start+n of the stream."""
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        return (np.uint64(1) + (z >> np.uint64(33)) % np.uint64(sigma - 1)).astype(np.uint8)


def repeat_families(n, seed, families):
    """uint8 array of n symbols in [1, 5): uniform noise with, for every (copies, length, divergence) of `families`,
    that many copies of one random element at non-overlapping places, each symbol of a copy replaced by a random one
    with the given probability.  The suffixes at one offset of an element's copies share long prefixes: after the
    prefix-key sort they sit in groups of up to `copies` members (what a family of interspersed repeats does to a genome)."""
    rng = np.random.default_rng(seed)
    x = rng.integers(1, 5, size=n, dtype=np.uint8)
    for copies, length, divergence in families:
        element = rng.integers(1, 5, size=length, dtype=np.uint8)
        for pos in rng.choice(n // length - 1, size=copies, replace=False) * length:
            copy = element.copy()
            changed = rng.random(length) < divergence
            copy[changed] = rng.integers(1, 5, size=int(changed.sum()), dtype=np.uint8)
            x[pos:pos + length] = copy
    return x
