"""The checkers of stralg_amd/benchlegs (what bench.py's end_to_end leg and the full-size GPU tests use to pin the host-pointer
drop-in path to the reference) do catch what they claim to: run here on the CPU execution harness of the kernels, against a
fixture made from the oracle at a small size."""
import hashlib

import numpy as np

import oracle
from stralg_amd.synth import synth


def _small_fixture(n, sigma, seed):
    x = synth(n, sigma, seed)
    sa = oracle.sa_is_strict(x, sigma)
    chunk = 1 << 26
    counts = np.bincount(np.concatenate([x, [0]]), minlength=sigma).astype(np.uint64)
    z = {"sa_sha256": np.frombuffer(hashlib.sha256(sa.tobytes()).digest(), np.uint8),
         "sa_chunk_sha256": np.stack([np.frombuffer(hashlib.sha256(sa[s:s + chunk].tobytes()).digest(), np.uint8)
                                      for s in range(0, sa.size, chunk)]),
         "sa_sampled": np.concatenate([sa[:: 1 << 20], sa[-1:]]), "counts": counts}

    class Z(dict):
        files = property(lambda self: list(self))

    key = f"n{n.bit_length() - 1}/s{sigma}"
    return x, sa, Z({f"{key}/{k}": v for k, v in z.items()}), key


def test_host_table_pin_catches_a_wrong_entry(emu_ctx, monkeypatch):
    from stralg_amd.benchlegs import cabi, pins
    n, sigma = 1 << 14, 5
    x, sa, z, key = _small_fixture(n, sigma, 42)
    monkeypatch.setattr(pins, "fixture", lambda n_, s_, seed=42: (z, key))
    lib = cabi.declare(emu_ctx.lib)
    letters = np.concatenate([np.frombuffer(b"\0ACGT", np.uint8)[x], [0]]).astype(np.uint8)
    t = lib.build_complete_table(letters.ctypes.data, True)
    try:
        pin = pins.host_table_pin(t, n, rows_sampled=300)
        assert pin["match"] and pin["sa"]["sha256_match"] and pin["o_rows_sampled_one_hot_at_bwt"], pin
        N = n + 1
        arr = np.ctypeslib.as_array(t.contents.sa.contents.array, shape=(N,))
        arr[[77, 78]] = arr[[78, 77]]                       # a swapped pair: still a permutation, no longer THE array
        bad = pins.host_table_pin(t, n, rows_sampled=300)
        assert not bad["match"] and bad["sa"]["chunks_differing"] == [0] and not bad["sa"]["sha256_match"]
        arr[[77, 78]] = arr[[78, 77]]
        o = np.ctypeslib.as_array(t.contents.o_table, shape=((N + 1) * sigma,))
        o[N * sigma + 2] += 1                                # the row behind the last position (never compared by bwt.c:579)
        bad = pins.host_table_pin(t, n, rows_sampled=300)
        assert not bad["match"] and not bad["o_last_row_match"]
        o[N * sigma + 2] -= 1
        c = np.ctypeslib.as_array(t.contents.c_table, shape=(sigma,))
        c[3] += 1
        assert not pins.host_table_pin(t, n, rows_sampled=300)["c_table_match"]
        c[3] -= 1
        assert pins.host_table_pin(t, n, rows_sampled=300)["match"]
    finally:
        lib.completely_free_bwt_table(t)
    symbols = np.concatenate([x, [0]]).astype(np.uint8)
    a = lib.sa_is_construction(symbols.ctypes.data, sigma)
    got = np.ctypeslib.as_array(a.contents.array, shape=(n + 1,))
    assert pins.host_sa_pin(got, n, sigma)["match"]
    lib.free_suffix_array(a)
