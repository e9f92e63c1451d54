"""numpy model of the device algorithm (NOT the oracle, NOT a product path).

The HIP pipeline in stralg_amd/csrc does not run the reference's sequential
induced sort; it computes the same (unique) suffix array with data-parallel
steps.  This module restates those steps with numpy so that the *algorithm*
can be checked against the oracle on a machine without a GPU:

  A  S/L types, LMS positions              (stralg/sa_is.c:134-162)
  B  sample set = LMS positions + cut points every W symbols inside long
     LMS substrings; pieces of <= W+1 symbols packed into 64-bit keys
  C  radix sort of the pieces, dense names  (role of sa_is.c:295-336)
  D  suffix sort of the reduced string by prefix doubling
     (role of the recursion, sa_is.c:370-387)
  E  sorted LMS suffixes                    (role of sa_is.c:443-464)
  F  induce L / induce S, bucket at a time, each bucket in rounds
     (sa_is.c:220-263)

tests/test_model.py compares it with the oracle; the GPU tests compare the
HIP pipeline with the oracle directly.
"""
import numpy as np


def bitlen(v):
    return int(v).bit_length()


def key_layout(maxc):
    """(bits per symbol, slots per key, bits of the length field)."""
    bits = max(1, bitlen(maxc))
    slots = 2
    while (slots + 1) * bits + bitlen(slots + 1) + 1 <= 64:
        slots += 1
    assert slots * bits + bitlen(slots) + 1 <= 64, "alphabet too wide for a 64-bit key"
    return bits, slots, bitlen(slots)


def types_and_lms(T):
    """T includes the sentinel at T[n].  Returns (is_s[n+1], lms_flag[n+1])."""
    n = T.size - 1
    is_s = np.zeros(n + 1, dtype=bool)
    is_s[n] = True
    if n > 0:
        diff = T[:-1] != T[1:]
        idx = np.where(diff, np.arange(n), n + 1)
        e = np.minimum.accumulate(idx[::-1])[::-1]  # end of the run containing i
        is_s[:n] = T[e] < T[e + 1]
    lms = np.zeros(n + 1, dtype=bool)
    lms[1:] = is_s[1:] & ~is_s[:-1]
    return is_s, lms


def samples(lms, W):
    """Positions of LMS suffixes plus cut points; returns (pos, is_lms_sample)."""
    N = lms.size
    idx = np.where(lms, np.arange(N), -1)
    prev = np.maximum.accumulate(idx)
    d = np.arange(N) - prev
    cut = (prev >= 0) & ~lms & (d % W == 0)
    flag = lms | cut
    pos = np.nonzero(flag)[0].astype(np.int64)
    return pos, lms[pos]


def piece_keys(T, pos, is_lms_sample, bits, slots, lenbits):
    M = pos.size
    keys = np.zeros(M, dtype=np.uint64)
    if M <= 1:
        return keys
    start = pos[:-1]
    length = pos[1:] - pos[:-1] + 1
    assert length.min() >= 2 and length.max() <= slots
    acc = np.zeros(M - 1, dtype=np.uint64)
    ones = np.uint64((1 << bits) - 1)
    for t in range(slots):
        inside = t < length
        p = np.where(inside, start + t, 0)
        code = np.where(inside, T[p].astype(np.uint64), ones)
        acc = (acc << np.uint64(bits)) | code
    acc = (acc << np.uint64(lenbits)) | (slots - length).astype(np.uint64)
    acc = (acc << np.uint64(1)) | is_lms_sample[1:].astype(np.uint64)
    used = slots * bits + lenbits + 1
    keys[:-1] = acc << np.uint64(64 - used)
    keys[-1] = 0  # the sentinel piece
    assert (keys[:-1] > 0).all()
    return keys


def dense_names(keys):
    order = np.argsort(keys, kind="stable")
    ks = keys[order]
    flag = np.ones(ks.size, dtype=np.int64)
    flag[1:] = ks[1:] != ks[:-1]
    name_sorted = np.cumsum(flag) - 1
    R = np.empty(keys.size, dtype=np.int64)
    R[order] = name_sorted
    return R, int(name_sorted[-1]) + 1, order


def prefix_doubling(R, n_names):
    """Suffix array of R (R[-1] == 0 is the unique smallest symbol)."""
    M = R.size
    b = max(1, bitlen(n_names - 1))
    q = max(1, 64 // b)
    key = np.zeros(M, dtype=np.uint64)
    for t in range(q):
        sym = np.zeros(M, dtype=np.uint64)
        if t < M:
            sym[: M - t] = R[t:].astype(np.uint64)
        key = (key << np.uint64(b)) | sym
    SA = np.argsort(key, kind="stable").astype(np.int64)
    ks = key[SA]
    head = np.ones(M, dtype=bool)
    head[1:] = ks[1:] != ks[:-1]
    rounds = 0
    h = q
    rank = np.empty(M, dtype=np.int64)
    while True:
        gid = np.maximum.accumulate(np.where(head, np.arange(M), 0))  # head position of each slot
        rank[SA] = gid
        nxt_head = np.ones(M, dtype=bool)
        nxt_head[:-1] = head[1:]
        single = head & nxt_head
        act = np.nonzero(~single)[0]
        if act.size == 0:
            break
        rounds += 1
        idx = SA[act]
        assert (idx + h < M).all()
        k2 = (gid[act].astype(np.uint64) << np.uint64(32)) | rank[idx + h].astype(np.uint64)
        o = np.argsort(k2, kind="stable")
        SA[act] = idx[o]
        k2s = k2[o]
        newh = np.ones(act.size, dtype=bool)
        newh[1:] = k2s[1:] != k2s[:-1]
        head[act] = newh
        h *= 2
    return SA, rounds


def induce(T, sigma, sorted_lms, stats=None):
    """Final induced sort from the sorted LMS suffixes, bucket at a time."""
    N = T.size
    n = N - 1
    is_s, lms = types_and_lms(T)
    sizes = np.bincount(T, minlength=sigma).astype(np.int64)
    begin = np.concatenate(([0], np.cumsum(sizes)[:-1]))
    end = np.cumsum(sizes)
    n_l = np.bincount(T[~is_s], minlength=sigma).astype(np.int64)
    lms_first = T[sorted_lms]
    lms_cnt = np.bincount(lms_first, minlength=sigma).astype(np.int64)
    lms_off = np.concatenate(([0], np.cumsum(lms_cnt)))
    SA = np.full(N, -1, dtype=np.int64)
    SA[0] = n
    launches = 0

    def split(src, accept, head, direction):
        nonlocal launches
        launches += 1
        src = src[src > 0]
        j = src - 1
        c = T[j]
        ok = accept(c)
        j, c = j[ok], c[ok]
        added = {}
        for k in np.unique(c):
            sel = j[c == k]
            if direction > 0:
                SA[head[k]: head[k] + sel.size] = sel
                added[int(k)] = (head[k], head[k] + sel.size)
                head[k] += sel.size
            else:
                SA[head[k] - sel.size: head[k]] = sel[::-1]
                added[int(k)] = (head[k] - sel.size, head[k])
                head[k] -= sel.size
        return added

    # ---- L pass (sa_is.c:220-242), buckets ascending
    head = begin.copy()
    for c in range(sigma):
        if sizes[c] == 0:
            continue
        lo, hi = begin[c], head[c]
        while hi > lo:
            added = split(SA[lo:hi], lambda x, c=c: x >= c, head, +1)
            lo, hi = added.get(c, (hi, hi))
        seeds = sorted_lms[lms_off[c]: lms_off[c + 1]]
        if seeds.size:
            split(seeds, lambda x: x >= 0, head, +1)
    assert ((head - begin) == n_l).all()

    # ---- S pass (sa_is.c:245-263), buckets descending, each right to left
    tail = end.copy()
    for c in range(sigma - 1, -1, -1):
        if sizes[c] == 0:
            continue
        lo, hi = tail[c], end[c]
        while hi > lo:
            added = split(SA[lo:hi][::-1], lambda x, c=c: x <= c, tail, -1)
            lo, hi = added.get(c, (lo, lo))
        lreg = SA[begin[c]: begin[c] + n_l[c]]
        if lreg.size:
            split(lreg[::-1], lambda x, c=c: x < c, tail, -1)
    if stats is not None:
        stats["induce_launches"] = launches
    SA[0] = n
    assert (SA >= 0).all()
    return SA.astype(np.uint32)


def suffix_array(text, sigma, W=None, stats=None):
    """Suffix array of text + sentinel by the device algorithm's steps."""
    text = np.asarray(text, dtype=np.uint8)
    n = text.size
    T = np.concatenate((text, np.zeros(1, dtype=np.uint8)))
    if n == 0:
        return np.zeros(1, dtype=np.uint32)
    is_s, lms = types_and_lms(T)
    bits, slots, lenbits = key_layout(int(T.max()))
    if W is None:
        W = slots - 1
    else:
        assert 1 <= W <= slots - 1
    pos, is_lms_sample = samples(lms, W)
    M = pos.size
    keys = piece_keys(T, pos, is_lms_sample, bits, slots, lenbits)
    R, n_names, order = dense_names(keys)
    if n_names == M:
        SA_R, rounds = order, 0
    else:
        SA_R, rounds = prefix_doubling(R, n_names)
    sorted_lms = pos[SA_R[is_lms_sample[SA_R]]]
    if stats is not None:
        stats.update(n=n, m=int(lms.sum()), M=M, n_names=n_names, doubling_rounds=rounds,
                     bits=bits, slots=slots, W=W)
    return induce(T, sigma, sorted_lms, stats)
