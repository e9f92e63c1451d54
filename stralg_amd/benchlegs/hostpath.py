"""Host-buffer (PCIe-inclusive) legs, outside bench.py's timed region: what a caller of the reference API sees
(end_to_end; round 5: every entry is CHECKED against the reference's hashes, pins.host_table_pin) and the FASTA records'
egress (egress_leg)."""
import time

from . import cabi, pins

def end_to_end(ctx, sizes, seed):
    """What a caller of the reference API sees (stralg/bwt.c:134-161 hands over malloc'd host arrays):
    sx_build_tables on pageable host buffers, and build_complete_table itself (remap, tables, o_indices) with and
    without the reverse table.  Second call of each (the first pays hipMalloc of the staging slab)."""
    import ctypes as C
    import numpy as np
    import psutil
    from stralg_amd.synth import synth
    lib = cabi.declare(ctx.lib)
    out = {}
    for log2n in sizes:
        n = 1 << log2n
        N = n + 1
        need = (N + 1) * 5 * 4 * 2 + N * 4 + (N + 1) * 8 * 2 + 3 * n  # O + RO + SA + row pointers + strings
        if psutil.virtual_memory().available < need * 1.25:
            out[f"2^{log2n}"] = {"skipped": f"needs {need >> 30} GiB of free host memory"}
            continue
        x = synth(n, 5, seed)
        sa = np.empty(N, dtype=np.uint32)
        c = np.zeros(5, dtype=np.uint32)
        o = np.empty((N + 1) * 5, dtype=np.uint32)
        moved = n + 4 * N + 4 * 5 * (N + 1)
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            ctx._check(lib.sx_build_tables(ctx.h, x.ctypes.data, n, 5, sa.ctypes.data, c.ctypes.data, o.ctypes.data), "sx_build_tables")
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        # round 5: the arrays this call filled are looked at -- the suffix array against the reference's hashes, C and the row
        # behind the last position against the reference's counts (tests/golden/golden_big.npz)
        z, zkey = pins.fixture(n, 5, seed)
        pin_sx = None
        if z is not None:
            counts = z[zkey + "/counts"].astype(np.int64)
            sa_pin = pins.host_sa_pin(sa, n, 5, seed, whole=False)
            pin_sx = {"sa": sa_pin,
                      "c_table_match": bool((c.astype(np.int64) == np.concatenate([[0], np.cumsum(counts)[:-1]])).all()),
                      "o_last_row_match": bool((o[N * 5:(N + 1) * 5].astype(np.int64) == counts).all())}
            pin_sx["match"] = bool(sa_pin["match"] and pin_sx["c_table_match"] and pin_sx["o_last_row_match"])
        del sa, o
        letters = np.frombuffer(b"\0ACGT", dtype=np.uint8)[x].tobytes()  # NUL-terminated by bytes' own terminator
        del x
        res = {"sx_build_tables_ms": round(best * 1e3, 1), "pcie_GBps": round(moved / best / 1e9, 2),
               "sx_build_tables_Msuffixes_per_s": round(N / best / 1e6, 1)}
        if pin_sx is not None:
            res["sx_build_tables_reference_pin"] = pin_sx
        for key, rev in (("build_complete_table_ms", False), ("with_ro_ms", True)):
            best = None
            for _ in range(2):  # (the first call of a size pays the host's first touch of 30 GiB of result arrays)
                t0 = time.perf_counter()
                t = lib.build_complete_table(letters, rev)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
                if _ == 0 or rev:
                    # what the caller holds now (sa->array, c_table, the O / RO rows through o_indices / ro_indices)
                    # against the reference's hashes and counts: pins.host_table_pin
                    pin = pins.host_table_pin(t, n, seed, whole=False)
                    if pin is not None:
                        res[key.replace("_ms", "") + "_reference_pin"] = pin
                lib.completely_free_bwt_table(t)
                if rev:
                    break
            res[key] = round(best * 1e3, 1)
        res["build_complete_table_Msuffixes_per_s"] = round(N / (res["build_complete_table_ms"] * 1e-3) / 1e6, 1)
        res["bytes_over_pcie"] = moved
        pinned = [v for k, v in res.items() if k.endswith("reference_pin")]
        if pinned:
            res["reference_pin"] = {"match": all(p["match"] for p in pinned), "checked": len(pinned),
                                    "what": "host arrays of sx_build_tables, build_complete_table and build_complete_table(.., true): "
                                            "chunk SHA-256s + sampled entries of sa->array, C, O(a, N) via o_indices, sampled O rows "
                                            "one-hot at the BWT, RO rows, vs the reference's sa_is_mem_construction output"}
            res["verified"] = res["reference_pin"]["match"]
        if log2n == max(sizes):
            # the production caller's loop (tools/readmappers/bwt_readmapper/bwt_readmapper.c:54-62): build_complete_table(seq,
            # true) -> write_complete_bwt_info -> completely_free_bwt_table, record after record on one thread.  The freed
            # arrays go to the thread's block cache (stralg_host.c) and come back for the next record: no unmapping of
            # 52 GiB of huge pages (1.9 s a record in round 3), no first touch of as many by the next build.
            libc = C.CDLL(None)
            libc.fopen.restype = C.c_void_p
            libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
            libc.fclose.argtypes = [C.c_void_p]
            f = libc.fopen(b"/dev/null", b"wb")
            per, frees = [], []
            for _ in range(4):
                t0 = time.perf_counter()
                t = lib.build_complete_table(letters, True)
                lib.write_complete_bwt_info(f, t)
                t1 = time.perf_counter()
                lib.completely_free_bwt_table(t)
                t2 = time.perf_counter()
                per.append(round((t2 - t0) * 1e3, 1))
                frees.append(round((t2 - t1) * 1e3, 1))
            libc.fclose(f)
            lib.stralg_amd_release()  # (the calling thread's context and its cached host blocks)
            res["readmapper_loop"] = {"records": 4, "ms_per_record": per, "free_ms_per_record": frees,
                                      "steady_ms_per_record": round(sum(per[1:]) / 3, 1),
                                      "what": "build_complete_table(seq, true) + write_complete_bwt_info(/dev/null) + "
                                              "completely_free_bwt_table, four records in a row on one thread"}
        out[f"2^{log2n}"] = res
    out["note"] = ("pageable malloc'd host buffers as the reference's ownership rules require; build_complete_table "
                   "includes the host remap and the o_indices row-pointer table; with_ro_ms adds the reverse table")
    return out


def egress_leg(ctx, local_rank, text, n, sigma, world, red_dev, cuda=True, with_ro=False):
    """Every rank at once, between barriers: the record's whole index (suffix array, C, O: 24 bytes per base) leaves
    the GPU -- (1) stralg_amd_write_complete_bwt_info_stream into /dev/null (stralg/serialise.c:7-18's file, streamed
    through two pinned buffers: no host copy of the tables), (2) memory permitting, sx_build_tables into malloc'd host
    arrays as build_complete_table's caller owns them (stralg/bwt.c:134-161).  Both start from the record on the host
    (H2D included)."""
    import ctypes as C
    import numpy as np
    import psutil
    import torch
    from stralg_amd import farm
    lib = ctx.lib
    N = n + 1
    index_bytes = 4 * N + 4 * sigma + 4 * sigma * (N + 1)
    # symbols 1 .. 5 -> A C G T N (bytes.translate: no index array of eight bytes a base beside the record)
    letters = text.cpu().numpy().tobytes().translate(bytes([0]) + b"ACGTN" + b"N" * 250)  # (bytes' own terminator ends the string)
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    lib.stralg_amd_write_complete_bwt_info_stream.argtypes = [C.c_void_p, C.c_char_p, C.c_bool]
    lib.stralg_amd_write_complete_bwt_info_stream.restype = C.c_int
    lib.stralg_amd_set_device.argtypes = [C.c_int]
    lib.stralg_amd_set_device(local_rank)
    out = {"index_bytes_per_record": index_bytes}
    f = libc.fopen(b"/dev/null", b"wb")

    def stream():
        if lib.stralg_amd_write_complete_bwt_info_stream(f, letters, False) != 0:
            raise RuntimeError("stralg_amd_write_complete_bwt_info_stream failed")

    stream()  # (the first call pays the thread context's hipMalloc and the pinned buffers)
    t_own = farm.timed(stream, 1, 0, cuda=cuda)
    t_max, units = farm.reduce_scalars(t_own, N, device=red_dev)
    out.update(stream_ms_per_record=round(t_max * 1e3, 1),
               egress_inclusive_Msuffixes_per_s=round(units / t_max / 1e6, 3),
               d2h_GBps_per_rank=round(index_bytes / t_own / 1e9, 2),
               d2h_GBps_all_ranks=round(index_bytes * world / t_max / 1e9, 2))
    if with_ro:
        # the index file bwt_readmapper -p writes: write_complete_bwt_info of build_complete_table(seq, TRUE) -- the RO
        # table streams out behind the O table (44 bytes per base over PCIe instead of 24)
        def stream_ro():
            if lib.stralg_amd_write_complete_bwt_info_stream(f, letters, True) != 0:
                raise RuntimeError("stralg_amd_write_complete_bwt_info_stream (include_reverse) failed")

        t_own = farm.timed(stream_ro, 1, 0, cuda=cuda)
        t_max, units = farm.reduce_scalars(t_own, N, device=red_dev)
        ro_bytes = index_bytes + 4 * sigma * (N + 1)
        out["egress_with_ro"] = {"index_bytes_per_record": ro_bytes, "stream_ms_per_record": round(t_max * 1e3, 1),
                          "egress_inclusive_Msuffixes_per_s": round(units / t_max / 1e6, 3),
                          "d2h_GBps_per_rank": round(ro_bytes / t_own / 1e9, 2)}
    libc.fclose(f)
    lib.stralg_amd_release()
    # (2) into malloc'd host arrays, when every rank's 24 bytes per base fit the host
    need = (index_bytes + n) * world
    fits, _ = farm.reduce_scalars(0.0 if psutil.virtual_memory().available > need * 1.3 else 1.0, 0, device=red_dev)
    if fits == 0.0:
        x = text.cpu().numpy()
        sa = np.empty(N, dtype=np.uint32)
        c = np.zeros(sigma, dtype=np.uint32)
        o = np.empty((N + 1) * sigma, dtype=np.uint32)

        def host_tables():
            ctx._check(lib.sx_build_tables(ctx.h, x.ctypes.data, n, sigma, sa.ctypes.data, c.ctypes.data, o.ctypes.data),
                       "sx_build_tables")

        host_tables()  # (first touch of the result arrays, the staging slab)
        t_own = farm.timed(host_tables, 1, 0, cuda=cuda)
        t_max, units = farm.reduce_scalars(t_own, N, device=red_dev)
        out.update(host_tables_ms_per_record=round(t_max * 1e3, 1),
                   host_tables_Msuffixes_per_s=round(units / t_max / 1e6, 3),
                   host_tables_GBps_per_rank=round((index_bytes + n) / t_own / 1e9, 2))
    else:
        out["host_tables"] = f"skipped: {need >> 30} GiB of host memory needed for {world} ranks"
    out["egress_note"] = ("all ranks at once between barriers, starting from the record on the host: remap, H2D, build, then "
                   "SA + C + O (24 B per base) over PCIe; stream = the reference's index file into /dev/null")
    return out
