"""The other BASELINE.json configurations, measured after bench.py's timed region so that the driver's line carries them."""
from .pins import reference_pin
from .pmc import pmc_traffic

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md, chip-level parameters)

LMS_PATHS = {0: "none", 1: "prefix-key LMS sort + induced-sort passes", 2: "general path (pieces, names, reduced string) + "
             "induced-sort passes", 3: "direct prefix sort of all suffixes"}


def measure_config(ctx, dev, gen, n, sigma_arg, seed, steps, tables=True, no_direct=False, cuda=True, cpu_at_size=None):
    """`steps` timed steps of the hot path on one more text (generated on the device), one warm-up that doubles as the
    per-class profile, results verified on the device afterwards.  Outside bench.py's timed region."""
    import torch
    from stralg_amd import farm, verify, workloads
    text, sigma = workloads.make_text(ctx, gen, n, sigma_arg, seed, dev)
    if cuda:
        torch.cuda.synchronize()
    N = n + 1
    tables = tables and sigma <= 128
    sa = torch.empty(N, dtype=torch.int32, device=dev)
    c_tab = torch.zeros(sigma, dtype=torch.int32, device=dev) if tables else None
    o_tab = torch.empty((N + 1) * sigma, dtype=torch.int32, device=dev) if tables else None
    bwt = torch.empty(N, dtype=torch.uint8, device=dev) if tables else None

    def step():
        if tables:
            ctx.sa_bwt_build_dev(text, n, sigma, sa, bwt)
            ctx.bwt_tables_from_bwt_dev(bwt, N, sigma, c_tab, o_tab)
        else:
            ctx.sa_build_dev(text, n, sigma, sa)

    ctx.set_no_direct_sort(no_direct)
    try:
        ctx.profile_reset()
        ctx.profile_only(None)
        ctx.profile_enable(True)
        step()
        if cuda:
            torch.cuda.synchronize()
        ctx.profile_enable(False)
        table = ctx.profile_read()
        elapsed = farm.timed(step, steps, 0, cuda=cuda)
        stats = ctx.last_stats()
    finally:
        ctx.set_no_direct_sort(False)
    dom = max(table, key=lambda k: table[k]["ms"])
    d = table[dom]
    alg_total = sum(v["alg_bytes"] for v in table.values())
    ms = elapsed / steps * 1e3
    out = {"n": n, "alphabet_size": sigma, "tables": tables, "steps": steps, "ms_per_step": round(ms, 3),
           "Msuffixes_per_s": round(N / (ms * 1e-3) / 1e6, 1),
           "lms_path": stats.get("lms_path"), "algorithm": LMS_PATHS.get(stats.get("lms_path"), "?"),
           "induce_rounds": stats.get("induce_rounds"), "recursion_levels": stats.get("recursion_levels"),
           "dominant_class": dom, "dominant_ms_per_step": round(d["ms"], 3),
           "roofline_frac": round(d["alg_bytes"] / (d["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if d["ms"] > 0 else 0.0,
           "whole_step_frac_of_peak": round(alg_total / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    if n == 1 << (n.bit_length() - 1):
        # the dominant class's HBM bytes per launch from the committed rocprofv3 PMC passes of `bench.py --workload ...`
        tr, stale = pmc_traffic(gen + ("_induced" if no_direct else ""), n.bit_length() - 1, sigma, tables, dom)
        out["traffic"], out["traffic_stale"] = tr, stale
    ctx.trim()
    try:
        verify.verify_build_on_device(text, n, sigma, sa, bwt, c_tab if tables else None, o_tab)
        out["verified"] = True
    except AssertionError as e:
        out["verified"] = False
        out["error"] = str(e)
    if gen in ("dna", "bytes") and cuda:
        pin = reference_pin(sa, n, sigma, seed)
        if pin is not None:
            out["reference_pin"] = pin
    if cpu_at_size is not None:  # (bench.py's CPU-baseline leg: the one place that runs the reference / the oracle)
        out["cpu_reference_whole_record"] = cpu_at_size(n.bit_length() - 1, sigma, seed, sa)
    del text, sa, bwt, c_tab, o_tab
    if cuda:
        torch.cuda.empty_cache()
    return out


def other_configs(ctx, dev, steps, cuda=True, log2n=30, cpu_whole_record=None):
    """BASELINE.json configs[1] and [3] and one hard text, so that the driver's line carries them too:
    256 MiB DNA; 1 GiB of random bytes by the default path (direct prefix sort) and through the LMS sort + induced-sort
    passes (the "wide-alphabet LDS-histogram path" configs[3] names); a genome-like 1 GiB text; a Fibonacci string
    (every LMS substring repeats: the general path, its reduced strings sorted by the pipeline itself level below level)."""
    out = {}

    def size(n):
        return f"{n >> 30}GiB" if n >= 1 << 30 else (f"{n >> 20}MiB" if n >= 1 << 20 else f"{n}B")

    big, quarter = 1 << log2n, 1 << (log2n - 2)
    for name, gen, n, sig, tables, no_direct in (
            (f"dna_{size(quarter)}", "dna", quarter, 5, True, False),
            (f"bytes_{size(big)}", "bytes", big, 256, False, False),
            (f"bytes_{size(big)}_induced", "bytes", big, 256, False, True),
            (f"genome_like_{size(big)}", "genome_like", big, 5, True, False),
            (f"fibonacci_{size(big)}", "periodic", big, 3, True, False)):
        try:
            out[name] = measure_config(ctx, dev, gen, n, sig, 42, steps, tables, no_direct, cuda,
                                       cpu_at_size=cpu_whole_record if gen == "dna" and n == 1 << 28 else None)
        except Exception as e:  # noqa: BLE001 -- an extra must not take the headline line down with it
            out[name] = {"error": f"{type(e).__name__}: {e}"}
    return out
