#!/usr/bin/env python3
"""bench.py -- Msuffixes/s of suffix-array + BWT C/O-table construction on MI355X.

A "step" is one pass of the hot path over one record already resident in HBM:

  --gpus 1 (default)   BASELINE.json configs[2]: sa_is_construction's device path with the BWT
                       handed over by the induced-sort passes (sx_sa_bwt_build_dev), then the C/O
                       tables (sx_bwt_tables_from_bwt_dev), on 1 GiB of synthetic DNA.
  --gpus N > 1         BASELINE.json configs[4]: every rank builds its own FASTA record, the loop of
                       tools/readmappers/bwt_readmapper/bwt_readmapper.c:54-62 farmed one record per
                       GPU: FASTA image in HBM -> sx_fasta_pack_dev -> sx_remap_dev ->
                       sx_sa_bwt_build_dev -> sx_bwt_tables_from_bwt_dev.  Weak scaling, no collective
                       on the data path; torch.distributed only carries the timing barrier and the
                       max / sum of the bookkeeping scalars.  (--workload fasta gives the same step at N = 1.)

Run as `python bench.py --gpus N` it starts the N ranks itself (fresh child processes, before
anything touches a GPU); under torch.distributed.run it takes RANK / LOCAL_RANK / WORLD_SIZE from the
environment.  Rank 0 prints ONE JSON line.  After the timed region the last step's results are
verified on the device ("verified"), and outside it the line also reports the host-buffer
(PCIe-inclusive) rates ("end_to_end"), the FASTA-ingest-inclusive rate ("fasta_record") and the
reference's own CPU path on a bounded sample ("cpu_baseline").

Layout (round 5): this file holds the argument parser, the rank launcher, the CPU-baseline leg (the one place that runs the
reference / the oracle) and run_rank with the timed region; the other legs live in stralg_amd/benchlegs/ -- ceiling.py (the
box's measured memory ceiling, before the timed region), pins.py (SHA-256 pins to the reference), hostpath.py (host-buffer
rates, each entry checked), configs.py (the other BASELINE configurations), pmc.py (replayed counters).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from stralg_amd.benchlegs.configs import HBM_PEAK_GBS  # 8000 GB/s: MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)  # noqa: E402
from stralg_amd.benchlegs.pmc import PMC_TRAFFIC, pmc_traffic, pmc_whole_step  # noqa: E402


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2n", type=int, default=int(os.environ.get("STRALG_BENCH_LOG2N", "30")))
    ap.add_argument("--n", type=int, default=0, help="symbols per record (overrides --log2n)")
    ap.add_argument("--sigma", type=int, default=5, help="alphabet_size of the uniform workloads")
    ap.add_argument("--workload", default=None,
                    help="dna (default at --gpus 1) | fasta (default at --gpus > 1) | bytes | uniform | genome_like | "
                         "n_runs | text_like | pangenome | periodic")
    ap.add_argument("--no-tables", action="store_true", help="suffix array only")
    ap.add_argument("--no-direct-sort", action="store_true",
                    help="wide alphabets: LMS sort + induced-sort passes even where the direct sort of all suffixes applies")
    ap.add_argument("--sort-mode", type=int, default=0,
                    help="prefix-key sort: 0 choose, 1 LSD passes over all key bits, 2 hybrid (top bits in HBM passes, sub-buckets in LDS)")
    ap.add_argument("--prefix-symbols", type=int, default=0,
                    help="prefix-key sort: symbols of the first attempt's key (0: by the text's size; A/B)")
    ap.add_argument("--chain-max", type=int, default=-1,
                    help="induce rounds of up to this many entries take the single chained launch (default: by alphabet size)")
    ap.add_argument("--no-text-keys", action="store_true",
                    help="the hybrid sort's first pass reads keys a key kernel wrote instead of computing them from the text (A/B)")
    ap.add_argument("--no-induce-batch", action="store_true",
                    help="induced-sort passes: every self round of a bucket as a launch of its own (no eight-rounds-at-a-time form)")
    ap.add_argument("--induce-attended", action="store_true",
                    help="induced-sort passes: the host reads every bucket's last range back before it queues the next bucket (A/B)")
    ap.add_argument("--copy-text-first", action="store_true", help="the build copies the text before the classification (A/B)")
    ap.add_argument("--cpu-log2n", type=int, default=int(os.environ.get("STRALG_BENCH_CPU_LOG2N", "25")))
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--no-cpu-whole-record", action="store_true",
                    help="skip the reference's sa_is_mem_construction on the whole 2^28 record of configs[1] (~1-2 minutes of one core)")
    ap.add_argument("--no-verify", action="store_true", help="skip the device-side check of the last step's results")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-buffer (PCIe-inclusive) measurements")
    ap.add_argument("--e2e-log2n", default="28,30", help="sizes of the host-buffer measurements")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the other BASELINE.json configurations measured after the timed region (default run only)")
    ap.add_argument("--other-steps", type=int, default=3, help="timed steps of each of the other configurations")
    ap.add_argument("--no-ro", action="store_true",
                    help="FASTA records: skip the reverse direction (build_complete_table's include_reverse: the RO table)")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the probe of the box's memory ceiling before the timed region")
    ap.add_argument("--roofline-every", type=int, default=4, metavar="M",
                    help="the dominant kernel's launches carry HIP events in every M-th step of the timed region (1: every step; "
                         "a timed launch costs the queue ~11 us)")
    ap.add_argument("--no-reference-scale", action="store_true",
                    help="skip the row at the reference's published size (n = 49 000 / 65 536 through the host C API)")
    ap.add_argument("--no-egress", action="store_true",
                    help="FASTA records: skip the egress leg (the record's index leaving the GPU on every rank at once)")
    return ap.parse_args(argv)


# ---- launcher: python bench.py --gpus N starts its own ranks -------------------------------------------

def launch_ranks(args):
    """N fresh child processes, one per GPU; this process never touches a GPU (it only waits)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in procs:  # a rank failed: the others would wait at the barrier for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            p.kill()
    return rc


# ---- CPU baseline ----------------------------------------------------------------------------------------

def cpu_baseline(x, sigma, what):
    """the reference's own sa_is_construction (oracle/_ref; or the oracle port) on one host core, on a bounded
    sample of the record rank 0 built (x: numpy uint8): SA, plus C/O tables on a smaller one"""
    import oracle
    from oracle import pyoracle
    n = int(x.size)
    t0 = time.perf_counter()
    sa = oracle.sa_is(x, sigma)
    t1 = time.perf_counter()
    levels = oracle.last_levels()
    # C/O tables on a smaller sample: the O table takes (n+2) * sigma * 4 bytes of host memory
    out = {"unit": "Msuffixes/s", "cores": 1, "host_cpus": os.cpu_count(), "levels": levels[:6]}
    tables = ""
    if sigma <= 128:
        o_n = min(n, 1 << 22)
        xs = x[:o_n]
        sas = sa if o_n == n else oracle.sa_is(xs, sigma)
        t2 = time.perf_counter()
        oracle.c_table(xs, sigma)
        oracle.o_table(xs, sas, sigma)
        t3 = time.perf_counter()
        tables = (f"; C/O tables on {o_n} symbols ({t3 - t2:.2f} s = "
                  f"{(o_n + 1) / (t3 - t2) / 1e6:.1f} Mpositions/s)")
    port = (n + 1) / (t1 - t0) / 1e6
    out.update(value=round(port, 3), kind="port",
               sample=f"oracle_sa_is on {what} ({t1 - t0:.1f} s)" + tables)
    if pyoracle.have_ref():
        # the unmodified reference (oracle/_ref, built from the reference's sources by oracle/Makefile): its own
        # sa_is_construction on the same sample is the baseline proper; the port's rate stays in the text
        ref = pyoracle._Ref()
        t4 = time.perf_counter()
        sa_ref = ref.sa_is(x, sigma)
        t5 = time.perf_counter()
        if not (sa_ref == sa).all():
            raise RuntimeError("reference and oracle disagree on the baseline sample")
        out.update(value=round((n + 1) / (t5 - t4) / 1e6, 3), kind="reference",
                   sample=f"the reference's sa_is_construction on {what} "
                          f"({t5 - t4:.1f} s; the oracle port: {port:.1f} Msuffixes/s)" + tables)
    return out


def cpu_baseline_at_size(log2n, sigma, seed, sa_device=None):
    """The unmodified reference's sa_is_mem_construction (oracle/_ref) on the WHOLE record of BASELINE.json configs[1]
    (2^28 symbols; ~60-100 s on one core), next to the bounded sample above: the reference's rate falls with n
    (BASELINE.md section 2).  Its array is compared entry by entry with the device's (sa_device) when that is given."""
    import numpy as np
    from oracle import pyoracle
    from stralg_amd.synth import synth
    if not pyoracle.have_ref():
        return {"skipped": "oracle/_ref has not been built"}
    n = 1 << log2n
    x = synth(n, sigma, seed)
    ref = pyoracle._Ref()
    t0 = time.perf_counter()
    sa = ref.sa_is_mem(x, sigma)
    dt = time.perf_counter() - t0
    out = {"n": n, "alphabet_size": sigma, "fn": "sa_is_mem_construction (sa_is_mem.c:471-494)", "kind": "reference", "cores": 1,
           "seconds": round(dt, 1), "value": round((n + 1) / dt / 1e6, 3), "unit": "Msuffixes/s"}
    if sa_device is not None:
        out["device_array_identical"] = bool((sa_device.cpu().numpy().view(np.uint32) == sa).all())
    return out


def reference_scale(lib, sizes=(49000, 65536), calls=50, seed=42):
    """The reference's only published operating point (SURVEY.md section 8d "reference-scale sanity
    row"): performance/suffix_array_construction.c:81-146,196-200 times sa_is_construction on random DNA of n <= 49 000
    (performance/suffix_array_construction.txt:3153: 4.63 ms = 10.6 Msuffixes/s, hardware unstated).  Here: the same call
    through libstralg_amd.so's reference-named entry point on a HOST string (context warm; strlen, H2D, the launch chain, D2H
    into a malloc'd array: everything a stralg caller pays), median of `calls`, next to the unmodified reference (oracle/_ref)
    -- or the oracle port -- on the same string on one host core.  Arrays compared."""
    import ctypes as C
    import statistics
    import numpy as np
    import oracle
    from oracle import pyoracle
    from stralg_amd.benchlegs import cabi
    from stralg_amd.synth import synth
    cabi.declare(lib)
    ref = pyoracle._Ref() if pyoracle.have_ref() else None
    out = {"published": {"n": 49000, "ms": 4.63, "Msuffixes_per_s": 10.6, "hardware": "unstated",
                         "source": "performance/suffix_array_construction.txt:3153 (SA-IS, random DNA)"},
           "calls": calls, "cpu_kind": "reference" if ref else "port", "cpu_cores": 1, "rows": []}
    for n in sizes:
        x = np.concatenate([synth(n, 5, seed), np.zeros(1, np.uint8)])
        a = lib.sa_is_construction(x.ctypes.data, 5)  # warm: the thread's context, its slabs, the pinned staging
        got = np.ctypeslib.as_array(a.contents.array, shape=(n + 1,)).copy()
        lib.free_suffix_array(a)
        gpu = []
        for _ in range(calls):
            t0 = time.perf_counter()
            a = lib.sa_is_construction(x.ctypes.data, 5)
            gpu.append(time.perf_counter() - t0)
            lib.free_suffix_array(a)
        cpu = []
        for _ in range(max(5, calls // 5)):
            t0 = time.perf_counter()
            want = ref.sa_is(x[:n], 5) if ref else oracle.sa_is(x[:n], 5)
            cpu.append(time.perf_counter() - t0)  # (includes the wrapper's copy of the 4(n+1)-byte array: microseconds)
        g, c = statistics.median(gpu), statistics.median(cpu)
        out["rows"].append({"n": n, "gpu_call_ms": round(g * 1e3, 3), "gpu_call_ms_min": round(min(gpu) * 1e3, 3),
                            "gpu_Msuffixes_per_s": round((n + 1) / g / 1e6, 1),
                            "cpu_ms": round(c * 1e3, 3), "cpu_Msuffixes_per_s": round((n + 1) / c / 1e6, 1),
                            "gpu_over_cpu": round(c / g, 2), "arrays_identical": bool((got == want).all())})
    lib.stralg_amd_release()
    out["what"] = ("sa_is_construction(host string, 5) through libstralg_amd.so (median of the calls, warm context: strlen, H2D, "
                   "launch chain, D2H into a malloc'd array) next to the same call of the CPU baseline on one host core")
    return out


def _cfg5_worker(args):
    """one CPU process of the cfg5 row: the reference's (or the port's) construction on one record's sample"""
    seed, log2n, sigma = args
    import oracle
    from oracle import pyoracle
    from stralg_amd.synth import synth
    x = synth(1 << log2n, sigma, seed)
    ref = pyoracle._Ref() if pyoracle.have_ref() else None
    t0 = time.perf_counter()
    sa = ref.sa_is_mem(x, sigma) if ref else oracle.sa_is(x, sigma)
    return time.perf_counter() - t0, int(sa[1])


def cpu_cfg5_row(records, log2n, sigma=5, seed=42):
    """SURVEY.md section 8d's CPU row for configs[4]: `records` CPU processes at once, one per record (a bounded sample of
    each record: its first 2^log2n symbols; the reference is single-threaded, so this is what `records` cores give), next to
    the single-thread figure.  Aggregate = all samples' suffixes / the slowest process's time."""
    import multiprocessing as mp
    from oracle import pyoracle
    procs = max(1, min(records, os.cpu_count() or 1))
    jobs = [(seed + r, log2n, sigma) for r in range(records)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(procs) as pool:  # (spawn: the children never see this process's GPU state)
        res = pool.map(_cfg5_worker, jobs)
    wall = time.perf_counter() - t0
    per = [r[0] for r in res]
    N = (1 << log2n) + 1
    return {"records": records, "processes": procs, "host_cpus": os.cpu_count(), "kind": "reference" if pyoracle.have_ref() else "port",
            "fn": "sa_is_mem_construction" if pyoracle.have_ref() else "oracle_sa_is", "sample": f"the first 2^{log2n} symbols of each record",
            "seconds_per_process": [round(v, 2) for v in per],
            "aggregate_Msuffixes_per_s": round(records * N / max(max(per), 1e-9) / 1e6, 3) if procs == records else
            round(records * N / wall / 1e6, 3),
            "single_thread_Msuffixes_per_s": round(N / min(per) / 1e6, 3), "unit": "Msuffixes/s",
            "note": "suffix arrays only (the reference's O table overflows its uint32_t size beyond 204.8 Mi positions, bwt.c:50-51)"}


# ---- one rank ----------------------------------------------------------------------------------------------

def run_rank(args):
    import torch
    import stralg_amd
    from stralg_amd import farm, workloads
    from stralg_amd.benchlegs.ceiling import measured_ceiling
    from stralg_amd.benchlegs.configs import other_configs
    from stralg_amd.benchlegs.hostpath import egress_leg, end_to_end
    from stralg_amd.benchlegs.pins import reference_pin

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # STRALG_BENCH_BACKEND=gloo + STRALG_BENCH_SHARE_GPU=1 let the N > 1 path be exercised on a single-GPU box: every
    # rank then uses cuda:0 and the scalars travel on the CPU.  STRALG_BENCH_EMU=1 (tests, no GPU) runs the same
    # code over the CPU execution harness of the kernels (tests/emu): torch CPU tensors, gloo.
    emu = os.environ.get("STRALG_BENCH_EMU") == "1"
    backend = os.environ.get("STRALG_BENCH_BACKEND", "gloo" if emu else "nccl")
    if os.environ.get("STRALG_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if emu:
        dev = torch.device("cpu")
        ctx = stralg_amd.Context(0, lib_path=os.path.join(ROOT, "tests", "emu", "libstralg_amd_emu.so"))
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        ctx = stralg_amd.Context(local_rank)
    if args.no_direct_sort:
        ctx.set_no_direct_sort(True)
    if args.sort_mode:
        ctx.set_sort_mode(args.sort_mode)
    if args.prefix_symbols:
        ctx.set_prefix_symbols(args.prefix_symbols)
    if args.chain_max >= 0:
        ctx.set_chain_max_entries(args.chain_max)
    if args.no_induce_batch:
        ctx.set_induce_batch(False)
    if args.no_text_keys:
        ctx.set_text_keys(False)
    if args.copy_text_first:
        ctx.set_copy_text_first(True)
    if args.induce_attended:
        ctx.set_induce_attended(1)
    # host work of a rank (staging copies, page faults of pinned and malloc'd buffers) next to its GPU's PCIe root
    numa_node = ctx.bind_to_numa_node()
    cuda = dev.type == "cuda"
    coll = None
    if world > 1:
        import torch.distributed as dist
        # gloo is the control plane and always comes up; RCCL ("nccl") is tried beside it and carries the timing barrier
        # and the scalar reductions when the attempt succeeds on EVERY rank (farm.init_collectives) -- a first RCCL run
        # that fails (no such run has been possible on the builder's one-GPU boxes) costs the line a label, not the run
        coll = farm.init_collectives(rank, world, dev if cuda else None, prefer=backend)
        ident = {"rank": rank, "local_rank": local_rank, "device": str(dev)}
        if cuda:
            pr = torch.cuda.get_device_properties(dev)
            ident["device_name"] = pr.name
            ident["pci_bus_id"] = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0),
                                                      getattr(pr, "pci_device_id", 0))
            ident["hbm_GiB"] = round(pr.total_memory / 2**30, 1)
        coll["ranks"] = farm.gather_objects(ident)
        # N ranks must mean N GPUs: two ranks on one device (a launcher that handed out the same LOCAL_RANK, a masked
        # HIP_VISIBLE_DEVICES) would report a batch throughput no batch of GPUs has.  Fail loudly: one JSON line with `error`
        # from rank 0, a non-zero exit on every rank.  ($STRALG_BENCH_SHARE_GPU=1: the builder's one-GPU rehearsal.)
        if cuda and os.environ.get("STRALG_BENCH_SHARE_GPU") != "1":
            seen = [(r.get("pci_bus_id"), r.get("device_name")) for r in coll["ranks"]]
            shared = sorted({p[0] for p in seen if seen.count(p) > 1})
            if shared:
                if rank == 0:
                    print(json.dumps({"metric": "Msuffixes/s", "value": None, "unit": "Msuffixes/s", "n_gpus": world,
                                      "error": f"{world} ranks but GPUs {shared} are used by more than one of them "
                                               "(set STRALG_BENCH_SHARE_GPU=1 to rehearse on one GPU)",
                                      "ranks": coll["ranks"]}), flush=True)
                dist.destroy_process_group()
                ctx.close()
                return 3

    def sync():
        if cuda:
            torch.cuda.synchronize()

    workload = args.workload or ("dna" if world == 1 else "fasta")
    n = args.n if args.n > 0 else 1 << args.log2n
    seed = 42 + rank
    gen = "dna" if workload == "fasta" else workload
    text, sigma = workloads.make_text(ctx, gen, n, args.sigma, seed, dev)
    sync()  # (torch's generators run on torch's stream, the library on its own)
    N = n + 1
    tables = (not args.no_tables) and sigma <= 128
    cpu_n = min(n, 1 << args.cpu_log2n)
    cpu_sample = text[:cpu_n].cpu().numpy() if rank == 0 and world == 1 and not args.no_cpu else None
    job = None
    if workload == "fasta":
        image = workloads.fasta_image(text, f"record{rank}")
        h_image = image.cpu()
        if cuda:
            h_image = h_image.pin_memory()
        del image
        job = farm.FastaRecordJob(ctx, h_image, dev, tables)
        job.upload()
        sa = bwt = c_tab = o_tab = None
    else:
        sa = torch.empty(N, dtype=torch.int32, device=dev)
        c_tab = torch.zeros(sigma, dtype=torch.int32, device=dev)
        o_tab = torch.empty((N + 1) * sigma, dtype=torch.int32, device=dev) if tables else None
        bwt = torch.empty(N, dtype=torch.uint8, device=dev) if tables else None

    def step():
        # build_complete_table's device work (stralg/bwt.c:143,154): suffix array, then C and O.
        # The induced-sort passes hand the BWT over with the suffix array (sx_sa_bwt_build_dev).
        if job is not None:
            job.build()
        elif tables:
            ctx.sa_bwt_build_dev(text, n, sigma, sa, bwt)
            ctx.bwt_tables_from_bwt_dev(bwt, N, sigma, c_tab, o_tab)
        else:
            ctx.sa_build_dev(text, n, sigma, sa)

    def profiled_step():
        """one untimed step with HIP events around every launch: the per-class table, and which class dominates"""
        ctx.profile_reset()
        ctx.profile_only(None)
        ctx.profile_enable(True)
        step()
        sync()
        ctx.profile_enable(False)
        return ctx.profile_read()

    # the box's measured memory ceiling, a few milliseconds before the timed region (rank 0 reports it)
    ceiling = measured_ceiling(ctx, dev, nbytes=(1 << 20) if emu else (2 << 30)) if (rank == 0 and not args.no_ceiling) else None
    for _ in range(args.warmup):
        step()
    # one more untimed step, with events around every launch: the per-class table and the dominant class
    table = profiled_step()
    dom = max(table, key=lambda k: table[k]["ms"])  # the class with the largest summed time, whichever it is
    # Timed region: events only on the dominant kernel's launches (the roofline figure is measured live, on the library's
    # own stream), and only in every M-th of the timed steps: a launch that carries events leaves the queue idle for ~6 us
    # in front of it and ~5 us behind it (rocprofv3 timeline; hipEventRecord pairs and hipExtLaunchKernelGGL's events alike),
    # 0.39 ms of a 20 ms step for this class's 35 launches -- and two events on each of a step's ~300 launches ~5 % of it.
    # Between the sampled steps the events are switched to a class no build launches (no synchronisation: the switch is a word).
    every = max(1, args.roofline_every)
    timed_steps = {"i": 0, "n": 0}

    def timed_step():
        sampled = timed_steps["i"] % every == 0
        ctx.profile_only(dom if sampled else "search")
        timed_steps["i"] += 1
        timed_steps["n"] += 1 if sampled else 0
        step()

    ctx.profile_reset()
    ctx.profile_only("search")
    ctx.profile_enable(True)
    # barrier + torch.cuda.synchronize() on both sides of exactly `steps` steps
    elapsed_own = farm.timed(timed_step, args.steps, 0, cuda=cuda)
    ctx.profile_enable(False)
    prof = ctx.profile_read()
    ctx.profile_only(None)
    stats = ctx.last_stats()
    # max time over ranks, total suffixes over ranks (the only collectives; none on the data path)
    red_dev = None  # (farm.reduce_scalars uses the group and device farm.init_collectives chose)
    elapsed, total_units = farm.reduce_scalars(elapsed_own, args.steps * N, device=red_dev)

    # ---- after the timed region: check the last step's results on the device ------------------------------
    verified, checks = None, []
    if not args.no_verify:
        from stralg_amd import verify
        ctx.trim()  # the checks need the memory more than the cached workspace does
        try:
            if job is not None:
                checks = verify.verify_build_on_device(job.d_text, job.n, job.sigma, job.sa, job.bwt if tables else None,
                                                       job.c if tables else None, job.o)
                # the record that came out of the FASTA image is the record that went in
                if job.n != n or not bool((job.d_text[:n] == text).all()):
                    raise AssertionError("FASTA ingest changed the record")
                checks.append("FASTA pack + remap reproduce the record")
            else:
                checks = verify.verify_build_on_device(text, n, sigma, sa, bwt, c_tab if tables else None, o_tab)
            verified = True
        except AssertionError as e:
            verified = False
            checks = [f"FAILED on rank {rank}: {e}"]
            print(f"bench.py: verification failed on rank {rank}: {e}", file=sys.stderr)
    ref_pin = None
    if rank == 0 and cuda and workload in ("dna", "bytes") and sa is not None and not args.no_verify:
        ref_pin = reference_pin(sa, n, sigma, seed)  # (a fixture exists for 2^28 and 2^30 symbols, sigma 5 and 256, seed 42)
        if ref_pin is not None and not ref_pin["match"]:
            verified = False
            checks.append("FAILED: the suffix array's SHA-256 differs from the reference's")
    bad_ranks, _ = farm.reduce_scalars(1.0 if verified is False else 0.0, 0, device=red_dev)

    # ---- the FASTA record with its ingest, every rank at once (what limits config 5: PCIe / host) --------
    fasta = None
    if job is not None:
        def ingest_step():
            job.upload()
            job.build()
        t_in = farm.timed(ingest_step, max(1, min(args.steps, 3)), 0, cuda=cuda)
        k = max(1, min(args.steps, 3))
        t_in_max, units_in = farm.reduce_scalars(t_in, k * N, device=red_dev)
        fasta = {"file_bytes_per_record": job.file_len, "steps": k,
                 "kernel_only_Msuffixes_per_s": round(total_units / elapsed / 1e6, 3),
                 "ingest_inclusive_Msuffixes_per_s": round(units_in / t_in_max / 1e6, 3),
                 "ingest_inclusive_ms_per_record": round(t_in_max / k * 1e3, 3),
                 "h2d_GBps_per_rank": None,
                 "note": "ingest = H2D copy of the pinned file image, all ranks at once, then the same step"}
        t0 = time.perf_counter()
        job.upload()
        fasta["h2d_GBps_per_rank"] = round(job.file_len / (time.perf_counter() - t0) / 1e9, 2)
        fasta["numa_node_per_rank"] = farm.gather_ints(numa_node if numa_node is not None else -1)
        if tables and not args.no_ro:
            # What bwt_readmapper.c:57 actually asks for: build_complete_table(rec.seq, true) -- the reversed string's suffix
            # array and the RO table too (bwt.c:147-158), on the device: sx_reverse_dev, a second sx_sa_bwt_build_dev,
            # sx_bwt_tables_from_bwt_dev.  Timed like the forward step (kernel-only, then with the ingest), verified.
            job.set_include_reverse(True)
            job.build()  # (allocates the reverse direction's buffers)
            k = max(1, min(args.steps, 3))
            t_ro = farm.timed(job.build, k, 0, cuda=cuda)
            t_ro_max, units_ro = farm.reduce_scalars(t_ro, k * N, device=red_dev)
            t_ro_in = farm.timed(ingest_step, k, 0, cuda=cuda)
            t_ro_in_max, units_ro_in = farm.reduce_scalars(t_ro_in, k * N, device=red_dev)
            ro = {"steps": k, "kernel_only_ms_per_record": round(t_ro_max / k * 1e3, 3),
                  "kernel_only_Msuffixes_per_s": round(units_ro / t_ro_max / 1e6, 3),
                  "ingest_inclusive_ms_per_record": round(t_ro_in_max / k * 1e3, 3),
                  "ingest_inclusive_Msuffixes_per_s": round(units_ro_in / t_ro_in_max / 1e6, 3),
                  "forward_only_ms_per_record": round(elapsed / args.steps * 1e3, 3),
                  "note": "build_complete_table(seq, true): forward SA + BWT + C/O, then reversal, the reverse suffix array and RO; "
                          "Msuffixes/s counts the record's n + 1 suffixes once"}
            if not args.no_verify:
                from stralg_amd import verify
                ctx.trim()
                try:
                    # RO is the O table of the reversed string (bwt.c:67-88): its rows by the one-hot property on that
                    # string's BWT, its suffix array by the O(n) order check, the reversal itself entry by entry
                    if not bool((job.d_rev[:job.n] == torch.flip(job.d_text[:job.n], dims=[0])).all()) or int(job.d_rev[job.n]) != 0:
                        raise AssertionError("the reversed string is not the reverse")
                    verify.verify_build_on_device(job.d_rev, job.n, job.sigma, job.rsa, job.rbwt, job.rc, job.ro)
                    if not bool((job.rc == job.c).all()):
                        raise AssertionError("the reverse direction's C table differs from the forward one")
                    ro["verified"] = True
                except AssertionError as e:
                    ro["verified"] = False
                    ro["error"] = str(e)
                    print(f"bench.py: reverse-direction verification failed on rank {rank}: {e}", file=sys.stderr)
                bad_ro, _ = farm.reduce_scalars(0.0 if ro["verified"] else 1.0, 0, device=red_dev)
                ro["verified"] = ro["verified"] and bad_ro == 0.0
            fasta["with_ro"] = ro
            job.set_include_reverse(False)
        if not args.no_egress:
            job_n, job_sigma = job.n, job.sigma
            job = None  # the index is rebuilt from the host's copy of the record: free the device-resident one
            ctx.trim()
            if cuda:
                torch.cuda.empty_cache()
            eg = egress_leg(ctx, local_rank, text, job_n, job_sigma, world, red_dev, cuda, with_ro=tables and not args.no_ro)
            fasta.update(eg)

    if rank == 0:
        value = total_units / elapsed / 1e6
        # dominant kernel class (by summed HIP-event time of the profiled step), measured in the timed steps
        d = prof[dom]
        achieved = d["alg_bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0
        size_label = (f"{n >> 30} GiB" if n >= (1 << 30) and n % (1 << 30) == 0 else
                      (f"{n >> 20} MiB" if n >= (1 << 20) and n % (1 << 20) == 0 else f"{n} B"))
        alpha_label = {"dna": "DNA", "fasta": "DNA, FASTA records", "bytes": "sigma=256"}.get(workload, workload if workload != "uniform" else f"sigma={sigma}")
        # name the algorithm that ran: lms_path 3 is the direct prefix sort of all suffixes (sx_build.hip), not SA-IS
        algo = "direct prefix sort" if stats.get("lms_path") == 3 else "SA-IS"
        what = f"{algo} + BWT C/O tables" if tables else algo
        if workload == "fasta":
            wl = (f"one FASTA record of 2^{args.log2n} bases per GPU (bwt_readmapper.c:54-62): image in HBM -> pack -> remap -> "
                  f"sa_is_construction + init_bwt_table (C, O)" if args.n == 0 else f"one FASTA record of {n} bases per GPU")
        elif tables:
            wl = (f"sa_is_construction + init_bwt_table (C, O) on {n} symbols ({workload}), alphabet_size={sigma}, "
                  f"one independent record per GPU")
        else:
            wl = f"sa_is_construction on {n} symbols ({workload}), alphabet_size={sigma}"
        traffic, traffic_stale = pmc_traffic(workload, args.log2n, sigma, tables, dom) if args.n == 0 else (None, None)
        out = {
            "metric": f"Msuffixes/s ({what}, {size_label} {alpha_label})",
            "value": round(value, 3),
            "unit": "Msuffixes/s",
            "n_gpus": world,
            **({"collective_backend": coll["collective_backend"], "n_ranks_seen": coll["n_ranks_seen"]} if coll else {}),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8 text / u32 indices",
            "data": "synthetic",
            "config": {
                "workload": wl,
                "n": n, "alphabet_size": sigma, "records_per_gpu": 1, "parallelism": f"batch x{world}, no collectives",
                "seed": 42, "generator": gen, "numa_node_rank0": numa_node,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_stale": traffic_stale,  # the kernel sources have changed since the PMC passes (None: no figure)
                "traffic_source": (f"{PMC_TRAFFIC}: rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this command, committed by the "
                                   "builder; not measured in this run") if traffic is not None else None,
                "launches": d["launches"],
                "avg_ms": round(d["ms"] / max(1, d["launches"]), 4),
                "timed_steps": timed_steps["n"],  # the steps of the timed region whose launches of this class carried the events
                "of_steps": args.steps,
                # the same fraction against what THIS box streams (sx_membw_probe before the timed region): boxes of the pool
                # differ by 12 - 14 % on unchanged kernels, a fraction of the data sheet's 8 TB/s cannot tell which moved
                **({"peak_measured": ceiling["peak_measured"],
                    "frac_of_measured": round(achieved / ceiling["peak_measured"], 4),
                    "measured": {k: ceiling[k] for k in ("read", "fill", "copy", "split4")}}
                   if ceiling and ceiling.get("peak_measured") else ({"peak_measured": None, "ceiling_error": ceiling.get("error")}
                                                                      if ceiling else {})),
            },
            "verified": verified if bad_ranks == 0.0 else False,
            "verified_checks": checks,
            **({"reference_pin": ref_pin} if ref_pin is not None else {}),
            # per-class times of ONE untimed step with events around every launch (run between warm-up and timing)
            "kernels": {k: {"ms_per_step": round(v["ms"], 3), "launches_per_step": v["launches"],
                            "GBps": round(v["alg_bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else 0.0}
                        for k, v in table.items() if v["launches"]},
            "build_stats": stats,
        }
        # whole-step algorithmic traffic over the step time: sum of the classes' bytes
        alg_total = sum(v["alg_bytes"] for v in table.values())
        out["whole_step"] = {"alg_GB": round(alg_total / 1e9, 2),
                             "GBps": round(alg_total / (elapsed / args.steps) / 1e9, 1),
                             "frac_of_peak": round(alg_total / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4)}
        if ceiling and ceiling.get("peak_measured"):
            out["whole_step"]["frac_of_measured"] = round(alg_total / (elapsed / args.steps) / 1e9 / ceiling["peak_measured"], 4)
            out["memory_ceiling"] = ceiling
        # the same from the hardware counters (BASELINE's target is stated on rocprof's HBM bytes): replayed like roofline.traffic
        hbm_total, hbm_stale, hbm_missing = pmc_whole_step(workload, args.log2n if args.n == 0 else -1, sigma, tables,
                                                           {k: v["launches"] for k, v in table.items()})
        if hbm_total is not None:
            out["whole_step"].update({"hbm_GB_pmc": round(hbm_total / 1e9, 2),
                                      "hbm_GBps_pmc": round(hbm_total / (elapsed / args.steps) / 1e9, 1),
                                      "hbm_frac_of_peak_pmc": round(hbm_total / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                                      "hbm_pmc_stale": hbm_stale, "hbm_pmc_classes_without_figure": hbm_missing})
        if fasta is not None:
            if not args.no_cpu:
                try:  # SURVEY.md section 8d's CPU row for configs[4]: one CPU process per record, all at once
                    fasta["cpu_cfg5"] = cpu_cfg5_row(max(world, 1) if world > 1 else 8, min(args.cpu_log2n, args.log2n))
                except Exception as e:  # noqa: BLE001
                    fasta["cpu_cfg5"] = {"error": f"{type(e).__name__}: {e}"}
            out["fasta_record"] = fasta
        if coll is not None:
            # who took part: the driver can check "RCCL saw N ranks" and that N distinct GPUs did the work
            out["collectives"] = {"backend": coll["collective_backend"], "n_ranks_seen": coll["n_ranks_seen"],
                                  "nccl_error": coll["nccl_error"], "control_plane": "gloo", "ranks": coll["ranks"],
                                  "distinct_devices": len({(r.get("pci_bus_id"), r.get("device")) for r in coll["ranks"]}),
                                  "note": "no collective on the data path; this group carries the timing barrier and the "
                                          "max / sum of the bookkeeping scalars only"}
        if world == 1 and not args.no_e2e and workload in ("dna", "fasta"):
            del text, sa, bwt, c_tab, o_tab
            job = None
            ctx.trim()
            if cuda:
                torch.cuda.empty_cache()
            out["end_to_end"] = end_to_end(ctx, [int(v) for v in args.e2e_log2n.split(",") if v], 42)
        if (world == 1 and not args.no_other_configs and workload == "dna" and args.n == 0 and not args.no_tables
                and (args.log2n == 30 or emu)):
            try:
                del text, sa, bwt, c_tab, o_tab
            except NameError:
                pass
            job = None
            ctx.trim()
            if cuda:
                torch.cuda.empty_cache()
            out["other_configs"] = other_configs(ctx, dev, max(1, args.other_steps), cuda, args.log2n,
                                                 cpu_whole_record=cpu_baseline_at_size if (not args.no_cpu and not args.no_cpu_whole_record
                                                                                           and not emu) else None)
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(cpu_sample, sigma, f"the first {cpu_n} symbols of the record rank 0 built")
            whole = next((c["cpu_reference_whole_record"] for c in out.get("other_configs", {}).values()
                          if isinstance(c, dict) and "cpu_reference_whole_record" in c), None)
            if whole is not None:  # the same baseline at a benchmark size: configs[1]'s whole 2^28-symbol record
                out["cpu_baseline"]["whole_record"] = whole
            if not args.no_reference_scale:
                try:  # the reference's published operating point through the host C API (outside the timed region)
                    out["cpu_baseline"]["reference_scale"] = reference_scale(ctx.lib, calls=3 if emu else 50)
                except Exception as e:  # noqa: BLE001 -- an extra must not take the headline line down with it
                    out["cpu_baseline"]["reference_scale"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        farm.fence(cuda)
        dist.destroy_process_group()
    ctx.close()
    return 1 if bad_ranks else 0


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)  # (this process stays off the GPU)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
