"""Copies the summaries of tools/r05_final.sh's runs (gpurun_out/r05_final/{a,b,c}) into profiles/ and rebuilds
profiles/pmc_traffic.json (stamped with the kernel sources' hash).   python tools/r05_collect.py"""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r05_final")
DST = os.path.join(ROOT, "profiles")


def cp(src, dst):
    src = os.path.join(SRC, src)
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copyfile(src, os.path.join(DST, dst))
        print("->", dst)
    else:
        print("MISSING", src)


def last_line(src, dst):
    src = os.path.join(SRC, src)
    if not os.path.exists(src):
        print("MISSING", src)
        return
    lines = [l for l in open(src).read().splitlines() if l.startswith("{")]
    if lines:
        open(os.path.join(DST, dst), "w").write(lines[-1] + "\n")
        print("->", dst)


def kernel_stats(prof_dir, dst):
    """the kernel-stats part of a prof step's summary"""
    p = os.path.join(SRC, prof_dir, "summary.txt")
    if os.path.exists(p):
        open(os.path.join(DST, dst), "w").write(open(p).read())
        print("->", dst)
    else:
        print("MISSING", p)


def pmc(prof_dir, key):
    d = os.path.join(SRC, prof_dir)
    if glob.glob(os.path.join(d, "pmc_fetch", "**", "*counter_collection.csv"), recursive=True):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_to_json.py"), d, key,
                               os.path.join(DST, "pmc_traffic.json")], stdout=subprocess.DEVNULL)
        print("pmc:", key)
    else:
        print("MISSING PMC", d)


# part a
last_line("a/bench_1.json", "r05_final_bench_1GiB_dna.json")
kernel_stats("a/prof_2", "r05_final_bench_1GiB_dna_rocprofv3_summary.txt")
for f in glob.glob(os.path.join(SRC, "a", "prof_2", "trace", "**", "*kernel_stats.csv"), recursive=True)[:1]:
    shutil.copyfile(f, os.path.join(DST, "r05_final_bench_1GiB_dna_kernel_stats.csv"))
cp("a/py_3.log", "r05_e2e_phases.txt")
if os.path.exists(os.path.join(DST, "pmc_traffic.json")):
    os.unlink(os.path.join(DST, "pmc_traffic.json"))  # (every key is measured again: no figure of an earlier round is carried over)
pmc("a/prof_2", "log2n=30 sigma=5 tables=1")
# part b
for k, (name, key) in enumerate((("dna_256MiB", "log2n=28 sigma=5 tables=1"),
                                 ("bytes", "workload=bytes log2n=30 sigma=256 tables=0"),
                                 ("bytes_induced", "workload=bytes_induced log2n=30 sigma=256 tables=0"),
                                 ("genome_like", "workload=genome_like log2n=30 sigma=5 tables=1"),
                                 ("fasta", "workload=fasta log2n=30 sigma=5 tables=1")), start=1):
    kernel_stats(f"b/prof_{k}", f"r05_final_{name}_rocprofv3_kernel_stats.txt")
    pmc(f"b/prof_{k}", key)
# part c
for k, name in enumerate(("bytes", "bytes_induced", "sigma21", "sigma21_induced", "sigma6", "genome_like", "n_runs", "text_like",
                          "pangenome", "periodic_1GiB", "fasta", "dna_4MiB"), start=1):
    last_line(f"c/bench_{k}.json", f"r05_final_bench_{name}.json")
last_line("c/bench2n_13.json", "r05_bench_2rank_rccl_refused.json")
last_line("c/bench2_14.json", "r05_final_bench_2rank_shared_gpu.json")
cp("c/py_15.log", "r05_next_rows_bench.txt")
kernel_stats("c/prof_16", "r05_next_rows_rocprofv3_summary.txt")
