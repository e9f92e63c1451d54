"""HBM bytes from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json): replayed into the bench line, stamped."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PMC_TRAFFIC = os.path.join("profiles", "pmc_traffic.json")


def pmc_traffic(workload, log2n, sigma, tables, klass):
    """HBM bytes per launch of the dominant kernel class from the committed rocprofv3 PMC passes of this
    same command (profiles/pmc_traffic.json, made by tools/gpu_step.sh's prof step + tools/pmc_to_json.py).  PMC counters
    cannot be read from inside the process: the figure is replayed from that file, not measured in this run
    (roofline.traffic_source says so); None for workloads that were not profiled.  Returns (bytes, stale): stale when the
    kernel sources have changed since the passes were collected (the file is stamped with their SHA-256)."""
    try:
        from stralg_amd._lib import kernel_sources_sha16
        doc = json.load(open(os.path.join(ROOT, PMC_TRAFFIC)))
        key = f"log2n={log2n} sigma={sigma} tables={int(tables)}"
        entry = doc[key if workload == "dna" else f"workload={workload} " + key]
        c = entry["classes"][klass]
        return round(c["hbm_bytes_per_launch"]), entry.get("kernel_sources_sha16") != kernel_sources_sha16()
    except (OSError, KeyError, ValueError):
        return None, None


def pmc_whole_step(workload, log2n, sigma, tables, launches_per_class):
    """HBM bytes of one whole step as rocprofv3's counters saw them: the committed per-class figures (pmc_traffic above)
    times this run's launches per class.  Returns (bytes, stale, classes the passes have no figure for) or (None, None, None)."""
    try:
        from stralg_amd._lib import kernel_sources_sha16
        doc = json.load(open(os.path.join(ROOT, PMC_TRAFFIC)))
        key = f"log2n={log2n} sigma={sigma} tables={int(tables)}"
        entry = doc[key if workload == "dna" else f"workload={workload} " + key]
        total, missing = 0.0, []
        for klass, launches in launches_per_class.items():
            c = entry["classes"].get(klass)
            if c is None:
                if launches:
                    missing.append(klass)  # (a class without a figure counts as no traffic: the total is a lower bound)
                continue
            total += c["hbm_bytes_per_launch"] * launches
        return total, entry.get("kernel_sources_sha16") != kernel_sources_sha16(), missing
    except (OSError, KeyError, ValueError):
        return None, None, None
