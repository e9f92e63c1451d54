"""The box's measured memory ceiling ("confirm on the box ... and state the figure used"; VERDICT
round 4, item 4): sx_membw_probe's four streaming shapes over 2 x 2 GiB, a few milliseconds, run before the timed region."""
import math


def measured_ceiling(ctx, dev, nbytes=2 << 30, reps=5):
    """{"read": GB/s, "fill", "copy", "split4", "peak_measured": the best of them, ...} or {"error": ...}"""
    import torch
    try:
        a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        b = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        a.random_(0, 256)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        r = ctx.membw_probe(a, b, nbytes, reps)
        del a, b
        if dev.type == "cuda":
            torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001 -- a probe must not take the bench line down
        return {"error": f"{type(e).__name__}: {e}"}
    out = {k: (round(v, 1) if math.isfinite(v) else None) for k, v in r.items()}
    finite = [v for v in out.values() if v is not None]
    out["peak_measured"] = max(finite) if finite else None
    out["unit"] = "GB/s"
    out["bytes_per_buffer"] = nbytes
    out["what"] = ("sx_membw_probe on this box before the timed region: 16-byte-a-lane read / fill / copy (bytes both ways) and a "
                   "four-way split of 4-byte entries (the induced-sort scatters' store shape); best of %d launches each" % reps)
    return out
