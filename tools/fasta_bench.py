"""FASTA packing of a synthetic image on the device (for rocprofv3 --kernel-trace --stats)"""
import sys, time
import numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stralg_amd
ctx = stralg_amd.Context(0)
rng = np.random.default_rng(1)
rec_n = (1 << 27) - 4096
seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=rec_n)
full = rec_n // 60
body = np.concatenate([seq[: full * 60].reshape(full, 60), np.full((full, 1), 10, dtype=np.uint8)], axis=1).tobytes() + seq[full * 60:].tobytes() + b"\n"
data = b"".join(b">chr%d\n" % k + body for k in range(8))
d_file = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
d_packed = torch.empty(len(data) + 1, dtype=torch.uint8, device="cuda")
d_term = torch.empty(64, dtype=torch.int32, device="cuda")
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); plen, nrec = ctx.fasta_pack_dev(d_file, len(data), d_packed, d_term, 64); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"FASTA pack, {len(data)/2**30:.2f} GiB image, {nrec} records: {dt*1e3:.1f} ms = {len(data)/dt/1e9:.0f} GB/s of file")
