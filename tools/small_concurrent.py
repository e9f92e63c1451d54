"""records too short to fill the GPU: K contexts (own stream, own workspace) in K host threads on one GPU, resident data --
what the batch farm could gain from more than one worker a device"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stralg_amd
K_MAX = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctxs = [stralg_amd.Context(0) for _ in range(K_MAX)]
for log2n in (13, 16, 18, 20, 22, 24):
    n = 1 << log2n; N = n + 1
    bufs = []
    for k in range(K_MAX):
        text = torch.empty(n, dtype=torch.uint8, device="cuda"); ctxs[k].synth_dev(text, n, 5, 7 + k)
        bufs.append((text, torch.empty(N, dtype=torch.int32, device="cuda"), torch.empty(N, dtype=torch.uint8, device="cuda"),
                     torch.zeros(5, dtype=torch.int32, device="cuda"), torch.empty((N + 1) * 5, dtype=torch.int32, device="cuda")))
    reps = 40 if log2n <= 20 else 8
    line = f"n = 2^{log2n}:"
    for K in (1, 2, 4, 8):
        if K > K_MAX: break
        def work(k):
            text, sa, bw, c, o = bufs[k]
            for _ in range(reps):
                ctxs[k].sa_bwt_build_dev(text, n, 5, sa, bw); ctxs[k].bwt_tables_from_bwt_dev(bw, N, 5, c, o)
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
            [t.start() for t in th]; [t.join() for t in th]
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        line += f"  K={K}: {dt / (reps * K) * 1e3:7.3f} ms/record = {N * reps * K / dt / 1e6:8.1f} Msuffixes/s"
    print(line, flush=True)
