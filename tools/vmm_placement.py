"""the O-table kernel over buffers obtained in different ways: torch (hipMalloc), and HIP's virtual memory management with one
physical allocation or with 1 GiB / 64 MiB / 2 MiB pieces mapped behind a 1 GiB-aligned range -- does the way a buffer is put
together decide how fast the scattered and many-window writers run?"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stralg_amd
hip = C.CDLL("libamdhip64.so")

class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]
class Flags(C.Structure):
    _fields_ = [("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]
class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", Loc), ("win32", C.c_void_p), ("allocFlags", Flags)]
class Access(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]

def chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: hip error {rc}")

def vmm_alloc(size, piece, align):
    prop = Prop(); prop.type = 1; prop.requestedHandleType = 0; prop.location = Loc(1, 0)
    gran = C.c_size_t(0)
    chk(hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), 1), "granularity")
    piece = max(piece, gran.value)
    size = (size + piece - 1) // piece * piece
    ptr = C.c_void_p(0)
    chk(hip.hipMemAddressReserve(C.byref(ptr), C.c_size_t(size), C.c_size_t(align), C.c_void_p(0), C.c_ulonglong(0)), "reserve")
    handles = []
    for off in range(0, size, piece):
        h = C.c_void_p(0)
        chk(hip.hipMemCreate(C.byref(h), C.c_size_t(piece), C.byref(prop), C.c_ulonglong(0)), "create")
        chk(hip.hipMemMap(C.c_void_p(ptr.value + off), C.c_size_t(piece), C.c_size_t(0), h, C.c_ulonglong(0)), "map")
        handles.append(h)
    acc = Access(Loc(1, 0), 3)
    chk(hip.hipMemSetAccess(ptr, C.c_size_t(size), C.byref(acc), C.c_size_t(1)), "access")
    return ptr.value, size, handles, gran.value

def vmm_free(ptr, size, handles, piece):
    chk(hip.hipMemUnmap(C.c_void_p(ptr), C.c_size_t(size)), "unmap")
    for h in handles:
        chk(hip.hipMemRelease(h), "release")
    chk(hip.hipMemAddressFree(C.c_void_p(ptr), C.c_size_t(size)), "addressfree")

ctx = stralg_amd.Context(0)
dev = torch.device("cuda:0")
n = 1 << 30; N = n + 1; sigma = 5
bwt = torch.randint(1, 5, (N,), dtype=torch.uint8, device=dev)
c_tab = torch.zeros(sigma, dtype=torch.int32, device=dev)
nbytes = (N + 1) * sigma * 4

def run(o_ptr, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.bwt_tables_from_bwt_dev(bwt, N, sigma, c_tab, o_ptr)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3

for rnd in range(3):
    big = torch.empty(nbytes // 4 + 1024, dtype=torch.int32, device=dev)
    line = [f"round {rnd}: torch {run(big.data_ptr()):.2f}"]
    del big; torch.cuda.empty_cache()
    for name, piece, align in (("vmm one piece", nbytes + (2 << 20), 1 << 30), ("vmm 1 GiB pieces", 1 << 30, 1 << 30),
                               ("vmm 64 MiB pieces", 64 << 20, 1 << 30), ("vmm 2 MiB pieces", 2 << 20, 2 << 20)):
        try:
            p, size, hs, gran = vmm_alloc(nbytes, piece, align)
            line.append(f"{name} {run(p):.2f}")
            vmm_free(p, size, hs, piece)
        except RuntimeError as e:
            line.append(f"{name}: {e}")
    print("  ".join(line), flush=True)
