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

dev = torch.device("cuda:0")
n = 1 << 30; N = n + 1; sigma = 5
ctx0 = stralg_amd.Context(0)
text = torch.empty(n, dtype=torch.uint8, device=dev)
ctx0.synth_dev(text, n, sigma, 42)
ctx0.close()
piece = 64 << 20
for trial in range(6):
    pad = torch.empty((trial * 611) << 20, dtype=torch.uint8, device=dev) if trial else None
    ctx = stralg_amd.Context(0)
    use_vmm = trial % 2 == 1
    if use_vmm:
        sa_p, sa_sz, sa_h, _ = vmm_alloc(4 * N, piece, 1 << 30)
        bw_p, bw_sz, bw_h, _ = vmm_alloc(N, piece, 1 << 30)
        sa, bwt = sa_p, bw_p
    else:
        sa_t = torch.empty(N, dtype=torch.int32, device=dev); bw_t = torch.empty(N, dtype=torch.uint8, device=dev)
        sa, bwt = sa_t.data_ptr(), bw_t.data_ptr()
    for _ in range(2):
        ctx.sa_bwt_build_dev(text, n, sigma, sa, bwt)
    torch.cuda.synchronize()
    ctx.profile_reset(); ctx.profile_only(None); ctx.profile_enable(True)
    ctx.sa_bwt_build_dev(text, n, sigma, sa, bwt)
    torch.cuda.synchronize()
    ctx.profile_enable(False)
    tab = ctx.profile_read()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.sa_bwt_build_dev(text, n, sigma, sa, bwt)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"context {trial} ({'vmm 64 MiB pieces' if use_vmm else 'torch'} for SA and BWT): step {ms:.2f} ms  " + "  ".join(f"{k} {v['ms']:.2f}" for k, v in tab.items() if v["ms"] > 0.3), flush=True)
    ctx.close()
    if use_vmm:
        vmm_free(sa_p, sa_sz, sa_h, piece); vmm_free(bw_p, bw_sz, bw_h, piece)
    else:
        del sa_t, bw_t
    del pad
    torch.cuda.empty_cache()
