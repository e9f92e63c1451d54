// sx_fasta.hip -- FASTA ingest and symbol remap on the device.
//
// bioinf/fasta.c:92-135 load_fasta_records packs the file image in place into
// "name\0sequence\0name\0sequence\0...": a header line loses every '>', ' ' and '\t' and ends at
// its newline (fasta.c:26-48); a sequence loses all white space and ends at the next '>',
// wherever it stands, or at the end of the file (fasta.c:50-70).  Whether a byte belongs to a
// header or to a sequence depends only on the last '\n' or '>' before it ('\n' -> sequence,
// '>' -> header, none -> header), so the sequential packing loop becomes three passes over
// 4096-byte tiles with 16 bytes per thread in registers:
//   1. (position, kind) of the last such byte of every tile; a max-scan over the tiles gives each
//      tile its entry state,
//   2. every thread walks its 16 bytes in the right state and counts what it emits (kept bytes,
//      terminators); sum-scans over the tiles give each tile its place in the packed image and in
//      the terminator table,
//   3. the same walk again, writing the packed image and the terminator positions (terminators
//      alternate header, sequence, header, ...: the record table).
// stralg/remap.c:8-31,102-114 (build table from the symbols present, relabel) is a presence
// histogram, a 256-entry table built on the host, and a streaming lookup.
#include "sx_common.hpp"
#include "sx_device.hpp"
#include "sx_scan.hpp"
#include "sx_internal.hpp"

namespace sx {

__device__ __forceinline__ bool fasta_space(uint32_t c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

// ---- packing: three passes over tiles of 4096 bytes, 16 bytes per thread in registers ------------------------
constexpr int kFaPer = 16, kFaTile = kBlock * kFaPer;

// the thread's 16 bytes (zero beyond `end`: position `end` itself is the terminating NUL of the reference's buffer).  In two
// steps, so that a workgroup can ask for the bytes of several tiles before it looks at the first (round 5): fasta_fetch16
// issues the load where one aligned 16-byte load does (everywhere but at the image's end), fasta_unpack16 spreads the bytes
// -- or reads them one by one.
__device__ __forceinline__ bool fasta_fetch16(const uint8_t *__restrict__ file, uint64_t i0, uint64_t end, uint4 &v)
{
    const bool fast = i0 + kFaPer <= end && ((uintptr_t)(file + i0) & 15u) == 0;
    if (fast) v = *reinterpret_cast<const uint4 *>(file + i0);
    return fast;
}
__device__ __forceinline__ void fasta_unpack16(const uint8_t *__restrict__ file, uint64_t i0, uint64_t end, bool fast, const uint4 &v,
                                               uint32_t (&b)[kFaPer])
{
    if (fast) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < kFaPer; ++k) b[k] = (w[k >> 2] >> (8 * (k & 3))) & 0xFFu;
    } else {
#pragma unroll
        for (int k = 0; k < kFaPer; ++k) b[k] = i0 + k < end ? (uint32_t)file[i0 + k] : 0u;
    }
}
// tiles a workgroup takes, one after the other, all of their loads in flight from the start.  Measured (round 5, 1 GiB image):
// 1, 2, 4 tiles a workgroup 2.2 ms each, 8: 2.4 -- the two kernels are not waiting for their loads: a tile costs each of its
// four waves some 600 vector instructions (the walk in both states, five block-wide reductions), 4 KiB per 1200 cycles of a
// CU = 2.1 TB/s, which is what they run at.  Left at one tile.
#ifndef SX_FASTA_SUB
#define SX_FASTA_SUB 1
#endif
constexpr int kFaSub = SX_FASTA_SUB;

// Which of a thread's 16 bytes are '\n', '>', white space (what a sequence drops: isspace()), ' ' or '\t' (what a header line
// drops besides '>'), NUL: bit k for byte k.  Where the 16 bytes came as one load they are classified four at a time in their
// words -- equality with a constant and "at least a constant" per byte without carries between the bytes, the four flag bits of
// a word gathered by a dot product (sx_classify.hip does the same for the type bits) --, 170 instructions where a compare and
// a shift for each byte and each class were 340 (round 5; the two kernels are bound by their instructions, 1.06 and 1.0 ms a GiB).
struct fa_masks {
    uint32_t nl, gt, sp, hdrop, zero;
};
__device__ __forceinline__ uint32_t fa_eq4(uint32_t w, uint32_t k4) // 0x80 in every byte of w that equals k4's
{
    const uint32_t x = w ^ k4;
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}
__device__ __forceinline__ uint32_t fa_ge4(uint32_t w, uint32_t k4) // 0x80 in every byte of w that is >= k4's (which are < 0x80)
{
    return (w | ((w | 0x80808080u) - k4)) & 0x80808080u;
}
__device__ __forceinline__ fa_masks fasta_masks16(const uint8_t *__restrict__ file, uint64_t i0, uint64_t end, bool fast, const uint4 &v)
{
    fa_masks m = {0, 0, 0, 0, 0};
    if (fast) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t nl[4], gt[4], sp[4], hd[4], ze[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t spc = fa_eq4(w[j], 0x20202020u), tab = fa_eq4(w[j], 0x09090909u);
            nl[j] = fa_eq4(w[j], 0x0A0A0A0Au);
            gt[j] = fa_eq4(w[j], 0x3E3E3E3Eu);
            ze[j] = fa_eq4(w[j], 0u);
            sp[j] = spc | (fa_ge4(w[j], 0x09090909u) & ~fa_ge4(w[j], 0x0E0E0E0Eu)); // ' ', or 9 ... 13
            hd[j] = spc | tab;
        }
        m.nl = gather16(nl[0], nl[1], nl[2], nl[3], 7);
        m.gt = gather16(gt[0], gt[1], gt[2], gt[3], 7);
        m.sp = gather16(sp[0], sp[1], sp[2], sp[3], 7);
        m.hdrop = gather16(hd[0], hd[1], hd[2], hd[3], 7);
        m.zero = gather16(ze[0], ze[1], ze[2], ze[3], 7);
    } else { // (the image's last bytes, or an image that does not start on a 16-byte boundary: zero beyond `end`)
#pragma unroll
        for (int k = 0; k < kFaPer; ++k) {
            const uint32_t c = i0 + k < end ? (uint32_t)file[i0 + k] : 0u;
            m.nl |= (c == '\n' ? 1u : 0u) << k;
            m.gt |= (c == '>' ? 1u : 0u) << k;
            m.sp |= ((c == ' ' || c - 9u < 5u) ? 1u : 0u) << k;
            m.hdrop |= ((c == ' ' || c == '\t') ? 1u : 0u) << k;
            m.zero |= (c == 0u ? 1u : 0u) << k;
        }
    }
    return m;
}
// the thread's positions in front of `end`
__device__ __forceinline__ uint32_t fasta_before(uint64_t i0, uint64_t end)
{
    return i0 >= end ? 0u : (end - i0 >= (uint64_t)kFaPer ? 0xFFFFu : (1u << (uint32_t)(end - i0)) - 1u);
}

// (position + 1, kind) of the last '\n' (kind 1: a sequence follows) or '>' (kind 0: a header follows) among the bytes
__device__ __forceinline__ uint32_t fasta_last_special(const fa_masks &m, uint64_t i0, uint64_t end)
{
    const uint32_t ev = (m.nl | m.gt) & fasta_before(i0, end);
    if (ev == 0) return 0;
    const uint32_t k = 31u - (uint32_t)__clz(ev);
    return ((uint32_t)(i0 + k + 1) << 1) | ((m.nl >> k) & 1u);
}

// What the packing loop of fasta.c:26-70 does with these bytes, entered in header or sequence state:
// emit: bytes that reach the packed image (kept characters and terminators), term: the terminators among them,
// name_end: header terminators; eof_in_name: the image ends inside a header line (MALFORMED_FILE).
struct fa_chunk {
    uint32_t emit, term, name_end;
    bool eof_in_name;
};
// The loop of fasta.c:26-70 is a two-state machine, but the state before a byte depends only on the last '\n'
// (a sequence follows, whatever the state was) or '>' (a header follows) before it, so the 16 bytes are classified
// without branches into bit masks, the states are filled in from the events by doubling (4 steps for 16 bits), and
// what is emitted follows from masks: a divergent 16-step walk per thread cost 1.2 - 1.4 ms per GiB and pass.
__device__ __forceinline__ fa_chunk fasta_walk(const fa_masks &m, uint64_t i0, uint64_t end, bool in_seq)
{
    const uint32_t nl = m.nl, gt = m.gt, sp = m.sp, hdrop = m.hdrop;
    // positions before `end`, and the position `end` itself (the terminating NUL of the reference's buffer)
    const uint32_t lt = fasta_before(i0, end);
    const uint32_t ate = (end >= i0 && end - i0 < (uint64_t)kFaPer) ? 1u << (uint32_t)(end - i0) : 0u;
    // state before every byte: events are "a sequence starts here" (after a newline) and "a header starts here"
    // (after a '>'); position 0 takes the entry state
    uint32_t val = (((nl & lt) << 1) | (in_seq ? 1u : 0u)) & 0xFFFFu;
    uint32_t have = (val | ((gt & lt) << 1) | 1u) & 0xFFFFu;
#pragma unroll
    for (int s = 1; s < kFaPer; s <<= 1) {
        const uint32_t fresh = ~have & (have << s) & 0xFFFFu;
        val |= (val << s) & fresh;
        have |= fresh;
    }
    const uint32_t seq = val, hdr = ~val & 0xFFFFu;
    fa_chunk r;
    const uint32_t seq_term = seq & ((gt & lt) | ate);
    r.name_end = hdr & nl & lt;
    r.term = seq_term | r.name_end;
    r.emit = r.term | (seq & lt & ~sp & ~gt) | (hdr & lt & ~(gt | hdrop | nl));
    r.eof_in_name = (hdr & ate) != 0;
    return r;
}

// Pass 1 of 2 (round 4: it was two passes): the last '\n' / '>' of every tile (for the max-scan that gives each tile its
// entry state), the first NUL byte (the reference reads a C string: io.c:15-18), the first '\n' of the image (where the
// first header line ends: the image starts in header state and a '>' keeps it there) -- and what the tile emits.  A
// thread's state is the kind of the last special byte before it; only the threads in front of the tile's first special
// byte -- four of them in a file of 60-column lines -- depend on the state the tile is entered in, and those are walked
// under both: tile_cnt[0] = what the other threads emit, [1] / [2] = what those threads add when the tile is entered in
// sequence / in header state (each: bytes emitted | terminators << 16).  The tile that holds the NUL counts up to it;
// the tiles behind it are never used.
__global__ __launch_bounds__(kBlock) void fasta_scan_kernel(const uint8_t *__restrict__ file, uint64_t end,
                                                            uint32_t *__restrict__ tile_last, uint32_t *__restrict__ tile_cnt /* [3][tiles] */,
                                                            uint32_t tiles_stride, uint32_t *__restrict__ scal /* [0] first NUL, [2] first newline */, uint32_t tiles)
{
    __shared__ uint32_t lds[kWavesPerBlock];
    uint4 raw[kFaSub];
    bool fast[kFaSub];
#pragma unroll
    for (int sub = 0; sub < kFaSub; ++sub) {
        raw[sub] = {0, 0, 0, 0};
        fast[sub] = fasta_fetch16(file, (((uint64_t)blockIdx.x * kFaSub + sub) * kBlock + threadIdx.x) * kFaPer, end, raw[sub]);
    }
#pragma unroll
    for (int sub = 0; sub < kFaSub; ++sub) {
    const uint32_t tile = blockIdx.x * (uint32_t)kFaSub + (uint32_t)sub;
    if (tile >= tiles) break; // uniform
    const uint64_t i0 = ((uint64_t)tile * kBlock + threadIdx.x) * kFaPer;
    const fa_masks m = fasta_masks16(file, i0, end, fast[sub], raw[sub]);
    const uint32_t in_image = fasta_before(i0, end);
    const uint32_t zero_at = (m.zero & in_image) ? (uint32_t)__ffs(m.zero & in_image) - 1u : (uint32_t)kFaPer;
    const uint32_t nl_at = (m.nl & in_image) ? (uint32_t)__ffs(m.nl & in_image) - 1u : (uint32_t)kFaPer;
    // the tile's first NUL ends the image for every thread of the tile (a NUL in an earlier tile: this tile is never used)
    const uint32_t my_nul = zero_at < (uint32_t)kFaPer ? (uint32_t)(i0 + zero_at) : 0xFFFFFFFFu;
    const uint32_t tile_nul = ~block_reduce<OpMax>(~my_nul, lds); // min as a max of complements
    if (threadIdx.x == 0 && tile_nul != 0xFFFFFFFFu) atomicMin(&scal[0], tile_nul);
    const uint64_t end_l = tile_nul < end ? (uint64_t)tile_nul : end;
    // (only a newline in front of the first one known so far is reported: one atomic a newline -- 18 million on one word for
    //  1 GiB of 60-column lines -- took 10 ms; after the image's first tiles have run, no thread has one to report)
    if (nl_at < (uint32_t)kFaPer && i0 + nl_at < end_l &&
        (uint32_t)(i0 + nl_at) < __hip_atomic_load(&scal[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMin(&scal[2], (uint32_t)(i0 + nl_at));
    uint32_t tot_last;
    const uint32_t before = block_exclusive_scan<OpMax>(fasta_last_special(m, i0, end_l), lds, tot_last);
    const bool known = before != 0;
    const fa_chunk c = fasta_walk(m, i0, end_l, known ? (before & 1u) != 0 : true);
    const uint32_t mine = (uint32_t)__popc(c.emit) | ((uint32_t)__popc(c.term) << 16);
    uint32_t other = 0;
    if (!known) { // (the first threads of the tile only)
        const fa_chunk h = fasta_walk(m, i0, end_l, false);
        other = (uint32_t)__popc(h.emit) | ((uint32_t)__popc(h.term) << 16);
    }
    const uint32_t s_known = block_reduce<OpAdd>(known ? mine : 0u, lds);
    const uint32_t s_seq = block_reduce<OpAdd>(known ? 0u : mine, lds);
    const uint32_t s_hdr = block_reduce<OpAdd>(other, lds);
    if (threadIdx.x == 0) {
        tile_last[tile] = tot_last;
        tile_cnt[tile] = s_known;
        tile_cnt[(uint64_t)tiles_stride + tile] = s_seq;
        tile_cnt[2ull * tiles_stride + tile] = s_hdr;
    }
    __syncthreads(); // (`lds` is the next tile's)
    }
}

// what tile i emits, now that its entry state is known (the max-scan's carry): bytes (shift 0) or terminators (shift 16)
struct InFaTileCount {
    const uint32_t *cnt, *carry;
    uint32_t stride, shift;
    __device__ __forceinline__ uint32_t operator()(uint64_t i) const
    {
        const uint32_t last = carry[i];
        const bool in_seq = last != 0 && (last & 1u);
        return ((cnt[i] + cnt[(in_seq ? (uint64_t)stride : 2ull * stride) + i]) >> shift) & 0xFFFFu;
    }
};

// state of every thread's first byte = kind of the last special byte before it (none: header)
__device__ __forceinline__ bool fasta_enter_state(const fa_masks &m, uint64_t i0, uint64_t end, uint32_t tile_carry,
                                                  uint32_t *lds)
{
    uint32_t tot;
    const uint32_t before = block_exclusive_scan<OpMax>(fasta_last_special(m, i0, end), lds, tot);
    const uint32_t last = before > tile_carry ? before : tile_carry;
    return last != 0 && (last & 1u);
}

// Pass 2: the walk again, now in the right state, and the packed image.  The tile's output is staged in LDS at the
// offset its first byte has inside a 16-byte piece of the destination, so that whole pieces leave as aligned 16-byte
// stores; only the two pieces the tile shares with its neighbours go byte by byte.  (A byte store per emitted byte,
// 16 store instructions a thread, each touching 64 lines: 1.6 ms per GiB, more than the two reading passes together.)
__global__ __launch_bounds__(kBlock) void fasta_write_kernel(const uint8_t *__restrict__ file, uint64_t end,
                                                             const uint32_t *__restrict__ tile_carry,
                                                             const uint32_t *__restrict__ tile_eoff,
                                                             const uint32_t *__restrict__ tile_toff, uint8_t *__restrict__ packed,
                                                             uint32_t *__restrict__ term_out, uint64_t term_cap,
                                                             uint32_t *__restrict__ scal /* [1] <- malformed; [2] first header end -> [0] its packed position */,
                                                             uint32_t tiles)
{
    __shared__ uint32_t lds[kWavesPerBlock];
    __shared__ __attribute__((aligned(16))) uint8_t stage[kFaTile + 48];
    uint4 raw[kFaSub];
    bool fast[kFaSub];
#pragma unroll
    for (int sub = 0; sub < kFaSub; ++sub) {
        raw[sub] = {0, 0, 0, 0};
        fast[sub] = fasta_fetch16(file, (((uint64_t)blockIdx.x * kFaSub + sub) * kBlock + threadIdx.x) * kFaPer, end, raw[sub]);
    }
#pragma unroll
    for (int sub = 0; sub < kFaSub; ++sub) {
    const uint32_t tile = blockIdx.x * (uint32_t)kFaSub + (uint32_t)sub;
    if (tile >= tiles) break; // uniform
    const uint64_t i0 = ((uint64_t)tile * kBlock + threadIdx.x) * kFaPer;
    uint32_t b[kFaPer];
    fasta_unpack16(file, i0, end, fast[sub], raw[sub], b);
    const fa_masks m = fasta_masks16(file, i0, end, fast[sub], raw[sub]);
    const bool in_seq = fasta_enter_state(m, i0, end, tile_carry[tile], lds);
    const fa_chunk c = fasta_walk(m, i0, end, in_seq);
    if (c.eof_in_name) atomicOr(&scal[1], 1u);
    uint32_t tot;
    const uint32_t ex = block_exclusive_scan<OpAdd>((uint32_t)__popc(c.emit) | ((uint32_t)__popc(c.term) << 16), lds, tot);
    const uint32_t eoff = tile_eoff[tile];
    const uint32_t sh = (uint32_t)((uintptr_t)(packed + eoff) & 15u); // the tile's first byte inside its 16-byte piece
    uint32_t at = sh + (ex & 0xFFFFu), out = eoff + (ex & 0xFFFFu), tq = tile_toff[tile] + (ex >> 16);
    const uint32_t first_end = scal[2];
#pragma unroll
    for (int k = 0; k < kFaPer; ++k) {
        if ((c.emit >> k) & 1u) {
            const bool is_term = (c.term >> k) & 1u;
            stage[at] = is_term ? (uint8_t)0 : (uint8_t)b[k];
            if (is_term) {
                if (term_out && tq < term_cap) term_out[tq] = out;
                ++tq;
                if (((c.name_end >> k) & 1u) && (uint32_t)(i0 + k) == first_end) scal[0] = out;
            }
            ++at;
            ++out;
        }
    }
    __syncthreads();
    const uint32_t total = tot & 0xFFFFu, last = sh + total; // staged bytes: [sh, last)
    uint8_t *dst = packed + eoff - sh;                        // 16-byte aligned
    for (uint32_t q = threadIdx.x; q * 16u < last; q += kBlock) {
        const uint32_t lo = q * 16u, hi = lo + 16u;
        if (lo >= sh && hi <= last) {
            *reinterpret_cast<uint4 *>(dst + lo) = *reinterpret_cast<const uint4 *>(stage + lo);
        } else {
            for (uint32_t e = lo < sh ? sh : lo; e < hi && e < last; ++e) dst[e] = stage[e];
        }
    }
    __syncthreads(); // (`stage` and `lds` are the next tile's)
    }
}

// which byte values occur: 256 flags as 8 words.  A workgroup marks the values it sees in a byte table in LDS with
// plain stores (every writer stores 1: the races are benign) and merges the table into the global flags once;
// keeping the flags in registers cost eight compares per byte for the static register indices (0.68 ms per GiB).
__global__ __launch_bounds__(kBlock) void remap_present_kernel(const uint8_t *__restrict__ in, uint64_t n,
                                                               uint32_t *__restrict__ present)
{
    __shared__ uint8_t seen[256];
    seen[threadIdx.x] = 0;
    __syncthreads();
    // (a record's sequence starts wherever its name ends in the packed image: 16-byte loads at any byte address,
    //  sx_device.hpp load_bytes16)
    for (uint64_t q = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q * 16 < n; q += (uint64_t)gridDim.x * kBlock) {
        if (q * 16 + 16 <= n) {
            uint32_t w[4];
            __builtin_memcpy(w, in + q * 16, 16);
#pragma unroll
            for (int k = 0; k < 16; ++k) seen[(w[k >> 2] >> (8 * (k & 3))) & 0xFFu] = 1;
        } else {
            for (uint64_t i = q * 16; i < n && i < q * 16 + 16; ++i) seen[in[i]] = 1;
        }
    }
    __syncthreads();
    if (seen[threadIdx.x]) atomicOr(&present[threadIdx.x >> 5], 1u << (threadIdx.x & 31u));
}

__global__ __launch_bounds__(kBlock) void remap_apply_kernel(const uint8_t *__restrict__ in, uint64_t n,
                                                             const uint8_t *__restrict__ table,
                                                             uint8_t *__restrict__ out)
{
    __shared__ uint8_t lut[256];
    lut[threadIdx.x] = table[threadIdx.x];
    __syncthreads();
    const bool aligned = ((uintptr_t)out & 15u) == 0; // (the input may start anywhere: unaligned 16-byte loads)
    for (uint64_t q = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q * 16 < n; q += (uint64_t)gridDim.x * kBlock) {
        if (aligned && q * 16 + 16 <= n) {
            uint32_t w[4];
            __builtin_memcpy(w, in + q * 16, 16);
            uint32_t o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] = (uint32_t)lut[w[k] & 0xFFu] | ((uint32_t)lut[(w[k] >> 8) & 0xFFu] << 8) |
                       ((uint32_t)lut[(w[k] >> 16) & 0xFFu] << 16) | ((uint32_t)lut[w[k] >> 24] << 24);
            uint4 r;
            r.x = o[0], r.y = o[1], r.z = o[2], r.w = o[3];
            *reinterpret_cast<uint4 *>(out + q * 16) = r;
        } else {
            for (uint64_t i = q * 16; i < n && i < q * 16 + 16; ++i) out[i] = lut[in[i]];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = 0; // the terminator remap() appends (remap.c:113)
}


// out[i] = in[n - 1 - i] (i < n), out[n] = 0: the reversed copy build_complete_table sorts for the RO table
// (stralg/bwt.c:147-151).  A thread turns 16 bytes around: one unaligned 16-byte load, two byte swaps, one 16-byte store.
__global__ __launch_bounds__(kBlock) void reverse_kernel(const uint8_t *__restrict__ in, uint64_t n, uint8_t *__restrict__ out)
{
    const uint64_t pieces = (n + 15) / 16;
    for (uint64_t q = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q < pieces; q += (uint64_t)gridDim.x * kBlock) {
        const uint64_t o0 = q * 16u; // out[o0 .. o0 + 16) = in[n - 16 - o0 .. n - o0) backwards
        if (o0 + 16u <= n && (((uintptr_t)out) & 15u) == 0) {
            uint64_t lo, hi;
            load_bytes16(in, n - 16u - o0, lo, hi);
            uint4 v;
            const uint64_t a = __builtin_bswap64(hi), b = __builtin_bswap64(lo);
            v.x = (uint32_t)a, v.y = (uint32_t)(a >> 32), v.z = (uint32_t)b, v.w = (uint32_t)(b >> 32);
            *reinterpret_cast<uint4 *>(out + o0) = v;
        } else {
            for (uint64_t i = o0; i < n && i < o0 + 16u; ++i) out[i] = in[n - 1u - i];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = 0;
}

} // namespace sx

using namespace sx;

extern "C" {

int sx_fasta_pack_dev(sx_ctx *ctx, const uint8_t *d_file, uint64_t file_len, uint8_t *d_packed_out,
                      uint64_t *packed_len_out, uint32_t *d_term_out, uint64_t term_cap, uint32_t *n_records_out)
{
    if (!ctx || !d_packed_out || !packed_len_out || !n_records_out || (file_len && !d_file)) return SX_E_ARG;
    if (file_len >= 0x7FFFFFFFull) return sx_fail_msg(ctx, SX_E_ARG, "FASTA image must be shorter than 2^31 - 1 bytes");
    SX_CHECK(hipSetDevice(ctx->device));
    *packed_len_out = 0;
    *n_records_out = 0;
    // scratch: a few scalars and eight u32 per 4096-byte tile
    const uint64_t span_max = file_len + 1;
    const uint32_t tiles_max = sx_div_up(span_max, kFaTile);
    SX_TRY(sx_slab_ensure(ctx, SX_SLAB_N, 256 + (size_t)8 * tiles_max * sizeof(uint32_t) + 1024));
    uint32_t *scal = (uint32_t *)ctx->slab[SX_SLAB_N].p; // [0] first NUL / packed position, [1] malformed, [2] first header end
    uint32_t *tile_last = scal + 64, *tile_carry = tile_last + tiles_max, *tile_cnt = tile_carry + tiles_max,
             *tile_eoff = tile_cnt + 3 * (size_t)tiles_max, *tile_toff = tile_eoff + tiles_max;
    const uint32_t init[4] = {0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u};
    SX_CHECK(hipMemcpyAsync(scal, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
    // pass 1 over the whole image: last '\n' / '>' of every tile, the first NUL (which ends the image), the first newline,
    // and every tile's emitted bytes and terminators for either entry state
    uint32_t first_nul = 0xFFFFFFFFu;
    const uint32_t tiles_file = sx_div_up(file_len ? file_len : 1, kFaTile);
    if (file_len) {
        sx_launch(ctx, SX_KC_FASTA, file_len, fasta_scan_kernel, dim3(sx_div_up(tiles_file, kFaSub)), dim3(kBlock), d_file, file_len, tile_last,
                  tile_cnt, tiles_max, scal, tiles_file);
        SX_TRY(sx_readback(ctx, scal, 1, &first_nul));
    }
    const uint64_t end = first_nul < file_len ? first_nul : file_len;
    const uint64_t span = end + 1; // with the terminating NUL of the reference's buffer
    const uint32_t tiles = sx_div_up(span, kFaTile);
    const dim3 grid(sx_div_up(tiles, kFaSub)), block(kBlock);
    if (!file_len || tiles > tiles_file) {
        // the span's last tile lies behind the image: it holds the terminator of the reference's buffer only -- a sequence's
        // terminator when the tile is entered in sequence state, nothing to emit in a header (MALFORMED, found by pass 2)
        const uint32_t t = tiles - 1;
        const uint32_t zero = 0, seq_term = 1u | (1u << 16);
        SX_CHECK(hipMemcpyAsync(tile_last + t, &zero, 4, hipMemcpyHostToDevice, ctx->stream));
        SX_CHECK(hipMemcpyAsync(tile_cnt + t, &zero, 4, hipMemcpyHostToDevice, ctx->stream));
        SX_CHECK(hipMemcpyAsync(tile_cnt + (size_t)tiles_max + t, &seq_term, 4, hipMemcpyHostToDevice, ctx->stream));
        SX_CHECK(hipMemcpyAsync(tile_cnt + 2 * (size_t)tiles_max + t, &zero, 4, hipMemcpyHostToDevice, ctx->stream));
    }
    // (the last tile's own entry is never used by the exclusive scan, so the tile that holds the NUL needs no second look
    //  although its last-special word saw the bytes behind the NUL too -- it did not: the scan kernel ends a tile at its NUL)
    SX_TRY((device_scan<OpMax>(ctx, tiles, InU32{tile_last}, OutExclusive{tile_carry}, nullptr, SX_KC_FASTA, 0)));
    SX_TRY((device_scan<OpAdd>(ctx, tiles, InFaTileCount{tile_cnt, tile_carry, tiles_max, 0u}, OutExclusive{tile_eoff}, scal + 3, SX_KC_FASTA, 0)));
    SX_TRY((device_scan<OpAdd>(ctx, tiles, InFaTileCount{tile_cnt, tile_carry, tiles_max, 16u}, OutExclusive{tile_toff}, scal + 4, SX_KC_FASTA, 0)));
    sx_launch(ctx, SX_KC_FASTA, span * 2, fasta_write_kernel, grid, block, d_file, end, (const uint32_t *)tile_carry,
              (const uint32_t *)tile_eoff, (const uint32_t *)tile_toff, d_packed_out, d_term_out, d_term_out ? term_cap : 0, scal, tiles);
    uint32_t h[5];
    SX_TRY(sx_readback(ctx, scal, 5, h));
    *packed_len_out = h[3];
    const uint32_t n_term = h[4];
    *n_records_out = n_term / 2;
    // MALFORMED_FILE (fasta.c:121-124): the image ends inside a header line -- or, because the reference packs in
    // place, the first header line had nothing to drop, so that its terminator overwrote the newline being examined
    const bool clobbered = h[2] != 0xFFFFFFFFu && h[0] == h[2];
    if (h[1] || clobbered) return sx_fail_msg(ctx, SX_E_MALFORMED, "FASTA image ends inside a header line");
    return 0;
}

int sx_fasta_pack(sx_ctx *ctx, const uint8_t *file, uint64_t file_len, uint8_t *packed_out, uint64_t *packed_len_out,
                  uint32_t *term_out, uint64_t term_cap, uint32_t *n_records_out)
{
    if (!ctx || !packed_out || !packed_len_out || !n_records_out || (file_len && !file)) return SX_E_ARG;
    if (file_len >= 0x7FFFFFFFull) return sx_fail_msg(ctx, SX_E_ARG, "FASTA image must be shorter than 2^31 - 1 bytes");
    SX_CHECK(hipSetDevice(ctx->device));
    const size_t file_b = (file_len + 256) & ~(size_t)255, term_b = (term_out ? term_cap * 4 + 255 : 0) & ~(size_t)255;
    SX_TRY(sx_slab_ensure(ctx, SX_SLAB_IO, 2 * file_b + term_b + 1024));
    uint8_t *d_file = (uint8_t *)ctx->slab[SX_SLAB_IO].p, *d_packed = d_file + file_b;
    uint32_t *d_term = term_out ? (uint32_t *)(d_packed + file_b) : nullptr;
    if (file_len) SX_CHECK(hipMemcpyAsync(d_file, file, file_len, hipMemcpyHostToDevice, ctx->stream));
    const int rc = sx_fasta_pack_dev(ctx, d_file, file_len, d_packed, packed_len_out, d_term, term_cap, n_records_out);
    if (rc != 0 && rc != SX_E_MALFORMED) return rc;
    if (*packed_len_out) SX_CHECK(hipMemcpyAsync(packed_out, d_packed, *packed_len_out, hipMemcpyDeviceToHost, ctx->stream));
    if (term_out) {
        const uint64_t nt = 2ull * *n_records_out < term_cap ? 2ull * *n_records_out : term_cap;
        if (nt) SX_CHECK(hipMemcpyAsync(term_out, d_term, nt * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    SX_TRY(sx_sync(ctx));
    return rc;
}

int sx_remap_dev(sx_ctx *ctx, const uint8_t *d_in, uint64_t n, uint8_t *d_out, int16_t *table_out,
                 uint32_t *alphabet_size_out)
{
    if (!ctx || !d_out || !alphabet_size_out || (n && !d_in)) return SX_E_ARG;
    SX_CHECK(hipSetDevice(ctx->device));
    SX_TRY(sx_slab_ensure(ctx, SX_SLAB_SCAN, 1024));
    uint32_t *present = (uint32_t *)ctx->slab[SX_SLAB_SCAN].p;
    uint8_t *lut = (uint8_t *)(present + 64);
    SX_CHECK(hipMemsetAsync(present, 0, 8 * sizeof(uint32_t), ctx->stream));
    uint32_t grid = sx_div_up(n ? n : 1, kBlock * 64);
    if (grid > 2048) grid = 2048;
    if (n) sx_launch(ctx, SX_KC_REMAP, n, remap_present_kernel, dim3(grid), dim3(kBlock), d_in, n, present);
    uint32_t h[8];
    SX_TRY(sx_readback(ctx, present, 8, h));
    // remap.c:8-31: symbols in increasing order get 1, 2, ...; 0 stays the sentinel
    uint8_t table[256];
    int16_t t16[256];
    uint32_t next = 1;
    table[0] = 0;
    t16[0] = 0;
    for (int c = 1; c < 256; ++c) {
        const bool seen = (h[c >> 5] >> (c & 31)) & 1u;
        t16[c] = seen ? (int16_t)next : (int16_t)-1;
        table[c] = seen ? (uint8_t)next : (uint8_t)0;
        if (seen) ++next;
    }
    if (table_out) memcpy(table_out, t16, sizeof t16);
    *alphabet_size_out = next;
    if (h[0] & 1u) return sx_fail_msg(ctx, SX_E_ARG, "remap: the input holds the sentinel symbol 0");
    if (next > 128) return sx_fail_msg(ctx, SX_E_ARG, "remap: more than 127 distinct symbols (stralg/remap.h:14-18)");
    SX_CHECK(hipMemcpyAsync(lut, table, 256, hipMemcpyHostToDevice, ctx->stream));
    sx_launch(ctx, SX_KC_REMAP, 2 * n, remap_apply_kernel, dim3(grid), dim3(kBlock), d_in, n, (const uint8_t *)lut, d_out);
    return sx_sync(ctx);
}


int sx_reverse_dev(sx_ctx *ctx, const uint8_t *d_in, uint64_t n, uint8_t *d_out)
{
    if (!ctx || !d_out || (n && !d_in)) return SX_E_ARG;
    // (the two buffers must not overlap: every byte moves)
    if (n && d_in < d_out + n + 1 && d_out < d_in + n) return sx_fail_msg(ctx, SX_E_ARG, "reverse: the buffers overlap");
    SX_CHECK(hipSetDevice(ctx->device));
    uint32_t grid = sx_div_up(sx_div_up(n ? n : 1, 16), kBlock);
    if (grid > 16384) grid = 16384;
    sx_launch(ctx, SX_KC_REMAP, 2 * n, reverse_kernel, dim3(grid), dim3(kBlock), d_in, n, d_out);
    return sx_sync(ctx);
}

} // extern "C"
