// sx_pager.hpp -- large downloads into pageable host memory.
//
// The reference's callers own malloc'd result arrays (stralg/bwt.c:134-161: 4 bytes per suffix and 4 sigma bytes per
// O row, 24 GiB for a 1 GiB text), usually fresh from malloc: every 4 KiB page of them is first touched by the
// copy that fills it, and a device-to-host copy into untouched pages runs at the kernel's single-threaded page
// fault rate (a few GB/s) instead of the PCIe rate (52 GB/s measured into touched pages).  The pager touches the
// pages of the output buffers with a few host threads, in the order the copies will want them, while the GPU is
// still building; the copies go chunk by chunk, each as soon as its pages are there.
#pragma once
#include "sx_common.hpp"

#include <atomic>
#include <memory>
#include <thread>
#include <vector>

#include <stdlib.h>
#include <unistd.h>

struct sx_host_pager {
    struct chunk {
        char *p;
        size_t bytes;
    };
    size_t chunk_bytes = (size_t)64 << 20;
    size_t min_bytes = (size_t)32 << 20; // smaller buffers are not worth a thread

    sx_host_pager()
    {
        // $STRALG_AMD_PARALLEL_MIN (tests): the threaded, chunked path for small buffers too
        if (const char *env = getenv("STRALG_AMD_PARALLEL_MIN")) {
            const long v = atol(env);
            if (v >= 1) {
                min_bytes = (size_t)v;
                chunk_bytes = (size_t)v < 4096 ? 4096 : ((size_t)v + 4095) & ~(size_t)4095;
            }
        }
    }

    std::vector<chunk> chunks;
    std::unique_ptr<std::atomic<unsigned char>[]> ready;
    std::atomic<size_t> next{0};
    std::vector<std::thread> workers;

    static int thread_count()
    {
        if (const char *env = getenv("STRALG_AMD_HOST_THREADS")) {
            const int v = atoi(env);
            if (v >= 1) return v > 64 ? 64 : v;
        }
        const long cpus = sysconf(_SC_NPROCESSORS_ONLN);
        return cpus >= 32 ? 16 : (cpus >= 4 ? (int)(cpus / 2) : 1);
    }

    // queue a buffer; returns the index of its first chunk (the buffer's chunks are consecutive), chunk boundaries
    // inside the buffer lie on page boundaries
    size_t add(void *dst, size_t bytes)
    {
        const size_t first = chunks.size();
        char *p = (char *)dst, *end = p + bytes;
        while (p < end) {
            char *stop = (char *)(((uintptr_t)p + chunk_bytes) & ~(uintptr_t)4095);
            if (stop > end || (size_t)(end - stop) < 4096) stop = end;
            chunks.push_back({p, (size_t)(stop - p)});
            p = stop;
        }
        return first;
    }

    void start()
    {
        size_t total = 0;
        for (const chunk &c : chunks) total += c.bytes;
        ready.reset(new std::atomic<unsigned char>[chunks.size() ? chunks.size() : 1]);
        for (size_t i = 0; i < chunks.size(); ++i) ready[i].store(0, std::memory_order_relaxed);
        if (total < min_bytes) { // nothing to gain: the copies touch the pages themselves
            for (size_t i = 0; i < chunks.size(); ++i) ready[i].store(1, std::memory_order_relaxed);
            return;
        }
        int nt = thread_count();
        if ((size_t)nt > chunks.size()) nt = (int)chunks.size();
        // Thread creation may fail (EAGAIN under a thread limit; a farm has a pager per GPU): an exception must not leave
        // through the C ABI.  The chunks are a shared queue, so the workers that did start touch all of them; with no
        // worker at all every chunk is marked ready and the copies fault the pages in themselves -- download() never
        // waits for a chunk nobody will complete.
        try {
            for (int t = 0; t < nt; ++t)
                workers.emplace_back([this] {
                for (;;) {
                    const size_t i = next.fetch_add(1, std::memory_order_relaxed);
                    if (i >= chunks.size()) return;
                    // one write per page: the buffers are outputs about to be overwritten
                    volatile char *p = chunks[i].p;
                    const size_t bytes = chunks[i].bytes;
                    for (size_t off = 0; off < bytes; off += 4096) p[off] = 0;
                    if (bytes) p[bytes - 1] = 0;
                    ready[i].store(1, std::memory_order_release);
                }
            });
        } catch (...) {
            if (workers.empty())
                for (size_t i = 0; i < chunks.size(); ++i) ready[i].store(1, std::memory_order_release);
        }
    }

    void wait(size_t i) const
    {
        while (!ready[i].load(std::memory_order_acquire)) std::this_thread::yield();
    }

    // device -> host copy of a queued buffer, chunk by chunk as the pages arrive
    int download(sx_ctx *ctx, size_t first_chunk, void *dst, const void *d_src, size_t bytes)
    {
        size_t off = 0;
        for (size_t i = first_chunk; off < bytes; ++i) {
            wait(i);
            const size_t len = chunks[i].bytes;
            SX_CHECK(hipMemcpyAsync((char *)dst + off, (const char *)d_src + off, len, hipMemcpyDeviceToHost, ctx->stream));
            off += len;
        }
        return 0;
    }

    void join()
    {
        for (std::thread &t : workers)
            if (t.joinable()) t.join();
        workers.clear();
    }
    ~sx_host_pager() { join(); }
};
