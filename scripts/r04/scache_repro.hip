// scache_repro.hip — does the scalar data cache serve a line that an EARLIER operation of the same stream has since rewritten?
//
// r03 verdict, item 4a.  r03 explained a 1-in-30,000 wrong bitmap word in scan_wide_kernel's append path by "the scalar data cache is
// not invalidated between the kernels of a stream".  This program tests exactly that, outside the library:
//     writer (value v)  ->  reader kernel: every wave s_loads words of the buffer and compares them with v
// repeated N times on one stream with v changing every iteration, so that every reader finds the PREVIOUS value's lines in whatever
// cache was not invalidated.  Writers: a kernel (vector stores), hipMemcpyAsync from pinned memory, hipMemcpyAsync from pageable
// memory (what fsgpu_search_topk_batched does with the caller's allow bitmap), a blocking hipMemcpy on the null stream while the
// readers use a non-blocking stream (what fsgpu_index_set_live_bitmap does), a kernel on ANOTHER stream ordered by an event.
// Readers: s_load_dwordx2 / s_load_dwordx2 glc / s_dcache_inv + s_load_dwordx2 / global_load_dwordx2 (control).
//
//   hipcc --offload-arch=gfx950 -O2 -o scache_repro scache_repro.hip && ./scache_repro [iterations, default 1000000]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned long long u64;

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            std::exit(2);                                                                      \
        }                                                                                      \
    } while (0)

__host__ __device__ inline u64 word_of(u64 v, u64 i) { return v * 0x9e3779b97f4a7c15ull + i; }

__global__ void writer_kernel(u64* buf, u64 n, u64 v) {
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) buf[i] = word_of(v, i);
}

// MODE 0 s_load, 1 s_load glc, 2 s_dcache_inv first, 3 vector load
template <int MODE>
__global__ void reader_kernel(const u64* buf, u64 n, u64 v, unsigned* bad, u64* first_bad) {
    const unsigned wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64);
    if (MODE == 2) asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {   // four words per wave: two cache lines apart, so that several lines per CU get cached
        const u64 i = ((u64)wave * 97u + (u64)r * 1031u) % n;
        const u64* p = buf + i;
        u64 w;
        if (MODE == 0 || MODE == 2) asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(w) : "s"(p) : "memory");
        else if (MODE == 1) asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(w) : "s"(p) : "memory");
        else asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(w) : "v"(p) : "memory");
        if (w != word_of(v, i) && (threadIdx.x & 63) == 0) {
            if (atomicAdd(bad, 1u) == 0) {
                first_bad[0] = v;
                first_bad[1] = i;
                first_bad[2] = w;
            }
        }
    }
}

enum Writer { W_KERNEL = 0, W_H2D_PINNED, W_H2D_PAGEABLE, W_NULL_STREAM_BLOCKING, W_KERNEL_OTHER_STREAM, W_COUNT };
static const char* writer_name[] = {"kernel (vector stores), same stream", "hipMemcpyAsync from pinned host memory, same stream",
                                    "hipMemcpyAsync from pageable host memory, same stream",
                                    "blocking hipMemcpy on the null stream, reader on a non-blocking stream",
                                    "kernel on another stream, ordered by an event"};
static const char* reader_name[] = {"s_load_dwordx2", "s_load_dwordx2 glc", "s_dcache_inv + s_load_dwordx2", "global_load_dwordx2 (control)"};

static void launch_reader(int mode, const u64* buf, u64 n, u64 v, unsigned* bad, u64* first, hipStream_t s) {
    const dim3 g(1024), b(256);   // 4,096 waves: every CU's scalar cache sees the buffer
    switch (mode) {
        case 0: hipLaunchKernelGGL(reader_kernel<0>, g, b, 0, s, buf, n, v, bad, first); break;
        case 1: hipLaunchKernelGGL(reader_kernel<1>, g, b, 0, s, buf, n, v, bad, first); break;
        case 2: hipLaunchKernelGGL(reader_kernel<2>, g, b, 0, s, buf, n, v, bad, first); break;
        default: hipLaunchKernelGGL(reader_kernel<3>, g, b, 0, s, buf, n, v, bad, first); break;
    }
}

int main(int argc, char** argv) {
    const long iters_full = argc > 1 ? std::atol(argv[1]) : 1000000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    std::printf("device: %s (%s), %d CUs; %ld iterations for the kernel / pinned-copy writers, %ld for the others\n", prop.name,
                prop.gcnArchName, prop.multiProcessorCount, iters_full, iters_full / 5);
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ev, ev2;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
    unsigned* bad;
    u64* first;
    CK(hipMalloc(&bad, 4));
    CK(hipMalloc(&first, 24));
    long total_bad = 0;
    for (u64 n : {(u64)16, (u64)156250}) {   // 128 bytes; 1.25 MB (the allow bitmap of a 10M-row index)
        u64* buf;
        CK(hipMalloc(&buf, n * 8));
        u64 *pinned[2], *pageable[2];
        for (int i = 0; i < 2; ++i) {
            CK(hipHostMalloc(&pinned[i], n * 8));
            pageable[i] = static_cast<u64*>(std::malloc(n * 8));
        }
        for (int w = 0; w < W_COUNT; ++w) {
            for (int mode = 0; mode < 4; ++mode) {
                const bool copy_writer = w != W_KERNEL && w != W_KERNEL_OTHER_STREAM;
                long iters = (w == W_KERNEL || w == W_H2D_PINNED) && mode < 2 ? iters_full : iters_full / 5;
                if (n > 16 && copy_writer) iters = iters / 4;   // (the host fills 1.25 MB per iteration)
                CK(hipMemsetAsync(bad, 0, 4, s));
                CK(hipStreamSynchronize(s));
                const auto t0 = std::chrono::steady_clock::now();
                for (long it = 0; it < iters; ++it) {
                    const u64 v = (u64)it + 1;
                    const int hb = (int)(it & 1);
                    if (copy_writer) {
                        u64* src = w == W_H2D_PINNED ? pinned[hb] : pageable[hb];
                        // the previous copy out of this host buffer must be over before it is refilled (two buffers: wait every other iteration)
                        if (w != W_NULL_STREAM_BLOCKING && it >= 2 && hb == 0) CK(hipStreamSynchronize(s));
                        for (u64 i = 0; i < n; ++i) src[i] = word_of(v, i);
                        if (w == W_NULL_STREAM_BLOCKING) CK(hipMemcpy(buf, src, n * 8, hipMemcpyHostToDevice));
                        else CK(hipMemcpyAsync(buf, src, n * 8, hipMemcpyHostToDevice, s));
                    } else if (w == W_KERNEL) {
                        hipLaunchKernelGGL(writer_kernel, dim3(64), dim3(256), 0, s, buf, n, v);
                    } else {
                        CK(hipStreamWaitEvent(s2, ev2, 0));   // the previous reader is done with the buffer
                        hipLaunchKernelGGL(writer_kernel, dim3(64), dim3(256), 0, s2, buf, n, v);
                        CK(hipEventRecord(ev, s2));
                        CK(hipStreamWaitEvent(s, ev, 0));
                    }
                    launch_reader(mode, buf, n, v, bad, first, s);
                    if (w == W_KERNEL_OTHER_STREAM) CK(hipEventRecord(ev2, s));
                    if (w == W_NULL_STREAM_BLOCKING) CK(hipStreamSynchronize(s));   // (the next blocking copy must not overtake the reader)
                    if ((it & 1023) == 1023) CK(hipStreamSynchronize(s));
                }
                CK(hipStreamSynchronize(s));
                CK(hipStreamSynchronize(s2));
                const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                unsigned hb = 0;
                u64 hf[3] = {0, 0, 0};
                CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hf, first, 24, hipMemcpyDeviceToHost));
                total_bad += hb;
                std::printf("buffer %8llu B | writer: %-72s | reader: %-30s | %8ld iterations, %5.1f s | stale words: %u", n * 8, writer_name[w],
                            reader_name[mode], iters, sec, hb);
                if (hb) std::printf("  (first: iteration %llu word %llu read 0x%llx want 0x%llx)", hf[0], hf[1], hf[2], word_of(hf[0], hf[1]));
                std::printf("\n");
                std::fflush(stdout);
            }
        }
        CK(hipFree(buf));
        for (int i = 0; i < 2; ++i) {
            CK(hipHostFree(pinned[i]));
            std::free(pageable[i]);
        }
    }
    std::printf("total stale words: %ld\n", total_bad);
    return 0;
}
