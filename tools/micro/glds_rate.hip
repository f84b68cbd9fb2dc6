// Micro-benchmark: sustained global -> LDS (LDS-DMA, global_load_lds_dwordx4) and global -> VGPR (global_load_dwordx4)
// rate of one CU as a function of waves per CU and instructions in flight per wave, for L2-resident and for streaming
// (every line touched once) sources.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/glds_rate.hip -o tools/micro/glds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int D, bool TOLDS>
__global__ __launch_bounds__(256) void k(const unsigned char* src, size_t window, size_t stride_wg, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* base = src + (size_t)blockIdx.x * stride_wg + wave * (window / 4);
    const size_t wwin = window / 4;               // per-wave window
    unsigned char* ldsw = smem + wave * D * 1024;
    float acc = 0.f;
    size_t off = 0;
    typedef __attribute__((ext_vector_type(4))) float f4;
    f4 r[D];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const unsigned char* p = base + off + lane * 16;
            if (TOLDS) glds16(p, ldsw + d * 1024);
            else r[d] = *(const f4*)p;
            off += 1024;
            if (off >= wwin) off = 0;
        }
        if (TOLDS) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        else {
#pragma unroll
            for (int d = 0; d < D; ++d) acc += r[d][0];
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int D, bool TOLDS>
double run(const unsigned char* src, size_t window, size_t stride, int wgs, int iters, float* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = 4 * D * 1024;
    hipFuncSetAttribute((const void*)k<D, TOLDS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    k<D, TOLDS><<<wgs, 256, lds>>>(src, window, stride, iters, sink);
    hipEventRecord(e0);
    k<D, TOLDS><<<wgs, 256, lds>>>(src, window, stride, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)wgs * 4 * iters * D * 1024.0;
    return bytes / (ms * 1e-3) / 1e9;     // GB/s aggregate
}

int main(int argc, char**) {
    const size_t total = 1ull << 30;
    unsigned char* buf; hipMalloc(&buf, total); hipMemset(buf, 1, total);
    float* sink; hipMalloc(&sink, 4);
    printf("mode, src, wg/CU, D(in flight per wave), GB/s aggregate, GB/s per CU\n");
    if (argc > 1) {
        // L2 rates (round 3): windows that MISS the 32 KB vector L1 and HIT the 4 MB L2 of the XCD -- private 64 KB per
        // workgroup (32 CUs x 64 KB = 2 MB per XCD) and one 1 MB window shared by every workgroup (the GEMM pattern:
        // all CUs of an XCD stream the same operand panel)
        printf("mode, src, wg/CU, D(in flight per wave), GB/s aggregate, GB/s per CU\n");
        for (int shared = 0; shared <= 1; ++shared)
            for (int wgpc : {1, 2}) {
                const int wgs = 256 * wgpc;
                const size_t window = shared ? (1u << 20) : (64u << 10);
                const size_t stride = shared ? 0 : (64u << 10);
#define RUN2(DD) do { const int iters = 4000 / DD; double a = run<DD, true>(buf, window, stride, wgs, iters, sink); \
                printf("LDS-DMA, %s, %d, %d, %.0f, %.1f\n", shared ? "L2 shared 1 MB window" : "L2 private 64 KB windows", wgpc, DD, a, a / 256); } while (0)
                RUN2(4); RUN2(8); RUN2(16);
            }
        return 0;
    }
    for (int resident = 1; resident >= 0; --resident) {
        for (int wgpc : {1, 2, 4}) {
            const int wgs = 256 * wgpc;
            // "L2-resident" (label kept for the committed CSV): every WG cycles a private 16 KB window -- which fits the CU's
            // 32 KB vector L1, so these rows are L1-HIT rates; a true L2 figure needs a window above 32 KB per CU
            const size_t window = resident ? 16 * 1024 : (total / wgs);
            const size_t stride = resident ? 16 * 1024 : (total / wgs);
            const int iters_base = resident ? 4000 : 0;
#define RUN(DD) do { const int iters = resident ? iters_base / DD : (int)(window / 4 / 1024 / DD); \
                double a = run<DD, true>(buf, window, stride, wgs, iters, sink); double b = run<DD, false>(buf, window, stride, wgs, iters, sink); \
                printf("LDS-DMA, %s, %d, %d, %.0f, %.1f\n", resident ? "L2-resident" : "streaming", wgpc, DD, a, a / 256); \
                printf("to-VGPR, %s, %d, %d, %.0f, %.1f\n", resident ? "L2-resident" : "streaming", wgpc, DD, b, b / 256); } while (0)
            RUN(4); RUN(8); RUN(16);
        }
    }
    return 0;
}
