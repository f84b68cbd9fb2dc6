// Micro-benchmark: LDS-DMA (global_load_lds_dwordx4) rate of a CU for the GEMM operand pattern -- one wave-instruction = 8 rows x 128 B,
// the rows `rowstride` bytes apart (rowstride = 128: the 8 rows are one contiguous 1-KB block, i.e. a K-tile-major operand layout;
// 2560 / 10240: row-major operands with K = 1280 / 5120) -- against each other, L2-resident (one 224-row panel shared by every
// workgroup, as the CUs of an XCD share a weight panel).  Every workgroup = 4 loader waves, D instructions in flight per wave, no
// consumers.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/glds_stride.hip -o tools/micro/glds_stride
// Output: CSV  rowstride, wg/CU, D, GB/s aggregate, GB/s per CU, cycles per 128-B line per CU (at 2.1 GHz)
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

constexpr int ROWS = 224;          // 64 + 160: one K tile of the 64 x 160 workgroup = 28 wave-instructions
constexpr int GROUPS = ROWS / 8;

// NWV loader waves per workgroup (4: the kernels' producer group; 7 / 14 / 28: more issuing waves, same bytes in flight per CU when
// D is scaled down); stagger: workgroup b starts at K tile (b % stagger) -- CUs that share the panel are not on the same lines at
// the same time
template <int D, int NWV>
__global__ __launch_bounds__(64 * NWV) void k(const unsigned char* src, size_t rowstride, size_t tilestride, int ktiles, int iters, int stagger) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* ldsw = smem + wave * D * 1024;
    int g = wave, kt = stagger > 1 ? (int)(blockIdx.x % stagger) % ktiles : 0;      // wave w takes row groups w, w + NWV, ... of every K tile
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const unsigned char* p = src + ((size_t)(g * 8 + (lane >> 3))) * rowstride + (size_t)kt * tilestride + (lane & 7) * 16;
            glds16(p, ldsw + d * 1024);
            g += NWV;
            if (g >= GROUPS) { g = wave; if (++kt == ktiles) kt = 0; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

template <int D, int NWV = 4>
double run(const unsigned char* src, size_t rowstride, size_t tilestride, int ktiles, int wgs, int iters, int stagger = 1) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = (size_t)NWV * D * 1024;
    hipFuncSetAttribute((const void*)k<D, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    k<D, NWV><<<wgs, 64 * NWV, lds>>>(src, rowstride, tilestride, ktiles, iters, stagger);
    hipEventRecord(e0);
    k<D, NWV><<<wgs, 64 * NWV, lds>>>(src, rowstride, tilestride, ktiles, iters, stagger);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)wgs * NWV * iters * D * 1024.0 / (ms * 1e-3) / 1e9;
}

int main() {
    const size_t total = 64ull << 20;
    unsigned char* buf; hipMalloc(&buf, total); hipMemset(buf, 1, total);
    printf("rowstride_bytes, ktiles, wg_per_cu, D, GBps_aggregate, GBps_per_cu, cycles_per_line_per_cu_at_2.1GHz\n");
    // (rowstride, ktiles): the panel is ROWS x ktiles x 128 B; contiguous: ktiles tiles of ROWS x 128 B back to back
    const size_t strides[] = {128, 2560, 10240, 10240 + 128, 2560 + 128};
    for (size_t rs : strides) {
        // 20 K tiles either way (560 KB: misses the 32-KB L1, sits in L2); rs = 128 = K-tile-major: tile kt is the contiguous block kt
        const int ktiles = 20;
        const size_t ts = rs == 128 ? (size_t)ROWS * 128 : 128;
        for (int wgpc : {1, 2}) {
            const int wgs = 256 * wgpc;
#define RUN(DD) do { const int iters = 4200 / DD; double a = run<DD>(buf, rs, ts, ktiles, wgs, iters); \
            printf("%zu, %d, %d, %d, %.0f, %.1f, %.2f\n", rs, ktiles, wgpc, DD, a, a / 256, 128.0 / (a / 256 / 2.1)); } while (0)
            RUN(7); RUN(14); RUN(21);
        }
    }
    // second table: more issuing waves in ONE workgroup per CU (bytes in flight per CU held at 84-112 KB), and a K stagger between workgroups
    printf("\nrowstride_bytes, waves_per_wg, D, stagger_tiles, GBps_aggregate, GBps_per_cu\n");
    {
        const size_t rs = 10240, ts = 128; const int ktiles = 20, wgs = 256;
#define RUNW(DD, NW, ST) do { const int iters = 16800 / (DD * NW); double a = run<DD, NW>(buf, rs, ts, ktiles, wgs, iters, ST); \
        printf("%zu, %d, %d, %d, %.0f, %.1f\n", rs, NW, DD, ST, a, a / 256); } while (0)
        RUNW(21, 4, 1); RUNW(21, 4, 2); RUNW(21, 4, 4); RUNW(21, 4, 8); RUNW(21, 4, 20);
        RUNW(12, 7, 1); RUNW(6, 14, 1); RUNW(3, 28, 1); RUNW(12, 7, 4); RUNW(6, 14, 4);
    }
    return 0;
}
