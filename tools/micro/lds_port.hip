// Micro-benchmark (round 5): does the K tile of the wave-specialised 64 x 160 GEMM cost
//     LDS-DMA writes (64 B / clk) + fragment reads (256 B / clk)  IN SERIES on the LDS port,
// and do plain global_load_dwordx4 -> VGPR loads share the 64 B / clk vector-memory address unit with the LDS-DMA stream?
// One workgroup per CU with the kernel's structure (4 consumer waves + NP producer waves, S-slot LDS ring, one s_barrier per K tile,
// counted vmcnt), each role reduced to its instruction mix per K tile:
//     producer wave : NG  global_load_lds_dwordx4 (1 KB each) of L2-resident lines
//     consumer wave : NR  ds_read_b128 (conflict-free, 1 KB each)  +  NV  global_load_dwordx4 -> VGPR (1 KB contiguous each, three
//                     tiles in flight)  +  NM  v_mfma_f32_16x16x32_bf16
// Rows of the output CSV = instruction mixes; `current` is the 2464 x 160 kernel's K tile (28 DMA + 4 x 14 reads + 4 x 20 MFMA),
// `proposed` the weights-straight-to-VGPR form (8 DMA + 4 x 4 reads + 4 x 5 VGPR loads + 4 x 20 MFMA).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/lds_port.hip -o tools/micro/lds_port
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int S = 4;                 // ring slots
constexpr int SLOT = 28 * 1024;      // bytes per slot (the 64 x 160 kernel's stage)
constexpr int XWIN = 160 * 1024;     // token-row window of an m tile (64 rows x 1280 k x 2 B)
constexpr int WWIN = 400 * 1024;     // weight window of an n tile (160 rows x 1280 k x 2 B)

template <int NP, int NG, int NR, int NV, int NM, int PM = 0, bool IL = false>
__global__ __launch_bounds__(64 * (4 + NP), 1) void k(const unsigned char* xsrc, const unsigned char* wsrc, int tiles,
                                                      unsigned long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* xw = xsrc + (size_t)(blockIdx.x / 8) * XWIN;     // m tile = b / 8, n tile = b % 8 (one n tile per XCD)
    const unsigned char* ww = wsrc + (size_t)(blockIdx.x % 8) * WWIN;

    if (wave >= 4) {
        // ---------------------------------------------------------------- producer
        const int pw = wave - 4;
        // the DMA stream of a tile = (NP * NG) KB: the first 8 KB from the token window, the rest from the weight window
        unsigned xo = 0, wo = 0;                  // running window offsets (wave-uniform), wrapped without a division
        constexpr unsigned WSTEP = (NP * NG > 8 ? NP * NG - 8 : 0) * 1024;
        auto issue = [&](int slot, int) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int piece = g * NP + pw;
                const unsigned char* src = piece < 8 ? xw + xo + piece * 1024 : ww + wo + (piece - 8) * 1024;
                glds16(src + lane * 16, smem + slot * SLOT + piece * 1024);
            }
            xo += 8 * 1024; if (xo + 8 * 1024 > XWIN) xo = 0;
            wo += WSTEP; if (wo + WSTEP > WWIN) wo = 0;
        };
        if constexpr (PM >= 1 && NG > 0) {
            // ---- the same stream through VGPRs: global_load_dwordx4 (D = PM + 2 tiles in flight) -> ds_write_b128 into the slot
            constexpr int D = PM + 2;
            u32x4 rg[D][NG];
            auto pload = [&](const int set) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int piece = g * NP + pw;
                    const unsigned char* src = piece < 8 ? xw + xo + piece * 1024 : ww + wo + (piece - 8) * 1024;
                    rg[set][g] = *(const u32x4*)(src + lane * 16);
                }
                xo += 8 * 1024; if (xo + 8 * 1024 > XWIN) xo = 0;
                wo += WSTEP; if (wo + WSTEP > WWIN) wo = 0;
            };
            auto pwrite = [&](const int set, int slot) {
#pragma unroll
                for (int g = 0; g < NG; ++g) *(u32x4*)(smem + slot * SLOT + (g * NP + pw) * 1024 + lane * 16) = rg[set][g];
            };
#pragma unroll
            for (int i = 0; i < D; ++i) pload(i);
            wait_vmcnt<(D - 1) * NG>(); pwrite(0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            int slot = 1;
            for (int t = 0; t < tiles; t += D) {
#pragma unroll
                for (int i = 0; i < D; ++i) {          // load tile t + i + D into the set tile t + i left, write tile t + i + 1
                    if constexpr (IL) {                // piece by piece: ds_write of tile t + i + 1 between the loads of tile t + i + D
                        wait_vmcnt<(D - 2) * NG>();    // (tile t + i + 1 has landed: the D - 2 younger tiles may be in flight)
#pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            const int piece = g * NP + pw;
                            const unsigned char* src = piece < 8 ? xw + xo + piece * 1024 : ww + wo + (piece - 8) * 1024;
                            rg[i][g] = *(const u32x4*)(src + lane * 16);
                            __builtin_amdgcn_sched_barrier(0);
                            *(u32x4*)(smem + slot * SLOT + piece * 1024 + lane * 16) = rg[(i + 1) % D][g];
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        xo += 8 * 1024; if (xo + 8 * 1024 > XWIN) xo = 0;
                        wo += WSTEP; if (wo + WSTEP > WWIN) wo = 0;
                    } else {
                    pload(i);
                    wait_vmcnt<(D - 1) * NG>();
                    pwrite((i + 1) % D, slot);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (++slot == S) slot = 0;
                }
            }
            wait_vmcnt<0>();
            return;
        }
        if (NG > 0) {
#pragma unroll
            for (int s = 0; s < S - 1; ++s) issue(s, s);
            wait_vmcnt<(S - 2) * NG>();
        }
        __builtin_amdgcn_s_barrier();
        int slot = S - 1;
        for (int t = 0; t < tiles; ++t) {
            if (NG > 0) {
                issue(slot, t + S - 1);
                wait_vmcnt<(S - 2) * NG>();
            }
            __builtin_amdgcn_s_barrier();
            if (++slot == S) slot = 0;
        }
        wait_vmcnt<0>();
        return;
    }

    // -------------------------------------------------------------------- consumer
    f32x4 acc[NM > 0 ? NM : 1];
#pragma unroll
    for (int i = 0; i < (NM > 0 ? NM : 1); ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xf[NR > 0 ? NR : 1];
    u32x4 wv[3][NV > 0 ? NV : 1];                 // VGPR weight ring: tile t lives in set t % 3
#pragma unroll
    for (int i = 0; i < (NR > 0 ? NR : 1); ++i) xf[i] = bf16x8{};
    // this wave's VGPR pieces of tile t: NV contiguous KB of the weight window (no two waves share a piece)
    unsigned vo = 0;                               // running window offset (wave-uniform)
    constexpr unsigned VSTEP = 4 * NV * 1024;
    auto vload = [&](auto SET, int) {
        constexpr int set = decltype(SET)::value;
#pragma unroll
        for (int j = 0; j < NV; ++j)
            wv[set][j] = *(const u32x4*)(ww + vo + (wave * NV + j) * 1024 + lane * 16);
        vo += VSTEP; if (vo + VSTEP > WWIN) vo = 0;
    };
    auto tile = [&](auto SET, auto SET2, int slot, int t) {
        constexpr int set = decltype(SET)::value;
        if (NV > 0) vload(SET2, t + 2);            // two tiles ahead
        const unsigned char* st = smem + slot * SLOT + wave * (NR > 0 ? (SLOT / 4 / 1024) * 1024 : 0);
#pragma unroll
        for (int r = 0; r < NR; ++r) xf[r] = *(const bf16x8*)(st + (r % 7) * 1024 + lane * 16);
        if (NM > 0) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                bf16x8 a;
                if (NV > 0) a = __builtin_bit_cast(bf16x8, wv[set][i % (NV > 0 ? NV : 1)]);
                else a = xf[(i + 1) % (NR > 0 ? NR : 1)];
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xf[i % (NR > 0 ? NR : 1)], acc[i], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < NR; ++r) asm volatile("" ::"v"(xf[r]));
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("" ::"v"(wv[set][j]));
        }
    };
    const std::integral_constant<int, 0> I0{};
    const std::integral_constant<int, 1> I1{};
    const std::integral_constant<int, 2> I2{};
    if (NV > 0) { vload(I0, 0); vload(I1, 1); }
    __builtin_amdgcn_s_barrier();
    int slot = 0;
    const unsigned long long c0 = __builtin_readcyclecounter();
    auto step = [&](auto SET, auto SET2, int t) {
        tile(SET, SET2, slot, t);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (++slot == S) slot = 0;
    };
    for (int t = 0; t < tiles; t += 3) {
        step(I0, I2, t);
        step(I1, I0, t + 1);
        step(I2, I1, t + 2);
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < (NM > 0 ? NM : 1); ++i) s += acc[i][0];
    if (NV > 0) { s += (float)wv[0][0][0] + (float)wv[1][0][0] + (float)wv[2][0][0]; }
    if (s == 123.456f) sink[0] = s;
    if (wave == 0 && lane == 0) cyc[blockIdx.x] = c1 - c0;
}

struct Res { double cyc_med, ns; };

template <int NP, int NG, int NR, int NV, int NM, int PM = 0, bool IL = false>
Res run(const unsigned char* x, const unsigned char* w, unsigned long long* cyc, float* sink) {
    const int tiles = 3000;
    auto kern = k<NP, NG, NR, NV, NM, PM, IL>;
    const int threads = 64 * (4 + NP);
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, S * SLOT);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<256, threads, S * SLOT>>>(x, w, tiles, cyc, sink);
    hipEventRecord(e0);
    kern<<<256, threads, S * SLOT>>>(x, w, tiles, cyc, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    return {(double)h[128] / tiles, ms * 1e6 / tiles};
}

int main() {
    unsigned char *x, *w; unsigned long long* cyc; float* sink;
    hipMalloc(&x, 32 * (size_t)XWIN + (1 << 20)); hipMalloc(&w, 8 * (size_t)WWIN + (1 << 20));
    hipMemset(x, 0x3c, 32 * (size_t)XWIN + (1 << 20)); hipMemset(w, 0x3c, 8 * (size_t)WWIN + (1 << 20));
    hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 4);
    printf("mix, producers, DMA KB/tile, ds_read_b128 KB/tile, VGPR-load KB/tile, MFMA/tile/wave, cycles/tile (median WG), ns/tile (wall)\n");
#define ROW(name, NP, NG, NR, NV, NM) do { Res r = run<NP, NG, NR, NV, NM>(x, w, cyc, sink); \
    printf("%s, %d, %d, %d, %d, %d, %.0f, %.0f\n", name, NP, NP * NG, 4 * NR, 4 * NV, NM, r.cyc_med, r.ns); fflush(stdout); } while (0)
    // ---- the present kernel, piece by piece
    ROW("mfma only", 4, 0, 0, 0, 20);
    ROW("dma only (28 KB)", 4, 7, 0, 0, 0);
    ROW("dma only (28 KB) two producers", 2, 14, 0, 0, 0);
    ROW("reads only (56 KB)", 4, 0, 14, 0, 0);
    ROW("dma + reads", 4, 7, 14, 0, 0);
    ROW("reads + mfma", 4, 0, 14, 0, 20);
    ROW("current: dma + reads + mfma", 4, 7, 14, 0, 20);
    // ---- LDS port: DMA writes against reads at other ratios (serial model: sum; parallel: max)
    ROW("dma 28 KB + reads 16 KB", 4, 7, 4, 0, 0);
    ROW("dma 8 KB + reads 56 KB", 4, 2, 14, 0, 0);
    ROW("dma 8 KB only", 4, 2, 0, 0, 0);
    ROW("reads only (16 KB)", 4, 0, 4, 0, 0);
    // ---- VGPR loads: their own rate, and against the DMA stream (shared address unit: sum; separate: max)
    ROW("vgpr loads only (20 KB)", 4, 0, 0, 5, 0);
    ROW("vgpr loads only (28 KB)", 4, 0, 0, 7, 0);
    ROW("vgpr 20 KB + dma 8 KB", 4, 2, 0, 5, 0);
    ROW("vgpr 20 KB + dma 28 KB", 4, 7, 0, 5, 0);
    ROW("vgpr 20 KB + reads 16 KB", 4, 0, 4, 5, 0);
    ROW("vgpr 20 KB + mfma", 4, 0, 0, 5, 20);
    // ---- the proposed K tile: 8 KB DMA (token rows) + 16 KB reads + 20 KB VGPR loads (weights) + MFMA
    ROW("proposed, 4 producers", 4, 2, 4, 5, 20);
    ROW("proposed, 2 producers", 2, 4, 4, 5, 20);
    ROW("proposed without mfma", 2, 4, 4, 5, 0);
    // ---- the LDS-DMA stream replaced by producer-wave global_load_dwordx4 -> VGPR -> ds_write_b128 (same bytes, same ring)
#define ROWV(name, NP, NG, NR, NM, PM, IL) do { Res r = run<NP, NG, NR, 0, NM, PM, IL>(x, w, cyc, sink); \
    printf("%s, %d, %d, %d, %d, %d, %.0f, %.0f\n", name, NP, NP * NG, 4 * NR, 0, NM, r.cyc_med, r.ns); fflush(stdout); } while (0)
    ROWV("via vgpr (3 tiles in flight): load + ds_write only (28 KB)", 4, 7, 0, 0, 1, false);
    ROWV("via vgpr (3): 8 KB only", 4, 2, 0, 0, 1, false);
    ROWV("via vgpr (3) + reads + mfma (= current mix)", 4, 7, 14, 20, 1, false);
    ROWV("via vgpr (6 tiles in flight): load + ds_write only (28 KB)", 4, 7, 0, 0, 4, false);
    ROWV("via vgpr (6) + reads + mfma", 4, 7, 14, 20, 4, false);
    ROWV("via vgpr (4), loads and writes interleaved: 28 KB only", 4, 7, 0, 0, 2, true);
    ROWV("via vgpr (4), interleaved + reads + mfma", 4, 7, 14, 20, 2, true);
    ROWV("via vgpr (6), interleaved: 28 KB only", 4, 7, 0, 0, 4, true);
    ROWV("via vgpr (6), interleaved: 8 KB only", 4, 2, 0, 0, 4, true);
    ROWV("via vgpr (6), interleaved + reads", 4, 7, 14, 0, 4, true);
    ROWV("via vgpr (6), interleaved + reads + mfma", 4, 7, 14, 20, 4, true);
    ROWV("via vgpr (6), interleaved, seven producers + reads + mfma", 7, 4, 14, 20, 4, true);
    return 0;
}
