// Multi-stage ("ring") variant of the MFMA GEMM / implicit-GEMM conv kernel for gfx950.
//
// Same operand layout, swizzles, fragment maps and epilogue as gemm.hip, but:
//   * S LDS stages with COUNTED vmcnt: S-1 tiles are in flight while one is consumed, the LDS-DMA
//     queue is never drained inside the K loop (cdna_hip_programming.md T3/T4: "never vmcnt(0) in the
//     main loop"), one raw s_barrier per K-tile;
//   * WM x WN waves (up to 8) and tiles up to 256x256, i.e. fewer L2->LDS bytes per FLOP.
// gemm.hip's two-stage kernel is load-latency bound (one 32 KB tile in flight per workgroup,
// profiles/r01_*); this kernel exists to lift that.
#include "imh_common.h"
#include "imh_kernels.h"
#include "imh_gemm_epilogue.h"
#include "imh_lnstats.h"
#include <type_traits>

namespace imh {

int g_ws_early = 1;     // imh_debug_set key 6 (A/B): 1 = the residual-add launches of the wave-specialised kernel fetch their residual rows before the K loop

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int BM, int BN, int WM, int WN, int S, bool CONV>
__global__ __launch_bounds__(64 * WM * WN, 1) void gemm_ring_kernel(const GemmParams p) {
    constexpr int NW = WM * WN;
    constexpr int NT = 64 * NW;
    constexpr int TM = BM / WM, TN = BN / WN;      // wave tile
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int RPR = NW * 8;                    // tile rows staged per round by the whole block
    constexpr int RX = BM / RPR, RW = BN / RPR;
    constexpr int LP = RX + RW;                    // LDS-DMA instructions per thread per K-tile
    constexpr int XT_BYTES = BM * GEMM_ROW_BYTES;
    constexpr int WT_BYTES = BN * GEMM_ROW_BYTES;
    constexpr int STAGE = XT_BYTES + WT_BYTES;
    static_assert(FN == 4 || FN == 2 || FN == 10 || FN == 5, "lane owns 8, 16, 20 or 40 output columns");
    static_assert(BM % RPR == 0 && BN % RPR == 0, "staging rounds");
    static_assert((S - 2) * LP <= 63, "vmcnt range");
    typedef typename Vec<T>::v8 v8;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    int m0, n0;
    if (!xcd_tile<BM, BN>(p, blockIdx.x, m0, n0)) return;     // the whole workgroup exits together

    const int nkt = p.K / GEMM_BK;
    const int z = blockIdx.y;
    const int per = (nkt + p.splits - 1) / p.splits;
    const int kt0 = z * per;
    const int kt1 = min(nkt, kt0 + per);
    const int nt = max(0, kt1 - kt0);

    const unsigned char* zero = g_zero_page;
    const int srow = wave * 8 + (lane >> 3);

    const unsigned char* xbase[RX];
    int xstep[RX];
    int cb[RX], coy[RX], cox[RX];
    bool cvalid[RX];
#pragma unroll
    for (int i = 0; i < RX; ++i) {
        const int row = i * RPR + srow;
        const int c = stage_chunk_x(row, lane);
        const int m = m0 + row;
        const bool ok = m < p.M;
        if (!CONV) {
            xbase[i] = ok ? (const unsigned char*)p.X + ((size_t)m * p.ldx) * sizeof(T) + c * 16 : zero + c * 16;
            xstep[i] = ok ? GEMM_BK * (int)sizeof(T) : 0;
        } else {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw;
            const int rem = m - b * hw;
            const int oy = rem / p.Wo;
            cb[i] = b; coy[i] = oy; cox[i] = rem - oy * p.Wo; cvalid[i] = ok;
            xbase[i] = zero + c * 16;
            xstep[i] = 0;
        }
    }
    const unsigned char* wbase[RW];
    int wstep[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int row = i * RPR + srow;
        const int c = stage_chunk_w(row, lane, FN);
        const int n = n0 + row;
        const bool ok = n < p.N;
        wbase[i] = ok ? (const unsigned char*)p.W + ((size_t)n * p.ldw) * sizeof(T) + c * 16 : zero + c * 16;
        wstep[i] = ok ? GEMM_BK * (int)sizeof(T) : 0;
    }
    const int cpt = CONV ? p.Cin / GEMM_BK : 1;

    auto stage = [&](int slot, int kt) {
        unsigned char* xs = smem + slot * STAGE;
        unsigned char* ws = xs + XT_BYTES;
        if (CONV) {
            const int tap = kt / cpt;
            const int ct = kt - tap * cpt;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int Hv = p.H << p.up, Wv = p.Wd << p.up;
#pragma unroll
            for (int i = 0; i < RX; ++i) {
                const int c = stage_chunk_x(i * RPR + srow, lane);
                const int iy = coy[i] * p.stride + ky - 1;
                const int ix = cox[i] * p.stride + kx - 1;
                const bool ok = cvalid[i] && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
                const size_t pix = ((size_t)cb[i] * p.H + (iy >> p.up)) * p.Wd + (ix >> p.up);
                const unsigned char* src = ok
                    ? (const unsigned char*)p.X + (pix * p.Cin + (size_t)ct * GEMM_BK) * sizeof(T) + c * 16
                    : zero + c * 16;
                glds16(src, xs + (i * RPR + wave * 8) * GEMM_ROW_BYTES);
            }
        } else {
#pragma unroll
            for (int i = 0; i < RX; ++i)
                glds16(xbase[i] + (size_t)kt * xstep[i], xs + (i * RPR + wave * 8) * GEMM_ROW_BYTES);
        }
#pragma unroll
        for (int i = 0; i < RW; ++i)
            glds16(wbase[i] + (size_t)kt * wstep[i], ws + (i * RPR + wave * 8) * GEMM_ROW_BYTES);
    };

    int xoff[2], woff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int xr = wm * TM + (lane & 15);
        const int wr = wn * TN + w_frag_row(lane & 15, 0, FN);
        xoff[kk] = tile_off(xr, kk * 4 + (lane >> 4), swz_x(xr));
        woff[kk] = tile_off(wr, kk * 4 + (lane >> 4), swz_w(wr, FN));
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: S-1 tiles in flight
#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < nt) stage(s, kt0 + s);

    int slot = 0;
    for (int t = 0; t < nt; ++t) {
        // tile t must have landed; tiles t+1 .. t+S-2 may stay in flight
        if (t + S - 2 < nt) wait_vmcnt<(S - 2) * LP>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();              // everyone's part of tile t is in LDS; tile t-1 fully consumed
        asm volatile("" ::: "memory");
        if (t + S - 1 < nt) {
            int ns = slot + S - 1;
            if (ns >= S) ns -= S;
            stage(ns, kt0 + t + S - 1);           // refill the slot tile t-1 used
        }
        const unsigned char* xs = smem + slot * STAGE;
        const unsigned char* ws = xs + XT_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            v8 xf[FM], wf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) xf[i] = *(const v8*)(xs + xoff[kk] + i * 16 * GEMM_ROW_BYTES);
#pragma unroll
            for (int j = 0; j < FN; ++j) wf[j] = *(const v8*)(ws + woff[kk] + j * 4 * GEMM_ROW_BYTES);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(wf[j], xf[i], acc[i][j]);
        }
        asm volatile("" ::: "memory");
        if (++slot == S) slot = 0;
    }

    const int nb = n0 + wn * TN + (lane >> 4) * 4 * FN;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * TM + i * 16 + (lane & 15);
        if (m >= p.M || nb >= p.N) continue;
        float v[4 * FN];
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j * 4 + r] = acc[i][j][r];
        if (p.splits > 1) {
            float* o = p.partial + ((size_t)z * p.M + m) * p.N + nb;
            if (nb + 4 * FN <= p.N && (p.N & 3) == 0) {
#pragma unroll
                for (int q0 = 0; q0 < 4 * FN; q0 += 4) *(f32x4*)(o + q0) = f32x4{v[q0], v[q0 + 1], v[q0 + 2], v[q0 + 3]};
            } else {
#pragma unroll
                for (int q = 0; q < 4 * FN; ++q) if (nb + q < p.N) o[q] = v[q];
            }
        } else {
            epilogue_store<T, FN>(p, v, m, nb);
        }
    }
    if (z == 0) tail_prefetch(p.pf_ptr, p.pf_bytes, blockIdx.x, gridDim.x, tid, blockDim.x);
}

// ---------------------------------------------------------------------------------------------
// "KG2": one 8-wave workgroup per output tile, the K range split between two 4-wave groups that each run
// the two-stage LDS pipeline of gemm.hip on alternate K-tiles; the two partial accumulators are exchanged
// through LDS and each group finishes half of the tile.  Same LDS/occupancy footprint per CU as two
// co-resident workgroups of gemm.hip, but every tile finishes in half the K iterations -- for the many
// SDXL GEMMs whose tile count is below the CU count (M = 2048, N = 1280) the kernel time is the length of
// one workgroup's K loop, not the throughput.
template <typename T, int BM, int BN, bool CONV>
__global__ __launch_bounds__(512, 2) void gemm_kg2_kernel(const GemmParams p) {
    constexpr int FM = BM / 32, FN = BN / 32;
    constexpr int RX = BM / 32, RW = BN / 32;
    constexpr int XT_BYTES = BM * GEMM_ROW_BYTES;
    constexpr int WT_BYTES = BN * GEMM_ROW_BYTES;
    constexpr int STAGE = XT_BYTES + WT_BYTES;
    constexpr int GROUP_BYTES = 2 * STAGE;
    static_assert(FM % 2 == 0, "the two groups split the token fragments");
    static_assert((FM / 2) * FN * 4 * 256 * 4 <= GROUP_BYTES, "exchange buffer fits a group's stages");
    typedef typename Vec<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave8 >> 2;               // K-group
    const int wave = wave8 & 3;               // wave inside the group (2x2 over the tile)
    const int wm = wave >> 1, wn = wave & 1;
    unsigned char* gmem = smem + grp * GROUP_BYTES;

    int m0, n0;
    if (!xcd_tile<BM, BN>(p, blockIdx.x, m0, n0)) return;     // the whole workgroup exits together
    const int nkt = p.K / GEMM_BK;
    const int z = blockIdx.y;
    const int per = (nkt + p.splits - 1) / p.splits;
    const int kt0 = z * per;
    const int kt1 = min(nkt, kt0 + per);
    const int niter = (max(0, kt1 - kt0) + 1) / 2;    // lockstep iterations; group g takes kt0 + 2*i + g

    const unsigned char* zero = g_zero_page;
    const unsigned char* xbase[RX];
    int xstep[RX];
    int cb[RX], coy[RX], cox[RX];
    bool cvalid[RX];
#pragma unroll
    for (int i = 0; i < RX; ++i) {
        const int row = stage_row(i, wave, lane);
        const int c = stage_chunk_x(row, lane);
        const int m = m0 + row;
        const bool ok = m < p.M;
        if (!CONV) {
            xbase[i] = ok ? (const unsigned char*)p.X + ((size_t)m * p.ldx) * sizeof(T) + c * 16 : zero + c * 16;
            xstep[i] = ok ? GEMM_BK * (int)sizeof(T) : 0;
        } else {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw;
            const int rem = m - b * hw;
            const int oy = rem / p.Wo;
            cb[i] = b; coy[i] = oy; cox[i] = rem - oy * p.Wo; cvalid[i] = ok;
            xbase[i] = zero + c * 16;
            xstep[i] = 0;
        }
    }
    const unsigned char* wbase[RW];
    int wstep[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int row = stage_row(i, wave, lane);
        const int c = stage_chunk_w(row, lane, FN);
        const int n = n0 + row;
        const bool ok = n < p.N;
        wbase[i] = ok ? (const unsigned char*)p.W + ((size_t)n * p.ldw) * sizeof(T) + c * 16 : zero + c * 16;
        wstep[i] = ok ? GEMM_BK * (int)sizeof(T) : 0;
    }
    const int cpt = CONV ? p.Cin / GEMM_BK : 1;

    auto stage = [&](int buf, int kt) {
        unsigned char* xs = gmem + buf * STAGE;
        unsigned char* ws = xs + XT_BYTES;
        if (CONV) {
            const int tap = kt / cpt;
            const int ct = kt - tap * cpt;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int Hv = p.H << p.up, Wv = p.Wd << p.up;
#pragma unroll
            for (int i = 0; i < RX; ++i) {
                const int c = stage_chunk_x(stage_row(i, wave, lane), lane);
                const int iy = coy[i] * p.stride + ky - 1;
                const int ix = cox[i] * p.stride + kx - 1;
                const bool ok = cvalid[i] && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
                const size_t pix = ((size_t)cb[i] * p.H + (iy >> p.up)) * p.Wd + (ix >> p.up);
                const unsigned char* src = ok
                    ? (const unsigned char*)p.X + (pix * p.Cin + (size_t)ct * GEMM_BK) * sizeof(T) + c * 16
                    : zero + c * 16;
                glds16(src, xs + stage_lds_off(i, wave));
            }
        } else {
#pragma unroll
            for (int i = 0; i < RX; ++i) glds16(xbase[i] + (size_t)kt * xstep[i], xs + stage_lds_off(i, wave));
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) glds16(wbase[i] + (size_t)kt * wstep[i], ws + stage_lds_off(i, wave));
    };

    int xoff[2], woff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        xoff[kk] = xfrag_off(lane, wm, BM, kk);
        woff[kk] = wfrag_off(lane, wn, BN, kk);
    }
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (niter > 0) {
        if (kt0 + grp < kt1) stage(0, kt0 + grp);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int cur = 0;
        for (int it = 0; it < niter; ++it) {
            const int kt = kt0 + 2 * it + grp;
            if (kt + 2 < kt1) stage(cur ^ 1, kt + 2);
            if (kt < kt1) {
                const unsigned char* xs = gmem + cur * STAGE;
                const unsigned char* ws = xs + XT_BYTES;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    v8 xf[FM], wf[FN];
#pragma unroll
                    for (int i = 0; i < FM; ++i) xf[i] = *(const v8*)(xs + xoff[kk] + i * 16 * GEMM_ROW_BYTES);
#pragma unroll
                    for (int j = 0; j < FN; ++j) wf[j] = *(const v8*)(ws + woff[kk] + j * 4 * GEMM_ROW_BYTES);
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(wf[j], xf[i], acc[i][j]);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur ^= 1;
        }
    }

    // ---- exchange: group g keeps token fragments [g*FM/2, (g+1)*FM/2) and receives the partner's partials.
    //      Every accumulator index below is a compile-time constant (a runtime index would push acc[] to
    //      scratch): the group id is lifted to a template constant. ----
    const int nb = n0 + out_col(lane, wn, BN);
    auto finish = [&](auto G) {
        constexpr int g = decltype(G)::value;
        float* mine = (float*)(smem + g * GROUP_BYTES);                 // written by me, read by the partner
        const float* theirs = (const float*)(smem + (g ^ 1) * GROUP_BYTES);
        const int t256 = tid & 255;
#pragma unroll
        for (int ii = 0; ii < FM / 2; ++ii)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    mine[((ii * FN + j) * 4 + r) * 256 + t256] = acc[(g ^ 1) * (FM / 2) + ii][j][r];
        __syncthreads();
#pragma unroll
        for (int ii = 0; ii < FM / 2; ++ii) {
            constexpr int dummy = 0; (void)dummy;
            const int m = m0 + out_row(lane, wm, BM, g * (FM / 2) + ii);
            float v[4 * FN];
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    v[j * 4 + r] = acc[g * (FM / 2) + ii][j][r] + theirs[((ii * FN + j) * 4 + r) * 256 + t256];
            if (m >= p.M || nb >= p.N) continue;
            if (p.splits > 1) {
                float* o = p.partial + ((size_t)z * p.M + m) * p.N + nb;
                if (nb + 4 * FN <= p.N && (p.N & 3) == 0) {
#pragma unroll
                    for (int q0 = 0; q0 < 4 * FN; q0 += 4) *(f32x4*)(o + q0) = f32x4{v[q0], v[q0 + 1], v[q0 + 2], v[q0 + 3]};
                } else {
#pragma unroll
                    for (int q = 0; q < 4 * FN; ++q) if (nb + q < p.N) o[q] = v[q];
                }
            } else {
                epilogue_store<T, FN>(p, v, m, nb);
            }
        }
    };
    if (grp == 0) finish(std::integral_constant<int, 0>{});
    else finish(std::integral_constant<int, 1>{});
    if (z == 0) tail_prefetch(p.pf_ptr, p.pf_bytes, blockIdx.x, gridDim.x, tid, blockDim.x);
}

// ---------------------------------------------------------------------------------------------
// "WS": wave-specialised variant for the M = 2048, N = 1280 layers (to_out, ff.out, the 32 x 32 convolutions).
// Those problems tile 256 ways only as 64 x 160, and at that size the vector-memory front end is the longest
// pipe of the CU: every 16-B piece of both operands goes through the address unit at 64 B / clk / CU
// (tools/micro/glds_rate.hip), 448 cycles per K tile against 320 cycles of MFMA time, and a wave that is queued
// behind that unit with an LDS-DMA instruction cannot issue its MFMAs (round-2 probe with cache-hot operands
// change nothing).  So the roles are split: NP producer waves do nothing but issue the LDS-DMA ring (own vmcnt
// counters, S - 1 tiles ahead), CM x CN consumer waves only read fragments and issue MFMAs, with the fragments of
// k step kk + 1 read into a second register set while the MFMAs of k step kk run, so that a consumer never waits
// on LDS latency either.  One s_barrier per K tile carries both hand-overs: "tile i + 1 has landed" (producers
// wait for it before arriving) and "tile i has been read" (consumers drain lgkmcnt before arriving), after which
// the producers refill tile i's slot.
// WS_TIMING (tools/ws_phase_probe.py only): cycle counter at the segment boundaries of the K loop; consumer wave 0 and producer wave 0
// of workgroup 0 write their per-segment totals to p.pf_ptr instead of prefetching --
//   consumer: [fragment reads + MFMAs] [s_waitcnt lgkmcnt(0)] [s_barrier]     producer: [LDS-DMA issue] [s_waitcnt vmcnt] [s_barrier]
#ifndef WS_TIMING
#define WS_TIMING 0
#endif
#if WS_TIMING
#define WS_TICK(i) do { asm volatile("s_nop 0" ::: "memory"); const unsigned long long tn_ = __builtin_readcyclecounter(); tacc[i] += tn_ - tm0; tm0 = tn_; } while (0)
#else
#define WS_TICK(i) do {} while (0)
#endif
// CM x CN consumer waves over the tile.  LN = 1: folded LayerNorm, row form, statistics handed over by the GEMM that wrote the
// token rows (imh_lnstats.h): before their first hand-over the consumer threads merge the slot partials of the tile's BM rows
// (one thread per row, loads in flight beside the producers' ring prologue) into an LDS area behind the ring -- no statistics
// work inside the K loop, all NP producers load.  LN = 2: the column form (tokens = the W operand's rows), same hand-over.
// (Rounds 2-3 carried a third form whose spare producer waves summed rows and squares of every landed K tile: E[x^2] - mean^2,
// removed with the other in-loop forms.)
template <typename T, int BM, int BN, int CM, int CN, int S, int NP, bool CONV, int LN>
__device__ __forceinline__ void gemm_ws_body(const GemmParams& p, const int bid, const int nblocks, unsigned char* smem) {
    constexpr int NC = CM * CN;
    constexpr int TM = BM / CM, TN = BN / CN;
    static_assert(LN == 0 || !CONV, "folded LayerNorm is a Linear-layer form");
    constexpr int NL = NP;                         // loader waves
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int NIX = BM / 8;                    // LDS-DMA wave instructions per K tile: token rows
    constexpr int NI = (BM + BN) / 8;              // ... and in total
    constexpr int LP = NI / NL, KX = NIX / NL;     // per loader wave
    constexpr int XT_BYTES = BM * GEMM_ROW_BYTES;
    constexpr int STAGE = (BM + BN) * GEMM_ROW_BYTES;
    static_assert(NI % NL == 0 && NIX % NL == 0, "instructions split evenly between the loaders");
    static_assert((S - 2) * LP <= 63, "vmcnt range");
    static_assert(TN % (4 * FN) == 0, "weight fragment rows");
    typedef typename Vec<T>::v8 v8;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

#if WS_TIMING
    const unsigned long long ts_entry = __builtin_amdgcn_s_memrealtime();      // 100 MHz, the same counter on every XCD
#endif
    int m0, n0;
    if (!xcd_tile<BM, BN>(p, bid, m0, n0)) return;     // the whole workgroup exits together
    const int nkt = p.K / GEMM_BK;
    const int z = blockIdx.y;
    const int per = (nkt + p.splits - 1) / p.splits;
    const int kt0 = z * per;
    const int kt1 = min(nkt, kt0 + per);
    const int nt = max(0, kt1 - kt0);

    if (wave >= NC) {
        // ------------------------------------------------------------------ producer
        const int pw = wave - NC;
        const unsigned char* zero = g_zero_page;
        const unsigned char* base[LP];
        const unsigned char* base2[KX > 0 ? KX : 1];      // second token source (p.X2: K columns [Cin1, K), dense rows)
        const int kt_x2 = (!CONV && p.X2) ? p.Cin1 / GEMM_BK : 0x7fffffff;
        int step[LP];
        int cb[KX > 0 ? KX : 1], coy[KX > 0 ? KX : 1], cox[KX > 0 ? KX : 1];
        bool cvalid[KX > 0 ? KX : 1];
#pragma unroll
        for (int k = 0; k < LP; ++k) {
            const int r = (k * NL + pw) * 8 + (lane >> 3);    // row of the stage: token rows, then weight rows
            if (k < KX) {
                const int c = stage_chunk_x(r, lane);
                const int m = m0 + r;
                const bool ok = m < p.M;
                base2[k] = zero + c * 16;
                if (!CONV) {
                    base[k] = ok ? (const unsigned char*)p.X + ((size_t)m * p.ldx) * sizeof(T) + c * 16 : zero + c * 16;
                    if (ok && p.X2) base2[k] = (const unsigned char*)p.X2 + ((size_t)m * (p.K - p.Cin1)) * sizeof(T) + c * 16;
                    step[k] = ok ? GEMM_BK * (int)sizeof(T) : 0;
                } else {
                    const int hw = p.Ho * p.Wo;
                    const int b = m / hw;
                    const int rem = m - b * hw;
                    const int oy = rem / p.Wo;
                    cb[k] = b; coy[k] = oy; cox[k] = rem - oy * p.Wo; cvalid[k] = ok;
                    base[k] = zero + c * 16;
                    step[k] = 0;
                }
            } else {
                const int row = r - BM;
                const int c = stage_chunk_w(row, lane, FN);
                const int n = n0 + row;
                const bool ok = n < p.N;
                base[k] = ok ? (const unsigned char*)p.W + ((size_t)n * p.ldw) * sizeof(T) + c * 16 : zero + c * 16;
                step[k] = ok ? GEMM_BK * (int)sizeof(T) : 0;
            }
        }
        const int cpt = CONV ? p.Cin / GEMM_BK : 1;
        auto issue = [&](int slot, int kt) {
            unsigned char* st = smem + slot * STAGE;
            int ky = 0, kx = 0, ct = 0;
            if (CONV) {
                const int tap = kt / cpt;
                ct = kt - tap * cpt;
                ky = tap / 3; kx = tap - ky * 3;
            }
#pragma unroll
            for (int k = 0; k < LP; ++k) {
                unsigned char* dst = st + (k * NL + pw) * 8 * GEMM_ROW_BYTES;
                if (CONV && k < KX) {
                    const int r = (k * NL + pw) * 8 + (lane >> 3);
                    const int c = stage_chunk_x(r, lane);
                    const int Hv = p.H << p.up, Wv = p.Wd << p.up;
                    const int iy = coy[k] * p.stride + ky - 1;
                    const int ix = cox[k] * p.stride + kx - 1;
                    const bool ok = cvalid[k] && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
                    const size_t pix = ((size_t)cb[k] * p.H + (iy >> p.up)) * p.Wd + (ix >> p.up);
                    const unsigned char* src = ok
                        ? (const unsigned char*)p.X + (pix * p.Cin + (size_t)ct * GEMM_BK) * sizeof(T) + c * 16
                        : zero + c * 16;
                    glds16(src, dst);
                } else if (!CONV && k < KX && kt >= kt_x2) {
                    glds16(base2[k] + (size_t)(kt - kt_x2) * step[k], dst);
                } else {
                    glds16(base[k] + (size_t)kt * step[k], dst);
                }
            }
        };
#pragma unroll
        for (int s = 0; s < S - 1; ++s)
            if (s < nt) issue(s, kt0 + s);
        if (S - 1 <= nt) wait_vmcnt<(S - 2) * LP>();          // tile 0 has landed
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        int slot = S - 1;                                     // slot of tile i + S - 1
#if WS_TIMING
        unsigned long long tacc[3] = {0, 0, 0}, tm0 = __builtin_readcyclecounter();
#endif
        for (int i = 0; i < nt; ++i) {
            if (i + S - 1 < nt) issue(slot, kt0 + i + S - 1); // the slot tile i - 1 was read from
            WS_TICK(0);
            if (i + S <= nt) wait_vmcnt<(S - 2) * LP>();      // tile i + 1 has landed
            else wait_vmcnt<0>();
            WS_TICK(1);
            __builtin_amdgcn_s_barrier();
            WS_TICK(2);
            if (++slot == S) slot = 0;
        }
#if WS_TIMING
        if (bid == 0 && pw == 0 && lane == 0 && p.pf_ptr) {
            unsigned long long* dbg = (unsigned long long*)p.pf_ptr + 4;
            dbg[0] = tacc[0]; dbg[1] = tacc[1]; dbg[2] = tacc[2]; dbg[3] = nt;
        }
#else
        if (z == 0) tail_prefetch(p.pf_ptr, p.pf_bytes, bid, nblocks, tid - 64 * NC, 64 * NL);
#endif
        return;
    }

    // ---------------------------------------------------------------------- consumer
    const int wm = wave / CN, wn = wave % CN;
    int xoff[2], woff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int xr = wm * TM + (lane & 15);
        const int wr = wn * TN + w_frag_row(lane & 15, 0, FN);
        xoff[kk] = tile_off(xr, kk * 4 + (lane >> 4), swz_x(xr));
        woff[kk] = XT_BYTES + tile_off(wr, kk * 4 + (lane >> 4), swz_w(wr, FN));
    }
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    v8 xf[2][FM], wf[2][FN];                       // register set kk holds k step kk of the tile being consumed
    auto rd = [&](auto KK, int slot) {
        constexpr int kk = decltype(KK)::value;
        const unsigned char* st = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < FM; ++i) xf[kk][i] = *(const v8*)(st + xoff[kk] + i * 16 * GEMM_ROW_BYTES);
#pragma unroll
        for (int j = 0; j < FN; ++j) wf[kk][j] = *(const v8*)(st + woff[kk] + j * 4 * GEMM_ROW_BYTES);
    };
    auto mm = [&](auto KK) {
        constexpr int kk = decltype(KK)::value;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(wf[kk][j], xf[kk][i], acc[i][j]);
    };
    const std::integral_constant<int, 0> K0{};
    const std::integral_constant<int, 1> K1{};
    // issue order of one half interval: the fragment reads of the next k step interleaved one-to-one with the first
    // MFMAs of the previous one, the remaining MFMAs behind them (the reads return under the MFMA pipe time)
    auto interleave = [&]() {
#pragma unroll
        for (int k = 0; k < FM + FN; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one DS read
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
        }
        __builtin_amdgcn_sched_group_barrier(0x008, FM * FN - (FM + FN), 0);
    };
    if constexpr (LN == 1) {                       // (mean, rstd) of the tile's rows -> LDS area behind the ring
        const int t = wave * 64 + lane;
        if (t < BM) {
            const int m = m0 + t;
            f32x2s mr = {0.f, 1.f};
            if (m < p.M) mr = merge_row_stats(p.ln_stats, m, p.ln_slots, p.K, p.ln_eps);
            *(f32x2s*)(smem + S * STAGE + t * 8) = mr;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if constexpr (LN == 2) {                       // column form (tokens = the W operand's rows = output columns): the tile's BN tokens
        const int t = wave * 64 + lane;
        if (t < BN) {
            const int n = n0 + t;
            f32x2s mr = {0.f, 1.f};
            if (n < p.N) mr = merge_row_stats(p.ln_stats, n, p.ln_slots, p.K, p.ln_eps);
            *(f32x2s*)(smem + S * STAGE + t * 8) = mr;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // The residual-add launches (to_out, ff.out, proj_out, conv2: bias + residual, nothing else) fetch the residual rows and the bias of
    // ALL of the lane's fragments HERE, beside the producers' ring prologue: they are inputs of the launch, and the epilogue then has no
    // load between the last MFMA and its stores (round 5: the dependent 5 MB read of 256 workgroups at once was ~1 us per launch)
    constexpr bool RES_EARLY = LN == 0 && NC + NP <= 8;
    constexpr int NVE = RES_EARLY ? 4 * FN : 2;
    RawRow<T, NVE> bs_e, rr_e[RES_EARLY ? FM : 1];   // packed: half the registers of the fp32 form, held across the whole K loop
    bool ok_e[RES_EARLY ? FM : 1];
    bool early = false;
    if constexpr (RES_EARLY) {
        const int nb_e = n0 + wn * TN + (lane >> 4) * 4 * FN;
        early = p.early_res && p.flags == 0 && !p.rowadd && p.residual && p.bias && p.splits == 1 && epilogue_fast<T, 4 * FN>(p, nb_e);
        if (early) {
            ldraw<T, NVE>((const T*)p.bias + nb_e, bs_e);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = m0 + wm * TM + i * 16 + (lane & 15);
                ok_e[i] = m < p.M;
                ldraw<T, NVE>((const T*)p.residual + (size_t)min(m, p.M - 1) * p.ldr + nb_e, rr_e[i]);
            }
        }
    }
    __builtin_amdgcn_s_barrier();                  // tile 0 has landed
    asm volatile("" ::: "memory");
    int slot = 0;
#if WS_TIMING
    unsigned long long tacc[3] = {0, 0, 0}, tm0 = __builtin_readcyclecounter();
    const unsigned long long ts_loop0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int i = 0; i < nt; ++i) {
        rd(K0, slot);                              // (i, k step 0)  beside the MFMAs of (i - 1, k step 1)
        if (i > 0) mm(K1);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        rd(K1, slot);                              // (i, k step 1)  beside the MFMAs of (i, k step 0)
        mm(K0);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        WS_TICK(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // tile i has been read: its slot may be refilled
        WS_TICK(1);
        __builtin_amdgcn_s_barrier();                            // ... and tile i + 1 has landed
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        WS_TICK(2);
        if (++slot == S) slot = 0;
    }
    if (nt > 0) mm(K1);
#if WS_TIMING
    const unsigned long long ts_loop1 = __builtin_amdgcn_s_memrealtime();
    if (bid == 0 && wave == 0 && lane == 0 && p.pf_ptr) {
        unsigned long long* dbg = (unsigned long long*)p.pf_ptr;
        dbg[0] = tacc[0]; dbg[1] = tacc[1]; dbg[2] = tacc[2]; dbg[3] = nt;
    }
    // every workgroup (pf_bytes == 0xfeed): [entry, K loop begin, K loop end] of consumer wave 0 on the chip-wide 100 MHz counter
    if (wave == 0 && lane == 0 && p.pf_ptr && p.pf_bytes == 0xfeed) {
        unsigned long long* dbg = (unsigned long long*)p.pf_ptr + 8 + (size_t)bid * 4;
        dbg[0] = ts_entry; dbg[1] = ts_loop0; dbg[2] = ts_loop1; dbg[3] = 0;
    }
#endif

    const int nb = n0 + wn * TN + (lane >> 4) * 4 * FN;
    // s_n, c_n (folded LayerNorm) or the bias of this lane's columns: fetched once, ahead of the stores (EpiPre; the LN
    // instantiations are at the register cap of twelve waves per CU and let the epilogue fetch the bias per row)
    float lnpre[8 * FN];
    const bool have_pre = ln_preload<4 * FN>(p, nb, lnpre);
    EpiPre<4 * FN> pre;
    pre.ok = !p.rowadd && !p.residual && p.splits == 1 && epilogue_fast<T, 4 * FN>(p, nb);
    if (pre.ok && p.bias) ldv<T, 4 * FN>((const T*)p.bias + nb, pre.bias);
    LnArgs<4 * FN> ln;
    float st_s[FM], st_q[FM];
    // GroupNorm partials of the output (for the GroupNorm that reads it; imh_lnstats.h gn_emit): sub-runs of 10 channels
    constexpr int GNV = (4 * FN) % 10 == 0 ? 4 * FN : 10;
    GnAcc<GNV> gna;
    gn_zero(gna);
    if constexpr (LN == 2) {
        const f32x2s* ex = (const f32x2s*)(smem + S * STAGE);
#pragma unroll
        for (int q = 0; q < 4 * FN; ++q) {
            const f32x2s mr = ex[wn * TN + (lane >> 4) * 4 * FN + q];
            ln.cm[q] = mr[0]; ln.cr[q] = mr[1];
        }
    }
    if constexpr (LN == 1) {                       // mean / rstd of the tile's rows, merged from the producer kernel's partials before the loop
        const float* ex = (const float*)(smem + S * STAGE);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int r = wm * TM + i * 16 + (lane & 15);
            st_s[i] = ex[r * 2 + 0];
            st_q[i] = ex[r * 2 + 1];
        }
    }
    if constexpr (LN == 1) {
        // the transformer blocks' launches (ff.net.0: LN + bias + GEGLU; to_q / to_k: LN only) take a lean epilogue: whole
        // 4*FN-column run in range, vector-aligned output, no row-add / residual / activation -- operands fetched once
        // above, then per row the LN formula, the GEGLU product and ONE store
        if ((p.flags & ~(GF_LN_ROW | GF_GEGLU)) == 0 && pre.ok && have_pre) {
            constexpr int NV = 4 * FN;
            const bool geglu = p.flags & GF_GEGLU;
            const bool hb = p.bias != nullptr;          // (the transformer blocks' launches carry none: the GEGLU bias lives in ln_c)
            if (!hb) {
#pragma unroll
                for (int q = 0; q < NV; ++q) pre.bias[q] = 0.f;
            }
            if (p.Yt && n0 >= p.yt_col0) {
                // TRANSPOSED store (the V third of the self-attention's [Q|K|V] launch): column n of this tile becomes row
                // n - yt_col0 of Yt [., ldyt], token m its column, the 16 tokens of every fragment row group stored in the order
                // [0-3, 8-11, 4-7, 12-15] (imh_layout.h vt_perm16: the V^T layout the attention kernels' PV operand reads with one
                // ds_read_b128).  Through the wave's own slab of the dead ring: [TN channels][TM tokens], written as 2-byte values
                // (lane = token, 20 / 40 channels), read back as whole TM-token rows and stored coalesced.  The launcher guarantees
                // whole tiles (M % BM == 0, N % BN == 0) and yt_col0 % BN == 0.
                T* stg = (T*)(smem + wave * (TM * TN * (int)sizeof(T)));
                const int g = lane >> 4, r = lane & 15;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int pos = i * 16 + vt_perm16(r);
#pragma unroll
                    for (int j = 0; j < FN; ++j)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const int q = j * 4 + rr;
                            const float y = fma_nopk(st_q[i], fma_nopk(-st_s[i], lnpre[q], acc[i][j][rr]), lnpre[NV + q]) + pre.bias[q];
                            stg[(g * NV + q) * TM + pos] = from_f32<T>(y);
                        }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // same-wave LDS hand-off (DS operations retire in order)
                constexpr int LPR = TM / 8;                              // lanes per transposed row (16 B each)
                constexpr int RPP = 64 / LPR;                            // rows per pass
                T* yt = (T*)p.Yt + (size_t)(n0 - p.yt_col0 + wn * TN) * p.ldyt + (m0 + wm * TM);
#pragma unroll
                for (int ps = 0; ps < TN / RPP; ++ps) {
                    const int row = ps * RPP + lane / LPR, ch = lane % LPR;
                    const v8 t8 = *(const v8*)(stg + row * TM + ch * 8);
                    *(v8*)(yt + (size_t)row * p.ldyt + ch * 8) = t8;
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = m0 + wm * TM + i * 16 + (lane & 15);
                if (m >= p.M) continue;
                float v[NV];
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = j * 4 + r;
                        v[q] = fma_nopk(st_q[i], fma_nopk(-st_s[i], lnpre[q], acc[i][j][r]), lnpre[NV + q]) + pre.bias[q];
                    }
                if (geglu) {
                    float o[NV / 2];
                    geglu_quads<NV>(v, o);
                    stv<T, NV / 2>((T*)p.Y + (size_t)m * p.ldy + (nb >> 1), o);
                } else {
                    stv<T, NV>((T*)p.Y + (size_t)m * p.ldy + nb, v);
                }
            }
            return;
        }
    }
    if constexpr (LN == 0 && NC + NP <= 8) {
        // the residual-add launches (to_out, ff.out, proj_out: bias + residual, nothing else) likewise: the residual rows of
        // ALL of the lane's fragments are fetched before the first store -- the generic epilogue's per-row loads queue behind
        // the previous row's stores (possible aliasing as far as the compiler knows; in place, Y == residual, is fine here:
        // every element is read before its own store, by the same lane)
        if (p.flags == 0 && !p.rowadd && p.residual && p.splits == 1 && epilogue_fast<T, 4 * FN>(p, nb)) {
            constexpr int NV = 4 * FN;
            float bs[NV], rr[FM][NV];
            bool ok[FM];
            if (early) {                            // fetched before the K loop
                unraw<T, NV>(bs_e, bs);
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    ok[i] = ok_e[i];
                    unraw<T, NV>(rr_e[i], rr[i]);
                }
            } else {
                if (p.bias) ldv<T, NV>((const T*)p.bias + nb, bs);
                else {
#pragma unroll
                    for (int q = 0; q < NV; ++q) bs[q] = 0.f;
                }
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int m = m0 + wm * TM + i * 16 + (lane & 15);
                    ok[i] = m < p.M;
                    if (ok[i]) ldv<T, NV>((const T*)p.residual + (size_t)m * p.ldr + nb, rr[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                if (!ok[i]) continue;
                const int m = m0 + wm * TM + i * 16 + (lane & 15);
                float v[NV];
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[j * 4 + r] = acc[i][j][r] + bs[j * 4 + r] + rr[i][j * 4 + r];
                stv<T, NV>((T*)p.Y + (size_t)m * p.ldy + nb, v);
                if (p.ln_stats_out) emit_row_stats<T, NV>(p.ln_stats_out, p.ln_slots_out, m, nb, v, lane);
                if constexpr (NV % 10 == 0) {
                    if (p.gn_out) gn_accumulate<T, NV>(v, gna, i == 0);
                }
            }
            if constexpr (NV % 10 == 0) {
                if (p.gn_out) {
                    const int mw = m0 + wm * TM;
                    if (mw < p.M) gn_emit<NV>(p.gn_out, p.gn_nblk, p.N / 10, mw / p.gn_hw, (mw % p.gn_hw) / TM, nb, gna, FM, lane);      // (wave-uniform guard: the butterflies inside need the whole wave)
                }
            }
            return;
        }
    }
    if constexpr (LN == 0 && NC + NP > 8) {
        // Round 6: the twelve-wave forms (256 x 160: the UNet-batch-8 to_out / ff.out / proj_out) take the same shortcut with the residual rows
        // kept PACKED (FM x 10 registers beside the 80 accumulators; the fragment registers are dead by now): ALL of the lane's residual
        // rows and the bias are requested before the first store.  Through the generic epilogue below each row's residual load queued behind
        // the previous row's stores -- 9.6 us from the end of the K loop to the last store of an 8192 x 1280 x 1280 launch against 5.0 us
        // without a residual (profiles/r06_ws_timeline.txt)
        if (p.flags == 0 && !p.rowadd && p.residual && p.bias && p.splits == 1 && epilogue_fast<T, 4 * FN>(p, nb)) {
            constexpr int NV = 4 * FN;
            RawRow<T, NV> bs_r, rr_r[FM];
            bool ok[FM];
            ldraw<T, NV>((const T*)p.bias + nb, bs_r);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = m0 + wm * TM + i * 16 + (lane & 15);
                ok[i] = m < p.M;
                ldraw<T, NV>((const T*)p.residual + (size_t)min(m, p.M - 1) * p.ldr + nb, rr_r[i]);
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                if (!ok[i]) continue;
                const int m = m0 + wm * TM + i * 16 + (lane & 15);
                float v[NV], bs[NV], rr[NV];
                unraw<T, NV>(bs_r, bs);
                unraw<T, NV>(rr_r[i], rr);
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[j * 4 + r] = acc[i][j][r] + bs[j * 4 + r] + rr[j * 4 + r];
                stv<T, NV>((T*)p.Y + (size_t)m * p.ldy + nb, v);
                if (p.ln_stats_out) emit_row_stats<T, NV>(p.ln_stats_out, p.ln_slots_out, m, nb, v, lane);
                if constexpr (NV % 10 == 0) {
                    if (p.gn_out) gn_accumulate<T, NV>(v, gna, i == 0);
                }
            }
            if constexpr (NV % 10 == 0) {
                if (p.gn_out) {
                    const int mw = m0 + wm * TM;
                    if (mw < p.M) gn_emit<NV>(p.gn_out, p.gn_nblk, p.N / 10, mw / p.gn_hw, (mw % p.gn_hw) / TM, nb, gna, FM, lane);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * TM + i * 16 + (lane & 15);
        if (m >= p.M || nb >= p.N) continue;
        float v[4 * FN];
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j * 4 + r] = acc[i][j][r];
        if (p.splits > 1) {
            float* o = p.partial + ((size_t)z * p.M + m) * p.N + nb;
            if (nb + 4 * FN <= p.N && (p.N & 3) == 0) {
#pragma unroll
                for (int q0 = 0; q0 < 4 * FN; q0 += 4) *(f32x4*)(o + q0) = f32x4{v[q0], v[q0 + 1], v[q0 + 2], v[q0 + 3]};
            } else {
#pragma unroll
                for (int q = 0; q < 4 * FN; ++q) if (nb + q < p.N) o[q] = v[q];
            }
        } else {
            if constexpr (LN == 1) { ln.mean = st_s[i]; ln.rstd = st_q[i]; }
            epilogue_store_pre<T, FN>(p, v, m, nb, lnpre, have_pre, LN != 0 ? &ln : nullptr, LN != 0 ? nullptr : &pre, lane,
                                      (LN == 0 && (4 * FN) % 10 == 0) ? &gna : nullptr, i == 0);
        }
    }
    if constexpr (LN == 0 && (4 * FN) % 10 == 0) {
        if (p.gn_out && p.splits == 1) {
            const int mw = m0 + wm * TM;
            if (mw < p.M) gn_emit<4 * FN>(p.gn_out, p.gn_nblk, p.N / 10, mw / p.gn_hw, (mw % p.gn_hw) / TM, nb, gna, FM, lane);
        }
    }
}

// OCC = workgroups per CU the register allocation is held to (2: two co-resident workgroups with half-depth rings -- one's
// prologue / epilogue runs beside the other's K loop)
template <typename T, int BM, int BN, int CM, int CN, int S, int NP, bool CONV, int LN, int OCC = 1>
__global__ __launch_bounds__(64 * (CM * CN + NP), OCC == 1 ? 1 : OCC * (CM * CN + NP) / 4) void gemm_ws_kernel(const GemmParams p) {      // (HIP: the second value is waves per SIMD)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    gemm_ws_body<T, BM, BN, CM, CN, S, NP, CONV, LN>(p, blockIdx.x, gridDim.x, smem);
#if WS_TIMING
    // per-workgroup time line: consumer wave 0's stores have left the wave -> exit stamp (tools/ws_timeline_probe.py)
    if (threadIdx.x == 0 && p.pf_ptr && p.pf_bytes == 0xfeed) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ((unsigned long long*)p.pf_ptr)[8 + (size_t)blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

#ifdef IMH_EXPERIMENTAL
// Self-attention's projections in ONE wave-specialised launch, LayerNorm statistics handed over (imh_lnstats.h):
//   a: [Q|K] = LN(x) [Wq;Wk]^T   row form, 128 x 160 tiles (M = 2048 x N = 2560 -> 256 tiles)
//   b: V^T   = Wv LN(x)^T        column form + V^T key permutation, 128 x 128 tiles (1280 x 2048 -> 160 tiles)
// workgroups [0, grid_a) run a, the rest b (same 4 consumer + 4 producer waves, 4-stage rings).
template <typename T>
__global__ __launch_bounds__(512, 1) void gemm_ws_dual_kernel(const GemmParams a, const GemmParams b, const int grid_a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((int)blockIdx.x < grid_a) gemm_ws_body<T, 128, 160, 2, 2, 4, 4, false, 1>(a, blockIdx.x, gridDim.x, smem);
    else gemm_ws_body<T, 128, 128, 2, 2, 4, 4, false, 2>(b, blockIdx.x - grid_a, gridDim.x, smem);
}

template <typename T>
static int launch_ws_dual(const GemmParams& a, const GemmParams& b, hipStream_t stream) {
    GemmParams qa = a, qb = b;
    int ga, gb;
    xcd_partition(qa, 128, 160, &ga);
    xcd_partition(qb, 128, 128, &gb);
    const size_t smem = 4 * (size_t)(128 + 160) * GEMM_ROW_BYTES + 128 * 8;
    auto kern = gemm_ws_dual_kernel<T>;
    static DynLdsOnce lds_once;
    lds_once.ensure((const void*)kern, (int)smem);
    hipLaunchKernelGGL(kern, dim3(ga + gb), dim3(512), smem, stream, qa, qb, ga);
    return check_launch("gemm_ws_dual_kernel");
}

int gemm_ws_dual_launch(const GemmParams& a, const GemmParams& b, int dtype, hipStream_t stream) {
    if (dtype == IMH_DT_BF16) return launch_ws_dual<bf16_t>(a, b, stream);
    if (dtype == IMH_DT_F16) return launch_ws_dual<f16_t>(a, b, stream);
    set_error("gemm_ws_dual: unknown dtype %d", dtype);
    return IMH_ERR_DTYPE;
}
#else
int gemm_ws_dual_launch(const GemmParams&, const GemmParams&, int, hipStream_t) { return experimental_refused("the wave-specialised two-problem launch (gemm_dual variant 24128)"); }
#endif

template <typename T, int BM, int BN, int CM, int CN, int S, int NP, bool CONV, int LN, int OCC = 1>
static int launch_ws_ln(const GemmParams& p, hipStream_t stream) {
    GemmParams q = p;
    int tiles;
    xcd_partition(q, BM, BN, &tiles);
    const size_t smem = (size_t)S * (BM + BN) * GEMM_ROW_BYTES + (LN == 1 ? BM * 8 : (LN == 2 ? BN * 8 : 0));
    auto kern = gemm_ws_kernel<T, BM, BN, CM, CN, S, NP, CONV, LN, OCC>;
    static DynLdsOnce lds_once;
    lds_once.ensure((const void*)kern, (int)smem);
    hipLaunchKernelGGL(kern, dim3(tiles, p.splits, 1), dim3(64 * (CM * CN + NP)), smem, stream, q);
    return check_launch("gemm_ws_kernel");
}

template <typename T, int BM, int BN, int CM, int CN, int S, int NP, bool CONV, int OCC = 1>
static int launch_ws(const GemmParams& p, hipStream_t stream) {
    if constexpr (!CONV) {
        if (p.flags & GF_LN_ROW) return launch_ws_ln<T, BM, BN, CM, CN, S, NP, false, 1, OCC>(p, stream);   // (gemm_launch: ln_stats is there)
    }
    return launch_ws_ln<T, BM, BN, CM, CN, S, NP, CONV, 0, OCC>(p, stream);
}

template <typename T, int BM, int BN, bool CONV>
static int launch_kg2(const GemmParams& p, hipStream_t stream) {
    GemmParams q = p;
    int tiles;
    xcd_partition(q, BM, BN, &tiles);
    const size_t smem = 2 * 2 * (size_t)(BM + BN) * GEMM_ROW_BYTES;
    auto kern = gemm_kg2_kernel<T, BM, BN, CONV>;
    static DynLdsOnce lds_once;
    lds_once.ensure((const void*)kern, (int)smem);
    hipLaunchKernelGGL(kern, dim3(tiles, p.splits, 1), dim3(512), smem, stream, q);
    return check_launch("gemm_kg2_kernel");
}

template <typename T, int BM, int BN, int WM, int WN, int S, bool CONV>
static int launch_ring(const GemmParams& p, hipStream_t stream) {
    GemmParams q = p;
    int tiles;
    xcd_partition(q, BM, BN, &tiles);
    const size_t smem = (size_t)S * (BM + BN) * GEMM_ROW_BYTES;
    auto kern = gemm_ring_kernel<T, BM, BN, WM, WN, S, CONV>;
    static DynLdsOnce lds_once;
    lds_once.ensure((const void*)kern, (int)smem);
    hipLaunchKernelGGL(kern, dim3(tiles, p.splits, 1), dim3(64 * WM * WN), smem, stream, q);
    return check_launch("gemm_ring_kernel");
}

// variant codes (bm field of the config): 256 -> ring 256x{128,256}; 3128 / 3064 -> KG2 128x128 / 64x64.
// (128x128 rings with 3-4 stages at one workgroup per CU and 256x128 with 2 stages were measured and dropped.)
template <typename T, bool CONV>
static int ring_typed(const GemmParams& p, int bm, int bn, hipStream_t stream) {
    // small-tile rings (4 waves, several workgroups per CU): the N = 1280 projections are latency-bound in the two-stage
    // kernel (one 16 KB tile in flight per workgroup); 4000 + BM = 4-stage / 3-stage ring, 5000 + BM = 3-stage 64x64
    IMH_EXP_ONLY(if (bm == 4064 && bn == 64) return launch_ring<T, 64, 64, 2, 2, 4, CONV>(p, stream);)
    if (bm == 5064 && bn == 64) return launch_ring<T, 64, 64, 2, 2, 3, CONV>(p, stream);
    IMH_EXP_ONLY(if (bm == 4064 && bn == 128) return launch_ring<T, 64, 128, 2, 2, 3, CONV>(p, stream);)
    IMH_EXP_ONLY(if (bm == 4128 && bn == 64) return launch_ring<T, 128, 64, 2, 2, 3, CONV>(p, stream);)
    // exact 256-way tilings of the M = 2048 / 8192 Linear layers (every CU gets the same number of equal tiles):
    // 128 x 320 (8 waves, N = 10240 -> 512 tiles) and 64 x 160 (4 waves, N = 1280 -> 256 tiles, deep ring)
    IMH_EXP_ONLY(if (bm == 5256 && bn == 320) return launch_ring<T, 256, 320, 2, 2, 2, CONV>(p, stream);)      // 4 waves, 320 accumulator registers
    if (bm == 5258 && bn == 320) return launch_ring<T, 256, 320, 2, 4, 2, CONV>(p, stream);   // 8 waves, 160
    IMH_EXP_ONLY(if (bm == 6128 && bn == 320) return launch_ring<T, 128, 320, 4, 2, 2, CONV>(p, stream);)
    IMH_EXP_ONLY(if (bm == 6064 && bn == 160) return launch_ring<T, 64, 160, 4, 1, 4, CONV>(p, stream);)
    IMH_EXP_ONLY(if (bm == 7064 && bn == 160) return launch_ring<T, 64, 160, 4, 1, 3, CONV>(p, stream);)
    IMH_EXP_ONLY(if (bm == 256 && bn == 128) return launch_ring<T, 256, 128, 4, 2, 3, CONV>(p, stream);)
    IMH_EXP_ONLY(if (bm == 256 && bn == 256) return launch_ring<T, 256, 256, 2, 4, 2, CONV>(p, stream);)
    // wave-specialised 64 x 160, 4-stage ring, four consumer waves + two (1464) / four (2464) producer waves
    // (a fifth stage measured no different)
    if (bm == 1464 && bn == 160) return launch_ws<T, 64, 160, 2, 2, 4, 2, CONV>(p, stream);
    if (bm == 2464 && bn == 160) return launch_ws<T, 64, 160, 2, 2, 4, 4, CONV>(p, stream);   // four producer waves
    if (bm == 24128 && bn == 160) return launch_ws<T, 128, 160, 2, 2, 4, 4, CONV>(p, stream);  // M = 8192, N = 640: 256 tiles
    IMH_EXP_ONLY(if (bm == 24128 && bn == 128) return launch_ws<T, 128, 128, 2, 2, 4, 4, CONV>(p, stream);)
    // 128 x 160 with a two-slot ring (74 KB) and 128 registers: TWO workgroups per CU, out of phase with each other
    IMH_EXP_ONLY(if (bm == 22128 && bn == 160) return launch_ws<T, 128, 160, 2, 2, 2, 2, CONV, 2>(p, stream);)
    // 256 x 160, eight consumer waves (4 x 2, 64 x 80 each) + four producers, 3 stages (156 KB): N = 10240 -> 512 tiles
    if (bm == 23256 && bn == 160) return launch_ws<T, 256, 160, 4, 2, 3, 4, CONV>(p, stream);
    // 256 x 128 (64 x 64 wave tiles): N = 3840 -> 240 tiles where 256 x 160 makes 192 -- the one-launch [Q|K|V] of the L = 1024 layers (round 5)
    if (bm == 23256 && bn == 128) return launch_ws<T, 256, 128, 4, 2, 3, 4, CONV>(p, stream);
    IMH_EXP_ONLY(if (bm == 3128 && bn == 128) return launch_kg2<T, 128, 128, CONV>(p, stream);)
    if (bm == 3064 && bn == 64) return launch_kg2<T, 64, 64, CONV>(p, stream);
#ifndef IMH_EXPERIMENTAL
    if (bm == 4064 || bm == 4128 || bm == 5256 || bm == 6128 || bm == 6064 || bm == 7064 || bm == 256 || bm == 22128 || bm == 3128 || (bm == 24128 && bn == 128))
        return experimental_refused("this gemm_ring / gemm_ws / KG2 variant");
#endif
    set_error("gemm_ring: unsupported variant %dx%d", bm, bn);
    return IMH_ERR_ARG;
}

int gemm_ring_launch(const GemmParams& p, int dtype, int conv, int bm, int bn, hipStream_t stream) {
    // the sixteen-wave 256 x 320 tile of ff.net.0 (gemm_w16.hip, round 6)
    if (bm == 26256 && bn == 320) {
        if (conv) { set_error("gemm: variant 26256 x 320 is a Linear-layer form"); return IMH_ERR_ARG; }
        return gemm_w16_launch(p, dtype, stream);
    }

    if (dtype == IMH_DT_BF16) return conv ? ring_typed<bf16_t, true>(p, bm, bn, stream) : ring_typed<bf16_t, false>(p, bm, bn, stream);
    if (dtype == IMH_DT_F16) return conv ? ring_typed<f16_t, true>(p, bm, bn, stream) : ring_typed<f16_t, false>(p, bm, bn, stream);
    set_error("gemm_ring: unknown dtype %d", dtype);
    return IMH_ERR_DTYPE;
}

}  // namespace imh
