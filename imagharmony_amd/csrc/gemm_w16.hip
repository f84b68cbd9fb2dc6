// ff.net.0 (diffusers FeedForward / GEGLU behind norm3; call site ip_adapter/custom_pipelines.py:338-345) as ONE round of workgroups:
// 256 x 320 output tiles on SIXTEEN waves.
//
// The launch is M = 2048 x N = 10240 x K = 1280: on the 256 x 160 wave-specialised tile (gemm_ring.hip, variant 23256) it is 512 workgroups =
// two rounds at one per CU, each round paying its own prologue (2.7 us), K loop (19.5 us) and epilogue (5.3 us) -- 58.6 us warm, 22 % of the
// UNet forward (profiles/r06_ws_timeline.txt).  The K loop of those launches runs at what a CU's vector-memory path delivers (53 KB per ~2000
// cycles), so what shortens it is fewer operand bytes per FLOP: a 256 x 320 tile stages 73.7 KB per K tile for twice the MFMAs (36.9 KB per
// 256 x 160 equivalent, -30 %), and M x N / (256 x 320) = 256 workgroups is exactly one round.
//   * 16 waves as 4 (M) x 4 (N), wave tile 64 x 80 = 4 x 5 fragments of v_mfma_f32_16x16x32 -- the wave tile of the 23256 kernel's consumers --
//     80 accumulator registers; 1024 threads hold a CU's whole register file at 128 registers each, so there are no dedicated producer
//     waves and no second fragment set: four waves per SIMD hide each other's LDS latency and LDS-DMA issue instead.
//   * two 73.7-KB stages (147 KB): tile t + 1 is in flight while tile t is consumed, one vmcnt(0) + s_barrier per K tile.
//   * operands, swizzles, fragment maps and the folded-LayerNorm + GEGLU arithmetic are the 23256 kernel's (same order of operations per
//     output element: results are bit-identical to it).
// Only the launch it was built for: row-form folded LayerNorm with handed-over statistics (+ GEGLU), whole tiles, no residual / row-add.
#include "imh_common.h"
#include "imh_kernels.h"
#include "imh_gemm_epilogue.h"
#include "imh_lnstats.h"

namespace imh {

int g_w16_pf = 1;        // imh_debug_set key 7: 1 (default) = the next launch's weights prefetched inside the K loop, 0 = behind the epilogue (A/B)

#ifndef W16_TIMING
#define W16_TIMING 0
#endif

// (Measured and not kept, profiles/r06_w16_probe.txt: K tiles of 32 in FOUR 36.9-KB stages with counted vmcnt -- three half tiles in flight
// across every barrier instead of one tile issued behind the barrier -- 59.2 vs 54.1 us: the K loop takes ~2250 cycles per 32 k either way
// (1280 of MFMA).  It is not the LDS-DMA round trip: sixteen 64 x 80 wave tiles read every staged byte four times (144 KB of fragment reads per
// 36.9 KB staged, twice the 23256 kernel's), each wave has registers for ONE weight fragment of look-ahead, and four waves per SIMD do not
// cover the LDS latency that leaves exposed.)
template <typename T>
__global__ __launch_bounds__(1024, 4) void gemm_w16_kernel(const GemmParams p) {
    constexpr int BM = 256, BN = 320, TM = 64, TN = 80, FM = 4, FN = 5;
    constexpr int XT_BYTES = BM * GEMM_ROW_BYTES;
    constexpr int STAGE = (BM + BN) * GEMM_ROW_BYTES;       // 73,728 B
    constexpr int NI = (BM + BN) / 8;                       // 72 LDS-DMA wave instructions per K tile: 32 token pieces, 40 weight pieces
    typedef typename Vec<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
#if W16_TIMING
    const unsigned long long ts_entry = __builtin_amdgcn_s_memrealtime();
#endif
    int m0, n0;
    if (!xcd_tile<BM, BN>(p, blockIdx.x, m0, n0)) return;
    const int nkt = p.K / GEMM_BK;

    // ---- staging: wave w issues pieces w, w + 16, w + 32, w + 48 (and w + 64 for w < 8); piece q = tile rows 8 q .. 8 q + 7, rows [0, 256) =
    //      tokens, [256, 576) = weights.  The launcher guarantees whole tiles: no bounds, one base pointer per operand.
    const int r0 = wave * 8 + (lane >> 3);                  // row of piece `wave`; piece wave + 16 k adds 128 k rows
    const unsigned char* const xb = (const unsigned char*)p.X + ((size_t)(m0 + r0) * p.ldx) * sizeof(T) + stage_chunk_x(r0, lane) * 16;
    const size_t xstride = (size_t)128 * p.ldx * sizeof(T);
    const unsigned char* const wb = (const unsigned char*)p.W + (size_t)n0 * p.ldw * sizeof(T);
    unsigned woffk[3];                                      // weight pieces wave + 32, + 48, + 64 -> weight rows r0, r0 + 128, r0 + 256 (the swizzle differs per piece)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int row = min(r0 + 128 * k, BN - 1);
        woffk[k] = (unsigned)(row * p.ldw * (int)sizeof(T)) + stage_chunk_w(row, lane, FN) * 16;
    }
    auto stage = [&](int slot, int kt) {
        unsigned char* st = smem + slot * STAGE + wave * 1024;
        const size_t ko = (size_t)kt * (GEMM_BK * sizeof(T));
        glds16(xb + ko, st);
        glds16(xb + xstride + ko, st + 16 * 1024);
        glds16(wb + woffk[0] + ko, st + 32 * 1024);
        glds16(wb + woffk[1] + ko, st + 48 * 1024);
        if (wave < NI - 64) glds16(wb + woffk[2] + ko, st + 64 * 1024);
    };

    int xoff[2], woff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int xr = wm * TM + (lane & 15);
        const int wr = wn * TN + w_frag_row(lane & 15, 0, FN);
        xoff[kk] = tile_off(xr, kk * 4 + (lane >> 4), swz_x(xr));
        woff[kk] = XT_BYTES + tile_off(wr, kk * 4 + (lane >> 4), swz_w(wr, FN));
    }

    stage(0, 0);
    // (mean, rstd) of the tile's 256 token rows, merged from the producer GEMM's slot partials (imh_lnstats.h) beside tile 0's flight
    if (tid < BM) {
        const f32x2s mr = merge_row_stats(p.ln_stats, m0 + tid, p.ln_slots, p.K, p.ln_eps);
        *(f32x2s*)(smem + 2 * STAGE + tid * 8) = mr;
    }
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#if W16_TIMING
    const unsigned long long ts_loop0 = __builtin_amdgcn_s_memrealtime();
#endif
    // The next launch's weights (ff.net.2: 13 MB) -- this kernel has no producer waves whose tail prefetch would run beside the epilogue, and a
    // prefetch behind the epilogue's stores cost the launch 3.3 us (profiles/r06_forward_ab_prefetch.log, `pf_off`).  So this workgroup's slice travels INSIDE
    // the K loop instead (g_w16_pf, imh_debug_set key 7; 0 = the tail form.  ff.net.0 67.6 -> 64.3 us in the forward, bit-identical; 603.1 ->
    // 600.3 ms per denoise over alternating bench.py processes: profiles/r06_forward_ab_w16_pf_loop.log, r06_bench_ab_w16_pf_loop.txt): one LDS-DMA piece (1 KB, landing in a scratch KB of LDS nobody
    // reads) per wave behind every PFS-th tile's operand pieces; it is the youngest entry of the wave's queue, so the next tile's wait leaves
    // exactly it in flight (vmcnt(1)) and the wait after that retires it with the operands it precedes.
    const bool pf_loop = p.early_res == 2 && p.pf_ptr != nullptr;       // (early_res carries the mode: the launcher sets 2)
    const unsigned pf_per = pf_loop ? ((((p.pf_bytes + gridDim.x - 1) / gridDim.x) + 1023u) & ~1023u) : 0u;     // this workgroup's slice, whole KB
    const unsigned pf_s0 = blockIdx.x * pf_per;
    const int pf_n = pf_loop ? (int)((min(p.pf_bytes, pf_s0 + pf_per) > pf_s0 ? min(p.pf_bytes, pf_s0 + pf_per) - pf_s0 : 0u) >> 10) : 0;     // KB pieces
    const int pf_rounds = (pf_n + 15) >> 4;                             // tiles that carry a piece per wave
    const int PFS = pf_rounds > 0 ? max(1, (nkt - 1) / pf_rounds) : 1;
    unsigned char* const pf_lds = smem + 2 * STAGE + 256 * 8;
    bool pf_in_flight = false;
    for (int t = 0; t < nkt; ++t) {
        if (pf_in_flight) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");     // tile t has landed; the prefetch piece issued behind it stays in flight
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                  // this wave's pieces of tile t have landed (and its statistics row is written)
        __builtin_amdgcn_s_barrier();                                    // ... everyone's; every wave is past its reads of tile t - 1
        asm volatile("" ::: "memory");
        if (t + 1 < nkt) stage((t + 1) & 1, t + 1);
        pf_in_flight = false;
        if (pf_loop && t + 1 < nkt && t % PFS == 0) {
            const int q = (t / PFS) * 16 + wave;
            if (q < pf_n) {                                              // (wave-uniform)
                glds16((const unsigned char*)p.pf_ptr + pf_s0 + (size_t)q * 1024 + lane * 16, pf_lds);
                pf_in_flight = true;
            }
        }
        const unsigned char* st = smem + (t & 1) * STAGE;
        // 128 registers per wave: 80 accumulators leave room for the four token fragments of a k step and TWO weight fragments -- the weight
        // fragment of column block j + 1 is read while the four MFMAs of block j issue; the fences keep hipcc from hoisting all nine reads of
        // a k step (it then spills 92 registers)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            v8 xf[FM], wf[2];
#pragma unroll
            for (int i = 0; i < FM; ++i) xf[i] = *(const v8*)(st + xoff[kk] + i * 16 * GEMM_ROW_BYTES);
            wf[0] = *(const v8*)(st + woff[kk]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if (j + 1 < FN) wf[(j + 1) & 1] = *(const v8*)(st + woff[kk] + (j + 1) * 4 * GEMM_ROW_BYTES);
#pragma unroll
                for (int i = 0; i < FM; ++i) acc[i][j] = mfma16(wf[j & 1], xf[i], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("" ::: "memory");
    }
#if W16_TIMING
    const unsigned long long ts_loop1 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- epilogue: y = rstd_m (acc - mean_m s_n) + c_n, then out[2k], out[2k + 1] = value * gelu(gate) per quad (GEGLU) -- fragment by
    //      fragment (its four (s_n, c_n) pairs fetched once for the four rows), a row's ten outputs packed and stored as 16 + 4 bytes ----
    const int nb = n0 + wn * TN + (lane >> 4) * (4 * FN);
    const float* ex = (const float*)(smem + 2 * STAGE);
    float mean[FM], rstd[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int r = wm * TM + i * 16 + (lane & 15);
        mean[i] = ex[r * 2]; rstd[i] = ex[r * 2 + 1];
    }
    const bool geglu = p.flags & GF_GEGLU;
    if (geglu) {
        typename Pk2<T>::t outp[FM][FN];
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const f32x4 s4 = *(const f32x4*)(p.ln_s + nb + 4 * j), c4 = *(const f32x4*)(p.ln_c + nb + 4 * j);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                float v[4], o[2];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fma_nopk(rstd[i], fma_nopk(-mean[i], s4[e], acc[i][j][e]), c4[e]);      // (no bias: fold_ln puts it into c_n)
                geglu_quads<4>(v, o);
                outp[i][j][0] = from_f32<T>(o[0]);
                outp[i][j][1] = from_f32<T>(o[1]);
            }
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = m0 + wm * TM + i * 16 + (lane & 15);
            T* y = (T*)p.Y + (size_t)m * p.ldy + (nb >> 1);
            typedef typename Pk2<T>::t pk2;
            struct alignas(4) Row { pk2 q[FN]; } row;
#pragma unroll
            for (int j = 0; j < FN; ++j) row.q[j] = outp[i][j];
            // 10 outputs = 20 B at a 4-B-aligned address (nb / 2 is a multiple of 10 elements): five 4-B pieces would be five store instructions;
            // stv's 16 + 4 split needs the first piece 16-B aligned, which (nb / 2) * 2 B = 20 k B is not -- so: 4-B pieces, merged by the compiler where it can
#pragma unroll
            for (int j = 0; j < FN; ++j) *(pk2*)(y + 2 * j) = row.q[j];
        }
    } else {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = m0 + wm * TM + i * 16 + (lane & 15);
            float v[4 * FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const f32x4 s4 = *(const f32x4*)(p.ln_s + nb + 4 * j), c4 = *(const f32x4*)(p.ln_c + nb + 4 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * j + e] = fma_nopk(rstd[i], fma_nopk(-mean[i], s4[e], acc[i][j][e]), c4[e]);
            }
            stv<T, 4 * FN>((T*)p.Y + (size_t)m * p.ldy + nb, v);
        }
    }
#if W16_TIMING
    if (tid == 0 && p.pf_ptr && p.pf_bytes == 0xfeed) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* dbg = (unsigned long long*)p.pf_ptr + 8 + (size_t)blockIdx.x * 4;
        dbg[0] = ts_entry; dbg[1] = ts_loop0; dbg[2] = ts_loop1; dbg[3] = __builtin_amdgcn_s_memrealtime();
    }
#else
    if (!pf_loop) tail_prefetch(p.pf_ptr, p.pf_bytes, blockIdx.x, gridDim.x, tid, 1024);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The same 256 x 320 tile on EIGHT fat waves (round 6, third structural attempt at ff.net.0; imh_debug_set key 9).  The sixteen-wave kernel above
// and the 256 x 160 wave-specialised kernel share the 64 x 80 wave tile: 18 fragment reads per 40 MFMAs, and a wave's reads and MFMAs add up
// rather than overlap (profiles/r05_lds_port_microbench.csv: reads + MFMA 521 cycles, MFMA alone 333, reads alone 216) -- both run ~4500 cycles
// per 64-k tile of this workgroup tile for 2560 of MFMA.  Eight waves as 2 (M) x 4 (N) own 128 x 80 each: 13 reads per 40 MFMAs (208 per K tile
// instead of 288), half the barrier arrivals, half the LDS-DMA issuers.  Measured: K loop 40.8 -> 34.8 us, launch 57.2 -> 50.8 us warm.
//   * 160 accumulator registers + the weight fragments of BOTH k steps (40) + four rotating token-fragment registers (16): 256 registers, two
//     waves per SIMD, no producer waves -- every wave issues nine of the 72 LDS-DMA pieces of a K tile, one per 5-MFMA block.
//   * one K tile = 16 blocks of 5 MFMAs (block = k step, 16-row token fragment); the block's token fragment was read three blocks earlier, the
//     weight fragments of k step 1 under k step 0's blocks.  ONE barrier per K tile, at the start of block 13: by then every read of the tile
//     has been issued (lgkmcnt(0)), this wave's pieces of tile t + 1 have landed (vmcnt(0)); behind it the tile's slot is refilled with tile
//     t + 2 (pieces 0-2 in blocks 13-15, 3-8 in blocks 0-5 of the next tile) and the last three blocks' MFMAs run beside the first reads of
//     tile t + 1.  No branch in the loop body: past the end the pieces re-fetch the last tile into a slot nobody reads.
//   * same operands, swizzles, fragment maps, accumulation order and epilogue arithmetic as the kernels above: bit-identical results.
int g_w16_form = 1;      // imh_debug_set key 9: 1 (default) = eight waves of 128 x 80, 0 = sixteen waves of 64 x 80 (A/B)
template <typename T>
__global__ __launch_bounds__(512, 2) void gemm_f8_kernel(const GemmParams p) {
    constexpr int BM = 256, BN = 320, TM = 128, TN = 80, FM = 8, FN = 5;
    constexpr int XT_BYTES = BM * GEMM_ROW_BYTES;
    constexpr int STAGE = (BM + BN) * GEMM_ROW_BYTES;       // 73,728 B
#ifndef F8_AHEAD
#define F8_AHEAD 3      // (2 measured the same: 54.9 / 67.5 / 193.7 vs 53.1 / 66.9 / 191.6 us on the three ff.net.0 shapes)
#endif
    constexpr int NB = 2 * FM, AHEAD = F8_AHEAD, BAR = NB - AHEAD; // blocks per K tile, token-fragment look-ahead, the block that starts with the barrier
    constexpr int NPC = 9;                                  // LDS-DMA pieces per wave and K tile (72 / 8)
    typedef typename Vec<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
#if W16_TIMING
    const unsigned long long ts_entry = __builtin_amdgcn_s_memrealtime();
#endif
    int m0, n0;
    if (!xcd_tile<BM, BN>(p, blockIdx.x, m0, n0)) return;
    const int nkt = p.K / GEMM_BK;

    // ---- staging: wave w issues pieces w + 8 k, k < 9; piece q = tile rows 8 q .. 8 q + 7, rows [0, 256) = tokens (k < 4: rows r0 + 64 k, the
    //      swizzle has period 16), [256, 576) = weights (k >= 4: weight rows r0 + 64 (k - 4), the swizzle differs per piece).  32-bit offsets
    //      from the two uniform bases (the launcher checks the operands stay below 4 GB).
    const int r0 = wave * 8 + (lane >> 3);
    const unsigned xofs = (unsigned)((m0 + r0) * p.ldx * (int)sizeof(T)) + stage_chunk_x(r0, lane) * 16;
    const unsigned xstride = (unsigned)(64 * p.ldx * (int)sizeof(T));
    unsigned wofs[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int row = r0 + 64 * k;
        wofs[k] = (unsigned)((n0 + row) * p.ldw * (int)sizeof(T)) + stage_chunk_w(row, lane, FN) * 16;
    }
    // (issuing the five weight pieces -- cold in the forward -- ahead of the token pieces measured the same: 67.7 / 66.1 vs 67.2 / 64.6 us cold)
    auto piece = [&](auto K_, int slot, int kt) {          // piece K_ of this wave of K tile kt -> its slot
        constexpr int k = decltype(K_)::value;
        unsigned char* d = smem + slot * STAGE + (wave + 8 * k) * 1024;
        const unsigned ko = (unsigned)kt * (GEMM_BK * (unsigned)sizeof(T));
        if constexpr (k < 4) glds16((const unsigned char*)p.X + (size_t)(xofs + k * xstride + ko), d);
        else glds16((const unsigned char*)p.W + (size_t)(wofs[k - 4] + ko), d);
    };

    int xoff[2], woff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int xr = wm * TM + (lane & 15);
        const int wr = wn * TN + w_frag_row(lane & 15, 0, FN);
        xoff[kk] = tile_off(xr, kk * 4 + (lane >> 4), swz_x(xr));
        woff[kk] = XT_BYTES + tile_off(wr, kk * 4 + (lane >> 4), swz_w(wr, FN));
    }

    // prologue: tile 0 entirely, pieces 0-2 of tile 1 (as blocks 13-15 of "tile -1" would have issued them)
    piece(std::integral_constant<int, 0>{}, 0, 0); piece(std::integral_constant<int, 1>{}, 0, 0); piece(std::integral_constant<int, 2>{}, 0, 0);
    piece(std::integral_constant<int, 3>{}, 0, 0); piece(std::integral_constant<int, 4>{}, 0, 0); piece(std::integral_constant<int, 5>{}, 0, 0);
    piece(std::integral_constant<int, 6>{}, 0, 0); piece(std::integral_constant<int, 7>{}, 0, 0); piece(std::integral_constant<int, 8>{}, 0, 0);
    {
        const int k1 = min(1, nkt - 1);
        piece(std::integral_constant<int, 0>{}, 1, k1); piece(std::integral_constant<int, 1>{}, 1, k1);
        if constexpr (AHEAD >= 3) piece(std::integral_constant<int, 2>{}, 1, k1);
    }
    // (mean, rstd) of the tile's 256 token rows, merged from the producer GEMM's slot partials (imh_lnstats.h) beside tile 0's flight
    if (tid < BM) {
        const f32x2s mr = merge_row_stats(p.ln_stats, m0 + tid, p.ln_slots, p.K, p.ln_eps);
        *(f32x2s*)(smem + 2 * STAGE + tid * 8) = mr;
    }
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Fragment reads as inline asm with HAND-COUNTED lgkmcnt: left to hipcc, every third block waited lgkmcnt(0) -- for the read it had issued
    // one MFMA earlier -- instead of the lgkmcnt(2) that leaves the two younger token fragments in flight.  In-order LDS returns; at the start of
    // block b the fragment of block b (read at b - 3) must be there, the reads of blocks b - 2 and b - 1 may fly: one token fragment each, plus
    // the five weight fragments read in block 0 (k step 1) and in block BAR (k step 0 of the next tile).
    v8 wf[2][FN], xr[4];
    const unsigned lds0 = (unsigned)(size_t)smem;           // (flat -> LDS offset: the low 32 bits)
    auto rd16 = [&](v8& dst, unsigned addr, auto OFF) {
        constexpr int off = decltype(OFF)::value;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off));
    };
    auto ldw = [&](auto KK, unsigned sta) {                  // sta = LDS address of the stage
        constexpr int kk = decltype(KK)::value;
        const unsigned a = sta + woff[kk];
        rd16(wf[kk][0], a, std::integral_constant<int, 0 * 4 * GEMM_ROW_BYTES>{}); rd16(wf[kk][1], a, std::integral_constant<int, 1 * 4 * GEMM_ROW_BYTES>{});
        rd16(wf[kk][2], a, std::integral_constant<int, 2 * 4 * GEMM_ROW_BYTES>{}); rd16(wf[kk][3], a, std::integral_constant<int, 3 * 4 * GEMM_ROW_BYTES>{});
        rd16(wf[kk][4], a, std::integral_constant<int, 4 * 4 * GEMM_ROW_BYTES>{});
    };
    auto ldx = [&](auto BI, unsigned sta) {
        constexpr int bi = decltype(BI)::value;
        rd16(xr[bi % 4], sta + xoff[bi / FM], std::integral_constant<int, (bi % FM) * 16 * GEMM_ROW_BYTES>{});
    };
    auto mm1 = [&](auto BI, auto J) {
        constexpr int bi = decltype(BI)::value, j = decltype(J)::value;
        acc[bi % FM][j] = mfma16(wf[bi / FM][j], xr[bi % 4], acc[bi % FM][j]);
    };
    const std::integral_constant<int, 0> K0{};
    const std::integral_constant<int, 1> K1{};
    const std::integral_constant<int, 0> J0{}; const std::integral_constant<int, 1> J1{}; const std::integral_constant<int, 2> J2{};
    const std::integral_constant<int, 3> J3{}; const std::integral_constant<int, 4> J4{};
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(AHEAD) : "memory");     // this wave's pieces of tile 0 have landed (and its statistics row is written)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ldw(K0, lds0);
    ldx(std::integral_constant<int, 0>{}, lds0); ldx(std::integral_constant<int, 1>{}, lds0);
    if constexpr (AHEAD >= 3) ldx(std::integral_constant<int, 2>{}, lds0);

    // the next launch's weights (ff.net.2) travel inside the K loop, as in the sixteen-wave kernel: one extra 1-KB piece per wave behind block 5's
    // operand piece of every PFS-th tile, left in flight across that tile's barrier (vmcnt(1)) and retired with the next tile's pieces
    const bool pf_loop = p.early_res == 2 && p.pf_ptr != nullptr && !W16_TIMING;
    const unsigned pf_per = pf_loop ? ((((p.pf_bytes + gridDim.x - 1) / gridDim.x) + 1023u) & ~1023u) : 0u;
    const unsigned pf_s0 = blockIdx.x * pf_per;
    const int pf_n = pf_loop ? (int)((min(p.pf_bytes, pf_s0 + pf_per) > pf_s0 ? min(p.pf_bytes, pf_s0 + pf_per) - pf_s0 : 0u) >> 10) : 0;
    const int pf_rounds = (pf_n + 7) >> 3;
    const int PFS = pf_rounds > 0 ? max(1, (nkt - 1) / pf_rounds) : 1;
    unsigned char* const pf_lds = smem + 2 * STAGE + 256 * 8;
#if W16_TIMING
    const unsigned long long ts_loop0 = __builtin_amdgcn_s_memrealtime();
#endif

    for (int t = 0; t < nkt; ++t) {
        const unsigned st = lds0 + (t & 1) * STAGE, stn = lds0 + ((t + 1) & 1) * STAGE;
        const int kt1 = min(t + 1, nkt - 1), kt2 = min(t + 2, nkt - 1);
        const int pfq = (t / PFS) * 8 + wave;
        const bool pf_now = pf_loop && t + 1 < nkt && t % PFS == 0 && pfq < pf_n;       // (wave-uniform)
        auto block = [&](auto BI) {
            constexpr int bi = decltype(BI)::value;
            if constexpr (bi == BAR) {
                if (pf_now) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // every read of tile t has completed; this wave's pieces of tile t + 1 have landed
                __builtin_amdgcn_s_barrier();                                    // ... everyone's
                asm volatile("" ::: "memory");
            } else if constexpr ((bi >= 1 && bi < AHEAD) || (bi > BAR && bi < BAR + AHEAD)) {
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(AHEAD - 1 + 5) : "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(AHEAD - 1) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            mm1(BI, J0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (bi == 0) ldw(K1, st);
            if constexpr (bi == BAR) ldw(K0, stn);
            if constexpr (bi + AHEAD < NB) ldx(std::integral_constant<int, bi + AHEAD>{}, st);
            else ldx(std::integral_constant<int, bi + AHEAD - NB>{}, stn);      // (bi >= BAR: behind the barrier)
            __builtin_amdgcn_sched_barrier(0);
            mm1(BI, J1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (bi < NPC - AHEAD) piece(std::integral_constant<int, bi + AHEAD>{}, (t + 1) & 1, kt1);        // pieces 3-8 of tile t + 1
            if constexpr (bi >= BAR) piece(std::integral_constant<int, bi - BAR>{}, t & 1, kt2);               // pieces 0-2 of tile t + 2 -> the slot just freed
            if constexpr (bi == NPC - AHEAD)
                if (pf_now) glds16((const unsigned char*)p.pf_ptr + pf_s0 + (size_t)pfq * 1024 + lane * 16, pf_lds);
            __builtin_amdgcn_sched_barrier(0);
            mm1(BI, J2); mm1(BI, J3); mm1(BI, J4);      // (s_setprio(1) over these: no change, 55.0 / 56.8 vs 55.7 / 55.4 us)
            __builtin_amdgcn_sched_barrier(0);
        };
        block(std::integral_constant<int, 0>{}); block(std::integral_constant<int, 1>{}); block(std::integral_constant<int, 2>{});
        block(std::integral_constant<int, 3>{}); block(std::integral_constant<int, 4>{}); block(std::integral_constant<int, 5>{});
        block(std::integral_constant<int, 6>{}); block(std::integral_constant<int, 7>{}); block(std::integral_constant<int, 8>{});
        block(std::integral_constant<int, 9>{}); block(std::integral_constant<int, 10>{}); block(std::integral_constant<int, 11>{});
        block(std::integral_constant<int, 12>{}); block(std::integral_constant<int, 13>{}); block(std::integral_constant<int, 14>{});
        block(std::integral_constant<int, 15>{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the re-fetched pieces and the look-ahead reads behind the last tile have landed: LDS and fragment registers may be reused
#if W16_TIMING
    const unsigned long long ts_loop1 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- epilogue (the sixteen-wave kernel's, eight row fragments per wave) ----
    const int nb = n0 + wn * TN + (lane >> 4) * (4 * FN);
    const float* ex = (const float*)(smem + 2 * STAGE);
    const bool geglu = p.flags & GF_GEGLU;
    typedef typename Pk2<T>::t pk2;
#pragma unroll
    for (int h = 0; h < 2; ++h) {                           // four row fragments at a time (the packed outputs of eight would not fit beside the accumulators)
        float mean[4], rstd[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = wm * TM + (4 * h + i) * 16 + (lane & 15);
            mean[i] = ex[r * 2]; rstd[i] = ex[r * 2 + 1];
        }
        if (geglu) {
            pk2 outp[4][FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const f32x4 s4 = *(const f32x4*)(p.ln_s + nb + 4 * j), c4 = *(const f32x4*)(p.ln_c + nb + 4 * j);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v[4], o[2];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fma_nopk(rstd[i], fma_nopk(-mean[i], s4[e], acc[4 * h + i][j][e]), c4[e]);      // (no bias: fold_ln puts it into c_n)
                    geglu_quads<4>(v, o);
                    outp[i][j][0] = from_f32<T>(o[0]);
                    outp[i][j][1] = from_f32<T>(o[1]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + wm * TM + (4 * h + i) * 16 + (lane & 15);
                T* y = (T*)p.Y + (size_t)m * p.ldy + (nb >> 1);
#pragma unroll
                for (int j = 0; j < FN; ++j) *(pk2*)(y + 2 * j) = outp[i][j];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + wm * TM + (4 * h + i) * 16 + (lane & 15);
                float v[4 * FN];
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const f32x4 s4 = *(const f32x4*)(p.ln_s + nb + 4 * j), c4 = *(const f32x4*)(p.ln_c + nb + 4 * j);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * j + e] = fma_nopk(rstd[i], fma_nopk(-mean[i], s4[e], acc[4 * h + i][j][e]), c4[e]);
                }
                stv<T, 4 * FN>((T*)p.Y + (size_t)m * p.ldy + nb, v);
            }
        }
    }
#if W16_TIMING
    if (tid == 0 && p.pf_ptr && p.pf_bytes == 0xfeed) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* dbg = (unsigned long long*)p.pf_ptr + 8 + (size_t)blockIdx.x * 4;
        dbg[0] = ts_entry; dbg[1] = ts_loop0; dbg[2] = ts_loop1; dbg[3] = __builtin_amdgcn_s_memrealtime();
    }
#else
    if (!pf_loop) tail_prefetch(p.pf_ptr, p.pf_bytes, blockIdx.x, gridDim.x, tid, 512);
#endif
}

template <typename T>
static int launch_w16(const GemmParams& p, hipStream_t stream) {
    GemmParams q = p;
    q.early_res = g_w16_pf ? 2 : 0;
    int tiles;
    xcd_partition(q, 256, 320, &tiles);
    const int smem = 2 * (256 + 320) * GEMM_ROW_BYTES + 256 * 8 + 1024;      // + the prefetch pieces' scratch KB
    if (g_w16_form == 1 && (size_t)p.M * p.ldx * sizeof(T) < (1ull << 31) && (size_t)p.N * p.ldw * sizeof(T) < (1ull << 31)) {
        auto kern = gemm_f8_kernel<T>;
        static DynLdsOnce lds_once8;
        lds_once8.ensure((const void*)kern, smem);
        hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), smem, stream, q);
        return check_launch("gemm_f8_kernel");
    }
    auto kern = gemm_w16_kernel<T>;
    static DynLdsOnce lds_once;
    lds_once.ensure((const void*)kern, smem);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(1024), smem, stream, q);
    return check_launch("gemm_w16_kernel");
}

// variant 26256 x 320
int gemm_w16_launch(const GemmParams& p, int dtype, hipStream_t stream) {
    if (!(p.flags & GF_LN_ROW) || (p.flags & ~(GF_LN_ROW | GF_GEGLU)) || !p.ln_stats || p.bias || p.residual || p.rowadd || p.splits > 1 || p.X2 || p.Yt ||
        p.ln_stats_out || p.gn_out || p.M % 256 || p.N % 320 || (p.ldy & 1) || (p.ldx & 7) || (p.ldw & 7)) {
        set_error("gemm 26256 x 320 (sixteen-wave 256 x 320 tile): row-form folded LayerNorm with handed-over statistics (+ GEGLU) only, whole tiles "
                  "(M %% 256, N %% 320), no bias (fold_ln folds it into ln_c) / residual / row-add / split-K / statistics epilogue (M=%d N=%d flags=%d)", p.M, p.N, p.flags);
        return IMH_ERR_ARG;
    }
    if (dtype == IMH_DT_BF16) return launch_w16<bf16_t>(p, stream);
    if (dtype == IMH_DT_F16) return launch_w16<f16_t>(p, stream);
    set_error("gemm_w16: unknown dtype %d", dtype);
    return IMH_ERR_DTYPE;
}

}  // namespace imh
