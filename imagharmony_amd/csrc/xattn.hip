// Fused QKV + image-prompt cross-attention for gfx950 -- the whole attention side of IPAttnProcessor2_0
// (ip_adapter/attention_processor.py:396-450) in ONE launch:
//
//     q   = to_q( LayerNorm(x) )                      (:396; diffusers' norm2 folded in, optional)
//     O   = softmax(q K^T / 8) V  [ + scale * softmax(q K_ip^T / 8) V_ip ]        (:416-450)
//
// K / V of the text tokens and of the image-prompt tokens do not depend on the denoise step (nor on the latent):
// they arrive as LDS-stageable caches (K row-major, V transposed), projected once per image.  `to_out` (:453) needs
// every head of a token and stays the next GEMM launch.
//
// One workgroup = (batch, head, 128 queries) = 4 waves x 32 queries.
//   prologue   Q^T[64 d, 32 q] per wave = Wq_h[64, C] . X[32 q, C]^T on v_mfma_f32_32x32x16: the head's weight
//              slice is the MFMA A operand (shared by the 4 waves through LDS), the wave's own 32 token rows the B
//              operand (wave-private LDS rows), K loop over C in 64-wide tiles, two LDS-DMA stages.  A query block
//              is re-read by the H heads (L2 hits), never by redundant arithmetic: the Q tile of a (batch, head,
//              128-query) item is a 128 x 64 x C GEMM with no overlap between items.
//   LayerNorm  the token rows' (mean, rstd) are merged from the (sum, M2) slot partials the GEMM that wrote the rows left
//              behind (imh_lnstats.h; never E[x^2] - mean^2), beside the first K tile's flight;
//              q = rstd * (acc - mean * s_d) + c_d with Wq pre-scaled by gamma -- norm2 never materialises.
//   hand-over  the accumulator layout of a 32x32 MFMA block IS the B-operand layout of the swapped QK^T product
//              (S^T = K Q^T) once the head dims inside every 16-group are ordered [0-3, 8-11, 4-7, 12-15]; the K
//              caches are written in that order by their projection GEMM (IMH_GF_VT_PERM), so Q goes from
//              accumulators to MFMA operands through a cvt only -- no LDS round trip, no global round trip.
//   attention  the online-softmax key loop shared with attention.hip (imh_attn_core.h): text pass, image-prompt
//              pass, text + scale * ip formed in registers, one coalesced store.
// Roofline: the prologue is a GEMM (MFMA-bound, 2*B*L*C^2 FLOP per call); the key loop over 77 + T keys is short.
#include "imh_attn_core.h"
#include "imh_lnstats.h"

namespace imh {

int g_xattn_mode = 0;   // imh_debug_set key 3 (test / A-B only, not thread-safe): 0 auto (by shape: 10 or 1), 1 one head per workgroup, 9 the same without half items,
                        // 10 the wide form (five heads per workgroup, round 6),
                        // 2 two heads, eight do-everything waves, 3 two heads + two producer waves, 4 two heads + four producer waves;
                        // 6 / 7 / 8 = 2 / 3 / 4 without the resident key tiles


constexpr int XQ_STAGE = 128 * 128 + 64 * 128;     // X tile (128 rows) + Wq tile (64 rows), 128 B per row
// items per XCD dealt as two 64-query halves (host and device agree on the grid: 8 * (per + split) workgroups)
__host__ __device__ __forceinline__ int xattn_split(const int per) { return (per > 32 && per < 64) ? per - 32 : 0; }

// XA_TIMING (tools/xattn_phase_probe.py only): every workgroup stamps entry / end of the to_q K loop / end of the key loops / exit on the
// chip-wide 100 MHz counter into p.pf_ptr[item * 4 ..] (instead of prefetching) -- the launch as a time line per workgroup
#ifndef XA_TIMING
#define XA_TIMING 0
#endif
#if XA_TIMING
#define XA_TICK(i) do { asm volatile("s_nop 0" ::: "memory"); const unsigned long long tn_ = __builtin_readcyclecounter(); tacc[i] += tn_ - tm0; tm0 = tn_; } while (0)
#else
#define XA_TICK(i) do {} while (0)
#endif
#ifndef XA_ABL
#define XA_ABL 0       // tools/xattn_phase_probe.py only (wrong results by design): 1 no MFMAs, 2 no fragment reads either, 4 no LDS-DMA in the loop, 8 no barrier
#endif

// LNQ: norm2 folded into to_q, row statistics handed over (xp.ln_stats)
template <typename T, int NPASS, bool LNQ>
__global__ __launch_bounds__(256, NPASS == 1 ? 3 : 2) void xattn_kernel(const XAttnParams xp) {
    constexpr int NW = 4;
    typedef typename Vec<T>::v8 v8;
    static_assert(2 * XQ_STAGE <= ATT_STAGES * 2 * ATT_TILE_BYTES, "the projection stages alias the K / V^T ring");
    __shared__ __attribute__((aligned(16))) unsigned char smem[ATT_STAGES * 2 * ATT_TILE_BYTES];
    const AttnParams& p = xp.a;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    // same XCD-aware head-major item order as attn_kernel: one XCD sees few heads (their Wq slices, K / V^T) and
    // streams the query blocks
    const int gx = (p.Lq + 32 * NW - 1) / (32 * NW);
    const int items = gx * p.H * p.B;
    const int per = (items + 7) >> 3;
    // Round 5: when an XCD's 32 CUs get between 32 and 64 items (the C = 1280 / L = 1024 layers at UNet batch 2: 40), the items beyond
    // the 32nd are dealt as HALVES (64 queries, two active waves) -- the launch is as long as its most loaded CU's operand stream
    // (r05_xattn_phase_probe.txt), and a CU with one item + one half pulls 819 KB where a CU with two items pulls 982 KB
    const int split = xp.split;                // xattn_split(per), or 0 (imh_debug_set(3, 9): A/B)
    const int jx = blockIdx.x >> 3;
    int item, qoff = 0, nq = 32 * NW;
    if (jx < per - split) item = (blockIdx.x & 7) * per + jx;
    else {
        const int k = jx - (per - split);
        if (k >= 2 * split) return;
        item = (blockIdx.x & 7) * per + (per - split) + (k >> 1);
        qoff = (k & 1) * 64; nq = 64;
    }
    if (item >= items) return;
    const int hb = item / gx, qblk = item - hb * gx;
    const int b = hb / p.H, h = hb - b * p.H;
    const int q0 = qblk * (32 * NW) + qoff;
    const bool act = wave * 32 < nq;           // (wave-uniform) the waves of a half item beyond its 64 queries only help staging the weights
#if XA_TIMING
    const unsigned long long ts_entry = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- prologue: Q^T = Wq_h X^T ----
    // staging: 4 LDS-DMA rows-of-8 for the wave's own 32 token rows, 2 for its quarter of the weight tile
    const unsigned char* xsrc[4];
    const unsigned char* wsrc[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = xq_stage_xrow(i, wave, lane);
        const int ch = stage_chunk_x(row, lane);
        xsrc[i] = (const unsigned char*)((const T*)xp.X + ((size_t)b * p.Lq + min(q0 + row, p.Lq - 1)) * xp.ldx + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = xq_stage_wrow(i, wave, lane);
        const int ch = stage_chunk_x(row, lane);
        wsrc[i] = (const unsigned char*)((const T*)xp.Wq + ((size_t)h * 64 + row) * xp.ldw + ch * 8);
    }
    auto stage = [&](int buf, int kt) {
        unsigned char* xs = smem + buf * XQ_STAGE;
        unsigned char* ws = xs + 128 * 128;
        if (act) {
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(xsrc[i] + (size_t)kt * 128, xs + (wave * 32 + i * 8) * 128);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(wsrc[i] + (size_t)kt * 128, ws + (wave * 16 + i * 8) * 128);
    };
    // fragment offsets inside a stage (same row / chunk / swizzle pattern as the K tile of the key loop)
    int xoff[4], woff[2][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        xoff[ks] = xq_x_off(wave, lane, ks);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) woff[dt][ks] = 128 * 128 + xq_w_off(dt, lane, ks);
    }
    f32x16 qa[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) qa[dt][r] = 0.f;
    float st_s = 0.f, st_q = 0.f;

    const int nkt = xp.C / 64;
    stage(0, 0);
    if constexpr (LNQ) {           // precomputed row statistics (imh_lnstats.h): lane (q, hi) merges its query row's slots
        const f32x2s mr = merge_row_stats(xp.ln_stats, b * p.Lq + min(q0 + wave * 32 + (lane & 31), p.Lq - 1), xp.ln_slots, xp.C, xp.ln_eps);
        st_s = mr[0]; st_q = mr[1];
    }
    // (Round 5, measured and not adopted -- NEGATIVE_RESULTS.md: a three-slot ring with counted vmcnt, the same with tile kt + 1's fragments
    // read under tile kt's MFMAs, and two producer waves owning the LDS-DMA stream.  Alone, this loop's LDS-DMA half takes 8.3 us and its
    // reads + MFMAs half 7.4 of the 10.7: r05_xattn_loop_ablation.txt.)
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's part of tile kt has landed
        if (!(XA_ABL & 8)) __builtin_amdgcn_s_barrier();      // ... everyone's has; tile kt-1 is fully consumed
        asm volatile("" ::: "memory");
        if (kt + 1 < nkt && !(XA_ABL & 4)) stage((kt + 1) & 1, kt + 1);
        const unsigned char* sb = smem + (kt & 1) * XQ_STAGE;
        if (act && !(XA_ABL & 2)) {
            v8 xf[4], wf[2][4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                xf[ks] = *(const v8*)(sb + xoff[ks]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) wf[dt][ks] = *(const v8*)(sb + woff[dt][ks]);
            }
            if (XA_ABL & 1) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(xf[ks]), "v"(wf[0][ks]), "v"(wf[1][ks]));
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) qa[dt] = mfma32(wf[dt][ks], xf[ks], qa[dt]);
            }
        }
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // every wave is done with the projection stages: the K / V^T ring may reuse them
#if XA_TIMING
    const unsigned long long ts_proj = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- accumulators -> Q^T B-operand fragments of the key loop.  MFMA step sd = 2*dt + u contracts the 16 head dims
    //      [32 dt + 16 u, +16); lane half hi supplies accumulator registers 8u .. 8u+7 of block dt, i.e. dims
    //      {0-3, 8-11} + 4 hi of that group -- the order the permuted K cache stores them in. ----
    v8 qf[4];
    {
        float mean = 0.f, rstd = 1.f;
        if constexpr (LNQ) { mean = st_s; rstd = st_q; }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = qa[dt][rg * 4 + e];
                if constexpr (LNQ) {
                    const int d = h * 64 + att_o_dim(dt, rg * 4, hi);          // 4 consecutive head dims
                    const f32x4 s4 = *(const f32x4*)(xp.ln_s + d), c4 = *(const f32x4*)(xp.ln_c + d);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fma_nopk(rstd, fma_nopk(-mean, s4[e], v[e]), c4[e]);   // scalar on purpose (IMH_KERNEL note)
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) qf[xq_sd(dt, rg * 4 + e)][xq_slot(rg * 4 + e)] = from_f32<T>(v[e]);
            }
    }

    f32x16 fin[2];
    attn_core<T, NW, NPASS>(p, smem, qf, b, h, wave, lane, item, fin);
#if XA_TIMING
    const unsigned long long ts_keys = __builtin_amdgcn_s_memrealtime();
#endif
    if (act) attn_store<T, NW>(p, smem, fin, b, h, q0, wave, lane);
#if XA_TIMING
    if (tid == 0 && p.pf_ptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* dbg = (unsigned long long*)p.pf_ptr + (size_t)blockIdx.x * 4;
        dbg[0] = ts_entry; dbg[1] = ts_proj; dbg[2] = ts_keys; dbg[3] = __builtin_amdgcn_s_memrealtime();
    }
#else
    tail_prefetch(p.pf_ptr, p.pf_bytes, item, items, tid, 64 * NW);
#endif
}



// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the WIDE form, shape-selected for the calls whose (batch, 128-query block, five-head group) items tile the chip (UNet batch 8:
// B = 8, L = 1024, H = 20 -> exactly 256 workgroups, one per CU; the C = 640 / L = 4096 layers -> 512).
//
// The one-head kernel above stages 24 KB per 1.05 MFLOP of its to_q tile (43 FLOP/B) and runs 2.5 rounds of 1280 workgroups at batch 8,
// each with its own vmcnt(0) + barrier per K tile: 57 us warm for work whose MFMA time is 10 us (profiles/r05_xattn_phase_probe.txt).
// Here one workgroup = (batch, 128 queries, FIVE heads): the to_q tile is 128 x 320 over C -- 56 KB per 5.2 MFLOP K tile (94 FLOP/B,
// what the 256 x 160 GEMM tile has), every 128-query block of X crosses the L2 -> CU path H / 5 times instead of H.
//   * 4 consumer waves, wave w = queries 32 w .. of ALL five heads: 10 accumulator blocks of v_mfma_f32_32x32x16 (160 registers), 40 MFMAs
//     per K tile per SIMD (one consumer per SIMD: the matrix pipes are evenly loaded, which no split of FIVE heads over 8 or 10 waves gives);
//   * 4 producer waves own every LDS-DMA (gemm_ws_kernel's structure: counted vmcnt, ONE s_barrier per K tile carries "tile i + 1 has
//     landed" and "tile i has been read").  Three 56-KB stages do not fit the 160 KB, so the two operands ride rings of DIFFERENT depth: the
//     weight tile (320 rows, 40 KB) three slots, the token tile (128 rows, 16 KB) two -- the producers issue X(i + 1) ahead of W(i + 2), so
//     that `vmcnt(own W pieces)` means "X(i + 1) and W(i + 1) have landed" and W(i + 2) stays in flight across the barrier.  (Measured and
//     not kept: a FOUR-slot weight ring with the token fragments fetched by their consumer wave straight into registers two tiles ahead --
//     25.6 vs 23.5 us for the projection; and the K loops of an XCD's workgroups started at rotated tiles -- the same to 1 us.  The K tile
//     runs at what one CU's vector-memory path delivers, 56 KB per ~2200 cycles = 25 B/clk: profiles/r06_xattn_phase_probe.txt)
//   * after the projection the producers STAY and stream the K / V^T tiles of the five heads (text tiles, then the image-prompt tile, head
//     after head) through a four-slot ring over the dead projection rings; a consumer runs the resident-tile key loop (attn_tile) on its 32
//     queries head after head -- no load wait inside a key loop, no workgroup-wide drain between heads -- and stores each head's O rows as
//     whole 128-B lines through its private staging rows.  The Q^T fragments of the five heads wait in LDS for their head's turn.
// Same arithmetic, same operand layouts and the same per-tile order of operations as the one-head kernel (Q^T = Wq_h X^T on 32x32x16 blocks,
// accumulators -> QK^T B operands through a cvt, attn_tile's online softmax): the two-pass results are bit-identical to it.
constexpr int XW_NH = 5;                            // heads per workgroup
constexpr int XW_XT = 128 * 128;                    // token tile: 128 rows x 128 B
constexpr int XW_WT = XW_NH * 64 * 128;             // weight tile: 320 rows x 128 B
constexpr int XW_XS = 2, XW_WS = 3;                 // ring depths (32 + 120 KB)
constexpr int XW_KVR = 4;                           // K / V^T ring slots (2 x 8 KB each) of the key phase
constexpr int XW_LDS = 160 * 1024;                  // the key phase needs all of it (ring 64 KB + O staging 16 KB + parked fragments 80 KB)
static_assert(XW_XS * XW_XT + XW_WS * XW_WT <= XW_LDS, "projection rings");
constexpr int XW_QS_OFF = XW_KVR * 2 * ATT_TILE_BYTES;              // O staging tile (128 rows x 128 B)
constexpr int XW_PARK_OFF = XW_QS_OFF + 128 * 128;                  // parked Q^T fragments: 4 waves x 5 heads x 4 KB
static_assert(XW_PARK_OFF + 4 * XW_NH * 4096 <= XW_LDS && XW_LDS <= 160 * 1024, "key ring + O staging rows + parked fragments fit the weight ring's bytes");

template <typename T, int NPASS, bool LNQ>
__global__ __launch_bounds__(512, 2) void xattn_wide_kernel(const XAttnParams xp) {
    typedef typename Vec<T>::v8 v8;
    constexpr int NH = XW_NH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const AttnParams& p = xp.a;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int gx = p.Lq >> 7;                       // (the launcher guarantees Lq % 128 == 0, H % 5 == 0)
    const int HG = p.H / NH;
    const int items = gx * HG * p.B;
    const int per = (items + 7) >> 3;
    // XCD-aware order as above: an XCD sees few (batch, head group) pairs -- their Wq slices, K / V^T caches -- and streams the query blocks
    const int item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || item >= items) return;
    const int hb = item / gx, qblk = item - hb * gx;
    const int b = hb / HG, h0 = (hb - b * HG) * NH;
    const int q0 = qblk * 128;
    const int nkt = xp.C / 64;
    unsigned char* const xr = smem;
    unsigned char* const wr = smem + XW_XS * XW_XT;
    const int nt1 = (p.Lk + ATT_KV - 1) / ATT_KV;
    const int nt2 = NPASS == 2 ? (p.Lk2 + ATT_KV - 1) / ATT_KV : 0;
    const int ntt = nt1 + nt2;
    const int NT = NH * ntt;                        // K / V^T tiles of the key phase, in consumption order
#if XA_TIMING
    const unsigned long long ts_entry = __builtin_amdgcn_s_memrealtime();
#endif

    if (wave >= 4) {
        // ------------------------------------------------------------------ producer
        const int pw = wave - 4;
        const unsigned char* xsrc[4];
        const unsigned char* wsrc[10];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = (k * 4 + pw) * 8 + (lane >> 3);
            const int ch = (lane & 7) ^ swz_x(r);
            xsrc[k] = (const unsigned char*)((const T*)xp.X + ((size_t)b * p.Lq + min(q0 + r, p.Lq - 1)) * xp.ldx + ch * 8);
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int r = (k * 4 + pw) * 8 + (lane >> 3);
            const int ch = (lane & 7) ^ swz_x(r);
            wsrc[k] = (const unsigned char*)((const T*)xp.Wq + ((size_t)h0 * 64 + r) * xp.ldw + ch * 8);
        }
        auto issue_x = [&](int slot, int kt) {
#pragma unroll
            for (int k = 0; k < 4; ++k) glds16(xsrc[k] + (size_t)kt * 128, xr + slot * XW_XT + (k * 4 + pw) * 1024);
        };
        auto issue_w = [&](int slot, int kt) {
#pragma unroll
            for (int k = 0; k < 10; ++k) glds16(wsrc[k] + (size_t)kt * 128, wr + slot * XW_WT + (k * 4 + pw) * 1024);
        };
        issue_x(0, 0);
        issue_w(0, 0);
        if (nkt > 1) { issue_w(1, 1); asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // tile 0 has landed
        int ws2 = 2;                                        // W slot of tile i + 2
#if XA_TIMING
        unsigned long long tacc[3] = {0, 0, 0}, tm0 = __builtin_readcyclecounter();
#endif
        for (int i = 0; i < nkt; ++i) {
            if (i + 1 < nkt) issue_x((i + 1) & 1, i + 1);   // the slot tile i - 1 was read from (every consumer is past the barrier that ended it)
            if (i + 2 < nkt) issue_w(ws2, i + 2);           // likewise
            XA_TICK(0);
            if (i + 2 < nkt) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // X(i + 1), W(i + 1) have landed; W(i + 2) stays in flight
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            XA_TICK(1);
            __builtin_amdgcn_s_barrier();                   // tile i + 1 has landed, tile i has been read
            XA_TICK(2);
            if (++ws2 == XW_WS) ws2 = 0;
        }
#if XA_TIMING
        if (blockIdx.x == 0 && pw == 0 && lane == 0 && p.pf_ptr) {
            unsigned long long* dbg = (unsigned long long*)p.pf_ptr + (size_t)gridDim.x * 4 + 4;
            dbg[0] = tacc[0]; dbg[1] = tacc[1]; dbg[2] = tacc[2]; dbg[3] = nkt;
        }
#endif
        // ---- key phase: the projection rings are dead.  Tile j = head j / ntt, text tiles first, then the image-prompt tiles ----
        auto issue_kv = [&](int j) {
            const int g = j / ntt, t = j - g * ntt;
            unsigned char* ks = smem + (j & (XW_KVR - 1)) * 2 * ATT_TILE_BYTES;
            if (NPASS == 1 || t < nt1)
                attn_stage_tile<T>((const T*)p.K, (const T*)p.Vt, p.Lk_pad, p.ldk, p.ldvt, b, h0 + g, t, ks, ks + ATT_TILE_BYTES, pw, lane);
            else
                attn_stage_tile<T>((const T*)p.K2, (const T*)p.Vt2, p.Lk2_pad, p.ldk2, p.ldvt2, b, h0 + g, t - nt1, ks, ks + ATT_TILE_BYTES, pw, lane);
        };
        for (int j = 0; j < XW_KVR - 1 && j < NT; ++j) issue_kv(j);
        wait_vmcnt_dyn(4 * min(XW_KVR - 2, NT - 1));        // tile 0 has landed (four LDS-DMA instructions per tile per producer)
        __builtin_amdgcn_s_barrier();
        for (int j = 0; j < NT; ++j) {
            if (j + XW_KVR - 1 < NT) issue_kv(j + XW_KVR - 1);          // into the slot of tile j - 1
            wait_vmcnt_of<4 * (XW_KVR - 2), 4, 0>(4 * max(0, min(XW_KVR - 2, NT - 2 - j)));    // tile j + 1 has landed
            __builtin_amdgcn_s_barrier();                   // ... and tile j has been read
        }
#if !XA_TIMING
        tail_prefetch(p.pf_ptr, p.pf_bytes, item, items, tid - 256, 256);
#endif
        return;
    }

    // ---------------------------------------------------------------------- consumer: queries q0 + 32 wave .., all five heads
    int xoff[4], woff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        xoff[ks] = xq_x_off(wave, lane, ks);
        woff[ks] = xq_w_off(0, lane, ks);           // + (2 g + dt) * 32 rows: the swizzle only sees the row inside a 32-block
    }
    f32x16 qa[NH][2];
#pragma unroll
    for (int g = 0; g < NH; ++g)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) qa[g][dt][r] = 0.f;
    float st_s = 0.f, st_q = 0.f;
    if constexpr (LNQ) {           // precomputed row statistics (imh_lnstats.h): lane (q, hi) merges its query row's slots, beside the ring prologue
        const f32x2s mr = merge_row_stats(xp.ln_stats, b * p.Lq + min(q0 + wave * 32 + (lane & 31), p.Lq - 1), xp.ln_slots, xp.C, xp.ln_eps);
        st_s = mr[0]; st_q = mr[1];
    }
    __builtin_amdgcn_s_barrier();                   // tile 0 has landed
    asm volatile("" ::: "memory");
    {
        // One consumer per SIMD: nothing else hides the LDS latency of its fragment reads, so the K tile is walked in eight UNITS
        // (k step ks, block row dt: the five heads' weight fragments against one token fragment = 5 MFMAs = 160 cycles of the matrix pipe)
        // with two fragment register sets -- the reads of unit u + 1 are issued one-to-one between the MFMAs of unit u (left to itself hipcc
        // reads two fragments at a time right in front of their MFMAs: the pipe then waits for every pair)
        v8 wf[2][NH], xf[2];
        int ws = 0;
#if XA_TIMING
        unsigned long long tacc[3] = {0, 0, 0}, tm0 = __builtin_readcyclecounter();
#endif
        for (int kt = 0; kt < nkt; ++kt) {
            const unsigned char* xb = xr + (kt & 1) * XW_XT;
            const unsigned char* wb = wr + ws * XW_WT;
            auto rd = [&](auto U) {
                constexpr int u = decltype(U)::value, ks = u >> 1, dt = u & 1;
                if constexpr (dt == 0) xf[ks & 1] = *(const v8*)(xb + xoff[ks]);
#pragma unroll
                for (int g = 0; g < NH; ++g) wf[u & 1][g] = *(const v8*)(wb + (g * 2 + dt) * (32 * 128) + woff[ks]);
            };
            auto step = [&](auto U) {
                constexpr int u = decltype(U)::value, ks = u >> 1, dt = u & 1;
                if constexpr (u + 1 < 8) rd(std::integral_constant<int, u + 1>{});
#pragma unroll
                for (int g = 0; g < NH; ++g) qa[g][dt] = mfma32(wf[u & 1][g], xf[ks & 1], qa[g][dt]);
                if constexpr (u + 1 < 8) {
#pragma unroll
                    for (int k = 0; k < NH; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one DS read
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                    }
                    if constexpr (((u + 1) & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // (+ the token fragment)
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            rd(std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
            step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
            XA_TICK(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // tile kt has been read: its slots may be refilled
            XA_TICK(1);
            __builtin_amdgcn_s_barrier();                            // ... and tile kt + 1 has landed
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            XA_TICK(2);
            if (++ws == XW_WS) ws = 0;
        }
#if XA_TIMING
        if (blockIdx.x == 0 && wave == 0 && lane == 0 && p.pf_ptr) {
            unsigned long long* dbg = (unsigned long long*)p.pf_ptr + (size_t)gridDim.x * 4;
            dbg[0] = tacc[0]; dbg[1] = tacc[1]; dbg[2] = tacc[2]; dbg[3] = nkt;
        }
#endif
    }
#if XA_TIMING
    const unsigned long long ts_proj = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- accumulators -> Q^T B-operand fragments of the key loops (as in xattn_kernel), head by head, parked in LDS (wave-private,
    //      lane-linear 16-B pieces: conflict-free) until their head's turn: five sets (80 registers) live beside a key loop's accumulators
    //      spilled 80 registers to scratch.  Behind the last projection barrier nobody reads the ring, and the producers' key ring stays
    //      below XW_QS_OFF. ----
    unsigned char* const park = smem + XW_PARK_OFF + wave * (NH * 4096) + lane * 16;
    {
        float mean = 0.f, rstd = 1.f;
        if constexpr (LNQ) { mean = st_s; rstd = st_q; }
#pragma unroll
        for (int g = 0; g < NH; ++g) {
            v8 qf[4];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = qa[g][dt][rg * 4 + e];
                    if constexpr (LNQ) {
                        const int d = (h0 + g) * 64 + att_o_dim(dt, rg * 4, hi);          // 4 consecutive head dims
                        const f32x4 s4 = *(const f32x4*)(xp.ln_s + d), c4 = *(const f32x4*)(xp.ln_c + d);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fma_nopk(rstd, fma_nopk(-mean, s4[e], v[e]), c4[e]);   // scalar on purpose (IMH_KERNEL note)
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) qf[xq_sd(dt, rg * 4 + e)][xq_slot(rg * 4 + e)] = from_f32<T>(v[e]);
                }
#pragma unroll
            for (int sd = 0; sd < 4; ++sd) *(v8*)(park + (g * 4 + sd) * 1024) = qf[sd];
        }
    }
    // ---- key phase: resident tiles from the producers' ring, one barrier per tile; O rows of a head leave as soon as the head is done ----
    unsigned char* const qs = smem + XW_QS_OFF;     // this wave's rows 32 wave .. of the O staging tile
    const float c = p.scale * LOG2E;
    const float wgt2 = NPASS == 2 ? (p.scale2_tab ? p.scale2_tab[*p.step] : p.scale2) : 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // tile 0 of the key phase has landed
    asm volatile("" ::: "memory");
    int j = 0;
#pragma unroll 1
    for (int g = 0; g < NH; ++g) {
        v8 qh[4];
#pragma unroll
        for (int sd = 0; sd < 4; ++sd) qh[sd] = *(const v8*)(park + (g * 4 + sd) * 1024);
        f32x16 fin[2];
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            f32x16 o[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
            float m_run = NEG_BIG, l_run = 0.f;
            const int nt = pass == 0 ? nt1 : nt2;
            const int Lk = pass == 0 ? p.Lk : p.Lk2;
            for (int t = 0; t < nt; ++t, ++j) {
                const unsigned char* ks = smem + (j & (XW_KVR - 1)) * 2 * ATT_TILE_BYTES;
                attn_tile<T>(ks, ks + ATT_TILE_BYTES, qh, lane, t * ATT_KV, Lk, c, o, m_run, l_run);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // tile j has been read
                __builtin_amdgcn_s_barrier();                        // ... and tile j + 1 has landed
                asm volatile("" ::: "memory");
            }
            const float inv = (pass == 0 ? 1.0f : wgt2) / (l_run + __shfl_xor(l_run, 32, 64));
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) fin[dt][r] = pass == 0 ? o[dt][r] * inv : fin[dt][r] + o[dt][r] * inv;
        }
        attn_store<T, 4>(p, qs, fin, b, h0 + g, q0, wave, lane);
    }
#if XA_TIMING
    if (tid == 0 && p.pf_ptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* dbg = (unsigned long long*)p.pf_ptr + (size_t)blockIdx.x * 4;
        dbg[0] = ts_entry; dbg[1] = ts_proj; dbg[2] = ts_proj; dbg[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

template <typename T>
static void launch_xattn_wide(const XAttnParams& xp, hipStream_t stream) {
    const AttnParams& p = xp.a;
    const int items = (p.Lq >> 7) * (p.H / XW_NH) * p.B;
    dim3 grid(8 * ((items + 7) / 8));
#define IMH_XW(NPV, LNV) do { auto kern = xattn_wide_kernel<T, NPV, LNV>; static DynLdsOnce once; once.ensure((const void*)kern, XW_LDS); \
        hipLaunchKernelGGL(kern, grid, dim3(512), XW_LDS, stream, xp); } while (0)
    if (p.K2) { if (xp.ln_s) IMH_XW(2, true); else IMH_XW(2, false); }
    else { if (xp.ln_s) IMH_XW(1, true); else IMH_XW(1, false); }
#undef IMH_XW
}
// the wide form's shape rule: whole 128-query blocks, heads in groups of five, and enough (batch, block, group) items to give every CU one
static bool xattn_wide_fits(const AttnParams& p) { return p.Lq % 128 == 0 && p.H % XW_NH == 0; }
static int xattn_wide_items(const AttnParams& p) { return (p.Lq >> 7) * (p.H / XW_NH) * p.B; }

#ifdef IMH_EXPERIMENTAL
// ---------------------------------------------------------------------------------------------------------------------
// Two heads per workgroup.  One workgroup = (batch, head PAIR, 128 queries) = 8 consumer waves (head g = wave >> 2, query
// group wave & 3) [+ NP producer waves].  Against the one-head kernel above: a 128-query block of X is fetched by H / 2
// items instead of H (16 KB of tokens + 16 KB of weights per 2.1 MFLOP K tile instead of 16 + 8 per 1.05), the K loop is an
// S-stage LDS-DMA ring with counted vmcnt (never drained; the one-head kernel waits vmcnt(0) + barrier every K tile), and
// with NP > 0 the LDS-DMA issue lives in producer waves that do nothing else (gemm_ws_kernel's structure: ONE s_barrier per
// K tile carries "tile i + 1 has landed" and "tile i has been read").  The producers exit after the projection; the two
// 4-wave groups then run the shared key loop (imh_attn_core.h) side by side, each on its own half of the LDS
// (s_barrier waits on surviving waves only).
constexpr int XQ2_STAGE = 128 * 128 + 128 * 128;   // X tile (128 token rows) + Wq tile (2 heads x 64 dims), 128 B per row

// RES: the key sets are short (<= 2 text tiles, <= 1 image-prompt tile -- every SDXL cross-attention layer): the text K / V^T
// tiles of both heads are staged ONCE into a dedicated 64 KB of LDS by the consumer waves at kernel entry (in flight under
// the whole projection), the image-prompt tile right after the projection (in flight under the text pass); the key loop then
// runs from resident tiles without a single load wait.  (The ring form below pays two dependent L2 round trips per pass.)
template <typename T, int NPASS, bool LNQ, int NP, int S, bool RES>
__global__ __launch_bounds__(64 * (8 + NP), 1) void xattn2_kernel(const XAttnParams xp) {
    typedef typename Vec<T>::v8 v8;
    constexpr int NI = 32;                              // LDS-DMA wave instructions per K tile (8 rows x 128 B each)
    constexpr int NISS = NP > 0 ? NP : 8;               // waves that issue them
    constexpr int LP = NI / NISS;
    constexpr int GROUP_BYTES = ATT_STAGES * 2 * ATT_TILE_BYTES;      // one head group's K / V^T ring (+ Q / O staging rows)
    static_assert(S * XQ2_STAGE >= 2 * GROUP_BYTES, "the projection ring covers both key-loop rings");
    static_assert(!RES || S * XQ2_STAGE + 2 * 4 * ATT_TILE_BYTES <= 160 * 1024, "ring + resident text tiles fit the LDS");
    static_assert((S - 2) * LP <= 63 && NI % NISS == 0, "vmcnt range / even split");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const AttnParams& p = xp.a;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int gx = (p.Lq + 127) / 128;
    const int HP = p.H >> 1;
    const int items = gx * HP * p.B;
    const int per = (items + 7) >> 3;
    const int item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || item >= items) return;
    const int hb = item / gx, qblk = item - hb * gx;
    const int b = hb / HP, h0 = (hb - b * HP) * 2;
    const int q0 = qblk * 128;
    const int nkt = xp.C / 64;
#if XA_TIMING
    const unsigned long long ts_entry = __builtin_amdgcn_s_memrealtime();
#endif

    // staging map: instruction rb (0..31) fills tile rows rb*8 .. rb*8+7; rows 0..127 = tokens, 128..255 = weight rows
    auto src_of = [&](int rb) -> const unsigned char* {
        const int r = rb * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ swz_x(r & 127);
        if (rb < 16) return (const unsigned char*)((const T*)xp.X + ((size_t)b * p.Lq + min(q0 + r, p.Lq - 1)) * xp.ldx + ch * 8);
        return (const unsigned char*)((const T*)xp.Wq + ((size_t)h0 * 64 + (r - 128)) * xp.ldw + ch * 8);
    };

    if (NP > 0 && wave >= 8) {
        // ------------------------------------------------------------------ producer
        const int pw = wave - 8;
        const unsigned char* src[LP];
#pragma unroll
        for (int k = 0; k < LP; ++k) src[k] = src_of(k * NISS + pw);
        auto issue = [&](int slot, int kt) {
#pragma unroll
            for (int k = 0; k < LP; ++k) glds16(src[k] + (size_t)kt * 128, smem + slot * XQ2_STAGE + (k * NISS + pw) * 1024);
        };
#pragma unroll
        for (int s = 0; s < S - 1; ++s)
            if (s < nkt) issue(s, s);
        if (S - 1 <= nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * LP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // tile 0 has landed
        int slot = S - 1;
        for (int i = 0; i < nkt; ++i) {
            if (i + S - 1 < nkt) issue(slot, i + S - 1);   // into the slot tile i - 1 was read from
            if (i + S <= nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * LP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                  // tile i + 1 has landed, tile i has been read
            if (++slot == S) slot = 0;
        }
#if !XA_TIMING
        tail_prefetch(p.pf_ptr, p.pf_bytes, item, items, tid - 512, 64 * NP);
#endif
        return;
    }

    // ---------------------------------------------------------------------- consumer
    const int g = wave >> 2, qg = wave & 3;
    const int h = h0 + g;
    unsigned char* const res = smem + S * XQ2_STAGE + g * (4 * ATT_TILE_BYTES);     // RES: this head's two text K / V^T tile pairs
    const int nt1 = (p.Lk + ATT_KV - 1) / ATT_KV;
    if constexpr (RES) {
        for (int t = 0; t < nt1; ++t)
            attn_stage_tile<T>((const T*)p.K, (const T*)p.Vt, p.Lk_pad, p.ldk, p.ldvt, b, h, t, res + t * 2 * ATT_TILE_BYTES,
                               res + t * 2 * ATT_TILE_BYTES + ATT_TILE_BYTES, qg, lane);
    }
    int xoff[4], woff[2][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        xoff[ks] = xq_x_off(qg, lane, ks);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) woff[dt][ks] = 128 * 128 + xq_w_off(g * 2 + dt, lane, ks);
    }
    f32x16 qa[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) qa[dt][r] = 0.f;
    float st_s = 0.f, st_q = 0.f;

    const unsigned char* src[NP > 0 ? 1 : LP];
    if constexpr (NP == 0) {
#pragma unroll
        for (int k = 0; k < LP; ++k) src[k] = src_of(k * 8 + wave);
    }
    auto issue = [&](int slot, int kt) {
        if constexpr (NP == 0) {
#pragma unroll
            for (int k = 0; k < LP; ++k) glds16(src[k] + (size_t)kt * 128, smem + slot * XQ2_STAGE + (k * 8 + wave) * 1024);
        }
    };
    if constexpr (NP == 0) {
#pragma unroll
        for (int s = 0; s < S - 1; ++s)
            if (s < nkt) issue(s, s);
    }
    if constexpr (LNQ) {           // precomputed row statistics: lane (q, hi) merges its query row's slots (loads in flight
        // beside the ring prologue)
        const f32x2s mr = merge_row_stats(xp.ln_stats, b * p.Lq + min(q0 + qg * 32 + (lane & 31), p.Lq - 1), xp.ln_slots, xp.C, xp.ln_eps);
        st_s = mr[0]; st_q = mr[1];
    }
    if constexpr (NP > 0) {
        __builtin_amdgcn_s_barrier();                      // tile 0 has landed
        asm volatile("" ::: "memory");
    }
    int slot = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if constexpr (NP == 0) {
            if (kt + S - 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * LP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                  // everyone's part of tile kt is in LDS; tile kt - 1 fully consumed
            asm volatile("" ::: "memory");
            if (kt + S - 1 < nkt) {
                int ns = slot + S - 1;
                if (ns >= S) ns -= S;
                issue(ns, kt + S - 1);
            }
        }
        const unsigned char* sb = smem + slot * XQ2_STAGE;
        v8 xf[4], wf[2][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            xf[ks] = *(const v8*)(sb + xoff[ks]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) wf[dt][ks] = *(const v8*)(sb + woff[dt][ks]);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) qa[dt] = mfma32(wf[dt][ks], xf[ks], qa[dt]);
        if constexpr (NP > 0) {
            if (RES && kt == nkt - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's resident text tiles landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // tile kt has been read: its slot may be refilled
            __builtin_amdgcn_s_barrier();                            // ... and tile kt + 1 has landed
        }
        asm volatile("" ::: "memory");
        if (++slot == S) slot = 0;
    }
    if constexpr (NP == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // every wave is done with the projection stages: the K / V^T rings may reuse them
    }

#if XA_TIMING
    const unsigned long long ts_proj = __builtin_amdgcn_s_memrealtime();
#endif
    v8 qf[4];
    {
        float mean = 0.f, rstd = 1.f;
        if constexpr (LNQ) { mean = st_s; rstd = st_q; }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = qa[dt][rg * 4 + e];
                if constexpr (LNQ) {
                    const int d = h * 64 + att_o_dim(dt, rg * 4, hi);
                    const f32x4 s4 = *(const f32x4*)(xp.ln_s + d), c4 = *(const f32x4*)(xp.ln_c + d);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fma_nopk(rstd, fma_nopk(-mean, s4[e], v[e]), c4[e]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) qf[xq_sd(dt, rg * 4 + e)][xq_slot(rg * 4 + e)] = from_f32<T>(v[e]);
            }
    }

    unsigned char* gs = smem + g * GROUP_BYTES;
    f32x16 fin[2];
    if constexpr (RES) {
        const float c = p.scale * LOG2E;
        if constexpr (NPASS == 2)      // the image-prompt tile: into the (dead) projection ring, in flight under the text pass
            attn_stage_tile<T>((const T*)p.K2, (const T*)p.Vt2, p.Lk2_pad, p.ldk2, p.ldvt2, b, h, 0, gs, gs + ATT_TILE_BYTES, qg, lane);
        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
        float m_run = NEG_BIG, l_run = 0.f;
        // the text tiles were issued before the projection by this group's own waves: every wave's vmcnt(0) before the last
        // projection barrier (NP == 0) or right here (NP > 0: consumers issue nothing else) + that barrier make them visible
        for (int t = 0; t < nt1; ++t)
            attn_tile<T>(res + t * 2 * ATT_TILE_BYTES, res + t * 2 * ATT_TILE_BYTES + ATT_TILE_BYTES, qf, lane, t * ATT_KV, p.Lk, c, o, m_run, l_run);
        {
            const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) fin[dt][r] = o[dt][r] * inv;
        }
        if constexpr (NPASS == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // every wave's part of the image-prompt tile has landed
            asm volatile("" ::: "memory");
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
            m_run = NEG_BIG; l_run = 0.f;
            attn_tile<T>(gs, gs + ATT_TILE_BYTES, qf, lane, 0, p.Lk2, c, o, m_run, l_run);
            const float wgt = p.scale2_tab ? p.scale2_tab[*p.step] : p.scale2;
            const float inv = wgt / (l_run + __shfl_xor(l_run, 32, 64));
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) fin[dt][r] += o[dt][r] * inv;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // the group is done reading the tile: its rows become O staging rows
        }
    } else {
        attn_core<T, 4, NPASS>(p, gs, qf, b, h, qg, lane, item, fin);
    }
#if XA_TIMING
    const unsigned long long ts_keys = __builtin_amdgcn_s_memrealtime();
#endif
    attn_store<T, 4>(p, gs, fin, b, h, q0, qg, lane);
#if XA_TIMING
    if (tid == 0 && p.pf_ptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* dbg = (unsigned long long*)p.pf_ptr + (size_t)item * 4;
        dbg[0] = ts_entry; dbg[1] = ts_proj; dbg[2] = ts_keys; dbg[3] = __builtin_amdgcn_s_memrealtime();
    }
#else
    if (NP == 0) tail_prefetch(p.pf_ptr, p.pf_bytes, item, items, tid, 512);
#endif
}

template <typename T, int NPASS, bool LNQ, int NP, int S, bool RES>
static void launch_xattn2(const XAttnParams& xp, hipStream_t stream) {
    const AttnParams& p = xp.a;
    const int items = ((p.Lq + 127) / 128) * (p.H / 2) * p.B;
    dim3 grid(8 * ((items + 7) / 8));
    const int lds = S * XQ2_STAGE + (RES ? 2 * 4 * ATT_TILE_BYTES : 0);
    auto kern = xattn2_kernel<T, NPASS, LNQ, NP, S, RES>;
    static DynLdsOnce once;
    once.ensure((const void*)kern, lds);
    hipLaunchKernelGGL(kern, grid, dim3(64 * (8 + NP)), lds, stream, xp);
}

template <typename T, int NP, int S, bool RES>
static void launch_xattn2_res(const XAttnParams& xp, hipStream_t stream) {
    if (xp.a.K2) {
        if (!xp.ln_s) launch_xattn2<T, 2, false, NP, S, RES>(xp, stream);
        else launch_xattn2<T, 2, true, NP, S, RES>(xp, stream);
    } else {
        if (!xp.ln_s) launch_xattn2<T, 1, false, NP, S, RES>(xp, stream);
        else launch_xattn2<T, 1, true, NP, S, RES>(xp, stream);
    }
}

// res: short key sets (see RES above) -> 3-stage projection ring + resident text tiles; else the 4-stage ring + key-loop rings
template <typename T, int NP>
static void launch_xattn2_np(const XAttnParams& xp, hipStream_t stream, bool res) {
    if (res) launch_xattn2_res<T, NP, 3, true>(xp, stream);
    else launch_xattn2_res<T, NP, 4, false>(xp, stream);
}

#endif      // IMH_EXPERIMENTAL

int xattn_launch(const XAttnParams& xp, int dtype, hipStream_t stream) {
    const AttnParams& p = xp.a;
    if (p.Lk <= 0 || p.Lk_pad % ATT_KV != 0 || p.Lk_pad < p.Lk) {
        set_error("cross_attention: Lk=%d Lk_pad=%d (pad must be a multiple of 64 and >= Lk)", p.Lk, p.Lk_pad);
        return IMH_ERR_SHAPE;
    }
    if (p.K2 && (p.Lk2 <= 0 || p.Lk2_pad % ATT_KV != 0 || p.Lk2_pad < p.Lk2)) {
        set_error("cross_attention: Lk2=%d Lk2_pad=%d invalid", p.Lk2, p.Lk2_pad);
        return IMH_ERR_SHAPE;
    }
    if (xp.C <= 0 || xp.C % 64) { set_error("cross_attention: C=%d must be a positive multiple of 64", xp.C); return IMH_ERR_SHAPE; }
    if ((xp.ldx & 7) || (xp.ldw & 7) || (p.ldk & 7) || (p.ldvt & 7) || (p.ldo & 7) || (p.K2 && ((p.ldk2 & 7) || (p.ldvt2 & 7)))) {
        set_error("cross_attention: leading dimensions must be multiples of 8 elements");
        return IMH_ERR_SHAPE;
    }
    if (p.B <= 0 || p.H <= 0 || p.Lq <= 0) { set_error("cross_attention: empty problem"); return IMH_ERR_SHAPE; }
    if (dtype != IMH_DT_BF16 && dtype != IMH_DT_F16) { set_error("cross_attention: unknown dtype %d", dtype); return IMH_ERR_DTYPE; }
    if (xp.ln_s && !xp.ln_stats) {
        set_error("cross_attention: the folded LayerNorm takes the token rows' statistics from ln_stats (imh_lnstats.h: the epilogue of "
                  "the GEMM that wrote the rows, or IMH_EW_ROW_STATS); there is no in-loop E[x^2] - mean^2 form");
        return IMH_ERR_ARG;
    }
    // auto = one head per workgroup: in the forward (operands cold in this XCD's L2) its 320 workgroups on all 256 CUs pull the
    // token rows and weights faster than the 160 workgroups of the two-head form, which only wins back-to-back on warm
    // operands (profiles/r03_attn_ab.json vs r03_forward_ab_*.json)
    int mode = g_xattn_mode;
    // auto: the wide form (round 6) when its items fill the chip -- UNet batch 8 (256 items at C = 1280 / L = 1024, 512 at C = 640 /
    // L = 4096: 57 -> XX us warm per launch); at UNet batch 2 its 64 / 128 items leave CUs idle and the one-head form stays.
    // imh_debug_set(3, 10) forces it wherever it fits, (3, 1) forces the one-head form (A/B, tests)
    if (mode == 0) mode = (xattn_wide_fits(p) && xattn_wide_items(p) >= 256) ? 10 : 1;
    if (mode == 10) {
        if (!xattn_wide_fits(p)) { set_error("cross_attention: the wide form takes Lq %% 128 == 0 and H %% 5 == 0 (Lq=%d H=%d)", p.Lq, p.H); return IMH_ERR_SHAPE; }
        if (dtype == IMH_DT_BF16) launch_xattn_wide<bf16_t>(xp, stream);
        else launch_xattn_wide<f16_t>(xp, stream);
        return check_launch("xattn_wide_kernel");
    }
    if ((p.H & 1) || mode == 1 || mode == 9) {
        const int items = ((p.Lq + 127) / 128) * p.H * p.B;
        const int per = (items + 7) / 8;
        XAttnParams xq = xp;
        xq.split = g_xattn_mode == 9 ? 0 : xattn_split(per);          // (mode 9, A/B: whole items only)
        dim3 grid(8 * (per + xq.split));
#define IMH_XA1(TT, NPV) do { \
            if (!xp.ln_s) hipLaunchKernelGGL((xattn_kernel<TT, NPV, false>), grid, dim3(256), 0, stream, xq); \
            else hipLaunchKernelGGL((xattn_kernel<TT, NPV, true>), grid, dim3(256), 0, stream, xq); } while (0)
#define IMH_XA(TT) do { if (p.K2) IMH_XA1(TT, 2); else IMH_XA1(TT, 1); } while (0)
        if (dtype == IMH_DT_BF16) IMH_XA(bf16_t);
        else IMH_XA(f16_t);
#undef IMH_XA
#undef IMH_XA1
    } else {
#ifndef IMH_EXPERIMENTAL
        return experimental_refused("the two-head fused cross-attention kernel (imh_debug_set(3, 2 .. 8))");
#else
        // modes 2 / 3 / 4: two heads per workgroup with 0 / 2 / 4 producer waves; +4 (6 / 7 / 8): the same without the resident
        // key tiles (A/B)
        const bool res = mode < 6 && p.Lk <= 2 * ATT_KV && (!p.K2 || p.Lk2 <= ATT_KV);
        const int m = mode >= 6 ? mode - 4 : mode;
        if (m == 2) { if (dtype == IMH_DT_BF16) launch_xattn2_np<bf16_t, 0>(xp, stream, res); else launch_xattn2_np<f16_t, 0>(xp, stream, res); }
        else if (m == 4) { if (dtype == IMH_DT_BF16) launch_xattn2_np<bf16_t, 4>(xp, stream, res); else launch_xattn2_np<f16_t, 4>(xp, stream, res); }
        else { if (dtype == IMH_DT_BF16) launch_xattn2_np<bf16_t, 2>(xp, stream, res); else launch_xattn2_np<f16_t, 2>(xp, stream, res); }
#endif
    }
    return check_launch("xattn_kernel");
}

}  // namespace imh
