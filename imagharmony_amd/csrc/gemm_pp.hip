// 256 x 256 "ping-pong" MFMA GEMM for gfx950: the large-N Linear layers of the SDXL transformer blocks (the GEGLU
// projection, diffusers BasicTransformerBlock.ff.net[0]; 2048 x 10240 x 1280 at 1024^2) with the shared epilogue (bias /
// GEGLU / residual ...).  (The folded LayerNorm these kernels carried in rounds 2-3 took its statistics as sum / sum of
// squares inside the K loop -- E[x^2] - mean^2 -- and went away with that form; the LayerNorm-folding launches run on the
// wave-specialised kernels of gemm_ring.hip, which take handed-over statistics.)
//
// Why another kernel: the 128 x 128 tiles of gemm.hip move 64 B of operands through the L2 -> LDS path per 64 FLOP-cycles
// of one CU -- the vector-L1 fill rate (64 B / clk / CU) is the bound, not the matrix pipe.  A
// 256 x 256 tile halves the bytes per FLOP.  Structure (cdna_hip_programming.md 5, "256^2 8-phase template"; written
// from its description, the schedule below is our own):
//   * 512 threads = 8 waves as 2 (token halves, wm) x 4 (weight slabs, wn); wave tile 128 tokens x 64 weight rows =
//     8 x 4 accumulator fragments of v_mfma_f32_16x16x32 (128 VGPRs); BK = 64; two LDS buffers of (256 + 256) rows x 128 B
//     = 128 KB, filled by global_load_lds_dwordx4 with the source-side XOR swizzles of imh_layout.h.
//   * a K tile is consumed in four PHASES of 16 MFMAs (one quadrant of the wave tile x K = 64); each phase =
//     [L: fragment reads of that quadrant + one 16 KB staging part of the NEXT tile] barrier [M: 16 MFMAs] barrier.
//     The two token halves run one barrier apart (wm = 1 executes one extra barrier up front), so on every SIMD one wave
//     is in its M segment while its partner is in its L segment: LDS reads and LDS-DMA issue hide under the partner's MFMAs.
//   * staging parts per K tile (16 KB each, 2 LDS-DMA instructions per thread), in issue order W01, XA, W23, XB:
//       XA / XB = token rows {0-63, 128-191} / {64-127, 192-255} (first / second half of each token half),
//       W01 / W23 = the 8-row groups of every 64-row slab that weight fragments j = 0,1 / j = 2,3 read (w_frag_row).
//     Part p of tile t+1 is issued in phase p of tile t and first read 3-5 phases later; counted vmcnt only (never a
//     drain inside the loop): a wave waits for ITS pieces at the end of an M segment two segments before the first read,
//     so that the partner half -- one barrier behind -- has also waited by then (RAW needs wait -> barrier -> read for
//     every writer).  WAR: every wave retires its fragment reads (lgkmcnt(0)) before the barrier that ends its L segment,
//     and a buffer is restaged one K tile after it was read.
// Roofline: MFMA-bound (2.5 PFLOP/s dense bf16 / f16); 2 * M * N * K FLOP per launch.
#include "imh_common.h"
#include "imh_kernels.h"
#include "imh_gemm_epilogue.h"
#include <type_traits>

namespace imh {

constexpr int PP_BM = 256, PP_BN = 256;
constexpr int PP_BUF = (PP_BM + PP_BN) * GEMM_ROW_BYTES;      // 64 KB per buffer
constexpr int PP_WOFF = PP_BM * GEMM_ROW_BYTES;               // weight tile offset inside a buffer

// first tile row of staging piece (part, q, wave): every piece is 8 consecutive rows (one wave instruction)
__host__ __device__ inline int pp_piece_row(int part, int q, int wave) {
    const int g = q * 8 + wave;                                // 16 pieces per part
    switch (part) {
    case 1: return (g < 8 ? 0 : 128) + (g & 7) * 8;            // XA: token rows 0-63, 128-191
    case 3: return (g < 8 ? 64 : 192) + (g & 7) * 8;           // XB: token rows 64-127, 192-255
    case 0: return (g >> 2) * 64 + (g & 3) * 16;               // W01: rows {0-7} of every 16-group of every slab
    default: return (g >> 2) * 64 + (g & 3) * 16 + 8;          // W23: rows {8-15}
    }
}

template <typename T>
__device__ __forceinline__ void gemm_pp_body(const GemmParams& p) {
    constexpr int FN = 4;
    typedef typename Vec<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    int m0, n0;
    if (!xcd_tile<PP_BM, PP_BN>(p, blockIdx.x, m0, n0)) {      // padding workgroup of a ragged partition
        tail_prefetch(p.pf_ptr, p.pf_bytes, blockIdx.x, gridDim.x, tid, 512);
        return;
    }
    const int nt = p.K / GEMM_BK;

    // ---- staging sources: piece (part, q) of this wave; lane -> (row = piece row + lane / 8, physical chunk = lane % 8) ----
    // (rows beyond M / N are clamped to the last valid row instead of a zero page: whatever they produce -- outputs and
    // LayerNorm statistics alike -- belongs to rows / columns the epilogue never stores)
    const unsigned char* src[8];
#pragma unroll
    for (int part = 0; part < 4; ++part)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = pp_piece_row(part, q, wave) + (lane >> 3);
            const bool isx = part & 1;
            const int c = isx ? stage_chunk_x(row, lane) : stage_chunk_w(row, lane, FN);
            const int g = min((isx ? m0 : n0) + row, (isx ? p.M : p.N) - 1);
            const unsigned char* base = (const unsigned char*)(isx ? p.X : p.W);
            const int ld = isx ? p.ldx : p.ldw;
            src[part * 2 + q] = base + (size_t)g * ld * sizeof(T) + c * 16;
        }
    auto stage = [&](auto PART, int buf, int kt) {
        constexpr int part = decltype(PART)::value;
        unsigned char* dst = smem + buf * PP_BUF + ((part & 1) ? 0 : PP_WOFF);
#pragma unroll
        for (int q = 0; q < 2; ++q)
            glds16(src[part * 2 + q] + (size_t)kt * (GEMM_BK * sizeof(T)), dst + pp_piece_row(part, q, wave) * GEMM_ROW_BYTES);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using P2 = std::integral_constant<int, 2>;
    using P3 = std::integral_constant<int, 3>;

    // ---- fragment read offsets (same maps as gemm.hip / gemm_ring.hip with TM = 128, TN = 64) ----
    int xoff[2], woff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int xr = wm * 128 + (lane & 15);
        const int wr = wn * 64 + w_frag_row(lane & 15, 0, FN);
        xoff[kk] = tile_off(xr, kk * 4 + (lane >> 4), swz_x(xr));
        woff[kk] = PP_WOFF + tile_off(wr, kk * 4 + (lane >> 4), swz_w(wr, FN));
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: the whole first K tile, then the skew barrier of the second token half ----
    stage(P0{}, 0, 0); stage(P1{}, 0, 0); stage(P2{}, 0, 0); stage(P3{}, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();

#define PP_WAIT_NEXT() do { if (nxt) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
#define PP_END_L() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_END_M() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
    int cur = 0;
    for (int t = 0; t < nt; ++t) {
        const bool nxt = t + 1 < nt;
        const unsigned char* sb = smem + cur * PP_BUF;
        v8 xf[4][2], wa[2][2], wb[2][2];

        // ---------------- phase 0: token fragments 0-3 x weight fragments 0,1 ----------------
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) wa[j][kk] = *(const v8*)(sb + woff[kk] + j * 4 * GEMM_ROW_BYTES);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) xf[i][kk] = *(const v8*)(sb + xoff[kk] + i * 16 * GEMM_ROW_BYTES);
        if (nxt) stage(P0{}, cur ^ 1, t + 1);                    // W01 of the next tile
        PP_END_L();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(wa[j][kk], xf[i][kk], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        PP_WAIT_NEXT();                                          // XB of THIS tile (issued in phase 3 of the previous one)
        PP_END_M();

        // ---------------- phase 1: token fragments 0-3 x weight fragments 2,3 ----------------
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) wb[j][kk] = *(const v8*)(sb + woff[kk] + (2 + j) * 4 * GEMM_ROW_BYTES);
        if (nxt) stage(P1{}, cur ^ 1, t + 1);                    // XA
        PP_END_L();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][2 + j] = mfma16(wb[j][kk], xf[i][kk], acc[i][2 + j]);
        __builtin_amdgcn_s_setprio(0);
        PP_END_M();

        // ---------------- phase 2: token fragments 4-7 x weight fragments 2,3 ----------------
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) xf[i][kk] = *(const v8*)(sb + xoff[kk] + (4 + i) * 16 * GEMM_ROW_BYTES);
        if (nxt) stage(P2{}, cur ^ 1, t + 1);                    // W23
        PP_END_L();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][2 + j] = mfma16(wb[j][kk], xf[i][kk], acc[4 + i][2 + j]);
        __builtin_amdgcn_s_setprio(0);
        PP_WAIT_NEXT();                                          // W01 + XA of the next tile
        PP_END_M();

        // ---------------- phase 3: token fragments 4-7 x weight fragments 0,1 ----------------
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) wa[j][kk] = *(const v8*)(sb + woff[kk] + j * 4 * GEMM_ROW_BYTES);
        if (nxt) stage(P3{}, cur ^ 1, t + 1);                    // XB
        PP_END_L();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][j] = mfma16(wa[j][kk], xf[i][kk], acc[4 + i][j]);
        __builtin_amdgcn_s_setprio(0);
        PP_WAIT_NEXT();                                          // W23 of the next tile
        PP_END_M();
        cur ^= 1;
    }
#undef PP_WAIT_NEXT
#undef PP_END_L
#undef PP_END_M
    if (wm == 0) __builtin_amdgcn_s_barrier();                  // balance the skew barrier: everyone is out of the loop

    // ---- epilogue: lane owns columns nb .. nb+15 of row m ----
    const int nb = n0 + wn * 64 + (lane >> 4) * 16;
    float lnpre[8 * FN];
    const bool have_pre = ln_preload<4 * FN>(p, nb, lnpre);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wm * 128 + i * 16 + (lane & 15);
        if (m >= p.M || nb >= p.N) continue;
        float v[4 * FN];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j * 4 + r] = acc[i][j][r];
        epilogue_store_pre<T, FN>(p, v, m, nb, lnpre, have_pre, nullptr);
    }
    tail_prefetch(p.pf_ptr, p.pf_bytes, blockIdx.x, gridDim.x, tid, 512);
}

// ---------------------------------------------------------------------------------------------------------------------
// 128 x 320 ping-pong tile: the EXACT 256-way tiling of the GEGLU projection (2048 x 10240 -> 16 x 32 = 512 tiles = two
// full rounds on 256 CUs; 8192 x 5120 -> 1024 tiles = four), where 256 x 256 tiles leave a quarter-filled second round.
//   * 8 waves as 2 weight halves (group = wave >> 2, 160 weight rows each) x 4 token quarters (32 tokens each); wave tile
//     32 x 160 = 2 x 10 accumulator fragments (80 VGPRs), lane owns 40 consecutive output columns.
//   * two phases per K tile: weight fragments 0-5 (24 MFMAs) and 6-9 (16 MFMAs); the token fragments are read once
//     (phase 0) and stay in registers.  The two weight halves run one barrier apart (ping-pong, as above).
//   * staging parts per K tile: X (16 KB, 2 LDS-DMA per thread), WA = rows 0-23 of every 40-row w_frag_row group (what
//     phase 0 reads; 3 per thread), WB = rows 24-39 (phase 1; 2 per thread).  Only WB of tile t+1 is issued during phase 0
//     of tile t; X and WA of tile t+2 are issued in phase 1 of tile t INTO THE BUFFER OF TILE t (their rows were last read
//     in phase 0 of tile t by both halves) -- 1.5 K tiles of prefetch depth out of two buffers.
//   RAW / WAR bookkeeping as in the 256 x 256 kernel: waits at the end of the M segment two segments before the first read
//   (vmcnt(2) after phase 0: X + WA of t+1 landed, WB of t+1 may fly; vmcnt(5) after phase 1: WB of t+1 landed, X + WA of
//   t+2 may fly), lgkmcnt(0) before the barrier that ends every L segment.
constexpr int PQ_BM = 128, PQ_BN = 320, PQ_FN = 10;
constexpr int PQ_BUF = (PQ_BM + PQ_BN) * GEMM_ROW_BYTES;      // 56 KB per buffer
constexpr int PQ_WOFF = PQ_BM * GEMM_ROW_BYTES;

// first tile row of staging piece q of `wave`: part 0 = X (q < 2), 1 = WA (q < 3), 2 = WB (q < 2)
__host__ __device__ inline int pq_piece_row(int part, int q, int wave) {
    const int g = q * 8 + wave;
    switch (part) {
    case 0: return g * 8;                                      // 16 pieces: token rows 0-127
    case 1: return (g / 3) * 40 + (g % 3) * 8;                 // 24 pieces: rows 0-23 of the 8 groups of 40
    default: return (g >> 1) * 40 + 24 + (g & 1) * 8;          // 16 pieces: rows 24-39
    }
}

template <typename T>
__device__ __forceinline__ void gemm_pq_body(const GemmParams& p) {
    constexpr int FN = PQ_FN;
    typedef typename Vec<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wm = wave & 3;                    // wn = ping-pong group

    int m0, n0;
    if (!xcd_tile<PQ_BM, PQ_BN>(p, blockIdx.x, m0, n0)) {
        tail_prefetch(p.pf_ptr, p.pf_bytes, blockIdx.x, gridDim.x, tid, 512);
        return;
    }
    const int nt = p.K / GEMM_BK;

    const unsigned char* sx[2];
    const unsigned char* sa[3];
    const unsigned char* sw[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = pq_piece_row(0, q, wave) + (lane >> 3);
        const int g = min(m0 + row, p.M - 1);
        sx[q] = (const unsigned char*)p.X + (size_t)g * p.ldx * sizeof(T) + stage_chunk_x(row, lane) * 16;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int row = pq_piece_row(1, q, wave) + (lane >> 3);
        const int g = min(n0 + row, p.N - 1);
        sa[q] = (const unsigned char*)p.W + (size_t)g * p.ldw * sizeof(T) + stage_chunk_w(row, lane, FN) * 16;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = pq_piece_row(2, q, wave) + (lane >> 3);
        const int g = min(n0 + row, p.N - 1);
        sw[q] = (const unsigned char*)p.W + (size_t)g * p.ldw * sizeof(T) + stage_chunk_w(row, lane, FN) * 16;
    }
    auto stage_xa = [&](int buf, int kt) {
        unsigned char* b = smem + buf * PQ_BUF;
        const size_t ko = (size_t)kt * (GEMM_BK * sizeof(T));
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(sx[q] + ko, b + pq_piece_row(0, q, wave) * GEMM_ROW_BYTES);
#pragma unroll
        for (int q = 0; q < 3; ++q) glds16(sa[q] + ko, b + PQ_WOFF + pq_piece_row(1, q, wave) * GEMM_ROW_BYTES);
    };
    auto stage_wb = [&](int buf, int kt) {
        unsigned char* b = smem + buf * PQ_BUF + PQ_WOFF;
        const size_t ko = (size_t)kt * (GEMM_BK * sizeof(T));
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(sw[q] + ko, b + pq_piece_row(2, q, wave) * GEMM_ROW_BYTES);
    };

    int xoff[2], woff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int xr = wm * 32 + (lane & 15);
        const int wr = wn * 160 + w_frag_row(lane & 15, 0, FN);
        xoff[kk] = tile_off(xr, kk * 4 + (lane >> 4), swz_x(xr));
        woff[kk] = PQ_WOFF + tile_off(wr, kk * 4 + (lane >> 4), swz_w(wr, FN));
    }

    f32x4 acc[2][10];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: tile 0 completely, X + WA of tile 1; then the skew barrier of the second half ----
    stage_xa(0, 0);
    stage_wb(0, 0);
    if (nt > 1) {
        stage_xa(1, 1);
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wn == 1) __builtin_amdgcn_s_barrier();

#define PQ_END_L() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PQ_END_M() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        const unsigned char* sb = smem + cur * PQ_BUF;
        v8 xf[2][2], wf[6][2];
        // ---------------- phase 0: weight fragments 0-5 ----------------
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) xf[i][kk] = *(const v8*)(sb + xoff[kk] + i * 16 * GEMM_ROW_BYTES);
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) wf[j][kk] = *(const v8*)(sb + woff[kk] + j * 4 * GEMM_ROW_BYTES);
        if (t + 1 < nt) stage_wb(cur ^ 1, t + 1);
        PQ_END_L();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = mfma16(wf[j][kk], xf[i][kk], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // X + WA of tile t+1 (WB of t+1 may fly)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PQ_END_M();
        // ---------------- phase 1: weight fragments 6-9 ----------------
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) wf[j][kk] = *(const v8*)(sb + woff[kk] + (6 + j) * 4 * GEMM_ROW_BYTES);
        if (t + 2 < nt) stage_xa(cur, t + 2);                   // into THIS tile's buffer: its X / WA rows are dead
        PQ_END_L();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][6 + j] = mfma16(wf[j][kk], xf[i][kk], acc[i][6 + j]);
        __builtin_amdgcn_s_setprio(0);
        if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");      // WB of tile t+1 (X + WA of t+2 may fly)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PQ_END_M();
    }
#undef PQ_END_L
#undef PQ_END_M
    if (wn == 0) __builtin_amdgcn_s_barrier();                  // balance the skew barrier

    // ---- epilogue: lane owns columns nb .. nb+39 of row m, handled as five 8-column pieces (keeps the live set small) ----
    const int nb = n0 + wn * 160 + (lane >> 4) * 40;
    // s_n, c_n of all five pieces and the bias are fetched once, ahead of the first store (see EpiPre)
    float lnpre[5][16];
    bool have_pre[5];
    EpiPre<8> pre[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        have_pre[c] = ln_preload<8>(p, nb + 8 * c, lnpre[c]);
        pre[c].ok = !p.rowadd && !p.residual && epilogue_fast<T, 8>(p, nb + 8 * c);
        if (pre[c].ok && p.bias) ldv<T, 8>((const T*)p.bias + nb + 8 * c, pre[c].bias);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 32 + i * 16 + (lane & 15);
        if (m >= p.M || nb >= p.N) continue;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            float v[8];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[jj * 4 + r] = acc[i][2 * c + jj][r];
            epilogue_store_pre<T, 2>(p, v, m, nb + 8 * c, lnpre[c], have_pre[c], nullptr, &pre[c]);
        }
    }
    tail_prefetch(p.pf_ptr, p.pf_bytes, blockIdx.x, gridDim.x, tid, 512);
}

template <typename T>
IMH_KERNEL __launch_bounds__(512, 2) void gemm_pq_kernel(const GemmParams p) {
    gemm_pq_body<T>(p);
}

template <typename T>
static int launch_pq(const GemmParams& p, hipStream_t stream) {
    GemmParams q = p;
    int tiles;
    xcd_partition(q, PQ_BM, PQ_BN, &tiles);
    const size_t smem = 2 * (size_t)PQ_BUF;
    auto kern = gemm_pq_kernel<T>;
    static DynLdsOnce lds_once;
    lds_once.ensure((const void*)kern, (int)smem);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), smem, stream, q);
    return check_launch("gemm_pq_kernel");
}

// ---------------------------------------------------------------------------------------------------------------------
// 256 x 320 ping-pong tile: ONE tile per CU for the GEGLU projection at 1024^2 (2048 x 10240 -> 8 x 32 = 256 tiles) -- the
// fewest operand bytes through the L2 -> LDS path per FLOP that 160 KB of LDS and 256 registers per wave allow
// (2 x 72 KB buffers; 160 accumulator registers), with no tile-count quantisation.
//   * 8 waves as 2 weight halves (group = wave >> 2, 160 weight rows each) x 4 token quarters (64 tokens each); wave tile
//     64 x 160 = 4 x 10 accumulator fragments; lane owns 40 consecutive output columns.
//   * four phases per K tile, ordered (k-half 0: weight fragments 0-5 | 6-9), (k-half 1: 0-5 | 6-9): 24 / 16 / 24 / 16
//     MFMAs; the four token fragments of a k-half are read in its first phase and reused in its second.
//   * staging, 9 LDS-DMA instructions per thread per K tile, 2-3 per load segment: phase 3 of tile t issues the first
//     half of X of tile t+2 INTO TILE t's OWN BUFFER (its token rows were last read in phase 2); phase 0 of tile t+1 the
//     second half, phase 1 WA (weight rows 0-23 of every 40-row group, what phases 0 / 2 read), phase 2 WB (rows 24-39)
//     of tile t+2's predecessor ... i.e. in tile t: phase 0 X[128:256)(t+1), phase 1 WA(t+1), phase 2 WB(t+1), phase 3
//     X[0:128)(t+2).  Waits: vmcnt(2) after phase 2 (X + WA of t+1 landed, WB of t+1 may fly) and vmcnt(2) after phase 3
//     (WB of t+1 landed, the first X half of t+2 may fly) -- each two segments before the first read.
constexpr int PR_BM = 256, PR_BN = 320;
constexpr int PR_BUF = (PR_BM + PR_BN) * GEMM_ROW_BYTES;      // 72 KB per buffer
constexpr int PR_WOFF = PR_BM * GEMM_ROW_BYTES;

template <typename T>
__device__ __forceinline__ void gemm_pr_body(const GemmParams& p) {
    constexpr int FN = PQ_FN;
    typedef typename Vec<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wm = wave & 3;                    // wn = ping-pong group

    int m0, n0;
    if (!xcd_tile<PR_BM, PR_BN>(p, blockIdx.x, m0, n0)) {
        tail_prefetch(p.pf_ptr, p.pf_bytes, blockIdx.x, gridDim.x, tid, 512);
        return;
    }
    const int nt = p.K / GEMM_BK;

    // staging sources as 32-bit byte offsets from the (uniform) operand bases
    unsigned ox[4], oa[3], ow[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = (q * 8 + wave) * 8 + (lane >> 3);       // 32 pieces: token rows 0-255
        ox[q] = (unsigned)min(m0 + row, p.M - 1) * (unsigned)(p.ldx * sizeof(T)) + stage_chunk_x(row, lane) * 16;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int row = pq_piece_row(1, q, wave) + (lane >> 3);
        oa[q] = (unsigned)min(n0 + row, p.N - 1) * (unsigned)(p.ldw * sizeof(T)) + stage_chunk_w(row, lane, FN) * 16;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = pq_piece_row(2, q, wave) + (lane >> 3);
        ow[q] = (unsigned)min(n0 + row, p.N - 1) * (unsigned)(p.ldw * sizeof(T)) + stage_chunk_w(row, lane, FN) * 16;
    }
    const unsigned char* Xb = (const unsigned char*)p.X;
    const unsigned char* Wb = (const unsigned char*)p.W;
    // X + WA of a tile go out in three groups (2 + 2 + 3 LDS-DMA instructions per thread) so that no load segment carries
    // more than three: an LDS-DMA instruction costs 60-185 cycles of issue time (MI355X_MICROARCH.md), and a load segment
    // longer than the partner's 16-24 MFMAs stretches the whole phase
    auto stage_x01 = [&](int buf, int kt) {
        unsigned char* b = smem + buf * PR_BUF;
        const unsigned ko = (unsigned)kt * (unsigned)(GEMM_BK * sizeof(T));
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(Xb + (ox[q] + ko), b + (q * 8 + wave) * 8 * GEMM_ROW_BYTES);
    };
    auto stage_x23 = [&](int buf, int kt) {
        unsigned char* b = smem + buf * PR_BUF;
        const unsigned ko = (unsigned)kt * (unsigned)(GEMM_BK * sizeof(T));
#pragma unroll
        for (int q = 2; q < 4; ++q) glds16(Xb + (ox[q] + ko), b + (q * 8 + wave) * 8 * GEMM_ROW_BYTES);
    };
    auto stage_wa = [&](int buf, int kt) {
        unsigned char* b = smem + buf * PR_BUF;
        const unsigned ko = (unsigned)kt * (unsigned)(GEMM_BK * sizeof(T));
#pragma unroll
        for (int q = 0; q < 3; ++q) glds16(Wb + (oa[q] + ko), b + PR_WOFF + pq_piece_row(1, q, wave) * GEMM_ROW_BYTES);
    };
    auto stage_wb = [&](int buf, int kt) {
        unsigned char* b = smem + buf * PR_BUF + PR_WOFF;
        const unsigned ko = (unsigned)kt * (unsigned)(GEMM_BK * sizeof(T));
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(Wb + (ow[q] + ko), b + pq_piece_row(2, q, wave) * GEMM_ROW_BYTES);
    };

    int xoff[2], woff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int xr = wm * 64 + (lane & 15);
        const int wr = wn * 160 + w_frag_row(lane & 15, 0, FN);
        xoff[kk] = tile_off(xr, kk * 4 + (lane >> 4), swz_x(xr));
        woff[kk] = PR_WOFF + tile_off(wr, kk * 4 + (lane >> 4), swz_w(wr, FN));
    }

    f32x4 acc[4][10];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: tile 0 completely; of tile 1 the first X group (what phase 3 of "tile -1" would have issued)
    stage_x01(0, 0); stage_x23(0, 0); stage_wa(0, 0); stage_wb(0, 0);
    if (nt > 1) {
        stage_x01(1, 1);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wn == 1) __builtin_amdgcn_s_barrier();

// PR_TIMING (tools/pp_phase_probe.py only): cycle counter at every segment boundary; wave 0 of workgroup 0 and wave 4
// write their per-segment totals [issue part, wait part] to p.pf_ptr instead of prefetching
#ifndef PR_TIMING
#define PR_TIMING 0
#endif
#if PR_TIMING
    unsigned long long tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tm0 = __builtin_readcyclecounter();
    int tseg = 0;
#define PR_TICK(i) do { asm volatile("s_nop 0" ::: "memory"); const unsigned long long tn_ = __builtin_readcyclecounter(); tacc[i] += tn_ - tm0; tm0 = tn_; } while (0)
#else
#define PR_TICK(i) do {} while (0)
#endif
#define PR_END_L() do { PR_TICK(tseg * 4 + 0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); PR_TICK(tseg * 4 + 1); } while (0)
#define PR_END_M() do { __builtin_amdgcn_sched_barrier(0); PR_TICK(tseg * 4 + 2); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); PR_TICK(tseg * 4 + 3); } while (0)
#define PR_MFMA(J0, NJ) do { __builtin_amdgcn_s_setprio(1); \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) _Pragma("unroll") for (int i = 0; i < 4; ++i) \
            acc[i][J0 + j] = mfma16(wf[j], xf[i], acc[i][J0 + j]); \
        __builtin_amdgcn_s_setprio(0); } while (0)
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        const unsigned char* sb = smem + cur * PR_BUF;
        v8 xf[4], wf[6];
        // ---------------- phase 0: k-half 0, weight fragments 0-5 ----------------
#if PR_TIMING
        tseg = 0;
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = *(const v8*)(sb + xoff[0] + i * 16 * GEMM_ROW_BYTES);
#pragma unroll
        for (int j = 0; j < 6; ++j) wf[j] = *(const v8*)(sb + woff[0] + j * 4 * GEMM_ROW_BYTES);
        if (t + 1 < nt) stage_x23(cur ^ 1, t + 1);
        PR_END_L();
        PR_MFMA(0, 6);
        PR_END_M();
        // ---------------- phase 1: k-half 0, weight fragments 6-9 ----------------
#if PR_TIMING
        tseg = 1;
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *(const v8*)(sb + woff[0] + (6 + j) * 4 * GEMM_ROW_BYTES);
        if (t + 1 < nt) stage_wa(cur ^ 1, t + 1);
        PR_END_L();
        PR_MFMA(6, 4);
        PR_END_M();
        // ---------------- phase 2: k-half 1, weight fragments 0-5 ----------------
#if PR_TIMING
        tseg = 2;
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = *(const v8*)(sb + xoff[1] + i * 16 * GEMM_ROW_BYTES);
#pragma unroll
        for (int j = 0; j < 6; ++j) wf[j] = *(const v8*)(sb + woff[1] + j * 4 * GEMM_ROW_BYTES);
        if (t + 1 < nt) stage_wb(cur ^ 1, t + 1);
        PR_END_L();
        PR_MFMA(0, 6);
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // X + WA of tile t+1 (WB of t+1 may fly)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PR_END_M();
        // ---------------- phase 3: k-half 1, weight fragments 6-9 ----------------
#if PR_TIMING
        tseg = 3;
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *(const v8*)(sb + woff[1] + (6 + j) * 4 * GEMM_ROW_BYTES);
        if (t + 2 < nt) stage_x01(cur, t + 2);                  // into THIS tile's buffer: its X rows are dead since phase 2
        PR_END_L();
        PR_MFMA(6, 4);
        if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // WB of tile t+1 (first X group of t+2 may fly)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PR_END_M();
    }
#undef PR_END_L
#undef PR_END_M
#undef PR_MFMA
#if PR_TIMING
    if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4) && p.pf_ptr) {
        unsigned long long* dbg = (unsigned long long*)p.pf_ptr + (wave >> 2) * 17;
        for (int i = 0; i < 16; ++i) dbg[i] = tacc[i];
        dbg[16] = nt;
    }
#endif
    if (wn == 0) __builtin_amdgcn_s_barrier();                  // balance the skew barrier

    // ---- epilogue: lane owns columns nb .. nb+39 of row m, handled as five 8-column pieces ----
    const int nb = n0 + wn * 160 + (lane >> 4) * 40;
    // (compile-time row / piece indices: a rolled loop here would turn acc[][] into a scratch array)
    auto piece = [&](auto I, auto C, int m) {
        constexpr int i = decltype(I)::value, c = decltype(C)::value;
        float v[8], lnpre[16];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[jj * 4 + r] = acc[i][2 * c + jj][r];
        const bool have_pre = ln_preload<8>(p, nb + 8 * c, lnpre);
        epilogue_store_pre<T, 2>(p, v, m, nb + 8 * c, lnpre, have_pre, nullptr);
    };
    auto row = [&](auto I) {
        constexpr int i = decltype(I)::value;
        const int m = m0 + wm * 64 + i * 16 + (lane & 15);
        if (m >= p.M || nb >= p.N) return;
        piece(I, std::integral_constant<int, 0>{}, m); piece(I, std::integral_constant<int, 1>{}, m);
        piece(I, std::integral_constant<int, 2>{}, m); piece(I, std::integral_constant<int, 3>{}, m);
        piece(I, std::integral_constant<int, 4>{}, m);
    };
    row(std::integral_constant<int, 0>{}); row(std::integral_constant<int, 1>{});
    row(std::integral_constant<int, 2>{}); row(std::integral_constant<int, 3>{});
    if (!PR_TIMING) tail_prefetch(p.pf_ptr, p.pf_bytes, blockIdx.x, gridDim.x, tid, 512);
}

template <typename T>
IMH_KERNEL __launch_bounds__(512, 2) void gemm_pr_kernel(const GemmParams p) {
    gemm_pr_body<T>(p);
}

template <typename T>
static int launch_pr(const GemmParams& p, hipStream_t stream) {
    GemmParams q = p;
    int tiles;
    xcd_partition(q, PR_BM, PR_BN, &tiles);
    const size_t smem = 2 * (size_t)PR_BUF;
    auto kern = gemm_pr_kernel<T>;
    static DynLdsOnce lds_once;
    lds_once.ensure((const void*)kern, (int)smem);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), smem, stream, q);
    return check_launch("gemm_pr_kernel");
}

template <typename T>
IMH_KERNEL __launch_bounds__(512, 2) void gemm_pp_kernel(const GemmParams p) {
    gemm_pp_body<T>(p);
}

template <typename T>
static int launch_pp(const GemmParams& p, hipStream_t stream) {
    GemmParams q = p;
    int tiles;
    xcd_partition(q, PP_BM, PP_BN, &tiles);
    const size_t smem = 2 * (size_t)PP_BUF;
    auto kern = gemm_pp_kernel<T>;
    static DynLdsOnce lds_once;
    lds_once.ensure((const void*)kern, (int)smem);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), smem, stream, q);
    return check_launch("gemm_pp_kernel");
}

// variant codes (bm field of the config): 8256 x 256, 9128 x 320 and 9256 x 320
int gemm_pp_launch(const GemmParams& p, int dtype, int conv, int bm, hipStream_t stream) {
    if (conv || p.splits > 1 || (p.flags & (GF_LN_ROW | GF_LN_COL | GF_VT_PERM))) {
        set_error("gemm_pp: plain GEMMs only (no conv, no folded LayerNorm, no V^T permutation), splits == 1");
        return IMH_ERR_ARG;
    }
    if (dtype != IMH_DT_BF16 && dtype != IMH_DT_F16) { set_error("gemm_pp: unknown dtype %d", dtype); return IMH_ERR_DTYPE; }
    const bool bf = dtype == IMH_DT_BF16;
    if (bm == 9256) {
#ifdef IMH_EXPERIMENTAL
        if ((size_t)p.M * p.ldx * 2 >= (1ull << 32) || (size_t)p.N * p.ldw * 2 >= (1ull << 32)) { set_error("gemm_pp: operand too large for 32-bit staging offsets"); return IMH_ERR_SHAPE; }
        return bf ? launch_pr<bf16_t>(p, stream) : launch_pr<f16_t>(p, stream);
#else
        return experimental_refused("the 256 x 320 ping-pong variant (9256)");
#endif
    }
    if (bm == 9128) return bf ? launch_pq<bf16_t>(p, stream) : launch_pq<f16_t>(p, stream);
    return bf ? launch_pp<bf16_t>(p, stream) : launch_pp<f16_t>(p, stream);
}

}  // namespace imh
