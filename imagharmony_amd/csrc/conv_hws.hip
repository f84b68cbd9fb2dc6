// conv3x3 (stride 1, pad 1) with an LDS-resident input halo, WAVE-SPECIALISED -- round 6, the ResnetBlock2D convolutions at the 128 x 128 and
// 64 x 64 latents (diffusers ResnetBlock2D.conv1 / conv2; call site ip_adapter/custom_pipelines.py:338-345).  Same tile geometry, LDS
// layouts, accumulation order (bit-identical results) and epilogue as conv_halo.hip's 16 x 16 / 8 x 16 patch x 160 couts forms; what
// changes is who does what inside a (chunk, tap) step:
//   * conv_halo.hip (lock-step): behind each step's barrier the eight MFMA waves issue the weight LDS-DMA, read their fragments, issue 40
//     MFMAs and meet the four halo waves (staging + GroupNorm/SiLU transform of the next chunk's halo, two 8-pixel pieces per tap) at the
//     next barrier: 2722 cycles per step for 1280 cycles of MFMA per SIMD at the 128 x 128 latent (profiles/r05_halo_phase_probe.txt).
//   * here: the eight CONSUMER waves (4 x 2, the same wave tile) read their fragments software-pipelined across the step barrier (k step 0
//     of step i beside the MFMAs of k step 1 of step i - 1: gemm_ring.hip's gemm_ws_body) and fill the weight ring THEMSELVES in the time
//     they would otherwise spend at the barrier; the four PRODUCER waves own the input side only (the next chunk's halo and its in-place
//     transform).  The first chunk's halo is transformed by all twelve waves.
// What the probes said on the way (tools/hws_phase_probe.py, profiles/r06_hws_phase_probe.txt; profiles/NEGATIVE_RESULTS.md):
//   * one wave sustains ONE LDS-DMA instruction (1 KB) per ~130 cycles whatever the prefetch depth: four producer waves that also run the
//     weight ring spend 650 cycles per step issuing five pieces each + 400 draining them, in series with their transform (950-1100), while
//     the consumers idle 700-1300 cycles at the barrier.  The slot of step - 1 is free during ALL of step, so any wave may fill it at any
//     time of the step: the second consumer wave of each SIMD (waves 4-7, which get the matrix pipe after waves 0-3) issues its 2 pieces
//     of step + S - 1 right behind the barrier, the first (3 pieces) behind its own MFMAs; each waits for its own pieces of step + 1.
//   * with the producers off the critical path the step is bound by the consumers (their second wave per SIMD arrives last): 18 | 14
//     fragment reads per 40 | 20 MFMAs per wave, and a wave's reads and MFMAs add up rather than overlap (profiles/
//     r05_lds_port_microbench.csv).  Four FAT consumers (8 | 4 patch rows x 80 couts, one per SIMD, 236 registers, token fragments
//     rotating through four registers: 13 | 9 reads per 40 | 20 MFMAs) lost 25 % of the matrix pipe to the instructions between their
//     MFMAs and could not take the weight ring: 81.8 vs 67.1 us per fused 128^2 launch -- not kept.  (Two fat waves per SIMD is what
//     works -- gemm_w16.hip's gemm_f8_kernel -- but needs a 256 x 320 workgroup tile.)
// Roofline: MFMA-bound, 2 * M * Cout * 9 Cin FLOP; algorithmic bytes per launch = input + weights + output (+ residual).
#include "imh_common.h"
#include "imh_kernels.h"
#include "imh_gemm_epilogue.h"
#include "imh_lnstats.h"
#include "imh_halo_norm.h"

namespace imh {

// HWS_TIMING (tools/hws_phase_probe.py): per-segment cycle totals of consumer wave 0 / producer wave 0 of workgroup 0, every wave's barrier wait -> p.pf_ptr
#ifndef HWS_TIMING
#define HWS_TIMING 0
#endif
#if HWS_TIMING
#define HWS_TICK(i) do { asm volatile("s_nop 0" ::: "memory"); const unsigned long long tn_ = __builtin_readcyclecounter(); tacc[i] += tn_ - tm0; tm0 = tn_; } while (0)
#else
#define HWS_TICK(i) do {} while (0)
#endif

constexpr int HW_PW = 16, HW_HW = HW_PW + 2;      // patch width (one MFMA token fragment), halo row length
constexpr int HW_FN = 5;                          // weight fragments per consumer wave: 80 of the workgroup's 160 couts
constexpr int HW_BN = 32 * HW_FN;
constexpr int HW_NC = 8, HW_NP = 4;               // consumer waves (4 patch-row groups x 2 cout halves), producer waves
constexpr int HW_W_BYTES = HW_BN * GEMM_ROW_BYTES;   // one (chunk, tap) weight tile: 20 KB
constexpr int HW_CWQ = (HW_BN / 8 + HW_NC - 1) / HW_NC;      // weight pieces (8 rows) per consumer wave per step: 3 (waves 0-3) | 2

// FM = patch rows per consumer wave: 4 -> 16 x 16 patch (256 pixels), 2 -> 8 x 16 patch (128 pixels)
template <typename T, int FM, int S>
__global__ __launch_bounds__(64 * (HW_NC + HW_NP), 1) void conv_hws_kernel(const GemmParams p, const int tiles_x, const int tiles_y, const int tiles_n) {
    constexpr int FN = HW_FN;
    constexpr int PH = 4 * FM;
    constexpr int HALO = (PH + 2) * HW_HW;             // 324 / 180 halo pixels
    constexpr int HPIECES = (HALO + 7) / 8;            // 41 / 23 staging pieces of 8 halo pixels
    constexpr int HQ = (HPIECES + HW_NP - 1) / HW_NP;  // 11 / 6 per producer wave
    constexpr int HALO_BYTES = HPIECES * 8 * GEMM_ROW_BYTES;
    constexpr int PPT = (HQ + 6) / 7;                  // pieces normalised per tap (taps 2 .. 8)
    constexpr int HQ0 = (HPIECES + HW_NC + HW_NP - 1) / (HW_NC + HW_NP);      // chunk 0: per wave of all twelve
    constexpr int NT = 64 * (HW_NC + HW_NP);
    static_assert(S >= 3, "the slot of step i - 1 is refilled while step i is read and step i + 1 has landed");
    typedef typename Vec<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const halo0 = smem;
    unsigned char* const wbuf0 = smem + 2 * HALO_BYTES;
    const float* const gtab = (const float*)(smem + 2 * HALO_BYTES + S * HW_W_BYTES);     // [Cin][2] (scale, shift) of this sample

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int t = blockIdx.x;                                  // cout tile fastest, then patch column, row, batch
    const int tn = t % tiles_n; t /= tiles_n;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int n0 = tn * HW_BN;
    const int cpt = p.Cin / GEMM_BK;                     // 64-channel chunks
    const int nsteps = 9 * cpt;
    const bool gn = p.gn_tab != nullptr || p.gn_src.partial != nullptr;
    const bool own_tab = p.gn_src.partial != nullptr;    // the table built here from the producers' partials (imh_gntable.h)

    // GroupNorm (+ SiLU) of staged pieces (8 halo pixels x 64 channels each), in place (imh_halo_norm.h); every piece of a lane holds the
    // same logical 16-B chunk (h & 7 == (lane >> 3) & 7 whatever the piece), i.e. the same 8 channels
    const int gch = ((lane & 7) ^ ((lane >> 3) & 7)) * 8;
    HaloNorm<T> hn;
    auto norm_pieces = [&](unsigned char* d, int ct, auto NMAX, const int (&piece)[decltype(NMAX)::value], const bool (&valid)[decltype(NMAX)::value],
                           const bool (&real)[decltype(NMAX)::value]) {
        constexpr int N = decltype(NMAX)::value;
        hn.load(gtab, ct, gch);
        unsigned char* addr[N];
#pragma unroll
        for (int k = 0; k < N; ++k) addr[k] = d + piece[k] * 8 * GEMM_ROW_BYTES + lane * 16;
        hn.template run<N>(addr, valid, real, p.gn_silu != 0);
    };
    auto halo_pixel_real = [&](int piece) {              // is this lane's pixel of `piece` inside the image?
        const int h = piece * 8 + (lane >> 3);
        const int hy = h / HW_HW, hx = h - hy * HW_HW;
        const int iy = ty * PH + hy - 1, ix = tx * HW_PW + hx - 1;
        return piece < HPIECES && h < HALO && iy >= 0 && iy < (p.H << p.up) && ix >= 0 && ix < (p.Wd << p.up);
    };
    auto norm_chunk0 = [&]() {                           // the first chunk's halo: every wave of the workgroup takes pieces wave, wave + 12, ...
        int piece[HQ0];
        bool valid[HQ0], real[HQ0];
#pragma unroll
        for (int q = 0; q < HQ0; ++q) { piece[q] = q * (HW_NC + HW_NP) + wave; valid[q] = piece[q] < HPIECES; real[q] = halo_pixel_real(piece[q]); }
        norm_pieces(halo0, 0, std::integral_constant<int, HQ0>{}, piece, valid, real);
    };
    if (gn && !own_tab) {                                // a table launch's output -> LDS before any LDS-DMA is in flight
        const f32x4* src = (const f32x4*)(p.gn_tab + (size_t)b * p.Cin * 2);
        f32x4* dst = (f32x4*)gtab;
        for (int i = tid; i < p.Cin / 2; i += NT) dst[i] = src[i];
        __syncthreads();
    }

    if (wave >= HW_NC) {
        // ---------------------------------------------------------------------------------------------- producer: the input side
        const int pw = wave - HW_NC;
        const int cpt1 = p.Cin1 / GEMM_BK;
        const int Hv = p.H << p.up, Wv = p.Wd << p.up;   // virtual (upsampled) input = output size
        // halo staging: piece q * 4 + pw; two sources (channels [0, Cin1) from X, the rest from X2: a channel concat read from its producers)
        const unsigned char* hsrc[HQ];
        const unsigned char* hsrc2[HQ];
        int hstep[HQ];
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const int piece = q * HW_NP + pw;
            const int h = piece * 8 + (lane >> 3);
            const int hy = h / HW_HW, hx = h - hy * HW_HW;
            const int iy = ty * PH + hy - 1, ix = tx * HW_PW + hx - 1;
            const bool ok = piece < HPIECES && h < HALO && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
            const int c = (lane & 7) ^ (h & 7);
            const size_t pix = ((size_t)b * p.H + (iy >> p.up)) * p.Wd + (ix >> p.up);
            hsrc[q] = ok ? (const unsigned char*)p.X + pix * p.Cin1 * sizeof(T) + c * 16 : g_zero_page + c * 16;
            hsrc2[q] = (ok && p.X2) ? (const unsigned char*)p.X2 + pix * (p.Cin - p.Cin1) * sizeof(T) + c * 16 : g_zero_page + c * 16;
            hstep[q] = ok ? GEMM_BK * (int)sizeof(T) : 0;
        }
        auto stage_halo = [&](int buf, int ct) {
            unsigned char* d = halo0 + buf * HALO_BYTES;
            const bool second = ct >= cpt1;              // wave-uniform
            const int cc = second ? ct - cpt1 : ct;
#pragma unroll
            for (int q = 0; q < HQ; ++q)
                if (q * HW_NP + pw < HPIECES) glds16((second ? hsrc2[q] : hsrc[q]) + (size_t)cc * hstep[q], d + (q * HW_NP + pw) * 8 * GEMM_ROW_BYTES);
        };
        unsigned realmask = 0;                           // bit q: this lane's pixel of piece q is inside the image (padding stays zero)
#pragma unroll
        for (int q = 0; q < HQ; ++q) realmask |= (hstep[q] != 0 ? 1u : 0u) << q;
        auto norm_halo = [&](int buf, int ct, int q0) {  // pieces q0 .. q0 + PPT - 1 of this wave
            int piece[PPT];
            bool valid[PPT], real[PPT];
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int q = q0 + k;
                piece[k] = q * HW_NP + pw; valid[k] = q < HQ && piece[k] < HPIECES; real[k] = (realmask >> q) & 1u;
            }
            norm_pieces(halo0 + buf * HALO_BYTES, ct, std::integral_constant<int, PPT>{}, piece, valid, real);
        };

        stage_halo(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of halo 0 have landed
        __builtin_amdgcn_s_barrier();                    // A: ... everyone's; the consumers' table is complete
        asm volatile("" ::: "memory");
        if (gn) norm_chunk0();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // B: halo 0 is normalised, step 0 may be read
        asm volatile("" ::: "memory");
#if HWS_TIMING
        unsigned long long tacc[4] = {0, 0, 0, 0}, tm0 = __builtin_readcyclecounter();
#endif
        for (int ct = 0; ct < cpt; ++ct) {
            const bool more = ct + 1 < cpt;
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                if (more) {
                    if (tap == 0) stage_halo((ct + 1) & 1, ct + 1);      // the buffer of chunk ct - 1 (every consumer is past its reads)
                    HWS_TICK(0);
                    if (tap == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // two steps of flight
                    HWS_TICK(1);
                    if (gn && tap >= 2) norm_halo((ct + 1) & 1, ct + 1, (tap - 2) * PPT);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's in-place writes are in LDS
                HWS_TICK(2);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                HWS_TICK(3);
            }
        }
#if HWS_TIMING
        if (blockIdx.x == 0 && pw == 0 && lane == 0 && p.pf_ptr) {
            unsigned long long* dbg = (unsigned long long*)p.pf_ptr + 8;
            for (int i = 0; i < 4; ++i) dbg[i] = tacc[i];
            dbg[4] = nsteps;
        }
        if (blockIdx.x == 0 && lane == 0 && p.pf_ptr) ((unsigned long long*)p.pf_ptr)[16 + wave] = tacc[3];      // every wave's barrier wait
#else
        tail_prefetch(p.pf_ptr, p.pf_bytes, blockIdx.x, gridDim.x, tid - 64 * HW_NC, 64 * HW_NP);
#endif
        return;
    }

    // -------------------------------------------------------------------------------------------------- consumer
    const int wm = wave >> 1, wn = wave & 1;
    // the weight ring: piece q * 8 + wave (< 20) of the step's [160 couts x 64] tile; rows past N fetch row N - 1 (their output columns are
    // never stored).  32-bit offsets from the uniform base keep the eight MFMA waves inside 168 registers.
    // (4 + 1 and 5 + 0 pieces per pair, and waves 4-7 issuing behind their MFMAs as well, measured the same or slower in the forward:
    // profiles/r06_forward_ab_conv_hws_weight_issue.log)
    const int nWc = wave < 4 ? HW_CWQ : HW_CWQ - 1;
    unsigned wofs[HW_CWQ];
#pragma unroll
    for (int q = 0; q < HW_CWQ; ++q) {
        const int row = min((q * HW_NC + wave) * 8 + (lane >> 3), HW_BN - 1);
        const int c = stage_chunk_w(row, lane, FN);
        wofs[q] = (unsigned)min(n0 + row, p.N - 1) * (unsigned)(p.ldw * sizeof(T)) + c * 16;
    }
    auto stage_w = [&](int st) {                         // this wave's pieces of step st = (chunk st / 9, tap st % 9) -> slot st % S
        const int ct = st / 9, tap = st - 9 * ct;
        const unsigned kb = (unsigned)(tap * cpt + ct) * (GEMM_BK * (unsigned)sizeof(T));      // packed weight K index = (ky * 3 + kx) * Cin + c
        unsigned char* d = wbuf0 + (st % S) * HW_W_BYTES;
#pragma unroll
        for (int q = 0; q < HW_CWQ; ++q)
            if (q * HW_NC + wave < HW_BN / 8) glds16((const unsigned char*)p.W + (size_t)(wofs[q] + kb), d + (q * HW_NC + wave) * 8 * GEMM_ROW_BYTES);
    };
    auto wait_w = [&](int ahead) {                       // all but this wave's pieces of the `ahead` youngest steps have landed
        if (nWc == HW_CWQ) wait_vmcnt_of<(S - 2) * HW_CWQ>(ahead * HW_CWQ);
        else wait_vmcnt_of<(S - 2) * (HW_CWQ - 1)>(ahead * (HW_CWQ - 1));
    };
#pragma unroll
    for (int j = 0; j < S - 1; ++j)
        if (j < nsteps) stage_w(j);
    if (own_tab) {                                       // beside the ring's prologue and the producers' first halo
        gn_table_of_sample(p.gn_src, b, (float*)gtab, tid / GN_GL, 64 * HW_NC / GN_GL, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the table rows are in LDS (the weight pieces stay in flight)
    }
    __builtin_amdgcn_s_barrier();                        // A
    asm volatile("" ::: "memory");
    if (gn) norm_chunk0();
    // token fragment i = patch row FM wm + i; tap (ky, kx): halo rows (FM wm + i + ky) * 18 + kx + (lane & 15)
    const int hrow0 = (FM * wm) * HW_HW + (lane & 15);
    int woff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int wr = wn * (16 * FN) + w_frag_row(lane & 15, 0, FN);
        woff[kk] = tile_off(wr, kk * 4 + (lane >> 4), swz_w(wr, FN));
    }
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    v8 xf[2][FM], wf[2][FN];                             // register set kk holds k step kk of the step being consumed
#pragma unroll
    for (int i = 0; i < FM; ++i) xf[1][i] = v8{};        // (step 0 issues the MFMAs of "step -1, k step 1" on zeros: no branch in the loop body)
#pragma unroll
    for (int j = 0; j < FN; ++j) wf[1][j] = v8{};
    auto rd = [&](auto KK, const unsigned char* hb, const unsigned char* wb, int rb) {
        constexpr int kk = decltype(KK)::value;
        const int kc = kk * 4 + (lane >> 4);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int r = rb + i * HW_HW;
            xf[kk][i] = *(const v8*)(hb + r * GEMM_ROW_BYTES + ((kc ^ (r & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) wf[kk][j] = *(const v8*)(wb + woff[kk] + j * 4 * GEMM_ROW_BYTES);
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
    // one half interval: the fragment reads of the next k step one-to-one with the first MFMAs of the previous one, the rest behind them
    auto interleave = [&]() {
#pragma unroll
        for (int k = 0; k < FM + FN; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one DS read
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
        }
        __builtin_amdgcn_sched_group_barrier(0x008, FM * FN - (FM + FN), 0);
    };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wait_w(max(0, min(S - 2, nsteps - 1)));              // this wave's pieces of weight step 0 have landed
    __builtin_amdgcn_s_barrier();                        // B: halo 0 is normalised, weight step 0 has landed
    asm volatile("" ::: "memory");
#if HWS_TIMING
    unsigned long long tacc[4] = {0, 0, 0, 0}, tm0 = __builtin_readcyclecounter();
#endif
    int step = 0;
    for (int ct = 0; ct < cpt; ++ct) {
        const unsigned char* hb = halo0 + (ct & 1) * HALO_BYTES;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap, ++step) {
            const int ky = (tap * 11) >> 5;              // tap / 3 for tap < 9
            const int rb = hrow0 + ky * HW_HW + (tap - ky * 3);
            const unsigned char* wb = wbuf0 + (step % S) * HW_W_BYTES;
            if (wave >= 4 && step + S - 1 < nsteps) stage_w(step + S - 1);       // into the slot of step - 1 (every consumer is past its reads)
            __builtin_amdgcn_sched_barrier(0);
            rd(K0, hb, wb, rb);                          // (step, k step 0)  beside the MFMAs of (step - 1, k step 1)
            mm(K1);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            rd(K1, hb, wb, rb);                          // (step, k step 1)  beside the MFMAs of (step, k step 0)
            mm(K0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            HWS_TICK(0);
            if (wave < 4 && step + S - 1 < nsteps) stage_w(step + S - 1);
            if (step + 1 < nsteps) wait_w(min(S - 2, nsteps - 2 - step));        // this wave's pieces of step + 1 have landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // step has been read: its weight slot (and, at tap 8, the halo) may be refilled
            HWS_TICK(1);
            __builtin_amdgcn_s_barrier();                            // ... and everyone's pieces of step + 1 have landed
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            HWS_TICK(2);
        }
    }
    mm(K1);
#if HWS_TIMING
    if (blockIdx.x == 0 && wave == 0 && lane == 0 && p.pf_ptr) {
        unsigned long long* dbg = (unsigned long long*)p.pf_ptr;
        dbg[0] = tacc[0]; dbg[1] = tacc[1]; dbg[2] = tacc[2]; dbg[3] = 0; dbg[4] = nsteps;
    }
    if (blockIdx.x == 0 && lane == 0 && p.pf_ptr) ((unsigned long long*)p.pf_ptr)[16 + wave] = tacc[2];      // every wave's barrier wait
#endif

    // ---- epilogue (conv_halo.hip's): lane owns couts nb .. nb + 19 (40 B) of output pixel (oy, ox); bias / time-embedding row / residual as
    //      16-B vectors once per run, 16-B stores; GroupNorm partials of the output for the GroupNorm that reads it (a wave's FM patch rows
    //      x 16 pixels are one partial block)
    const int nb = n0 + wn * (16 * FN) + (lane >> 4) * (4 * FN);
    const int ox = tx * HW_PW + (lane & 15);
    GnAcc<4 * FN> gna;
    gn_zero(gna);
    auto row = [&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (i < FM) {
            const int oy = ty * PH + FM * wm + i;
            if (oy >= p.Ho || ox >= p.Wo || nb >= p.N) return;
            const int m = (b * p.Ho + oy) * p.Wo + ox;
            float v[4 * FN];
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[j * 4 + r] = acc[i][j][r];
            const float none[8 * FN] = {};
            epilogue_store_pre<T, FN>(p, v, m, nb, none, false, nullptr, nullptr, lane, &gna, i == 0);
        }
    };
    row(std::integral_constant<int, 0>{});
    row(std::integral_constant<int, 1>{});
    row(std::integral_constant<int, 2>{});
    row(std::integral_constant<int, 3>{});
    if (p.gn_out) gn_emit<4 * FN>(p.gn_out, p.gn_nblk, p.N / 10, b, (ty * tiles_x + tx) * 4 + wm, nb, gna, FM, lane);
}

// ph = 16 | 8 (the 7256 / 7356 and 7128 variant codes at 160 couts); lds_tab = bytes of the GroupNorm table (0 without the fused front end)
int conv_hws_launch(const GemmParams& p, int dtype, int ph, int tiles_x, int tiles_y, int tiles_n, int B, int lds_tab, hipStream_t stream) {
    const int hpieces = ((ph + 2) * HW_HW + 7) / 8;
    const int S = ph == 16 ? 3 : 4;
    const int lds = 2 * hpieces * 8 * GEMM_ROW_BYTES + S * HW_W_BYTES + lds_tab;
    if (lds > 160 * 1024) { set_error("conv_hws: %d bytes of LDS (patch height %d, Cin=%d with the GroupNorm table)", lds, ph, p.Cin); return IMH_ERR_SHAPE; }
    if ((size_t)p.N * p.ldw * 2 >= (1ull << 32)) { set_error("conv_hws: weight matrix beyond 4 GB (32-bit row offsets)"); return IMH_ERR_SHAPE; }
    dim3 grid(B * tiles_y * tiles_x * tiles_n);
#define IMH_HWS(TT, FMV, SV) do { auto kern = conv_hws_kernel<TT, FMV, SV>; static DynLdsOnce lds_once; lds_once.ensure((const void*)kern, lds); \
        hipLaunchKernelGGL(kern, grid, dim3(64 * (HW_NC + HW_NP)), lds, stream, p, tiles_x, tiles_y, tiles_n); } while (0)
    if (dtype == IMH_DT_BF16) { if (ph == 16) IMH_HWS(bf16_t, 4, 3); else IMH_HWS(bf16_t, 2, 4); }
    else { if (ph == 16) IMH_HWS(f16_t, 4, 3); else IMH_HWS(f16_t, 2, 4); }
#undef IMH_HWS
    return check_launch("conv_hws_kernel");
}

}  // namespace imh
