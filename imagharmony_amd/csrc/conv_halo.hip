// conv3x3 (stride 1, pad 1, optional fused nearest x2 upsampling) with an LDS-resident input HALO for gfx950 -- the
// ResnetBlock2D convolutions of the SDXL UNet at the 128 x 128 latent (diffusers ResnetBlock2D.conv1 / conv2, Upsample2D.conv;
// call site ip_adapter/custom_pipelines.py:338-345).
//
// The implicit-GEMM kernels (gemm.hip, gemm_ring.hip) fetch the token operand once PER TAP: nine [pixels x 64 channel]
// tiles per 64-channel chunk, i.e. every input pixel crosses the L2 -> LDS path nine times and (profiles/r01_pmc_*,
// r02_pmc_*) 6-8x its size crosses the fabric.  Here a workgroup owns an 8 x 16 patch of output pixels: per 64-channel
// chunk it stages the (8+2) x (16+2) input halo ONCE (180 pixels x 128 B = 22.5 KB, zero page for padding) and issues
// all nine taps from it -- the MFMA token fragment of tap (ky, kx) for patch row py is the 16 consecutive halo pixels
// (py + ky, kx .. kx + 15), a shifted window of the same LDS rows.  Weights stream as before ([320 couts x 64] per
// (chunk, tap), double-buffered LDS-DMA).
//   * 512 threads = 8 waves as 4 (patch row pairs) x 2 (cout halves of 160); wave tile 32 pixels x 160 couts = 2 x 10
//     accumulator fragments of v_mfma_f32_16x16x32; lane owns 40 consecutive output channels (NHWC row).
//   * halo rows are swizzled with (row & 7) (conflict-free for ANY window start under the ds_read_b128 bank model;
//     the (row >> 1) & 7 swizzle of the GEMM tiles is 2-way for odd starts), weights with swz_w as in gemm_ring.hip.
//   * LDS: 2 x 23 KB halo + 2 x 40 KB weights = 126 KB (86 KB for the 160-cout form), one workgroup per CU; two-stage pipeline over the
//     9 * Cin / 64 (chunk, tap) steps, the next chunk's halo issued at tap 0 (nine steps of slack).
// Epilogue: the shared one (bias, time-embedding row-add, residual).  Roofline: MFMA-bound, 2 * M * Cout * 9 Cin FLOP.
//
// Round 4: the ResnetBlock2D front end (diffusers: norm -> SiLU -> conv, SURVEY.md 2.2 "GroupNorm(32)+SiLU") lives in the halo
// staging.  With p.gn_tab the halo of a 64-channel chunk is DMA-staged RAW and then normalised IN PLACE in LDS by the wave that
// staged it -- y = silu(x * scale[b, c] + shift[b, c]), the per-sample (scale, shift) table of norm.hip gn_table_kernel copied to LDS
// once per workgroup -- before the nine taps read it: one transform per staged pixel (amortised over 9 taps x all couts), padding
// pixels stay zero (the conv pads the NORMALISED tensor), and the normalised activation never exists in memory (the gn_apply pass:
// a read + write of the whole tensor per GroupNorm).  With p.X2 the input is a channel concat [X | X2] (the up path's skip
// connections) read from its two producers chunk by chunk: the concat pass disappears as well.
#include "imh_common.h"
#include "imh_kernels.h"
#include "imh_gemm_epilogue.h"
#include "imh_lnstats.h"
#include "imh_halo_norm.h"

namespace imh {

int g_halo_mode = 0;

constexpr int CH_PW = 16;                                 // output patch width (one MFMA token fragment)
constexpr int CH_HW = CH_PW + 2;                          // halo row length (18)
// FN = weight fragments per wave: 10 -> 320 couts per workgroup (one tile per CU at the 128^2 latent with 320 channels),
// 5 -> 160 couts (twice the workgroups: 256 at the 64^2 latent with 640 channels)
// FM = token fragments (patch rows) per wave: 2 -> 8 x 16 patch (128 pixels), 1 -> 4 x 16 patch (64 pixels: twice the
// workgroups again, for the 32^2 latent)
// S = weight-tile ring depth.  Round 2 ran two stages and drained the LDS-DMA queue (vmcnt(0)) before every step's barrier,
// which also waited for the next chunk's halo although it has nine steps of slack; now the weights are an S-slot ring with
// COUNTED vmcnt (S - 2 steps and, where one was issued inside the window, the next halo stay in flight across the barrier).
// S = 3 / 4 fit for the 160-cout form (107 / 127 KB); the 320-cout form stays at S = 2 (126 KB).
// CH_TIMING (tools/halo_phase_probe.py only): cycle counter at the segment boundaries of the (chunk, tap) loop; MFMA wave 0 and halo wave 0 of
// workgroup 0 write their per-segment totals to p.pf_ptr instead of prefetching --
//   MFMA wave: [counted vmcnt] [s_barrier] [weight LDS-DMA issue (+ eight-wave form: halo issue / transform)] [fragment reads + MFMAs]
//   halo wave: [lgkmcnt(0)] [s_barrier] [halo LDS-DMA issue / wait / in-place transform]
#ifndef CH_TIMING
#define CH_TIMING 0
#endif
#if CH_TIMING
#define CH_TICK(i) do { asm volatile("s_nop 0" ::: "memory"); const unsigned long long tn_ = __builtin_readcyclecounter(); tacc[i] += tn_ - tm0; tm0 = tn_; } while (0)
#else
#define CH_TICK(i) do {} while (0)
#endif

// HWV = 0: eight waves do everything (rounds 2-3).  HWV = 4 (round 4): four extra HALO waves own the input side -- they issue the
// halo LDS-DMA of the next chunk and normalise it in place (GroupNorm + SiLU, p.gn_tab), two pieces per tap, while the eight MFMA
// waves only stream weights, read fragments and issue MFMAs: the ~280 VALU-issue cycles per staged piece (41 pieces per chunk at the
// 16 x 16 patch) no longer stall a wave that feeds the matrix pipe (with all of it inside the MFMA waves the fused launch ran
// 13-21 us longer than the plain conv -- exactly what the stand-alone apply pass costs; profiles/r04_forward_ab_qkv_gn.json).
// One s_barrier per (chunk, tap) step carries every hand-over, as before.
// KS (round 5, the 32 x 32 latent): the two waves of a patch-row pair split the K range instead of the couts -- wave (wm, kh) takes k step
// kh of every (chunk, tap) tile for ALL 16 FN couts of the workgroup, the pair's accumulators are added once through the (dead) LDS after
// the loop.  A workgroup then owns 128 pixels x 80 couts: at M = 2048 pixels x 1280 couts that is 256 workgroups (one per CU) pulling
// 10 KB of weights + 2.6 KB of halo per 80-MFMA step, where the 64 x 160 implicit GEMM pulls 28 KB (the L2 -> LDS stream is the bound of
// that kernel: tools/micro/lds_port.hip) and the 4 x 16-patch x 160-cout form of round 4 read six fragments per five MFMAs.
// SVC (round 5): the HWV extra waves are SERVICE waves -- they own every LDS-DMA of the launch (the weight ring with its counted vmcnt as
// well as the halo) besides the in-place transform, and the eight MFMA waves only read fragments and issue MFMAs between barriers: an
// MFMA wave that issues its share of the weight pieces right after the barrier queues behind the other waves' pieces in the CU's one
// vector-memory front end for 400-600 cycles per step (profiles/r05_halo_phase_probe*.txt) before it reaches its first fragment read.
// SHARE (round 6, the K-split form): the next chunk's halo is issued AHEAD of tap 0's weight pieces and waited for before tap 1's barrier,
// so that barrier publishes it to every wave and all sixteen transform it -- the MFMA waves one piece each behind their MFMAs of tap 1 (they
// wait ~1500 cycles per step at the barrier), the service waves one piece each at taps 1 and 2 -- instead of the eight service waves taking
// two pieces at tap 1 and one at tap 2 in series with their 3-4 weight pieces per step (the step's critical path, tools/halo_phase_probe.py)
template <typename T, int FN, int FM, int S, int HWV, bool KS = false, bool SVC = false, bool SHARE = false>
__global__ __launch_bounds__(64 * (8 + HWV), HWV ? (8 + HWV) / 4 : (S == 2 ? 2 : 1)) void conv_halo_kernel(const GemmParams p, const int tiles_x, const int tiles_y, const int tiles_n) {
    constexpr int NSTG = HWV ? HWV : 8;                     // waves that stage the halo
    constexpr int CH_PH = 4 * FM;                           // output patch height
    constexpr int CH_HALO = (CH_PH + 2) * CH_HW;            // 180 / 108 halo pixels
    constexpr int HPIECES = (CH_HALO + 7) / 8;              // 23 / 14 staging pieces of 8 halo pixels
    constexpr int HQ = (HPIECES + NSTG - 1) / NSTG;         // ... per staging wave
    constexpr int CH_HALO_BYTES = HPIECES * 8 * GEMM_ROW_BYTES;
    constexpr int CH_BN = (KS ? 16 : 32) * FN;
    // TPS = taps per step.  One step = one s_barrier, one counted vmcnt, one round of weight LDS-DMA issue by the MFMA waves -- measured
    // (tools/halo_phase_probe.py, profiles/r05_halo_phase_probe.txt) at 1000-1400 cycles per step on top of the MFMAs, whatever the tile.
    // The KS form's step is only 10 MFMAs per wave, so it takes a whole kernel ROW per step (ky = step of the chunk, kx = 0 .. 2): the
    // weight stage holds three [80 x 64] tap tiles (30 KB, three slots).
    constexpr int TPS = KS ? 3 : 1;
    constexpr int SPC = 9 / TPS;                            // steps per chunk
    constexpr int CH_TAP_BYTES = CH_BN * GEMM_ROW_BYTES;    // one tap's weight tile
    constexpr int CH_W_BYTES = TPS * CH_TAP_BYTES;
    constexpr int TPIECES = CH_BN / 8;                      // 8-row staging pieces of one tap's weight tile
    constexpr int WPIECES = TPS * TPIECES;                  // ... of a step
    constexpr int NWS = SVC ? HWV : 8;                      // waves that stage the weights
    constexpr int WQ = (WPIECES + NWS - 1) / NWS;           // ... pieces per such wave
    static_assert(!KS || HWV > 0, "the KS form runs with halo waves");
    static_assert(!SVC || HWV > 0, "service waves are the extra waves");
    static_assert(!SHARE || (KS && SVC && HWV == 8), "sixteen waves share the transform of the K-split form");
    typedef typename Vec<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* halo0 = smem;
    unsigned char* wbuf0 = smem + 2 * CH_HALO_BYTES;
    const float* const gtab = (const float*)(smem + 2 * CH_HALO_BYTES + S * CH_W_BYTES);     // [Cin][2] (scale, shift) of this sample (p.gn_tab)
    static_assert(S >= 2 && (S - 2) * WQ + ((HWV && !SVC) ? 0 : HQ) <= 31, "vmcnt switch range");
    static_assert(HWV ? (HQ <= 14) : (S + HQ <= 9), "the next chunk's halo is normalised piece by piece inside the current chunk's nine taps");
    constexpr int NT = 64 * (8 + HWV);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave & 7) >> 1, wn = wave & 1;
    const int sw = HWV ? wave - 8 : wave;                  // index among the halo-staging waves (negative: an MFMA wave of the HWV form)
    const int widx = SVC ? (sw < 0 ? 0 : sw) : (wave & 7);  // index among the weight-staging waves

    // tile coordinates: cout tile fastest, then patch column, row, batch
    int t = blockIdx.x;
    const int tn = t % tiles_n; t /= tiles_n;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int n0 = tn * CH_BN;
    const int Hv = p.H << p.up, Wv = p.Wd << p.up;          // virtual (upsampled) input = output size (stride 1)

    // ---- halo staging: 23 pieces of 8 halo pixels; this wave takes pieces wave, wave + 8, wave + 16 ----
    // (two sources: channels [0, Cin1) from X, the rest from X2 -- a channel concat read from its producers; Cin1 = Cin without X2)
    const int cpt1 = p.Cin1 / GEMM_BK;
    const unsigned char* hsrc[HQ];
    const unsigned char* hsrc2[HQ];
    int hstep[HQ];
#pragma unroll
    for (int q = 0; q < HQ; ++q) {
        const int piece = q * NSTG + (sw < 0 ? 0 : sw);
        const int h = piece * 8 + (lane >> 3);              // halo pixel index (may run past the halo on the last piece)
        const int hy = h / CH_HW, hx = h - hy * CH_HW;
        const int iy = ty * CH_PH + hy - 1, ix = tx * CH_PW + hx - 1;
        const bool ok = piece < HPIECES && h < CH_HALO && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
        const int c = (lane & 7) ^ (h & 7);
        const size_t pix = ((size_t)b * p.H + (iy >> p.up)) * p.Wd + (ix >> p.up);
        hsrc[q] = ok ? (const unsigned char*)p.X + pix * p.Cin1 * sizeof(T) + c * 16 : g_zero_page + c * 16;
        hsrc2[q] = (ok && p.X2) ? (const unsigned char*)p.X2 + pix * (p.Cin - p.Cin1) * sizeof(T) + c * 16 : g_zero_page + c * 16;
        hstep[q] = ok ? GEMM_BK * (int)sizeof(T) : 0;
    }
    auto stage_halo = [&](int buf, int ct) {
        unsigned char* d = halo0 + buf * CH_HALO_BYTES;
        const bool second = ct >= cpt1;                     // wave-uniform
        const int cc = second ? ct - cpt1 : ct;
#pragma unroll
        for (int q = 0; q < HQ; ++q)
            if (q * NSTG + sw < HPIECES) glds16((second ? hsrc2[q] : hsrc[q]) + (size_t)cc * hstep[q], d + (q * NSTG + sw) * 8 * GEMM_ROW_BYTES);
    };
    // GroupNorm (+ SiLU) of the staged chunk, in place, by the wave that staged it (its own DMA pieces: its own vmcnt wait covers them);
    // every piece of a lane holds the same logical 16-B chunk (h & 7 == (lane >> 3) & 7 whatever the piece), i.e. the same 8 channels
    const int gch = ((lane & 7) ^ ((lane >> 3) & 7)) * 8;
    // pieces [q0, q1) of the wave (one piece = 8 halo pixels x 64 channels), q1 - q0 <= NMAX per call: imh_halo_norm.h
    HaloNorm<T> hn;
    unsigned realmask = 0;                               // bit q: this lane's pixel of piece q is inside the image
#pragma unroll
    for (int q = 0; q < HQ; ++q) realmask |= (hstep[q] != 0 ? 1u : 0u) << q;
    auto norm_halo_n = [&](int buf, int ct, int q0, int q1, auto NMAX) {
        constexpr int N = decltype(NMAX)::value;
        unsigned char* d = halo0 + buf * CH_HALO_BYTES;
        hn.load(gtab, ct, gch);
        unsigned char* addr[N];
        bool valid[N], real[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int q = q0 + k;
            valid[k] = q < q1 && q < HQ && q * NSTG + sw < HPIECES;
            real[k] = (realmask >> q) & 1u;
            addr[k] = d + (q * NSTG + sw) * 8 * GEMM_ROW_BYTES + lane * 16;
        }
        hn.template run<N>(addr, valid, real, p.gn_silu != 0);
    };
    auto norm_halo = [&](int buf, int ct, int q0, int q1) {      // any range, in groups of two
        for (int q = q0; q < q1; q += 2) norm_halo_n(buf, ct, q, min(q + 2, q1), std::integral_constant<int, 2>{});
    };
    // ---- weight staging: WPIECES pieces of 8 rows per step (tap-major); this wave takes pieces wave + 8 q ----
    const unsigned char* wsrc[WQ];
    int wstep[WQ], wtap[WQ];
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
        const int piece = q * NWS + widx;
        wtap[q] = piece / TPIECES;                          // tap of the step this piece belongs to
        const int row = (piece - wtap[q] * TPIECES) * 8 + (lane >> 3);
        const int c = stage_chunk_w(row, lane, FN);
        const bool ok = piece < WPIECES && n0 + row < p.N;
        wsrc[q] = ok ? (const unsigned char*)p.W + (size_t)(n0 + row) * p.ldw * sizeof(T) + c * 16 : g_zero_page + c * 16;
        wstep[q] = ok ? GEMM_BK * (int)sizeof(T) : 0;
    }
    const int cpt = p.Cin / GEMM_BK;                        // 64-channel chunks
    auto stage_w_q = [&](int buf, int ct, int st, auto Q) {      // piece Q of this wave; st: step of the chunk = first tap / TPS
        constexpr int q = decltype(Q)::value;
        if constexpr (q < WQ) {
            if (q * NWS + widx < WPIECES) {
                const size_t kt = (size_t)(st * TPS + wtap[q]) * cpt + ct;      // packed weight K index = (ky*3 + kx) * Cin + c
                glds16(wsrc[q] + kt * wstep[q], wbuf0 + buf * CH_W_BYTES + (q * NWS + widx) * 8 * GEMM_ROW_BYTES);
            }
        }
    };
    auto stage_w = [&](int buf, int ct, int st) {
        static_assert(WQ <= 10, "pieces per wave");
        stage_w_q(buf, ct, st, std::integral_constant<int, 0>{}); stage_w_q(buf, ct, st, std::integral_constant<int, 1>{});
        stage_w_q(buf, ct, st, std::integral_constant<int, 2>{}); stage_w_q(buf, ct, st, std::integral_constant<int, 3>{});
        stage_w_q(buf, ct, st, std::integral_constant<int, 4>{}); stage_w_q(buf, ct, st, std::integral_constant<int, 5>{});
        stage_w_q(buf, ct, st, std::integral_constant<int, 6>{}); stage_w_q(buf, ct, st, std::integral_constant<int, 7>{});
        stage_w_q(buf, ct, st, std::integral_constant<int, 8>{}); stage_w_q(buf, ct, st, std::integral_constant<int, 9>{});
    };

    // ---- fragment read offsets ----
    // token fragment i = patch row FM wm + i; tap (ky, kx): halo rows (FM wm + i + ky) * 18 + kx + (lane & 15)
    const int hrow0 = (FM * wm) * CH_HW + (lane & 15);
    int woff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int wr = (KS ? 0 : wn * (16 * FN)) + w_frag_row(lane & 15, 0, FN);
        woff[kk] = tile_off(wr, (KS ? wn : kk) * 4 + (lane >> 4), swz_w(wr, FN));     // KS: this wave's k step is wn, whatever kk
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nsteps = SPC * cpt;
    // LDS-DMA instructions this wave issues per weight step / per halo (wave-uniform: the last round of pieces is ragged)
    const int nW = (WQ - 1) + ((WQ - 1) * NWS + widx < WPIECES ? 1 : 0);      // (per step)
    const int nH = (HWV && !SVC) ? 0 : (HQ - 1) + ((HQ - 1) * NSTG + (sw < 0 ? 0 : sw) < HPIECES ? 1 : 0);
    auto step_ct = [&](int st) { return st / SPC; };
    const bool gn = p.gn_tab != nullptr || p.gn_src.partial != nullptr;
    // this sample's table built right here from the producers' partials (imh_gntable.h): no table launch.  With extra waves the EIGHT MFMA
    // waves build it (32 quarter waves = the 32 groups in one round) while the extra waves already have the first halo (and, as service
    // waves, the weight ring's prologue) in flight -- the cold read of the partials hides under the first LDS-DMA round trip; one raw
    // s_barrier hands the table over (a __syncthreads() would drain the extra waves' LDS-DMA queue)
    const bool own_tab = p.gn_src.partial != nullptr;
    if (own_tab && HWV == 0) {
        gn_table_of_sample(p.gn_src, b, (float*)gtab, tid / GN_GL, NT / GN_GL, lane);
        __syncthreads();
    } else if (gn && !own_tab) {                                    // ... or copied from a table launch's output; -> LDS before any LDS-DMA is in flight
        const f32x4* src = (const f32x4*)(p.gn_tab + (size_t)b * p.Cin * 2);
        f32x4* dst = (f32x4*)gtab;
        for (int i = tid; i < p.Cin / 2; i += NT) dst[i] = src[i];
        __syncthreads();
    }
    if constexpr (HWV > 0) {
        if (wave >= 8) {
            // ------------------------------------------------------------------ halo wave: the input side of every chunk
            constexpr int PPT = (HQ + 6) / 7;                   // pieces normalised per tap (taps 2 .. 8)
            if constexpr (SVC) {                                // the weight ring's prologue: S - 1 steps in flight
#pragma unroll
                for (int j = 0; j < S - 1; ++j)
                    if (j < nsteps) stage_w(j % S, step_ct(j), j - SPC * step_ct(j));
            }
            stage_halo(0, 0);
            if (own_tab) {                                      // the MFMA waves' table (below) is complete
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (gn) norm_halo(0, 0, 0, HQ);
#if CH_TIMING
            unsigned long long tacc[4] = {0, 0, 0, 0}, tm0 = __builtin_readcyclecounter();
#endif
            int step = 0;
            auto staged = [&](int st) { return (SVC && st + S - 1 < nsteps) ? nW : 0; };      // weight pieces this wave issues at step st
            for (int ct = 0; ct < cpt; ++ct) {
#pragma unroll 1
                for (int tap = 0; tap < SPC; ++tap, ++step) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's in-place writes are in LDS ...
                    if constexpr (SVC) {
                        // weight step `step` has landed.  Younger loads of this wave that may stay in flight: the weight steps behind it
                        // (at most S - 2) and the next chunk's halo if it was issued inside that window (at tap 0, behind that step's weights)
                        const int ahead = min(S - 2, nsteps - 1 - step);
                        wait_vmcnt_of<(S - 2) * WQ, (S - 2) * (WQ - 1), (S - 2) * WQ + HQ, (S - 2) * WQ + HQ - 1, (S - 2) * (WQ - 1) + HQ, (S - 2) * (WQ - 1) + HQ - 1>(
                            ahead * nW + ((!SHARE && tap >= 1 && tap <= S - 1 && ct + 1 < cpt) ? nH : 0));
                    }
                    CH_TICK(0);
                    __builtin_amdgcn_s_barrier();                        // ... before the step that may read them; the halo buffer
                    asm volatile("" ::: "memory");                       // of chunk ct - 1 is free from (ct, tap 0) on
                    CH_TICK(1);
                    if constexpr (SHARE) {                               // (SHARE: the halo ahead of this step's weight pieces, which have two steps of flight)
                        if (tap == 0 && ct + 1 < cpt) stage_halo((ct + 1) & 1, ct + 1);
                    }
                    if constexpr (SVC) {
                        if (step + S - 1 < nsteps) {
                            const int ns = step + S - 1, nct = step_ct(ns);
                            stage_w(ns % S, nct, ns - SPC * nct);        // into the slot of step - 1 (every MFMA wave has read it)
                        }
                    }
                    // (interleaving the weight pieces with the transform of the halo pieces -- one LDS-DMA, one piece, ... -- and the halo
                    // ahead of step 0's weights measured SLOWER: 96.0 vs 89.6 us per fused conv2 at 32 x 32, profiles/r05_halo_phase_probe.txt)
                    if (SHARE && ct + 1 < cpt) {
                        // of the 23 pieces the MFMA waves take 0 .. 7 at tap 1 and 12 .. 19 at tap 2; the service waves the rest, ONE per wave and
                        // chunk: waves 0-3 pieces 8 .. 11 (their own second DMA piece) at tap 1, waves 4-6 pieces 20 .. 22 (their third) at tap 2
                        if (gn && ((tap == 1 && sw < 4) || (tap == 2 && sw >= 4))) norm_halo_n((ct + 1) & 1, ct + 1, tap, tap + 1, std::integral_constant<int, 1>{});
                    } else if (ct + 1 < cpt) {
                        if (tap == 0) stage_halo((ct + 1) & 1, ct + 1);
                        if constexpr (TPS == 1) {
                            // two steps of flight; behind the halo in this wave's queue: the weights issued at taps 1 and 2
                            if (tap == 2) wait_vmcnt_of<0, 2 * WQ, 2 * (WQ - 1)>(staged(step - 1) + staged(step));
                            if (gn && tap >= 2) norm_halo((ct + 1) & 1, ct + 1, (tap - 2) * PPT, (tap - 1) * PPT);
                        } else {                             // three long steps per chunk: issue / first half / second half
                            if (tap == 1) wait_vmcnt_of<0, WQ, WQ - 1>(staged(step));
                            if (gn && tap >= 1) norm_halo((ct + 1) & 1, ct + 1, (tap - 1) * ((HQ + 1) / 2), tap * ((HQ + 1) / 2));
                        }
                    }
                    CH_TICK(2);
                }
            }
#if CH_TIMING
            if (blockIdx.x == 0 && wave == 8 && lane == 0 && p.pf_ptr) {
                unsigned long long* dbg = (unsigned long long*)p.pf_ptr + 8;
                dbg[0] = tacc[0]; dbg[1] = tacc[1]; dbg[2] = tacc[2]; dbg[3] = SPC * cpt;
            }
#else
            tail_prefetch(p.pf_ptr, p.pf_bytes, blockIdx.x, gridDim.x, tid - 512, 64 * HWV);
#endif
            return;
        }
    }
    if constexpr (HWV > 0) {
        if (own_tab) {
            gn_table_of_sample(p.gn_src, b, (float*)gtab, tid / GN_GL, 512 / GN_GL, lane);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the table rows are in LDS
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    if constexpr (HWV == 0) stage_halo(0, 0);           // oldest: whoever waits for weight step 0 has the first halo too
    if constexpr (!SVC) {
#pragma unroll
        for (int j = 0; j < S - 1; ++j)
            if (j < nsteps) stage_w(j % S, step_ct(j), j - SPC * step_ct(j));
    }
    if (HWV == 0 && gn) {                               // chunk 0: wait for the own halo pieces only (the weight steps behind them stay in flight)
        wait_vmcnt_dyn(min(S - 1, nsteps) * nW);
        norm_halo(0, 0, 0, HQ);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // written back before the barrier of step 0 publishes the halo
    }
    bool share_real = false, share_real2 = false;       // SHARE: is this lane's pixel of halo piece wave | 12 + wave inside the image (padding stays zero)?
    if constexpr (SHARE) {
        static_assert(!SHARE || (HPIECES == 23), "piece split of the 8 x 16 patch");
        auto inside = [&](int piece) {
            const int h = piece * 8 + (lane >> 3);
            const int hy = h / CH_HW, hx = h - hy * CH_HW;
            const int iy = ty * CH_PH + hy - 1, ix = tx * CH_PW + hx - 1;
            return h < CH_HALO && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
        };
        share_real = inside(wave); share_real2 = inside(12 + wave);
    }
    int step = 0;
#if CH_TIMING
    unsigned long long tacc[4] = {0, 0, 0, 0}, tm0 = __builtin_readcyclecounter();
#endif
    for (int ct = 0; ct < cpt; ++ct) {
        const unsigned char* hb = halo0 + (ct & 1) * CH_HALO_BYTES;
#pragma unroll 1
        for (int tap = 0; tap < SPC; ++tap, ++step) {       // (TPS == 1: tap = the tap; TPS == 3: the kernel row ky)
            // weight step `step` must have landed.  Younger loads that may stay in flight: the weight steps behind it (at most
            // S - 2) and the next chunk's halo if it was issued inside that window (at tap 0 of this chunk, taps 1 .. S - 1 ago)
            if constexpr (!SVC) {
                const int ahead = min(S - 2, nsteps - 1 - step);
                const int halo_in_window = (HWV == 0 && tap >= 1 && tap <= S - 1 && ct + 1 < cpt) ? nH : 0;
                wait_vmcnt_of<(S - 2) * WQ, (S - 2) * (WQ - 1), (S - 2) * WQ + HQ, (S - 2) * (WQ - 1) + HQ>(ahead * nW + halo_in_window);
            }
            CH_TICK(0);
            __builtin_amdgcn_s_barrier();                            // ... everyone's; the previous step is fully consumed
            asm volatile("" ::: "memory");
            CH_TICK(1);
            if constexpr (!SVC) {
                if (step + S - 1 < nsteps) {
                    const int ns = step + S - 1, nct = step_ct(ns);
                    stage_w(ns % S, nct, ns - SPC * nct);            // into the slot of step - 1
                }
            }
            if constexpr (HWV == 0) {
                if (tap == 0 && ct + 1 < cpt) stage_halo((ct + 1) & 1, ct + 1);   // next chunk's halo: nine steps of slack
                // ... and normalised in place once it has landed for this wave: the wait above stops counting it from tap S on (it is
                // older than every weight step still in flight); the barrier of the next chunk's tap 0 publishes the result
                // -- one piece per tap (taps S .. S + HQ - 1 <= 8), so that no step carries more than ~70 extra VALU instructions
                // beside its 40 MFMAs per wave
                if (gn && tap >= S && tap < S + HQ && ct + 1 < cpt) norm_halo((ct + 1) & 1, ct + 1, tap - S, tap - S + 1);
            }
            CH_TICK(2);
#pragma unroll
            for (int tp = 0; tp < TPS; ++tp) {
                const unsigned char* wb = wbuf0 + (step % S) * CH_W_BYTES + tp * CH_TAP_BYTES;
                const int ky = TPS == 3 ? tap : ((tap * 11) >> 5);   // tap / 3 for tap < 9
                const int kx = TPS == 3 ? tp : tap - ky * 3;
#pragma unroll
                for (int kk = 0; kk < (KS ? 1 : 2); ++kk) {
                    v8 xf[FM], wf[FN];
                    const int kc = (KS ? wn : kk) * 4 + (lane >> 4);
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        const int r = hrow0 + (i + ky) * CH_HW + kx;
                        xf[i] = *(const v8*)(hb + r * GEMM_ROW_BYTES + ((kc ^ (r & 7)) << 4));
                    }
#pragma unroll
                    for (int j = 0; j < FN; ++j) wf[j] = *(const v8*)(wb + woff[kk] + j * 4 * GEMM_ROW_BYTES);
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(wf[j], xf[i], acc[i][j]);
                }
                // (80 accumulator registers: one tap's fragments at a time -- hoisting the three taps' reads spills 130 registers)
                if constexpr (TPS > 1 && FM * FN >= 20) __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("" ::: "memory");
            if constexpr (SHARE) {
                if (gn && tap >= 1 && ct + 1 < cpt) {            // piece wave | 12 + wave of the next chunk's halo (landed and published by tap 1's barrier)
                    HaloNorm<T> hm;
                    hm.load(gtab, ct + 1, gch);
                    unsigned char* const addr[1] = {halo0 + ((ct + 1) & 1) * CH_HALO_BYTES + (tap == 1 ? wave : 12 + wave) * 8 * GEMM_ROW_BYTES + lane * 16};
                    const bool valid[1] = {true}, real[1] = {tap == 1 ? share_real : share_real2};
                    hm.template run<1>(addr, valid, real, p.gn_silu != 0);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
            CH_TICK(3);
        }
    }
#if CH_TIMING
    if (blockIdx.x == 0 && wave == 0 && lane == 0 && p.pf_ptr) {
        unsigned long long* dbg = (unsigned long long*)p.pf_ptr;
        dbg[0] = tacc[0]; dbg[1] = tacc[1]; dbg[2] = tacc[2]; dbg[3] = tacc[3]; dbg[4] = nsteps;
    }
#endif

    // ---- epilogue: lane owns couts nb .. nb + 4 FN - 1 (40 or 20 consecutive channels = 80 / 40 B) of output pixel (oy, ox):
    //      the whole run goes through the shared vector epilogue at once -- bias / time-embedding row / residual fetched as
    //      16-B vectors once per run and the result stored as 16-B pieces.  (Round 2 stored 4-channel = 8-B pieces, 64
    //      different 128-B lines per store instruction: profiles/r02_pmc_hbm_traffic.json showed 3.2x write amplification on
    //      this kernel.)  Compile-time fragment indices only: a rolled loop would turn acc[][] into a scratch array.
    if constexpr (KS) {
        // the pair's k halves: wave (wm, 1) hands its accumulators to (wm, 0) through LDS (everything staged there is dead once every
        // MFMA wave has left the loop -- the halo waves are past their last barrier or gone) and leaves
        f32x4* ex = (f32x4*)smem + (size_t)wm * (FM * FN) * 64 + lane;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (wn == 1) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) ex[(i * FN + j) * 64] = acc[i][j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (wn == 1) return;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] += ex[(i * FN + j) * 64];
    }
    const int nb = n0 + (KS ? 0 : wn * (16 * FN)) + (lane >> 4) * (4 * FN);
    const int ox = tx * CH_PW + (lane & 15);
    // GroupNorm partials of the output for the GroupNorm that reads it (norm2 after conv1, the next block's norm after conv2):
    // a wave's FM patch rows x 16 pixels are one partial block
    GnAcc<4 * FN> gna;
    gn_zero(gna);
    auto row = [&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (i < FM) {
            const int oy = ty * CH_PH + FM * wm + i;
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
#if !CH_TIMING
    if (HWV == 0) tail_prefetch(p.pf_ptr, p.pf_bytes, blockIdx.x, gridDim.x, tid, 512);      // (KS: the surviving half of the waves)
#endif
}

// variant codes (bm x bn fields of the config), conv only: 7128 = 8 x 16 patch, 7564 = 4 x 16 patch; bn = 320 | 160 couts;
// 7328 / 7428 = the 8 x 16 patch with a 3- / 4-slot weight ring (160 couts only); 7256 / 7356 = a 16 x 16 patch (256 pixels x 160
// couts per workgroup, wave tile 64 pixels x 80 couts like the wave-specialised GEMM's consumers: per MFMA 0.45 fragment reads
// instead of 0.6 and 24.6 KB instead of 42.5 KB of operands per step for the 320-channel convolutions at the 128 x 128 latent)
// with a 2- / 3-slot weight ring
int conv_hws_launch(const GemmParams& p, int dtype, int ph, int tiles_x, int tiles_y, int tiles_n, int B, int lds_tab, hipStream_t stream);     // conv_hws.hip
int conv_halo_launch(const GemmParams& p, int dtype, int bm, int bn, hipStream_t stream) {
    const int ph = bm == 7564 ? 4 : ((bm == 7256 || bm == 7356) ? 16 : 8);
    // bn = 80: k halves per wave pair, three taps per step, service waves -- 7128 x 80: 8 x 16 patch, 3-slot ring; 7256 x 80: 16 x 16 patch
    // (256 pixels x 80 couts: 44 KB of LDS-DMA per 240-MFMA step instead of 22.6 KB per 80 on 7128 x 160), 2-slot ring
    const bool ks = bn == 80;
    const int S = ks ? (ph == 16 ? 2 : 3) : ((bm == 7328 || bm == 7356) ? 3 : (bm == 7428 ? 4 : 2));
    if (p.stride != 1 || p.splits > 1 || p.Cin % GEMM_BK != 0 || p.K != 9 * p.Cin || (bn != 320 && bn != 160 && bn != 80) || ((S > 2 || ph == 16) && bn != 160 && !ks) ||
        (ks && bm != 7128 && bm != 7256)) {
        set_error("conv_halo: stride-1 conv3x3 with Cin %% 64 == 0, splits == 1, bn 320 | 160 (160 only for the 3- / 4-slot rings) | 80 (7128 only) (stride=%d splits=%d Cin=%d bm=%d bn=%d)", p.stride, p.splits, p.Cin, bm, bn);
        return IMH_ERR_ARG;
    }
    if (p.Ho != (p.H << p.up) || p.Wo != (p.Wd << p.up)) { set_error("conv_halo: output size must equal the (upsampled) input size"); return IMH_ERR_SHAPE; }
    if (dtype != IMH_DT_BF16 && dtype != IMH_DT_F16) { set_error("conv_halo: unknown dtype %d", dtype); return IMH_ERR_DTYPE; }
    const int B = p.M / (p.Ho * p.Wo);
    const int tiles_x = (p.Wo + CH_PW - 1) / CH_PW, tiles_y = (p.Ho + ph - 1) / ph, tiles_n = (p.N + bn - 1) / bn;
    dim3 grid(B * tiles_y * tiles_x * tiles_n);
    const bool gnf = p.gn_tab || p.gn_src.partial;
    if (gnf && p.up) { set_error("conv_halo: the fused GroupNorm front end and the fused upsampling are separate forms"); return IMH_ERR_ARG; }
    if (p.Cin1 % GEMM_BK != 0 || p.Cin1 <= 0 || p.Cin1 > p.Cin || (p.X2 == nullptr) != (p.Cin1 == p.Cin)) {
        set_error("conv_halo: Cin1=%d must be a positive multiple of 64, = Cin=%d exactly when there is no second source", p.Cin1, p.Cin);
        return IMH_ERR_ARG;
    }
    int lds = 2 * (((ph + 2) * CH_HW + 7) / 8) * 8 * GEMM_ROW_BYTES + S * (ks ? 3 : 1) * bn * GEMM_ROW_BYTES + (gnf ? p.Cin * 8 : 0);
    if (ks && lds < 4 * (ph / 4) * 5 * 64 * 16) lds = 4 * (ph / 4) * 5 * 64 * 16;       // the pairs' accumulator exchange (40 / 80 KB) reuses the staging area
    if (lds > 160 * 1024) { set_error("conv_halo: %d bytes of LDS (variant %d x %d, Cin=%d with the GroupNorm table)", lds, bm, bn, p.Cin); return IMH_ERR_SHAPE; }
    // the fused GroupNorm front end runs on the form with four halo waves (the input side off the MFMA waves); g_halo_mode
    // (imh_debug_set key 5, A/B): 1 forces the eight-wave form, 2 the halo-wave form for every launch
    // the four halo waves for every launch (round 5: the un-fused launches gain too -- upsample 8192 x 1280 x 11520 258 -> 245 us,
    // profiles/r05_forward_ab_conv32_ks80.json `halo_w12`); g_halo_mode 1 = the eight-wave form (A/B)
    const bool hw4 = g_halo_mode != 1;
    // round 6: the 16 x 16 / 8 x 16 patch x 160 couts forms run wave-specialised (conv_hws.hip: same geometry, bit-identical results);
    // g_halo_mode 6 = the kernels below (A/B, the bit-identity test)
    if (!ks && bn == 160 && (bm == 7128 || bm == 7256 || bm == 7356) && g_halo_mode == 0)
        return conv_hws_launch(p, dtype, ph, tiles_x, tiles_y, tiles_n, B, gnf ? p.Cin * 8 : 0, stream);
#define IMH_CH6(TT, FNV, FMV, SV, HV, KSV, SVCV) IMH_CH7(TT, FNV, FMV, SV, HV, KSV, SVCV, false)
#define IMH_CH7(TT, FNV, FMV, SV, HV, KSV, SVCV, SHV) do { auto kern = conv_halo_kernel<TT, FNV, FMV, SV, HV, KSV, SVCV, SHV>; static DynLdsOnce lds_once; \
        lds_once.ensure((const void*)kern, lds); \
        hipLaunchKernelGGL(kern, grid, dim3(64 * (8 + HV)), lds, stream, p, tiles_x, tiles_y, tiles_n); } while (0)
#define IMH_CH5(TT, FNV, FMV, SV, HV, KSV) IMH_CH6(TT, FNV, FMV, SV, HV, KSV, KSV)
    // default build: every form runs with its four halo waves (the 7128 x 80 form with eight service waves).  -DIMH_EXPERIMENTAL adds the
    // eight-wave form (g_halo_mode 1), the halo waves of the other forms as service waves (g_halo_mode 3: four, measured slower there --
    // four issuing waves pull 20-40 KB of weights per step more slowly than eight, 22.14 vs 21.86 ms per forward; g_halo_mode 4: eight, on
    // the 160-cout forms that fit 128 registers: the same, 21.279 vs 21.272 ms; profiles/r05_forward_ab_halo_svc*.json), the 4 x 16 patch
    // (7564), the 3- / 4-slot weight rings of the 8 x 16 patch (7328 / 7428) and the 16 x 16 patch x 80 couts K-split form (7256 x 80:
    // +0.03 ms at 64 x 64, +0.47 ms at 128 x 128, profiles/r05_forward_ab_p16ks80.json; g_halo_mode 5: with four service waves)
#ifdef IMH_EXPERIMENTAL
    // (g_halo_mode 4: EIGHT service waves on the forms whose MFMA waves fit 128 registers -- the 8 x 16 / 4 x 16 patch x 160 couts)
#define IMH_CH3(TT, FNV, FMV, SV) do { if (hw4) { if (g_halo_mode == 3) IMH_CH6(TT, FNV, FMV, SV, 4, false, true); \
            else if (g_halo_mode == 4 && FNV == 5 && FMV <= 2) IMH_CH6(TT, 5, (FMV <= 2 ? FMV : 2), SV, 8, false, true); \
            else IMH_CH6(TT, FNV, FMV, SV, 4, false, false); } \
        else IMH_CH6(TT, FNV, FMV, SV, 0, false, false); } while (0)
#define IMH_CH(TT) do { \
        if (ks && ph == 16) { if (g_halo_mode == 5) IMH_CH6(TT, 5, 4, 2, 4, true, true); else IMH_CH6(TT, 5, 4, 2, 8, true, true); } \
        else if (ks) { if (g_halo_mode == 8) IMH_CH5(TT, 5, 2, 3, 8, true); else IMH_CH7(TT, 5, 2, 3, 8, true, true, true); } \
        else if (ph == 16) { if (S == 3) IMH_CH3(TT, 5, 4, 3); else IMH_CH3(TT, 5, 4, 2); } \
        else if (S == 3) IMH_CH3(TT, 5, 2, 3); else if (S == 4) IMH_CH3(TT, 5, 2, 4); \
        else if (bn == 320) { if (ph == 8) IMH_CH3(TT, 10, 2, 2); else IMH_CH3(TT, 10, 1, 2); } \
        else { if (ph == 8) IMH_CH3(TT, 5, 2, 2); else IMH_CH3(TT, 5, 1, 2); } } while (0)
#else
    if (!hw4 || (g_halo_mode >= 3 && g_halo_mode < 6) || bm == 7564 || bm == 7328 || bm == 7428 || (ks && ph == 16))
        return experimental_refused("this LDS-halo conv form (eight-wave / service-wave A-B modes, variants 7564 / 7328 / 7428 / 7256 x 80)");
#define IMH_CH3(TT, FNV, FMV, SV) IMH_CH6(TT, FNV, FMV, SV, 4, false, false)
#define IMH_CH(TT) do { \
        if (ks) { if (g_halo_mode == 8) IMH_CH5(TT, 5, 2, 3, 8, true); else IMH_CH7(TT, 5, 2, 3, 8, true, true, true); } \
        else if (ph == 16) { if (S == 3) IMH_CH3(TT, 5, 4, 3); else IMH_CH3(TT, 5, 4, 2); } \
        else if (bn == 320) IMH_CH3(TT, 10, 2, 2); \
        else IMH_CH3(TT, 5, 2, 2); } while (0)
#endif
    if (dtype == IMH_DT_BF16) IMH_CH(bf16_t);
    else IMH_CH(f16_t);
#undef IMH_CH
#undef IMH_CH3
#undef IMH_CH5
#undef IMH_CH6
#undef IMH_CH7
    return check_launch("conv_halo_kernel");
}

}  // namespace imh
