// MFMA GEMM / implicit-GEMM conv3x3 for gfx950.
//
//   Y[m, n] = epilogue( sum_k X[m, k] * W[n, k] )          (both operands K-contiguous)
//
// One kernel family serves every dense contraction of the SDXL UNet hot path
// (SURVEY.md 2.2): Linear layers (to_q/to_k/to_v/to_out, proj_in/out, GEGLU, FF-out,
// time/add embeddings), the 3x3 convolutions of ResnetBlock2D / Down/Upsample2D / conv_out
// as an implicit GEMM over NHWC activations (K index = (ky*3+kx)*Cin + c), and the 1x1
// shortcut convs.  Reference call sites: ip_adapter/attention_processor.py:292-300,320,
// 396,410-411,432-433,453; diffusers UNet.
//
// Structure (cdna_hip_programming.md section 5, "step-3 + swizzle" with the T3/T4 minimum
// two-phase loop): 256 threads = 2x2 waves, BK = 64, two LDS stages filled by
// global_load_lds_dwordx4 (16 B/lane, destination lane-linear), XOR swizzle applied on the
// SOURCE chunk and on the fragment READ (rule 21), v_mfma_f32_16x16x32 with the weight tile
// as the MFMA A operand so that every lane ends up owning 4*FN consecutive output columns
// (16-B vector epilogue).  Out-of-range rows and conv padding taps read a zero page, so the
// K loop has no bounds branches.  Roofline: MFMA-bound (2.5 PFLOP/s dense bf16/f16).
#include "imh_common.h"
#include "imh_kernels.h"
#include "imh_gemm_epilogue.h"
#include <algorithm>
#include <type_traits>

namespace imh {

int g_xcd_mode = 0;


// LN: 0 = none; 1 = folded LayerNorm with the token rows as the X operand (GF_LN_ROW); 2 = token rows as the W operand
// (GF_LN_COL, the swapped-operand V^T projection).  BasicTransformerBlock.norm1/2/3 cost no launch and no pass over the
// residual stream: y = rstd * (acc - mean * s) + c with W pre-scaled by gamma, and (mean, rstd) come from the (sum, M2) slot
// partials left by the GEMM that wrote the token rows (p.ln_stats, imh_lnstats.h; Chan-merged, never E[x^2] - mean^2): the
// tile's token rows are merged by one thread each while the first K tile is in flight and parked in LDS behind the two
// stages; nothing statistical happens in the K loop.  Needs splits == 1.  (Rounds 1-3 also carried a form that accumulated
// sum / sum of squares inside the K loop; it cancelled catastrophically on rows with |mean| >> sigma and is gone.)
template <typename T, int BM, int BN, bool CONV, int LN>
__device__ __forceinline__ void gemm_body(const GemmParams& p, const int bid, const int z) {
    constexpr int FM = BM / 32;   // 16-row token fragments per wave
    constexpr int FN = BN / 32;   // 16-row weight fragments per wave
    constexpr int RX = BM / 32;   // staging rounds (32 rows per round per block)
    constexpr int RW = BN / 32;
    constexpr int XT_BYTES = BM * GEMM_ROW_BYTES;
    constexpr int WT_BYTES = BN * GEMM_ROW_BYTES;
    constexpr int STAGE = XT_BYTES + WT_BYTES;
    typedef typename Vec<T>::v8 v8;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- tile coordinates; blockIdx.x walks N fastest inside an M panel so that concurrently
    //      resident blocks share the token panel in L2 ----
    int m0, n0;
    if (!xcd_tile<BM, BN>(p, bid, m0, n0)) {           // padding workgroup of a ragged partition: only prefetches
        if (z == 0) tail_prefetch(p.pf_ptr, p.pf_bytes, bid, gridDim.x, threadIdx.x, 256);
        return;
    }

    // ---- split-K range ----
    const int nkt = p.K / GEMM_BK;
    const int per = (nkt + p.splits - 1) / p.splits;
    const int kt0 = z * per;
    const int kt1 = min(nkt, kt0 + per);

    // ---- per-thread staging state ----
    const unsigned char* zero = g_zero_page;

    const unsigned char* xbase[RX];
    const unsigned char* xbase2[RX];       // second token source (p.X2: K columns [Cin1, K) of a column concat), dense rows of K - Cin1
    int xstep[RX];
    const int kt_x2 = (!CONV && p.X2) ? p.Cin1 / GEMM_BK : 0x7fffffff;
    // conv decomposition of the output pixel handled by this staging row
    int cb[RX], coy[RX], cox[RX];
    bool cvalid[RX];
#pragma unroll
    for (int i = 0; i < RX; ++i) {
        const int row = stage_row(i, wave, lane);
        const int c = stage_chunk_x(row, lane);
        const int m = m0 + row;
        xbase2[i] = zero + c * 16;
        if (!CONV) {
            const bool ok = m < p.M;
            xbase[i] = ok ? (const unsigned char*)p.X + ((size_t)m * p.ldx) * sizeof(T) + c * 16 : zero + c * 16;
            if (ok && p.X2) xbase2[i] = (const unsigned char*)p.X2 + ((size_t)m * (p.K - p.Cin1)) * sizeof(T) + c * 16;
            xstep[i] = ok ? GEMM_BK * (int)sizeof(T) : 0;
        } else {
            const bool ok = m < p.M;
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
    const int cpt = CONV ? p.Cin / GEMM_BK : 1;   // k-tiles per conv tap

    auto stage = [&](int buf, int kt) {
        unsigned char* xs = smem + buf * STAGE;
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
            const bool second = kt >= kt_x2;            // wave-uniform
#pragma unroll
            for (int i = 0; i < RX; ++i)
                glds16(second ? xbase2[i] + (size_t)(kt - kt_x2) * xstep[i] : xbase[i] + (size_t)kt * xstep[i], xs + stage_lds_off(i, wave));
        }
#pragma unroll
        for (int i = 0; i < RW; ++i)
            glds16(wbase[i] + (size_t)kt * wstep[i], ws + stage_lds_off(i, wave));
    };

    // ---- per-lane fragment read offsets ----
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
    constexpr int NS = LN == 1 ? FM : 1;
    float st_s[NS], st_q[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) { st_s[i] = 0.f; st_q[i] = 0.f; }
    f32x2s* const lnst = (f32x2s*)(smem + 2 * STAGE);      // LN != 0: (mean, rstd) of the tile's token rows

    if (kt0 < kt1) {
        stage(0, kt0);
        if constexpr (LN != 0) {                           // beside the first tile's flight (the wait below covers both)
            constexpr int NTOK = LN == 1 ? BM : BN;
            if (tid < NTOK) {
                const int tok = (LN == 1 ? m0 : n0) + tid;
                f32x2s mr = {0.f, 1.f};
                if (tok < (LN == 1 ? p.M : p.N)) mr = merge_row_stats(p.ln_stats, tok, p.ln_slots, p.K, p.ln_eps);
                lnst[tid] = mr;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // LDS-DMA landed (also implied by the fence below)
        __syncthreads();
        int cur = 0;
        for (int kt = kt0; kt < kt1; ++kt) {
            if (kt + 1 < kt1) stage(cur ^ 1, kt + 1);
            const unsigned char* xs = smem + cur * STAGE;
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
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur ^= 1;
        }
    }

    // ---- epilogue: lane owns columns nb .. nb+4*FN-1 of row m ----
    const int nb = n0 + out_col(lane, wn, BN);
    // folded-LayerNorm operands of this lane's columns: fetched once here (after the K loop, so they cost no
    // registers inside it), not once per output row
    float lnpre[8 * FN];
    const bool have_pre = ln_preload<4 * FN>(p, nb, lnpre);
    LnArgs<4 * FN> ln;
    if constexpr (LN == 1) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const f32x2s mr = lnst[wm * (BM / 2) + i * 16 + (lane & 15)];
            st_s[i] = mr[0]; st_q[i] = mr[1];
        }
    }
    if constexpr (LN == 2) {
#pragma unroll
        for (int q = 0; q < 4 * FN; ++q) {
            const f32x2s mr = lnst[wn * (BN / 2) + (lane >> 4) * 4 * FN + q];
            ln.cm[q] = mr[0]; ln.cr[q] = mr[1];
        }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + out_row(lane, wm, BM, i);
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
            if constexpr (LN == 1) { ln.mean = st_s[i]; ln.rstd = st_q[i]; }     // fragment i's row (lane & 15) IS output row m
            epilogue_store_pre<T, FN>(p, v, m, nb, lnpre, have_pre, LN != 0 ? &ln : nullptr, nullptr, lane);
        }
    }
    if (z == 0) tail_prefetch(p.pf_ptr, p.pf_bytes, bid, gridDim.x, tid, 256);
}

template <typename T, int BM, int BN, bool CONV, int LN>
IMH_KERNEL __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
    gemm_body<T, BM, BN, CONV, LN>(p, blockIdx.x, blockIdx.y);
}

// Two independent problems in ONE launch (e.g. self-attention's [Q|K] = x [Wq;Wk]^T and V^T = Wv x^T, which
// share x): workgroups [0, grid_a) run problem a, the rest problem b.  Halves the launch count of the pair and
// lets the two sub-chip-sized grids fill the machine together.
// LNP: false = plain problems; true = a carries GF_LN_ROW and b GF_LN_COL (x un-normalised, LayerNorm folded into both, row
// statistics handed over in p.ln_stats).
template <typename T, int BM, int BN, bool LNP>
IMH_KERNEL __launch_bounds__(256, 2) void gemm_dual_kernel(const GemmParams a, const GemmParams b, const int grid_a) {
    if ((int)blockIdx.x < grid_a) gemm_body<T, BM, BN, false, LNP ? 1 : 0>(a, blockIdx.x, 0);
    else gemm_body<T, BM, BN, false, LNP ? 2 : 0>(b, blockIdx.x - grid_a, 0);
}

// split-K second pass: sum the fp32 slabs and run the same epilogue. One thread per 16 columns.
template <typename T>
IMH_KERNEL __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p) {
    const int groups_n = (p.N + 15) / 16;
    const size_t total = (size_t)p.M * groups_n;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / groups_n);
        const int nb = (int)(idx % groups_n) * 16;
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = 0.f;
        for (int z = 0; z < p.splits; ++z) {
            const float* s = p.partial + ((size_t)z * p.M + m) * p.N + nb;
            if (nb + 16 <= p.N && (p.N & 3) == 0) {
#pragma unroll
                for (int q0 = 0; q0 < 16; q0 += 4) {
                    f32x4 t = *(const f32x4*)(s + q0);
                    v[q0] += t[0]; v[q0 + 1] += t[1]; v[q0 + 2] += t[2]; v[q0 + 3] += t[3];
                }
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) if (nb + q < p.N) v[q] += s[q];
            }
        }
        GemmParams q = p;
        epilogue_store<T, 4>(q, v, m, nb);
    }
}

template <typename T, int BM, int BN, bool CONV>
static int launch_tile(const GemmParams& p, hipStream_t stream) {
    GemmParams q = p;
    int tiles;
    xcd_partition(q, BM, BN, &tiles);
    const size_t smem = 2 * (size_t)(BM + BN) * GEMM_ROW_BYTES + (p.ln_stats ? (BM > BN ? BM : BN) * 8 : 0);
    dim3 grid(tiles, p.splits, 1);
    if (!CONV && (p.flags & GF_LN_ROW)) {             // (gemm_launch has checked that p.ln_stats is there)
        static DynLdsOnce once;
        once.ensure((const void*)gemm_kernel<T, BM, BN, false, 1>, (int)smem);
        hipLaunchKernelGGL((gemm_kernel<T, BM, BN, false, 1>), grid, dim3(256), smem, stream, q);
    } else if (!CONV && (p.flags & GF_LN_COL)) {
        static DynLdsOnce once;
        once.ensure((const void*)gemm_kernel<T, BM, BN, false, 2>, (int)smem);
        hipLaunchKernelGGL((gemm_kernel<T, BM, BN, false, 2>), grid, dim3(256), smem, stream, q);
    }
    else hipLaunchKernelGGL((gemm_kernel<T, BM, BN, CONV, 0>), grid, dim3(256), smem, stream, q);
    return check_launch("gemm_kernel");
}

int gemm_ring_launch(const GemmParams& p, int dtype, int conv, int bm, int bn, hipStream_t stream);
int gemm_ws_dual_launch(const GemmParams& a, const GemmParams& b, int dtype, hipStream_t stream);   // gemm_ring.hip: variant code 24128
int conv_halo_launch(const GemmParams& p, int dtype, int bm, int bn, hipStream_t stream);       // conv_halo.hip: variant codes 7128 / 7564 x 320 / 160
int gemm_pp_launch(const GemmParams& p, int dtype, int conv, int bm, hipStream_t stream);      // gemm_pp.hip: variant codes 8256 x 256, 9128 x 320

template <typename T>
static int launch_typed(const GemmParams& p, int conv, int bm, int bn, hipStream_t stream) {
    int rc;
    if (bm == 7128 || bm == 7564 || bm == 7328 || bm == 7428 || bm == 7256 || bm == 7356) {
        if (!conv) { set_error("gemm: variant 7128 is the halo conv3x3 kernel"); return IMH_ERR_ARG; }
        rc = conv_halo_launch(p, sizeof(T) == 2 && std::is_same<T, bf16_t>::value ? IMH_DT_BF16 : IMH_DT_F16, bm, bn, stream);
    } else
    if (bm == 8256 || bm == 9128 || bm == 9256) {
        rc = gemm_pp_launch(p, sizeof(T) == 2 && std::is_same<T, bf16_t>::value ? IMH_DT_BF16 : IMH_DT_F16, conv, bm, stream);
    } else
    if (bm >= 256) {
        rc = gemm_ring_launch(p, sizeof(T) == 2 && std::is_same<T, bf16_t>::value ? IMH_DT_BF16 : IMH_DT_F16, conv, bm, bn, stream);
    } else
    if (conv) {
        if (bm == 128 && bn == 128) rc = launch_tile<T, 128, 128, true>(p, stream);
        else if (bm == 128 && bn == 64) rc = launch_tile<T, 128, 64, true>(p, stream);
        else if (bm == 64 && bn == 128) rc = launch_tile<T, 64, 128, true>(p, stream);
        else if (bm == 64 && bn == 64) rc = launch_tile<T, 64, 64, true>(p, stream);
        else { set_error("gemm: unsupported tile %dx%d", bm, bn); return IMH_ERR_ARG; }
    } else {
        if (bm == 128 && bn == 128) rc = launch_tile<T, 128, 128, false>(p, stream);
        else if (bm == 128 && bn == 64) rc = launch_tile<T, 128, 64, false>(p, stream);
        else if (bm == 64 && bn == 128) rc = launch_tile<T, 64, 128, false>(p, stream);
        else if (bm == 64 && bn == 64) rc = launch_tile<T, 64, 64, false>(p, stream);
        else { set_error("gemm: unsupported tile %dx%d", bm, bn); return IMH_ERR_ARG; }
    }
    if (rc != IMH_OK) return rc;
    if (p.splits > 1) {
        const size_t total = (size_t)p.M * ((p.N + 15) / 16);
        int blocks = (int)std::min<size_t>((total + 255) / 256, 2048);
        hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, stream, p);
        rc = check_launch("splitk_reduce_kernel");
    }
    return rc;
}

template <typename T, int BM, int BN>
static int launch_dual_tile(const GemmParams& a, const GemmParams& b, hipStream_t stream) {
    GemmParams qa = a, qb = b;
    int ga, gb;
    xcd_partition(qa, BM, BN, &ga);
    xcd_partition(qb, BM, BN, &gb);
    const size_t smem = 2 * (size_t)(BM + BN) * GEMM_ROW_BYTES + (a.ln_stats ? (BM > BN ? BM : BN) * 8 : 0);
    if (a.flags & GF_LN_ROW) {
        static DynLdsOnce once;
        once.ensure((const void*)gemm_dual_kernel<T, BM, BN, true>, (int)smem);
        hipLaunchKernelGGL((gemm_dual_kernel<T, BM, BN, true>), dim3(ga + gb), dim3(256), smem, stream, qa, qb, ga);
    }
    else hipLaunchKernelGGL((gemm_dual_kernel<T, BM, BN, false>), dim3(ga + gb), dim3(256), smem, stream, qa, qb, ga);
    return check_launch("gemm_dual_kernel");
}

template <typename T>
static int launch_dual_typed(const GemmParams& a, const GemmParams& b, int bm, int bn, hipStream_t stream) {
    if (bm == 128 && bn == 128) return launch_dual_tile<T, 128, 128>(a, b, stream);
    if (bm == 128 && bn == 64) return launch_dual_tile<T, 128, 64>(a, b, stream);
    if (bm == 64 && bn == 128) return launch_dual_tile<T, 64, 128>(a, b, stream);
    if (bm == 64 && bn == 64) return launch_dual_tile<T, 64, 64>(a, b, stream);
    set_error("gemm_dual: unsupported tile %dx%d", bm, bn);
    return IMH_ERR_ARG;
}

int gemm_dual_launch(GemmParams a, GemmParams b, int dtype, int bm, int bn, hipStream_t stream) {
    for (const GemmParams* p : {&a, &b}) {
        if (p->K % GEMM_BK != 0 || p->K <= 0 || p->M <= 0 || p->N <= 0 || (p->ldx & 7) || (p->ldw & 7)) {
            set_error("gemm_dual: bad shape M=%d N=%d K=%d", p->M, p->N, p->K);
            return IMH_ERR_SHAPE;
        }
        if ((p->flags & GF_GEGLU) && (p->N & 15)) { set_error("geglu: N must be a multiple of 16"); return IMH_ERR_SHAPE; }
        if (p->rowadd && p->rows_per_batch <= 0) { set_error("gemm_dual: rowadd needs rows_per_batch"); return IMH_ERR_ARG; }
        // fields only the single-problem launcher implements (gemm_launch validates and normalises them): refuse them here instead of
        // letting a tile kernel read Cin1 / Yt unchecked
        if (p->Yt || p->gn_tab || p->X2 || p->gn_out || p->ln_stats_out || p->splits > 1) {
            set_error("gemm_dual: Yt / gn_tab / X2 / gn_out / ln_stats_out / split-K are single-problem (imh_gemm) features");
            return IMH_ERR_ARG;
        }
    }
    a.splits = b.splits = 1;
    a.Cin1 = a.K; b.Cin1 = b.K;                 // (one token source each)
    const int lnf = GF_LN_ROW | GF_LN_COL;
    if (((a.flags | b.flags) & lnf) && !((a.flags & lnf) == GF_LN_ROW && (b.flags & lnf) == GF_LN_COL)) {
        set_error("gemm_dual: folded LayerNorm needs problem a in row form and problem b in column form");
        return IMH_ERR_ARG;
    }
    if (((a.flags | b.flags) & lnf) && (!a.ln_stats || !b.ln_stats)) {
        set_error("gemm_dual: the folded LayerNorm takes the token rows' statistics from ln_stats on both problems (imh_lnstats.h)");
        return IMH_ERR_ARG;
    }
    if (bm == 24128) {       // wave-specialised pair: [Q|K] row form on 128 x 160 tiles + V^T column form on 128 x 128 tiles
        if (!a.ln_stats || !b.ln_stats || (a.flags & ~GF_LN_ROW) != 0 || !(a.flags & GF_LN_ROW) ||
            (b.flags & ~(GF_LN_COL | GF_VT_PERM)) != 0 || !(b.flags & GF_LN_COL)) {
            set_error("gemm_dual: variant 24128 is the self-attention projection pair with handed-over LayerNorm statistics "
                      "(a: IMH_GF_LN_ROW, b: IMH_GF_LN_COL [| IMH_GF_VT_PERM], ln_stats on both)");
            return IMH_ERR_ARG;
        }
        return gemm_ws_dual_launch(a, b, dtype, stream);
    }
    if (((a.flags | b.flags) & GF_VT_PERM) && bn != 128) bn = 128;     // the permutation lives in 16-column groups
    if (dtype == IMH_DT_BF16) return launch_dual_typed<bf16_t>(a, b, bm, bn, stream);
    if (dtype == IMH_DT_F16) return launch_dual_typed<f16_t>(a, b, bm, bn, stream);
    set_error("gemm_dual: unknown dtype %d", dtype);
    return IMH_ERR_DTYPE;
}

// heuristic tile / split-K choice; overridable per call (bm/bn/splits > 0)
void gemm_pick_config(int M, int N, int K, int* bm, int* bn, int* splits) {
    const int CUS = 256;
    int cbm = 128, cbn = 128;
    auto tiles = [&](int a, int b) { return ((M + a - 1) / a) * ((N + b - 1) / b); };
    if (N <= 64) cbn = 64;
    if (M <= 64) cbm = 64;
    if (tiles(cbm, cbn) < 2 * CUS && cbn == 128 && N % 128 != 0 && N % 64 == 0) cbn = 64;
    if (tiles(cbm, cbn) < CUS && cbn == 128) cbn = 64;
    if (tiles(cbm, cbn) < CUS && cbm == 128) cbm = 64;
    int s = 1;
    const int nkt = K / GEMM_BK;
    while (tiles(cbm, cbn) * s < CUS && s * 2 <= 8 && nkt / (s * 2) >= 4) s *= 2;
    *bm = cbm; *bn = cbn; *splits = s;
}

// statistics epilogue (imh_lnstats.h): a wave's four 16-lane groups own one slot of consecutive columns
int gemm_stats_slot_width(int bm, int bn) {
    if ((bm == 64 || bm == 128) && (bn == 64 || bn == 128)) return bn / 2;          // plain tiles: 2 x 2 waves
    if ((bm == 1464 || bm == 2464 || bm == 24128 || bm == 23256 || bm == 22128) && bn == 160) return 80;   // wave-specialised, CN = 2
    if ((bm == 24128 || bm == 23256) && bn == 128) return 64;
    return 0;
}

// GroupNorm epilogue (imh_lnstats.h gn_emit): pixels per partial block = one wave's rows; needs lane runs of 20 / 40 channels
int gemm_gn_block_rows(int bm, int bn) {
    if ((bm == 1464 || bm == 2464) && bn == 160) return 32;
    if ((bm == 24128 || bm == 23256 || bm == 22128) && bn == 160) return 64;
    if (bm == 7128 || bm == 7328 || bm == 7428) return 32;
    if (bm == 7564) return 16;
    if (bm == 7256 || bm == 7356) return 64;
    return 0;
}

size_t gemm_workspace_bytes(int M, int N, int splits) {
    return splits > 1 ? (size_t)splits * M * N * sizeof(float) : 0;
}

int gemm_launch(GemmParams p, int dtype, int conv, int bm, int bn, hipStream_t stream) {
    if (p.K % GEMM_BK != 0 || p.K <= 0) { set_error("gemm: K=%d must be a positive multiple of 64", p.K); return IMH_ERR_SHAPE; }
    if (conv && (p.Cin % GEMM_BK != 0 || p.K != 9 * p.Cin)) { set_error("conv3x3: Cin=%d must be a multiple of 64 and K=9*Cin", p.Cin); return IMH_ERR_SHAPE; }
    if (!conv && ((p.ldx & 7) || (p.ldw & 7))) { set_error("gemm: ldx/ldw must be multiples of 8 elements"); return IMH_ERR_SHAPE; }
    if (p.M <= 0 || p.N <= 0) { set_error("gemm: empty problem M=%d N=%d", p.M, p.N); return IMH_ERR_SHAPE; }
    if ((p.flags & GF_GEGLU) && (p.N & 15)) { set_error("geglu: N must be a multiple of 16"); return IMH_ERR_SHAPE; }
    if (p.splits < 1) p.splits = 1;
    if (p.splits > 1 && !p.partial) { set_error("gemm: split-K needs a workspace"); return IMH_ERR_WORKSPACE; }
    if (p.rowadd && p.rows_per_batch <= 0) { set_error("gemm: rowadd needs rows_per_batch"); return IMH_ERR_ARG; }
    if ((p.flags & (GF_LN_ROW | GF_LN_COL)) && !p.ln_stats) {
        set_error("gemm: the folded LayerNorm takes the token rows' statistics from ln_stats (imh_lnstats.h: the epilogue of the GEMM that "
                  "wrote the rows, or IMH_EW_ROW_STATS); there is no in-loop E[x^2] - mean^2 form");
        return IMH_ERR_ARG;
    }
    if ((p.flags & (GF_LN_ROW | GF_LN_COL)) && (p.splits > 1 || (bm >= 256 && !((bm == 1464 || bm == 2464 || bm == 24128 || bm == 23256 || bm == 22128 || bm == 26256) && (p.flags & GF_LN_ROW))) || conv)) {
        set_error("gemm: folded LayerNorm needs a plain 64/128 tile or a wave-specialised variant (row form), splits == 1, no conv (bm=%d splits=%d conv=%d)", bm, p.splits, conv);
        return IMH_ERR_ARG;
    }
    if (p.ln_stats_out) {
        const int w = gemm_stats_slot_width(bm, bn);
        if (w == 0 || p.N % w || p.ln_slots_out != p.N / w || conv || p.splits > 1 || (p.flags & (GF_GEGLU | GF_VT_PERM | GF_OUT_F32))) {
            set_error("gemm: ln_stats_out needs a variant with a statistics epilogue (slot width %d for %dx%d), N %% width == 0, "
                      "ln_slots_out == N / width, a plain output (N=%d slots=%d flags=%d splits=%d conv=%d)", w, bm, bn, p.N, p.ln_slots_out, p.flags, p.splits, conv);
            return IMH_ERR_ARG;
        }
    }
    if (p.gn_out) {
        const int rows = gemm_gn_block_rows(bm, bn);
        const bool halo = bm >= 7000 && bm < 8000;
        if (rows == 0 || p.N % 10 || p.N % (halo ? bn : 80) || p.gn_hw <= 0 || p.gn_hw % rows || p.M % p.gn_hw || p.gn_nblk != p.gn_hw / rows ||
            p.splits > 1 || (p.flags & (GF_GEGLU | GF_VT_PERM | GF_OUT_F32 | GF_LN_ROW | GF_LN_COL)) ||
            (halo && (p.Ho % (rows * 4 / 16) || p.Wo % 16))) {
            set_error("gemm: gn_out needs a variant with a GroupNorm epilogue (%d rows per block for %dx%d), N a multiple of 10 and of the tile "
                      "width, whole tiles, gn_nblk == gn_hw / rows (N=%d hw=%d nblk=%d M=%d flags=%d)", rows, bm, bn, p.N, p.gn_hw, p.gn_nblk, p.M, p.flags);
            return IMH_ERR_ARG;
        }
    }
    {
        const bool halo = bm >= 7000 && bm < 8000;
        const bool ws = bm == 1464 || bm == 2464 || bm == 24128 || bm == 23256 || bm == 22128;
        if ((p.gn_tab || p.gn_src.partial) && !(conv && halo)) {
            set_error("gemm: the fused GroupNorm front end (gn_tab / gn_part) is a form of the LDS-halo conv3x3 (variant %d)", bm);
            return IMH_ERR_ARG;
        }
        if (p.gn_src.partial && (p.gn_tab || !gn_src_ok(p.gn_src))) {
            set_error("gemm: gn_part (in-kernel GroupNorm table) is exclusive with gn_tab and needs partials whose sub-runs tile the groups "
                      "(Cin=%d groups=%d C1=%d sub=%d/%d nblk=%d/%d), eps > 0", p.Cin, p.gn_src.groups, p.gn_src.C1, p.gn_src.sub, p.gn_src.sub2,
                      p.gn_src.nblk, p.gn_src.nblk2);
            return IMH_ERR_ARG;
        }
        if (p.X2 && conv && !halo) { set_error("gemm: a two-source conv input (X2) needs the LDS-halo conv3x3 (variant %d)", bm); return IMH_ERR_ARG; }
        if (p.X2 && !conv) {      // token operand = column concat [X | X2]: plain tiles and the wave-specialised variants
            if (!(bm <= 128 || ws) || p.Cin1 <= 0 || p.Cin1 >= p.K || p.Cin1 % GEMM_BK || (p.flags & (GF_LN_ROW | GF_LN_COL | GF_VT_PERM))) {
                set_error("gemm: X2 needs a plain or wave-specialised variant, 0 < Cin1 < K, Cin1 %% 64 == 0, no folded LayerNorm (bm=%d Cin1=%d K=%d)", bm, p.Cin1, p.K);
                return IMH_ERR_ARG;
            }
        }
        if (!p.X2) p.Cin1 = conv ? p.Cin : p.K;
        if (p.Yt) {      // transposed V^T store of the columns >= yt_col0: the lean LN epilogue of the wave-specialised kernels
            const int rows = bm == 23256 ? 256 : (bm == 24128 || bm == 22128 ? 128 : 64);
            const bool wide = bn == 160 || (bm == 23256 && bn == 128);
            if (!ws || !wide || conv || p.flags != GF_LN_ROW || p.residual || p.rowadd || p.splits > 1 || p.M % rows || p.N % bn ||
                p.yt_col0 <= 0 || p.yt_col0 >= p.N || p.yt_col0 % bn || (p.ldyt & 7) || p.ldyt < p.M || (p.ldy & 7)) {
                set_error("gemm: Yt (transposed V^T store) needs a wave-specialised bn = 160 variant or 23256 x 128, flags == IMH_GF_LN_ROW only, no residual / "
                          "row-add / split-K, whole tiles (M %% %d, N %% bn), 0 < yt_col0 < N a multiple of bn, ldyt >= M a multiple of 8 "
                          "(bm=%d bn=%d M=%d N=%d yt_col0=%d ldyt=%d flags=%d)", rows, bm, bn, p.M, p.N, p.yt_col0, p.ldyt, p.flags);
                return IMH_ERR_ARG;
            }
        }
    }
    if (dtype == IMH_DT_BF16) return launch_typed<bf16_t>(p, conv, bm, bn, stream);
    if (dtype == IMH_DT_F16) return launch_typed<f16_t>(p, conv, bm, bn, stream);
    set_error("gemm: unknown dtype %d", dtype);
    return IMH_ERR_DTYPE;
}

}  // namespace imh
