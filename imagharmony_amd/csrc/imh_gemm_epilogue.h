// Shared GEMM epilogue (bias / time-embedding row-add / activation / GEGLU / residual / V^T permutation),
// used by gemm.hip and gemm_ring.hip.
#pragma once
#include "imh_common.h"
#include "imh_kernels.h"
#include "imh_lnstats.h"

namespace imh {

// XCD-aware workgroup -> tile map.  Workgroup b runs on XCD b % 8 (observed dispatch order; used for L2
// locality only, never for correctness).  Each XCD owns one cell of a px x py partition of the tile grid
// (tmx x tny tiles per cell), so its private 4 MB L2 sees only M/px rows of X and N/py rows of W instead of
// every XCD re-fetching both operands (8x fetch amplification measured with FETCH_SIZE, profiles/r01_pmc_*).
// Returns false for the padding workgroups of ragged partitions.
template <int BM, int BN>
__device__ __forceinline__ bool xcd_tile(const GemmParams& p, int b, int& m0, int& n0) {
    if (p.px == 0) {                      // legacy row-major tile order (tuning / A-B only)
        m0 = (b / p.tny) * BM;
        n0 = (b % p.tny) * BN;
        return m0 < p.M;
    }
    const int xcd = b & 7, s = b >> 3;
    const int xi = xcd / p.py, yi = xcd - xi * p.py;
    const int ms = s / p.tny, ns = s - ms * p.tny;
    m0 = (xi * p.tmx + ms) * BM;
    n0 = (yi * p.tny + ns) * BN;
    return ms < p.tmx && m0 < p.M && n0 < p.N;
}

// host side: choose the partition that minimises the bytes the 8 L2s fetch together, py*|X| + px*|W|,
// penalising partitions whose ragged cells launch many padding workgroups
// g_xcd_mode (imh_kernels.h): 0 auto, 1 legacy row-major, 2..5 force (8,1) (4,2) (2,4) (1,8)   (imh_debug_set key 2)
inline void xcd_partition(GemmParams& p, int bm, int bn, int* grid) {
    const int tm = (p.M + bm - 1) / bm, tn = (p.N + bn - 1) / bn;
    if (g_xcd_mode == 1) { p.px = 0; p.py = 1; p.tmx = tm; p.tny = tn; *grid = tm * tn; return; }
    int bpx = 8, bpy = 1;
    double best = 1e300;
    const int cand[4][2] = {{8, 1}, {4, 2}, {2, 4}, {1, 8}};
    for (auto& c : cand) {
        const int tmx = (tm + c[0] - 1) / c[0], tny = (tn + c[1] - 1) / c[1];
        const double waste = (double)(tmx * c[0]) * (tny * c[1]) / ((double)tm * tn);
        const double cost = ((double)c[1] * p.M + (double)c[0] * p.N) * waste * waste;
        if (cost < best) { best = cost; bpx = c[0]; bpy = c[1]; }
    }
    const int forced = g_xcd_mode ? g_xcd_mode : p.xcd;          // the debug knob wins over the launch's own request
    if (forced >= 2 && forced <= 5) { bpx = cand[forced - 2][0]; bpy = cand[forced - 2][1]; }
    p.px = bpx; p.py = bpy;
    p.tmx = (tm + bpx - 1) / bpx;
    p.tny = (tn + bpy - 1) / bpy;
    *grid = 8 * p.tmx * p.tny;
}

enum : int {
    GF_GEGLU = 1,      // columns interleaved in quads (value, value, gate, gate): out[2k], out[2k+1] = v[4k], v[4k+1] * gelu(v[4k+2], v[4k+3])
    GF_ACT_GELU = 2,   // exact-erf GELU on the (biased) result
    GF_ACT_SILU = 4,
    GF_VT_PERM = 8,    // permute each 16-column group [0-3,8-11,4-7,12-15] (attention V^T layout)
    GF_OUT_F32 = 16,   // store fp32 instead of T
    GF_LN_ROW = 32,    // folded LayerNorm, stats per row m, ln_s / ln_c per column n
    GF_LN_COL = 64,    // folded LayerNorm, stats per column n, ln_s / ln_c per row m
};

// vector load / store of CNT (any even count) consecutive T values as floats: 16-B pieces, then one 8-B, then one 4-B piece
template <typename T, int CNT>
__device__ __forceinline__ void ldv(const T* p, float* o) {
    static_assert(CNT % 2 == 0, "vector epilogue works on even column counts");
#pragma unroll
    for (int q0 = 0; q0 + 8 <= CNT; q0 += 8) {
        typename Vec<T>::v8 t = *(const typename Vec<T>::v8*)(p + q0);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[q0 + e] = to_f32(t[e]);
    }
    constexpr int R4 = CNT / 8 * 8;
    if constexpr (CNT % 8 >= 4) {
        typename Vec<T>::v4 t = *(const typename Vec<T>::v4*)(p + R4);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[R4 + e] = to_f32(t[e]);
    }
    if constexpr (CNT % 4 == 2) {
        typename Pk2<T>::t t = *(const typename Pk2<T>::t*)(p + CNT - 2);
        o[CNT - 2] = to_f32(t[0]); o[CNT - 1] = to_f32(t[1]);
    }
}
// the same loads kept PACKED (CNT / 2 registers instead of CNT): for operands fetched long before their use (gemm_ws's early residual)
template <typename T, int CNT>
struct RawRow {
    typename Vec<T>::v8 a[CNT / 8 > 0 ? CNT / 8 : 1];
    typename Vec<T>::v4 b;
    typename Pk2<T>::t c;
};
template <typename T, int CNT>
__device__ __forceinline__ void ldraw(const T* p, RawRow<T, CNT>& r) {
    static_assert(CNT % 2 == 0, "vector epilogue works on even column counts");
#pragma unroll
    for (int q0 = 0; q0 + 8 <= CNT; q0 += 8) r.a[q0 / 8] = *(const typename Vec<T>::v8*)(p + q0);
    constexpr int R4 = CNT / 8 * 8;
    if constexpr (CNT % 8 >= 4) r.b = *(const typename Vec<T>::v4*)(p + R4);
    if constexpr (CNT % 4 == 2) r.c = *(const typename Pk2<T>::t*)(p + CNT - 2);
}
template <typename T, int CNT>
__device__ __forceinline__ void unraw(const RawRow<T, CNT>& r, float* o) {
#pragma unroll
    for (int q0 = 0; q0 + 8 <= CNT; q0 += 8)
#pragma unroll
        for (int e = 0; e < 8; ++e) o[q0 + e] = to_f32(r.a[q0 / 8][e]);
    constexpr int R4 = CNT / 8 * 8;
    if constexpr (CNT % 8 >= 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[R4 + e] = to_f32(r.b[e]);
    }
    if constexpr (CNT % 4 == 2) { o[CNT - 2] = to_f32(r.c[0]); o[CNT - 1] = to_f32(r.c[1]); }
}
template <typename T, int CNT>
__device__ __forceinline__ void stv(T* p, const float* v) {
    static_assert(CNT % 2 == 0, "vector epilogue works on even column counts");
#pragma unroll
    for (int q0 = 0; q0 + 8 <= CNT; q0 += 8) {
        typename Vec<T>::v8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = from_f32<T>(v[q0 + e]);
        *(typename Vec<T>::v8*)(p + q0) = t;
    }
    constexpr int R4 = CNT / 8 * 8;
    if constexpr (CNT % 8 >= 4) {
        typename Vec<T>::v4 t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = from_f32<T>(v[R4 + e]);
        *(typename Vec<T>::v4*)(p + R4) = t;
    }
    if constexpr (CNT % 4 == 2) {
        typename Pk2<T>::t t;
        t[0] = from_f32<T>(v[CNT - 2]); t[1] = from_f32<T>(v[CNT - 1]);
        *(typename Pk2<T>::t*)(p + CNT - 2) = t;
    }
}

// Folded LayerNorm (GF_LN_ROW / GF_LN_COL).  The statistics of the un-normalised token rows are handed over by the GEMM that
// wrote them (imh_lnstats.h merge_row_stats): LnArgs carries what one lane needs for one output row.
//   row form:  y[m, n] = rstd_m * (acc - mean_m * s_n) + c_n      (tokens = X rows; s, c per output column)
//   col form:  y[m, n] = rstd_n * (acc - mean_n * s_m) + c_m      (tokens = W rows = output columns; s, c per row)
template <int NV>
struct LnArgs {
    float mean, rstd;            // row form: statistics of output row m
    float cm[NV], cr[NV];        // col form: (mean, rstd) of the lane's NV output columns
};

// s_n, c_n of the lane's column range, fetched once per lane (row form); false if the range is ragged
template <int NV>
__device__ __forceinline__ bool ln_preload(const GemmParams& p, int nb, float (&pre)[2 * NV]) {
    if (!(p.flags & GF_LN_ROW) || nb + NV > p.N) return false;
#pragma unroll
    for (int q0 = 0; q0 < NV; q0 += 4) {
        const f32x4 s4 = *(const f32x4*)(p.ln_s + nb + q0), c4 = *(const f32x4*)(p.ln_c + nb + q0);
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) { pre[q0 + e2] = s4[e2]; pre[NV + q0 + e2] = c4[e2]; }
    }
    return true;
}

// Operands of the epilogue fetched AHEAD of the stores (epilogue_preload): a kernel that finishes a lane's columns in
// several pieces would otherwise run  load - wait - compute - store  once per piece -- the compiler may not hoist a load of
// bias / row-add / residual above the previous piece's store to Y (they could alias as far as C++ knows), and with one
// workgroup per CU nothing hides those round trips (measured: +12-15 us on the 256 x 320 / 128 x 320 GEGLU launches).
// Hoisting by hand is safe even for an in-place residual (Y == residual): every element is read before ITS OWN store,
// by the same lane.  Only valid on the fast path (vector-aligned operands, whole piece in range).
template <int NV>
struct EpiPre {
    float bias[NV], radd[NV], res[NV];      // res holds NV / 2 values for GEGLU
    bool ok;                                 // operands are in here (fast path); false -> the epilogue loads them itself
};
template <typename T, int NV>
__device__ __forceinline__ bool epilogue_fast(const GemmParams& p, int nb) {
    return (nb + NV <= p.N) && !(p.ldy & 7) && (!p.residual || !(p.ldr & 7)) && (!p.rowadd || !(p.ldra & 7));
}
template <typename T, int NV>
__device__ __forceinline__ void epilogue_preload(const GemmParams& p, int m, int nb, EpiPre<NV>& e, bool row_ok) {
    e.ok = row_ok && epilogue_fast<T, NV>(p, nb);
    if (!e.ok) return;
    if (p.bias) ldv<T, NV>((const T*)p.bias + nb, e.bias);
    if (p.rowadd) ldv<T, NV>((const T*)p.rowadd + (size_t)(m / p.rows_per_batch) * p.ldra + nb, e.radd);
    if (p.residual) {
        if (p.flags & GF_GEGLU) ldv<T, NV / 2>((const T*)p.residual + (size_t)m * p.ldr + (nb >> 1), e.res);
        else ldv<T, NV>((const T*)p.residual + (size_t)m * p.ldr + nb, e.res);
    }
}

// Epilogue of one lane: NV = 4*FN consecutive columns nb.. of row m.
// Fast path (whole vector in range, 16-B aligned operands): vector loads/stores only.
template <typename T, int FN>
__device__ __forceinline__ void epilogue_store_pre(const GemmParams& p, float (&v)[4 * FN], int m, int nb,
                                                   const float (&lnpre)[8 * FN], const bool have_pre,
                                                   const LnArgs<4 * FN>* ln = nullptr, const EpiPre<4 * FN>* pre = nullptr,
                                                   const int lane = -1, GnAcc<(4 * FN) % 10 == 0 ? 4 * FN : 10>* gn_acc = nullptr,
                                                   const bool gn_first = false) {
    constexpr int NV = 4 * FN;
    constexpr int NH = NV / 2;
    const T* bias = (const T*)p.bias;
    const T* rowadd = p.rowadd ? (const T*)p.rowadd + (size_t)(m / p.rows_per_batch) * p.ldra : nullptr;
    const T* res = (const T*)p.residual;
    const int N = p.N;
    const bool geglu = p.flags & GF_GEGLU;
    const bool fast = epilogue_fast<T, NV>(p, nb);
    const bool use_pre = pre && pre->ok;     // implies fast
    float t[NV];
    if (ln && (p.flags & (GF_LN_ROW | GF_LN_COL))) {
        if (p.flags & GF_LN_ROW) {       // y = rstd_m * (acc - mean_m * s_n) + c_n
            const float mean = ln->mean, rstd = ln->rstd;
            if (have_pre) {              // s_n, c_n preloaded once per lane by the caller (ln_preload)
#pragma unroll
                for (int q = 0; q < NV; ++q) v[q] = fma_nopk(rstd, fma_nopk(-mean, lnpre[q], v[q]), lnpre[NV + q]);
            } else {
#pragma unroll
                for (int q = 0; q < NV; ++q) if (nb + q < N) v[q] = fma_nopk(rstd, fma_nopk(-mean, p.ln_s[nb + q], v[q]), p.ln_c[nb + q]);
            }
        } else {                         // y = rstd_n * (acc - mean_n * s_m) + c_m
            const float sm = p.ln_s[m], cm = p.ln_c[m];
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] = fma_nopk(ln->cr[q], fma_nopk(-ln->cm[q], sm, v[q]), cm);
        }
    }
    if ((p.flags & GF_VT_PERM) && FN == 4) {   // V^T key permutation: swap the 2nd and 3rd run of 4
#pragma unroll
        for (int r = 0; r < 4; ++r) { float t = v[4 + r]; v[4 + r] = v[8 + r]; v[8 + r] = t; }
    }
    if (use_pre) {
        if (bias) {
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] += pre->bias[q]; }
        if (rowadd) {
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] += pre->radd[q]; }
    } else if (fast) {
        if (bias) { ldv<T, NV>(bias + nb, t);
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] += t[q]; }
        if (rowadd) { ldv<T, NV>(rowadd + nb, t);
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] += t[q]; }
    } else {
#pragma unroll
        for (int q = 0; q < NV; ++q) if (nb + q < N) {
            if (bias) v[q] += to_f32(bias[nb + q]);
            if (rowadd) v[q] += to_f32(rowadd[nb + q]);
        }
    }
    if (p.flags & GF_ACT_GELU) {
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = gelu_erf_f(v[q]);
    }
    if (p.flags & GF_ACT_SILU) {
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = silu_f(v[q]);
    }
    if (geglu) {       // (value, value, gate, gate) quads -> NH outputs at column nb/2 of an [M, N/2] result
        const int ob = nb >> 1, NO = N >> 1;
        {
            float o2[NH];
            geglu_quads<NV>(v, o2);
#pragma unroll
            for (int q = 0; q < NH; ++q) v[q] = o2[q];
        }
        T* y = (T*)p.Y + (size_t)m * p.ldy + ob;
        if (fast) {
            if (res && use_pre) {
#pragma unroll
                for (int q = 0; q < NH; ++q) v[q] += pre->res[q];
            } else if (res) { ldv<T, NH>(res + (size_t)m * p.ldr + ob, t);
#pragma unroll
                for (int q = 0; q < NH; ++q) v[q] += t[q]; }
            stv<T, NH>(y, v);
        } else {
#pragma unroll
            for (int q = 0; q < NH; ++q) if (ob + q < NO) {
                if (res) v[q] += to_f32(res[(size_t)m * p.ldr + ob + q]);
                y[q] = from_f32<T>(v[q]);
            }
        }
        return;
    }
    if (res && use_pre) {
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] += pre->res[q];
    } else if (res) {
        const T* rr = res + (size_t)m * p.ldr + nb;
        if (fast) { ldv<T, NV>(rr, t);
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] += t[q]; }
        else {
#pragma unroll
            for (int q = 0; q < NV; ++q) if (nb + q < N) v[q] += to_f32(rr[q]);
        }
    }
    if (p.flags & GF_OUT_F32) {
        float* y = (float*)p.Y + (size_t)m * p.ldy + nb;
        if (fast && !(p.ldy & 3)) {
#pragma unroll
            for (int q0 = 0; q0 < NV; q0 += 4) *(f32x4*)(y + q0) = f32x4{v[q0], v[q0 + 1], v[q0 + 2], v[q0 + 3]};
        } else {
#pragma unroll
            for (int q = 0; q < NV; ++q) if (nb + q < N) y[q] = v[q];
        }
        return;
    }
    T* y = (T*)p.Y + (size_t)m * p.ldy + nb;
    // statistics hand-over to the LayerNorm-folding consumer of Y (the launcher only sets ln_stats_out for variants that pass
    // their lane here, for full-width slots: N % (4 * NV) == 0, and for plain T outputs)
    if (p.ln_stats_out && lane >= 0) emit_row_stats<T, NV>(p.ln_stats_out, p.ln_slots_out, m, nb, v, lane);
    if constexpr (NV % 10 == 0) {      // GroupNorm partials of the output (the caller merges the lanes and stores them: gn_emit)
        if (gn_acc && p.gn_out) gn_accumulate<T, NV>(v, *gn_acc, gn_first);     // gn_acc itself must be a compile-time null / object (a selected pointer would pin it to scratch)
    }
    if (fast) stv<T, NV>(y, v);
    else {
#pragma unroll
        for (int q = 0; q < NV; ++q) if (nb + q < N) y[q] = from_f32<T>(v[q]);
    }
}

template <typename T, int FN>
__device__ __forceinline__ void epilogue_store(const GemmParams& p, float (&v)[4 * FN], int m, int nb) {
    const float none[8 * FN] = {};
    epilogue_store_pre<T, FN>(p, v, m, nb, none, false);
}

}  // namespace imh
