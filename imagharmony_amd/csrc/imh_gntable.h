// GroupNorm step 2 (partials -> per-channel (scale, shift)) as a device routine, shared by gn_table_kernel (norm.hip) and by the consumers
// that build the table of their sample in their own prologue instead of reading one a table launch wrote (round 5: conv_halo.hip's fused
// front end, gn_apply_kernel's IMH_GN_TABLE_APPLY mode -- 45 five-microsecond launches of the SDXL forward fold away).
//
// One group = the 16 lanes of a QUARTER wave (GN_GL: a 768-thread conv workgroup then covers the 32 groups of an SDXL GroupNorm in ONE
// round -- two rounds were two dependent L2 round trips in the prologue): lane l adds the group's partials e = l, l + 16, ... (pixel
// block, sub-run) in double -- S = sum_i sum_i, Q = sum_i (M2_i + sum_i^2 / n_i), N = sum_i n_i -- a four-step xor butterfly inside the
// quarter wave gives every lane the totals (fixed order: deterministic, and the same order wherever the routine runs, so a table launch and an in-kernel table are
// bit-identical), mean = S / N, var = (Q - S^2 / N) / N: the between-partial term in double, the within-partial terms already centred
// (imh_lnstats.h gn_emit / norm.hip gn_stats_kernel: never E[x^2] - mean^2).
// Source 1 covers channels [0, C1), source 2 (optional: the other half of a channel concat) [C1, C).  npart > 0: elements per partial (a
// producer epilogue: block rows x sub); npart == 0: the ragged blocks of gn_stats_kernel.
#pragma once
#include "imh_common.h"

namespace imh {

struct GnTabSrc {
    const float* partial;     // [B][nblk][C1 / sub][2]; null -> no in-kernel table
    const float* partial2;    // [B][nblk2][(C - C1) / sub2][2] or null
    const void* gamma;        // [C] in the activation dtype, or null
    const void* beta;
    float eps;
    int groups, C, C1, HW;
    int nblk, sub, npart;
    int nblk2, sub2, npart2;
    int dtype_f16;
};

constexpr int GN_GL = 16;                                      // lanes per group
__device__ __forceinline__ double half_sum_d(double v) {      // over the GN_GL lanes of this lane's quarter wave, result in every lane
#pragma unroll
    for (int o = GN_GL / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// (mean, rstd) of group gg of sample b; l = lane & (GN_GL - 1).  Call with all GN_GL lanes of the quarter wave.
__device__ __forceinline__ void gn_group_stats(const GnTabSrc& g, const int b, const int gg, const int l, float& mean_f, float& rstd_f) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    const int cpg = g.C / g.groups;
    const int c0 = gg * cpg, c1 = c0 + cpg;
    double S = 0.0, Q = 0.0, N = 0.0;
#pragma unroll 1
    for (int src = 0; src < 2; ++src) {
        const float* pp = src == 0 ? g.partial : g.partial2;
        if (!pp) continue;
        const int cb = src == 0 ? 0 : g.C1, ce = src == 0 ? g.C1 : g.C;          // channel range of this source
        const int lo = max(c0, cb), hi = min(c1, ce);
        if (lo >= hi) continue;
        const int sub = src == 0 ? g.sub : g.sub2, nblk = src == 0 ? g.nblk : g.nblk2, npart = src == 0 ? g.npart : g.npart2;
        const int nsub = (ce - cb) / sub;
        const int j0 = (lo - cb) / sub, nj = (hi - cb) / sub - j0;
        const int ppb = (g.HW + nblk - 1) / nblk;
        const f2* base = (const f2*)pp + (size_t)b * nblk * nsub + j0;
        const int total = nblk * nj;
        const double inv_const = npart > 0 ? 1.0 / (double)npart : 0.0;
        // UB = 8 partials in flight per lane, GN_GL = 16 lanes per group: one batch covers 128 partials, so the SDXL groups with 256 partials take two dependent
        // L2 round trips (UB = 16 would take one, at 16 more live registers in the 128-register conv forms) (all loads of a batch issued before the first add; the adds keep the ascending order of e,
        // so the sums are those of the plain loop -- which cost one L2 round trip PER partial: 8-10 us in a conv's prologue)
        constexpr int UB = 8;
        for (int e0 = l; e0 < total; e0 += GN_GL * UB) {
            f2 v[UB];
            int kk[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int e = min(e0 + GN_GL * u, total - 1);
                kk[u] = e / nj;
                v[u] = base[(size_t)kk[u] * nsub + (e - kk[u] * nj)];
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (e0 + GN_GL * u >= total) continue;
                double n, inv;
                if (npart > 0) { n = (double)npart; inv = inv_const; }
                else { n = (double)(min(g.HW, (kk[u] + 1) * ppb) - kk[u] * ppb) * sub; inv = n > 0.0 ? 1.0 / n : 0.0; }
                if (n > 0.0) {
                    S += (double)v[u][0];
                    Q += (double)v[u][1] + (double)v[u][0] * (double)v[u][0] * inv;
                    N += n;
                }
            }
        }
    }
    S = half_sum_d(S); Q = half_sum_d(Q); N = half_sum_d(N);
    const double mean = S / N;
    double var = (Q - S * S / N) / N;
    if (var < 0.0) var = 0.0;
    mean_f = (float)mean;
    rstd_f = (float)(1.0 / sqrt(var + (double)g.eps));
}

// the group's rows of the table: tab[c][0] = gamma[c] * rstd, tab[c][1] = beta[c] - mean * tab[c][0]   (tab = this sample's [C][2]).
// gamma / beta of the lane's rows are fetched BEFORE the statistics (GnRows: up to GN_RPL rows per lane, cpg <= 16 * GN_RPL), so the
// prologue pays one memory round trip, not two.
constexpr int GN_RPL = 8;
struct GnRows { float gm[GN_RPL], bt[GN_RPL]; };
__device__ __forceinline__ void gn_rows_fetch(const GnTabSrc& g, const int gg, const int l, GnRows& r) {
    const int cpg = g.C / g.groups;
#pragma unroll
    for (int u = 0; u < GN_RPL; ++u) {
        const int c = gg * cpg + min(l + GN_GL * u, cpg - 1);
        r.gm[u] = g.gamma ? (g.dtype_f16 ? to_f32(((const f16_t*)g.gamma)[c]) : to_f32(((const bf16_t*)g.gamma)[c])) : 1.f;
        r.bt[u] = g.beta ? (g.dtype_f16 ? to_f32(((const f16_t*)g.beta)[c]) : to_f32(((const bf16_t*)g.beta)[c])) : 0.f;
    }
}
__device__ __forceinline__ void gn_group_rows(const GnTabSrc& g, const int gg, const int l, const float mean, const float rstd, const GnRows& r, float* tab) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    const int cpg = g.C / g.groups;
#pragma unroll
    for (int u = 0; u < GN_RPL; ++u) {
        const int cl = l + GN_GL * u;
        if (cl < cpg) {
            const float sc = r.gm[u] * rstd;
            const f2 o = {sc, r.bt[u] - mean * sc};
            *(f2*)(tab + (size_t)(gg * cpg + cl) * 2) = o;
        }
    }
    for (int cl = l + GN_GL * GN_RPL; cl < cpg; cl += GN_GL) {      // wider groups than any SDXL / VAE norm: rows beyond the prefetched ones
        const int c = gg * cpg + cl;
        const float gm = g.gamma ? (g.dtype_f16 ? to_f32(((const f16_t*)g.gamma)[c]) : to_f32(((const bf16_t*)g.gamma)[c])) : 1.f;
        const float bt = g.beta ? (g.dtype_f16 ? to_f32(((const f16_t*)g.beta)[c]) : to_f32(((const bf16_t*)g.beta)[c])) : 0.f;
        const float sc = gm * rstd;
        const f2 o = {sc, bt - mean * sc};
        *(f2*)(tab + (size_t)c * 2) = o;
    }
}

// the whole table of sample b into `tab` (LDS or global) by a workgroup of nq quarter waves; q = this lane's quarter-wave index.  The caller
// synchronises before reading.
__device__ __forceinline__ void gn_table_of_sample(const GnTabSrc& g, const int b, float* tab, const int q, const int nq, const int lane) {
    for (int gg = q; gg < g.groups; gg += nq) {
        GnRows r;
        gn_rows_fetch(g, gg, lane & (GN_GL - 1), r);
        float mean, rstd;
        gn_group_stats(g, b, gg, lane & (GN_GL - 1), mean, rstd);
        gn_group_rows(g, gg, lane & (GN_GL - 1), mean, rstd, r, tab);
    }
}

// host-side validity of a table source (the launchers' checks)
static inline bool gn_src_ok(const GnTabSrc& g) {
    if (!g.partial || g.groups <= 0 || g.groups > 64 || g.C <= 0 || g.C % g.groups || !(g.eps > 0.f) || g.HW <= 0) return false;
    const int cpg = g.C / g.groups;
    const bool two = g.partial2 != nullptr;
    const int C1 = two ? g.C1 : g.C;
    if (g.sub <= 0 || g.nblk <= 0 || C1 <= 0 || C1 > g.C || C1 % g.sub || cpg % g.sub) return false;
    if (two && (g.sub2 <= 0 || g.nblk2 <= 0 || (g.C - C1) % g.sub2 || cpg % g.sub2 || C1 % g.sub2)) return false;
    return true;
}

}  // namespace imh
