// GroupNorm(32)+SiLU over NHWC activations and LayerNorm over token rows, gfx950.
// HBM-bound kernels: 16-B vector loads/stores, fp32 statistics (PyTorch semantics:
// GroupNorm / LayerNorm accumulate in fp32 whatever the activation dtype).
// Call sites being replaced: diffusers ResnetBlock2D.norm1/norm2 (+SiLU), Transformer2DModel.norm
// (eps 1e-6, no SiLU), conv_norm_out, BasicTransformerBlock.norm1/2/3;
// ImageProjModel.norm (ip_adapter/ip_adapter.py:39,47), Resampler norms (resampler.py:15,42-43,104),
// HarmonyAttention.ln (train.py:238).
#include "imh_common.h"
#include "imh_kernels.h"
#include "imh_gntable.h"
#include <algorithm>

namespace imh {

constexpr int GN_THREADS = 512;
constexpr int GN_ELEMS_PER_BLOCK = 8192;     // 32768 left the 32 x 32 levels (1.3 M elements per sample) on 80 workgroups of 256 CUs

// GroupNorm in three steps (diffusers ResnetBlock2D.norm1/norm2 + SiLU, Transformer2DModel.norm, conv_norm_out; torch.nn.GroupNorm
// semantics: fp32 statistics, biased variance):
//   1 statistics  (sum, M2) partials per (sample, pixel block, sub-run of `sub` consecutive channels), the format of
//                 imh_lnstats.h gn_emit.  Either from the epilogue of the launch that WROTE the tensor (imh_gemm_args.gn_out, sub = 10) or
//                 from gn_stats_kernel below (a pass over the tensor) -- Welford / Chan, never E[x^2] - mean^2.
//   2 table       gn_table_kernel: per sample the partials of one or TWO producers (the second = the other half of a channel
//                 concat) are merged per group in double precision, then scale[c] = gamma[c] * rstd[g(c)], shift[c] = beta[c] - mean *
//                 scale[c] -> table[b][c][2] fp32.
//   3 apply       y = silu?(x * scale + shift): gn_apply_kernel (a pass), or INSIDE the consuming conv3x3's halo staging
//                 (conv_halo.hip, imh_gemm_args.gn_tab): the normalised tensor never exists in memory.
static inline int gn_nblk(int HW, int C) {
    long long e = (long long)HW * C;
    int n = (int)((e + GN_ELEMS_PER_BLOCK - 1) / GN_ELEMS_PER_BLOCK);
    return n < 1 ? 1 : (n > 256 ? 256 : n);
}
// channels per sub-run of the stand-alone statistics kernel: 10 where the producers' epilogues use it (every SDXL UNet width), else
// the group width itself (VAE: 4 / 8 / 16 channels per group)
static inline int gn_sub(int C, int groups) {
    const int cpg = C / groups;
    return (cpg % 10 == 0) ? 10 : cpg;
}
int groupnorm_stats_blocks(int HW, int C) { return gn_nblk(HW, C); }
int groupnorm_stats_sub(int C, int groups) { return gn_sub(C, groups); }

size_t groupnorm_workspace_bytes(int B, int HW, int C, int groups) {
    const size_t part = (size_t)B * gn_nblk(HW, C) * (C / gn_sub(C, groups)) * 2 * sizeof(float);
    return ((part + 255) & ~(size_t)255) + (size_t)B * C * 2 * sizeof(float);
}

// generic Chan merge of (n, sum, M2) += (nb, sb, mb)
__device__ __forceinline__ void chan_merge(float& n, float& s, float& m2, const float nb, const float sb, const float mb) {
    if (nb <= 0.f) return;
    if (n <= 0.f) { n = nb; s = sb; m2 = mb; return; }
    const float nn = n + nb;
    const float d = sb / nb - s / n;
    m2 = m2 + mb + d * d * (n * nb / nn);
    s += sb;
    n = nn;
}

// step 1, stand-alone: thread t < CL*P: channel chunk cl = t % CL (8 channels), pixel lane pl = t / CL; per channel a pivot-shifted
// (sum, sum of squares) over the lane's pixels -> (n, sum, M2); thread j < C / sub then merges its sub-run's sub x P triples in a
// FIXED order (deterministic).  partial[((b * nblk + blk) * nsub + j) * 2 + {0, 1}]; n = (pixels of the block) * sub.
template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const NormParams p, int nblk, int sub) {
    typedef typename Vec<T>::v8 v8;
    extern __shared__ float lds[];   // [P*C] sums, [P*C] M2, [P] counts
    const int C = p.C, CL = C >> 3;
    const int P = max(1, GN_THREADS / CL);
    float* ls = lds;
    float* lm = lds + P * C;
    float* ln = lds + 2 * P * C;
    const int b = blockIdx.y, blk = blockIdx.x;
    const int ppb = (p.HW + nblk - 1) / nblk;
    const int start = blk * ppb, end = min(p.HW, start + ppb);
    const int t = threadIdx.x;
    if (t < CL * P) {
        const int cl = t % CL, pl = t / CL;
        float s[8], q[8], pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; pv[e] = 0.f; }
        const T* x = (const T*)p.x + ((size_t)b * p.HW) * C + cl * 8;
        int cnt = 0;
        for (int pix = start + pl; pix < end; pix += P) {
            const v8 v = *(const v8*)(x + (size_t)pix * C);
            if (cnt == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) pv[e] = to_f32(v[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = to_f32(v[e]) - pv[e]; s[e] += f; q[e] = __builtin_fmaf(f, f, q[e]); }
            ++cnt;
        }
        const float n = (float)cnt;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ls[pl * C + cl * 8 + e] = __builtin_fmaf(n, pv[e], s[e]);
            lm[pl * C + cl * 8 + e] = cnt ? fmaxf(q[e] - s[e] * s[e] / n, 0.f) : 0.f;
        }
        if (cl == 0) ln[pl] = n;
    }
    __syncthreads();
    const int nsub = C / sub;
    for (int j = t; j < nsub; j += GN_THREADS) {
        float n = 0.f, s = 0.f, m2 = 0.f;
        for (int pl = 0; pl < P; ++pl) {
            const float np = ln[pl];
            for (int c = j * sub; c < (j + 1) * sub; ++c) chan_merge(n, s, m2, np, ls[pl * C + c], lm[pl * C + c]);
        }
        float* o = p.partial + (((size_t)b * nblk + blk) * nsub + j) * 2;
        o[0] = s; o[1] = m2;
    }
}

// step 2 (imh_gntable.h): one QUARTER wave per (sample, group) -- its 16 lanes stride over the group's partials of one or two sources in
// double, a fixed-order butterfly, then the group's (scale, shift) rows.  The same routine runs in the prologue of the consumers that
// build their sample's table themselves (conv_halo.hip with imh_gemm_args.gn_part, gn_apply_kernel below with IMH_GN_TABLE_APPLY): the
// stand-alone launch is left for consumers that share one table (the implicit-GEMM convs, tests).
static inline GnTabSrc gn_src_of(const NormParams& p) {
    GnTabSrc g;
    g.partial = p.partial; g.partial2 = p.partial2; g.gamma = p.gamma; g.beta = p.beta; g.eps = p.eps;
    g.groups = p.groups; g.C = p.C; g.C1 = p.partial2 ? p.C1 : p.C; g.HW = p.HW;
    g.nblk = p.nblk; g.sub = p.sub; g.npart = p.npart; g.nblk2 = p.nblk2; g.sub2 = p.sub2; g.npart2 = p.npart2;
    g.dtype_f16 = p.dtype_f16;
    return g;
}
__global__ __launch_bounds__(256) void gn_table_kernel(const GnTabSrc g, float* table) {
    const int b = blockIdx.y, t = threadIdx.x;
    const int gg = blockIdx.x * (256 / GN_GL) + t / GN_GL;        // sixteen groups per workgroup
    if (gg >= g.groups) return;                                   // (whole quarter waves leave together)
    GnRows r;
    gn_rows_fetch(g, gg, t & (GN_GL - 1), r);
    float mean, rstd;
    gn_group_stats(g, b, gg, t & (GN_GL - 1), mean, rstd);
    gn_group_rows(g, gg, t & (GN_GL - 1), mean, rstd, r, table + (size_t)b * g.C * 2);
}

// step 3, stand-alone: y = silu?(x * scale[c] + shift[c]) with the per-sample table of step 2
// OWN: the table of the workgroup's sample is built here, in LDS, from the producers' partials (IMH_GN_TABLE_APPLY) -- no table launch
template <typename T, bool OWN>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const NormParams p, int nblk, const GnTabSrc g) {
    typedef typename Vec<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) float gn_lds_tab[];      // OWN: [C][2]
    tail_prefetch(p.pf_ptr, p.pf_bytes, blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y, threadIdx.x, GN_THREADS);
    const int C = p.C, CL = C >> 3;
    const int P = max(1, GN_THREADS / CL);
    const int b = blockIdx.y, blk = blockIdx.x;
    const int ppb = (p.HW + nblk - 1) / nblk;
    const int start = blk * ppb, end = min(p.HW, start + ppb);
    const int t = threadIdx.x;
    if constexpr (OWN) {
        gn_table_of_sample(g, b, gn_lds_tab, t / GN_GL, GN_THREADS / GN_GL, t & 63);
        __syncthreads();
    }
    if (t < CL * P) {
        const int cl = t % CL, pl = t / CL;
        float sc[8], sh[8];
        {
            const f32x4* tb = OWN ? (const f32x4*)(gn_lds_tab + (size_t)cl * 8 * 2) : (const f32x4*)(p.table + ((size_t)b * C + cl * 8) * 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const f32x4 v = tb[e]; sc[2 * e] = v[0]; sh[2 * e] = v[1]; sc[2 * e + 1] = v[2]; sh[2 * e + 1] = v[3]; }
        }
        const T* x = (const T*)p.x + ((size_t)b * p.HW) * C + cl * 8;
        T* y = (T*)p.y + ((size_t)b * p.HW) * C + cl * 8;
        auto one = [&](const v8& v, size_t off) {
            v8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = __builtin_fmaf(to_f32(v[e]), sc[e], sh[e]);       // (the same expression as conv_halo.hip's in-kernel apply: bit-equal)
                if (p.silu) f = silu_f(f);
                o[e] = from_f32<T>(f);
            }
            *(v8*)(y + off) = o;
        };
        int pix = start + pl;
        for (; pix + 3 * P < end; pix += 4 * P) {
            const size_t o0 = (size_t)pix * C, o1 = (size_t)(pix + P) * C, o2 = (size_t)(pix + 2 * P) * C, o3 = (size_t)(pix + 3 * P) * C;
            v8 v0 = *(const v8*)(x + o0), v1 = *(const v8*)(x + o1), v2 = *(const v8*)(x + o2), v3 = *(const v8*)(x + o3);
            one(v0, o0); one(v1, o1); one(v2, o2); one(v3, o3);
        }
        for (; pix < end; pix += P) one(*(const v8*)(x + (size_t)pix * C), (size_t)pix * C);
    }
}

// mode: IMH_GN_ALL statistics + table + apply (workspace in p.partial: partials, then the table); IMH_GN_STATS statistics only
// (-> p.partial); IMH_GN_TABLE table from the partials of one or two producers (-> p.table); IMH_GN_APPLY apply p.table;
// IMH_GN_TABLE_APPLY partials -> y in ONE launch (every workgroup builds its sample's table in LDS)
int groupnorm_launch(const NormParams& p0, int dtype, hipStream_t stream) {
    NormParams p = p0;
    if (p.C % 8 || p.groups <= 0 || p.groups > 64 || p.C % p.groups || (p.C >> 3) > GN_THREADS || p.B <= 0 || p.HW <= 0) {
        set_error("groupnorm: unsupported B=%d HW=%d C=%d groups=%d", p.B, p.HW, p.C, p.groups);
        return IMH_ERR_SHAPE;
    }
    if (dtype != IMH_DT_BF16 && dtype != IMH_DT_F16) { set_error("groupnorm: unknown dtype %d", dtype); return IMH_ERR_DTYPE; }
    p.dtype_f16 = dtype == IMH_DT_F16;
    const int nblk = gn_nblk(p.HW, p.C);
    const int mode = p.mode;
    if (mode < 0 || mode > 4) { set_error("groupnorm: unknown mode %d", mode); return IMH_ERR_ARG; }
    const int cpg = p.C / p.groups;
    if (mode == 4) {          // IMH_GN_TABLE_APPLY: partials -> (table in LDS, per workgroup) -> y
        if (!p.partial2) p.C1 = p.C;
        const GnTabSrc g = gn_src_of(p);
        if (!p.x || !p.y || !gn_src_ok(g)) {
            set_error("groupnorm table+apply: needs x, y and partials whose sub-runs tile the groups (C=%d groups=%d C1=%d sub=%d/%d nblk=%d/%d), eps > 0",
                      p.C, p.groups, p.C1, p.sub, p.sub2, p.nblk, p.nblk2);
            return IMH_ERR_ARG;
        }
        const size_t lds = (size_t)p.C * 2 * sizeof(float);
        dim3 grid(nblk, p.B);
        if (dtype == IMH_DT_BF16) hipLaunchKernelGGL((gn_apply_kernel<bf16_t, true>), grid, dim3(GN_THREADS), lds, stream, p, nblk, g);
        else hipLaunchKernelGGL((gn_apply_kernel<f16_t, true>), grid, dim3(GN_THREADS), lds, stream, p, nblk, g);
        return check_launch("groupnorm table+apply");
    }
    if (mode == 0 || mode == 1) {
        if (!p.x || !p.partial) { set_error("groupnorm: statistics need x and a partial / workspace buffer"); return p.x ? IMH_ERR_WORKSPACE : IMH_ERR_ARG; }
        const int sub = mode == 0 ? gn_sub(p.C, p.groups) : p.sub;
        if (sub <= 0 || p.C % sub || (mode == 0 && cpg % sub)) { set_error("groupnorm: sub-run width %d does not divide C=%d", sub, p.C); return IMH_ERR_ARG; }
        const int P = std::max(1, GN_THREADS / (p.C >> 3));
        const size_t lds = (2 * (size_t)p.C * P + P) * sizeof(float);
        dim3 grid(nblk, p.B);
        if (dtype == IMH_DT_BF16) hipLaunchKernelGGL((gn_stats_kernel<bf16_t>), grid, dim3(GN_THREADS), lds, stream, p, nblk, sub);
        else hipLaunchKernelGGL((gn_stats_kernel<f16_t>), grid, dim3(GN_THREADS), lds, stream, p, nblk, sub);
        if (mode == 1) return check_launch("gn_stats_kernel");
        // all-in-one: the table lives behind the partials in the workspace
        const size_t part = (size_t)p.B * nblk * (p.C / sub) * 2 * sizeof(float);
        p.table = (float*)((unsigned char*)p.partial + ((part + 255) & ~(size_t)255));
        p.partial2 = nullptr; p.C1 = p.C; p.sub = sub; p.nblk = nblk; p.npart = 0;
    }
    if (mode == 0 || mode == 2) {
        if (!p.partial || !p.table) { set_error("groupnorm: the table step needs partials and a table buffer"); return IMH_ERR_ARG; }
        if (!p.partial2) p.C1 = p.C;
        const GnTabSrc g = gn_src_of(p);
        if (!gn_src_ok(g)) {
            set_error("groupnorm table: sub-runs must tile the groups (C=%d groups=%d C1=%d sub=%d/%d nblk=%d/%d) and eps > 0", p.C, p.groups, p.C1, p.sub, p.sub2, p.nblk, p.nblk2);
            return IMH_ERR_ARG;
        }
        hipLaunchKernelGGL(gn_table_kernel, dim3((p.groups * GN_GL + 255) / 256, p.B), dim3(256), 0, stream, g, p.table);
        if (mode == 2) return check_launch("gn_table_kernel");
    }
    if (!p.x || !p.y || !p.table) { set_error("groupnorm: apply needs x, y and a table"); return IMH_ERR_ARG; }
    dim3 grid(nblk, p.B);
    if (dtype == IMH_DT_BF16) hipLaunchKernelGGL((gn_apply_kernel<bf16_t, false>), grid, dim3(GN_THREADS), 0, stream, p, nblk, GnTabSrc{});
    else hipLaunchKernelGGL((gn_apply_kernel<f16_t, false>), grid, dim3(GN_THREADS), 0, stream, p, nblk, GnTabSrc{});
    return check_launch("groupnorm");
}

// ---- LayerNorm: one wave handles ROWS rows at once (all loads issued before the first reduction), the rows
//      live in registers (C <= 4096) ----
template <typename T, int NCH, int ROWS>
__global__ __launch_bounds__(256) void ln_kernel(const NormParams p) {
    typedef typename Vec<T>::v8 v8;
    const int lane = threadIdx.x & 63;
    tail_prefetch(p.pf_ptr, p.pf_bytes, blockIdx.x, gridDim.x, threadIdx.x, 256);   // fire-and-forget, overlaps the row loads
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
    if (row0 >= p.rows) return;
    const int C = p.C, CL = C >> 3;
    float v[ROWS][NCH][8];
    float s[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int row = min(row0 + r, p.rows - 1);
        const T* x = (const T*)p.x + (size_t)row * C;
        s[r] = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + k * 64;
            if (ch < CL) {
                v8 t = *(const v8*)(x + ch * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[r][k][e] = to_f32(t[e]); s[r] += v[r][k][e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[r][k][e] = 0.f;
            }
        }
    }
    const T* ga = (const T*)p.gamma;
    const T* be = (const T*)p.beta;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const float mean = wave_sum(s[r]) / C;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + k * 64;
            if (ch < CL) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[r][k][e] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / C + p.eps);
        if (row0 + r >= p.rows) continue;
        T* y = (T*)p.y + (size_t)(row0 + r) * C;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + k * 64;
            if (ch < CL) {
                v8 g8, b8, o;
                if (ga) g8 = *(const v8*)(ga + ch * 8);
                if (be) b8 = *(const v8*)(be + ch * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = (v[r][k][e] - mean) * rstd;
                    if (ga) f *= to_f32(g8[e]);
                    if (be) f += to_f32(b8[e]);
                    o[e] = from_f32<T>(f);
                }
                *(v8*)(y + ch * 8) = o;
            }
        }
    }
}

template <typename T>
static int ln_typed(const NormParams& p, hipStream_t stream) {
    const int cl = p.C >> 3;
    if (cl <= 128) hipLaunchKernelGGL((ln_kernel<T, 2, 1>), dim3((p.rows + 3) / 4), dim3(256), 0, stream, p);
    else if (cl <= 256) hipLaunchKernelGGL((ln_kernel<T, 4, 1>), dim3((p.rows + 3) / 4), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((ln_kernel<T, 8, 1>), dim3((p.rows + 3) / 4), dim3(256), 0, stream, p);
    return check_launch("ln_kernel");
}

int layernorm_launch(const NormParams& p, int dtype, hipStream_t stream) {
    if (p.C % 8 || p.C > 4096 || p.rows <= 0) { set_error("layernorm: unsupported C=%d rows=%d", p.C, p.rows); return IMH_ERR_SHAPE; }
    if (dtype == IMH_DT_BF16) return ln_typed<bf16_t>(p, stream);
    if (dtype == IMH_DT_F16) return ln_typed<f16_t>(p, stream);
    set_error("layernorm: unknown dtype %d", dtype);
    return IMH_ERR_DTYPE;
}

}  // namespace imh
