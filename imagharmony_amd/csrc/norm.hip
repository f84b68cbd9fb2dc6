// GroupNorm(32)+SiLU over NHWC activations and LayerNorm over token rows, gfx950.
// HBM-bound kernels: 16-B vector loads/stores, fp32 statistics (PyTorch semantics:
// GroupNorm / LayerNorm accumulate in fp32 whatever the activation dtype).
// Call sites being replaced: diffusers ResnetBlock2D.norm1/norm2 (+SiLU), Transformer2DModel.norm
// (eps 1e-6, no SiLU), conv_norm_out, BasicTransformerBlock.norm1/2/3;
// ImageProjModel.norm (ip_adapter/ip_adapter.py:39,47), Resampler norms (resampler.py:15,42-43,104),
// HarmonyAttention.ln (train.py:238).
#include "imh_common.h"
#include "imh_kernels.h"
#include <algorithm>

namespace imh {

constexpr int GN_THREADS = 512;
constexpr int GN_ELEMS_PER_BLOCK = 8192;     // 32768 left the 32 x 32 levels (1.3 M elements per sample) on 80 workgroups of 256 CUs

static inline int gn_nblk(int HW, int C) {
    long long e = (long long)HW * C;
    int n = (int)((e + GN_ELEMS_PER_BLOCK - 1) / GN_ELEMS_PER_BLOCK);
    return n < 1 ? 1 : (n > 256 ? 256 : n);
}

size_t groupnorm_workspace_bytes(int B, int HW, int C, int groups) {
    return (size_t)B * gn_nblk(HW, C) * groups * 2 * sizeof(float);
}

// pass 1: per-(batch, block, group) partial sum / sum of squares.
// thread t < CL*P: channel chunk cl = t % CL (8 channels), pixel lane pl = t / CL.
template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const NormParams p, int nblk) {
    typedef typename Vec<T>::v8 v8;
    extern __shared__ float lds[];   // [2][P*C]: per pixel-lane channel sums, reduced in a FIXED order (deterministic)
    const int C = p.C, CL = C >> 3;
    const int P = max(1, GN_THREADS / CL);
    const int b = blockIdx.y, blk = blockIdx.x;
    const int ppb = (p.HW + nblk - 1) / nblk;
    const int start = blk * ppb, end = min(p.HW, start + ppb);
    const int t = threadIdx.x;
    if (t < CL * P) {
        const int cl = t % CL, pl = t / CL;
        float s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
        const T* x = (const T*)p.x + ((size_t)b * p.HW) * C + cl * 8;
        int pix = start + pl;
        for (; pix + 3 * P < end; pix += 4 * P) {
            v8 v0 = *(const v8*)(x + (size_t)pix * C);
            v8 v1 = *(const v8*)(x + (size_t)(pix + P) * C);
            v8 v2 = *(const v8*)(x + (size_t)(pix + 2 * P) * C);
            v8 v3 = *(const v8*)(x + (size_t)(pix + 3 * P) * C);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f0 = to_f32(v0[e]), f1 = to_f32(v1[e]), f2 = to_f32(v2[e]), f3 = to_f32(v3[e]);
                s[e] += (f0 + f1) + (f2 + f3);
                q[e] += (f0 * f0 + f1 * f1) + (f2 * f2 + f3 * f3);
            }
        }
        for (; pix < end; pix += P) {
            v8 v = *(const v8*)(x + (size_t)pix * C);
#pragma unroll
            for (int e = 0; e < 8; ++e) { float f = to_f32(v[e]); s[e] += f; q[e] += f * f; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            lds[pl * C + cl * 8 + e] = s[e];
            lds[P * C + pl * C + cl * 8 + e] = q[e];
        }
    }
    __syncthreads();
    if (t < p.groups) {
        const int cpg = C / p.groups;
        float s = 0.f, q = 0.f;
        for (int pl = 0; pl < P; ++pl)
            for (int c = t * cpg; c < (t + 1) * cpg; ++c) { s += lds[pl * C + c]; q += lds[P * C + pl * C + c]; }
        float* o = p.partial + (((size_t)b * nblk + blk) * p.groups + t) * 2;
        o[0] = s; o[1] = q;
    }
}

// pass 2: finalise statistics (double), then y = silu?(x * scale[c] + shift[c]).
template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const NormParams p, int nblk, int nstat) {
    typedef typename Vec<T>::v8 v8;
    __shared__ float mean_s[64], rstd_s[64];
    __shared__ double part_s[16][64], part_q[16][64];
    tail_prefetch(p.pf_ptr, p.pf_bytes, blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y, threadIdx.x, GN_THREADS);
    const int C = p.C, CL = C >> 3;
    const int P = max(1, GN_THREADS / CL);
    const int b = blockIdx.y, blk = blockIdx.x;
    const int ppb = (p.HW + nblk - 1) / nblk;
    const int start = blk * ppb, end = min(p.HW, start + ppb);
    const int t = threadIdx.x;
    {   // 16 slices x groups threads sum the per-block partials, then a fixed-order combine (deterministic)
        const int g = t % 32, sl = t / 32;           // GN_THREADS = 512 -> 16 slices of 32 lanes
        for (int gg = g; gg < p.groups; gg += 32) {
            // nstat partial blocks per sample: pass 1's, or the producing conv / GEMM's (up to 512 of them: eight independent
            // 8-B loads in flight per thread -- one dependent load per step left this prologue at ~nstat / 16 L2 latencies)
            typedef __attribute__((ext_vector_type(2))) float f2;
            const f2* pp = (const f2*)p.partial + ((size_t)b * nstat) * p.groups + gg;
            double s = 0.0, q = 0.0;
            int k = sl;
            for (; k + 7 * 16 < nstat; k += 8 * 16) {
                f2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = pp[(size_t)(k + u * 16) * p.groups];
#pragma unroll
                for (int u = 0; u < 8; ++u) { s += v[u][0]; q += v[u][1]; }
            }
            for (; k < nstat; k += 16) { const f2 v = pp[(size_t)k * p.groups]; s += v[0]; q += v[1]; }
            part_s[sl][gg] = s; part_q[sl][gg] = q;
        }
    }
    __syncthreads();
    if (t < p.groups) {
        double s = 0.0, q = 0.0;
        for (int sl = 0; sl < 16; ++sl) { s += part_s[sl][t]; q += part_q[sl][t]; }
        const double n = (double)p.HW * (C / p.groups);
        const double mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_s[t] = (float)mean;
        rstd_s[t] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
    __syncthreads();
    if (t < CL * P) {
        const int cl = t % CL, pl = t / CL;
        const int cpg = C / p.groups;
        float sc[8], sh[8];
        const T* ga = (const T*)p.gamma;
        const T* be = (const T*)p.beta;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cl * 8 + e;
            const int g = c / cpg;
            const float gm = ga ? to_f32(ga[c]) : 1.f;
            const float bt = be ? to_f32(be[c]) : 0.f;
            sc[e] = gm * rstd_s[g];
            sh[e] = bt - mean_s[g] * sc[e];
        }
        const T* x = (const T*)p.x + ((size_t)b * p.HW) * C + cl * 8;
        T* y = (T*)p.y + ((size_t)b * p.HW) * C + cl * 8;
        auto one = [&](const v8& v, size_t off) {
            v8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = to_f32(v[e]) * sc[e] + sh[e];
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

int groupnorm_launch(const NormParams& p, int dtype, hipStream_t stream) {
    if (p.C % 8 || p.groups <= 0 || p.groups > 64 || p.C % p.groups || (p.C >> 3) > GN_THREADS) {
        set_error("groupnorm: unsupported C=%d groups=%d", p.C, p.groups);
        return IMH_ERR_SHAPE;
    }
    if (!p.partial) { set_error("groupnorm: workspace missing"); return IMH_ERR_WORKSPACE; }
    const int nblk = gn_nblk(p.HW, p.C);
    // stats_blocks > 0: `partial` already holds the (sum, sum of squares) of stats_blocks pixel blocks per sample, left behind by the
    // epilogue of the GEMM / conv that wrote x (imh_gemm_args.gn_out): pass 1 is skipped
    const int nstat = p.stats_blocks > 0 ? p.stats_blocks : nblk;
    dim3 grid(nblk, p.B);
    const size_t lds = 2 * (size_t)p.C * std::max(1, GN_THREADS / (p.C >> 3)) * sizeof(float);
    if (dtype == IMH_DT_BF16) {
        if (p.stats_blocks <= 0) hipLaunchKernelGGL((gn_stats_kernel<bf16_t>), grid, dim3(GN_THREADS), lds, stream, p, nblk);
        hipLaunchKernelGGL((gn_apply_kernel<bf16_t>), grid, dim3(GN_THREADS), 0, stream, p, nblk, nstat);
    } else if (dtype == IMH_DT_F16) {
        if (p.stats_blocks <= 0) hipLaunchKernelGGL((gn_stats_kernel<f16_t>), grid, dim3(GN_THREADS), lds, stream, p, nblk);
        hipLaunchKernelGGL((gn_apply_kernel<f16_t>), grid, dim3(GN_THREADS), 0, stream, p, nblk, nstat);
    } else { set_error("groupnorm: unknown dtype %d", dtype); return IMH_ERR_DTYPE; }
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
