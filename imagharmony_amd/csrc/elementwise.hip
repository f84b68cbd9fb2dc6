// Small memory-/latency-bound kernels around the UNet forward and the denoise loop, gfx950.
//   EW_TIMESTEP   diffusers Timesteps(dim, flip_sin_to_cos=True, shift 0)      
//   EW_SILU       SiLU on the time embedding before time_emb_proj               (App. A ResnetBlock2D)
//   EW_CONCAT     channel concat of NHWC tensors (up-block skip connections)    (App. A.6)
//   EW_CONV_IN    conv_in 3x3 (4 -> C0) reading NCHW fp32 latents, fusing CFG duplication
//                 (custom_pipelines.py:332) and scale_model_input (:334); writes NHWC
//   EW_CFG_STEP   CFG combine (custom_pipelines.py:348-350) + linear scheduler update
//                 x' = cx*x + ce*eps (DDIM eta=0 / Euler), noise pred read
//                 NHWC, latents kept NCHW fp32 (custom_pipelines.py:357)
//   EW_CFG_RESCALE per-sample factor of rescale_noise_cfg (custom_pipelines.py:351-354; arXiv 2305.08891 3.4):
//                 phi * std(eps_text) / std(eps_cfg) + (1 - phi), consumed by EW_CFG_STEP through `w`
//   EW_SOFTMAX    row softmax of fp32 scores -> T probabilities (the VAE mid-block attention: one head of width
//                 512 over 4096-16384 tokens, materialised as GEMM -> softmax -> GEMM; once per image)
//   EW_ROW_STATS  (sum, M2) of token rows, the stand-alone LayerNorm-statistics producer (imh_lnstats.h)
//   EW_STEP_ROW   y[0:n] = a[*step * n + 0:n]: the denoise step's row of a per-schedule table (the stacked time-embedding projections,
//                 computed once per schedule for all steps instead of five launches per step)
//   EW_CAST_F32   T -> fp32 copy (debug / host-side plumbing)
//   EW_STEP_SET   the device-resident step counter (lets 30 graph replays run with no host updates)
// Per-step scalars (timestep, scheduler coefficients, input scale) may come from device tables
// indexed by *step instead of the immediate f0..f2 fields.
#include <algorithm>

#include "imh_common.h"
#include "imh_kernels.h"

namespace imh {

enum : int { EW_TIMESTEP = 0, EW_SILU = 1, EW_CONCAT = 2, EW_CONV_IN = 3, EW_CFG_STEP = 4, EW_CAST_F32 = 5,
             EW_ADD = 6, EW_STEP_SET = 7, EW_CFG_RESCALE = 8, EW_SOFTMAX = 9, EW_ROW_STATS = 10, EW_STEP_ROW = 11 };

// a: fp32 values [n_vals]; y: T [n_vals, dim]; cos first, then sin.
template <typename T>
__global__ void timestep_kernel(const EwParams p) {
    const int dim = p.i0, half = dim >> 1;
    const long long total = p.n * half;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(i / half), j = (int)(i % half);
        const float t = p.step ? ((const float*)p.a)[*p.step] : ((const float*)p.a)[v];
        const float f = expf(-9.210340371976184f * (float)j / (float)half);   // ln(10000)
        const float a = t * f;
        T* y = (T*)p.y + (size_t)v * dim;
        y[j] = from_f32<T>(cosf(a));
        y[half + j] = from_f32<T>(sinf(a));
    }
}

template <typename T>
__global__ void silu_kernel(const EwParams p) {
    typedef typename Vec<T>::v8 v8;
    const long long nv = p.n >> 3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        v8 v = ((const v8*)p.a)[i], o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(silu_f(to_f32(v[e])));
        ((v8*)p.y)[i] = o;
    }
}

template <typename T>
__global__ void add_kernel(const EwParams p) {
    typedef typename Vec<T>::v8 v8;
    const long long nv = p.n >> 3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        v8 u = ((const v8*)p.a)[i], v = ((const v8*)p.b)[i], o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(to_f32(u[e]) + to_f32(v[e]));
        ((v8*)p.y)[i] = o;
    }
}

// y[pix, 0:C1] = a[pix], y[pix, C1:C1+C2] = b[pix]   (n = pixels, C1 = i0, C2 = i1; multiples of 8)
template <typename T>
__global__ void concat_kernel(const EwParams p) {
    typedef typename Vec<T>::v8 v8;
    const int c1 = p.i0 >> 3, c2 = p.i1 >> 3, ct = c1 + c2;
    const long long total = p.n * ct;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long pix = i / ct;
        const int c = (int)(i - pix * ct);
        v8 v = c < c1 ? ((const v8*)p.a)[pix * c1 + c] : ((const v8*)p.b)[pix * c2 + (c - c1)];
        ((v8*)p.y)[i] = v;
    }
}

// conv_in: a = latents fp32 NCHW [S, 4, H, W]; w = weights T [C0][4][3][3]; bias T [C0];
// y = NHWC T [Bout, H, W, C0] with Bout = i4 (batch b reads latent b % S: CFG duplication).
// i0 = S, i1 = H, i2 = W, i3 = C0, f0 = input scale.  Eight threads per pixel, each 8 channels at a time.
template <typename T>
__global__ __launch_bounds__(256) void conv_in_kernel(const EwParams p) {
    typedef typename Vec<T>::v8 v8;
    extern __shared__ float wl[];   // [36][C0] transposed weights, then [C0] bias
    const int S = p.i0, H = p.i1, W = p.i2, C0 = p.i3, Bout = p.i4;
    const T* w = (const T*)p.w;
    for (int co = threadIdx.x; co < C0; co += blockDim.x) {      // one output channel (36 contiguous taps) per thread
#pragma unroll
        for (int k = 0; k < 36; ++k) wl[k * C0 + co] = to_f32(w[co * 36 + k]);
    }
    for (int i = threadIdx.x; i < C0; i += blockDim.x) wl[36 * C0 + i] = p.bias ? to_f32(((const T*)p.bias)[i]) : 0.f;
    __syncthreads();
    const int cg = C0 >> 3;
    const float in_scale = p.tab ? p.tab[*p.step] : p.f0;
    // 8 threads per pixel: each fetches the pixel's 36 taps once (all loads in flight together; the 8 copies hit
    // L1), then produces the channel octets l8, l8+8, .. -- the 8 threads of a pixel store 128 contiguous bytes
    const long long total = (long long)Bout * H * W * 8;
    const float* lat = (const float*)p.a;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int l8 = (int)(i & 7);
        const long long pix = i >> 3;
        const int x = (int)(pix % W);
        const int y = (int)((pix / W) % H);
        const int b = (int)(pix / ((long long)W * H));
        const float* src = lat + (size_t)(b % S) * 4 * H * W;
        float tap[36];
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int iy = y + ky - 1, ix = x + kx - 1;
                    const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
                    const float v = in ? src[((size_t)ci * H + iy) * W + ix] * in_scale : 0.f;
                    // the model sees the latent rounded to the compute dtype (pipeline casts latents)
                    tap[ci * 9 + ky * 3 + kx] = to_f32(from_f32<T>(v));
                }
        for (int c8 = l8; c8 < cg; c8 += 8) {
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = wl[36 * C0 + c8 * 8 + e];
#pragma unroll
            for (int k = 0; k < 36; ++k) {
                const float* wr = wl + k * C0 + c8 * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += tap[k] * wr[e];
            }
            v8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(acc[e]);
            ((v8*)p.y)[pix * cg + c8] = o;
        }
    }
}

template <typename T>
__global__ void cfg_step_kernel(const EwParams p) {
    const int S = p.i0, HW = p.i1;
    const long long total = (long long)S * HW * 4;
    const T* np_ = (const T*)p.a;
    float* lat = (float*)p.y;
    const float cx = p.tab ? p.tab[*p.step * 2] : p.f0;
    const float ce = p.tab ? p.tab[*p.step * 2 + 1] : p.f1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int pix = (int)(i % HW);
        const int ch = (int)((i / HW) % 4);
        const int s = (int)(i / ((long long)HW * 4));
        float eps;
        if (p.i3) {
            const float u = to_f32(np_[((size_t)s * HW + pix) * 4 + ch]);
            const float c = to_f32(np_[((size_t)(S + s) * HW + pix) * 4 + ch]);
            eps = u + p.f2 * (c - u);
        } else {
            eps = to_f32(np_[((size_t)s * HW + pix) * 4 + ch]);
        }
        if (p.w) eps *= ((const float*)p.w)[s];        // guidance_rescale factor of this sample (EW_CFG_RESCALE)
        if (p.b) ((float*)p.b)[i] = eps;
        lat[i] = cx * lat[i] + ce * eps;
    }
}

// One workgroup per sample: unbiased std over (C, H, W) of the text-conditioned prediction and of the CFG
// combination (torch.std semantics, as diffusers' rescale_noise_cfg), double accumulation, fixed-order tree.
// a = noise prediction NHWC T [2S, HW, 4] ([uncond | cond]); y = fp32 [S]; f2 = guidance scale, f3 = guidance_rescale.
template <typename T>
__global__ __launch_bounds__(256) void cfg_rescale_kernel(const EwParams p) {
    __shared__ double red[4][256];
    const int S = p.i0, HW = p.i1, s = blockIdx.x;
    const long long n = (long long)HW * 4;
    const T* u = (const T*)p.a + (size_t)s * n;
    const T* c = (const T*)p.a + (size_t)(S + s) * n;
    double st = 0, st2 = 0, sc = 0, sc2 = 0;
    for (long long i = threadIdx.x; i < n; i += 256) {
        const float uu = to_f32(u[i]), cc = to_f32(c[i]);
        const float g = uu + p.f2 * (cc - uu);
        st += cc; st2 += (double)cc * cc; sc += g; sc2 += (double)g * g;
    }
    red[0][threadIdx.x] = st; red[1][threadIdx.x] = st2; red[2][threadIdx.x] = sc; red[3][threadIdx.x] = sc2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
#pragma unroll
            for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double N = (double)n;
        const double vt = (red[1][0] - red[0][0] * red[0][0] / N) / (N - 1.0);
        const double vc = (red[3][0] - red[2][0] * red[2][0] / N) / (N - 1.0);
        const double ratio = sqrt(fmax(vt, 0.0)) / sqrt(fmax(vc, 1e-300));
        ((float*)p.y)[s] = (float)(p.f3 * ratio + (1.0 - p.f3));
    }
}

// y[r, :] = softmax(f0 * a[r, :]) ; a fp32 [rows, ld_in], y T [rows, ld_out]; one workgroup per row, three passes
// over a row that stays in L2 (<= 64 KB).  i0 rows, i1 cols (multiple of 4), i2 ld_in, i3 ld_out.
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const EwParams p) {
    __shared__ float red[4];
    const int cols = p.i1, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* a = (const float*)p.a + (size_t)blockIdx.x * p.i2;
    T* y = (T*)p.y + (size_t)blockIdx.x * p.i3;
    const float c = p.f0 * 1.4426950408889634f;
    float m = -3.0e38f;
    for (int i = threadIdx.x * 4; i < cols; i += 1024) {
        const f32x4 v = *(const f32x4*)(a + i);
        m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
    }
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    const float mc = m * c;       // f0 > 0: the maximum of the scaled row is the scaled maximum
    float s = 0.f;
    for (int i = threadIdx.x * 4; i < cols; i += 1024) {
        const f32x4 v = *(const f32x4*)(a + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) s += __builtin_amdgcn_exp2f(v[e] * c - mc);
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
    for (int i = threadIdx.x * 4; i < cols; i += 1024) {
        const f32x4 v = *(const f32x4*)(a + i);
        typename Vec<T>::v4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(__builtin_amdgcn_exp2f(v[e] * c - mc) * inv);
        *(typename Vec<T>::v4*)(y + i) = o;
    }
}

template <typename T>
__global__ void step_row_kernel(const EwParams p) {
    typedef typename Vec<T>::v8 v8;
    const long long nv = p.n >> 3;
    const v8* src = (const v8*)p.a + (long long)(*p.step) * nv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) ((v8*)p.y)[i] = src[i];
}

__global__ void step_set_kernel(int* step, int value, int set) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *step = set ? value : *step + 1;
}

template <typename T>
__global__ void cast_f32_kernel(const EwParams p) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (long long)gridDim.x * blockDim.x)
        ((float*)p.y)[i] = to_f32(((const T*)p.a)[i]);
}

static inline int grid_for(long long work, int threads) {
    long long g = (work + threads - 1) / threads;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// EW_ROW_STATS: y[row] = (sum, M2 about the row mean) of a[row, 0:i0] (row stride i1 elements), ONE slot per row, in the
// format of imh_lnstats.h -- the stand-alone producer of LayerNorm statistics for token rows whose writing GEMM variant
// has no statistics epilogue (odd shapes / tiles).  One wave per row, two passes (the second hits L1).
template <typename T>
__global__ __launch_bounds__(256) void row_stats_kernel(const EwParams p) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.n) return;
    const T* x = (const T*)p.a + row * p.i1;
    const int C = p.i0;
    float s = 0.f;
    for (int c = lane * 8; c < C; c += 512) {
        const typename Vec<T>::v8 t = *(const typename Vec<T>::v8*)(x + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += to_f32(t[e]);
    }
    s = wave_sum(s);
    const float mean = s / (float)C;
    float m2 = 0.f;
    for (int c = lane * 8; c < C; c += 512) {
        const typename Vec<T>::v8 t = *(const typename Vec<T>::v8*)(x + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = to_f32(t[e]) - mean; m2 = __builtin_fmaf(d, d, m2); }
    }
    m2 = wave_sum(m2);
    if (lane == 0) { float* y = (float*)p.y + row * 2; y[0] = s; y[1] = m2; }
}

template <typename T>
static int ew_typed(int op, const EwParams& p, hipStream_t stream) {
    switch (op) {
        case EW_TIMESTEP:
            if (p.i0 <= 0 || (p.i0 & 1)) { set_error("timestep: dim must be even"); return IMH_ERR_SHAPE; }
            hipLaunchKernelGGL((timestep_kernel<T>), dim3(grid_for(p.n * (p.i0 >> 1), 256)), dim3(256), 0, stream, p);
            break;
        case EW_SILU:
            if (p.n & 7) { set_error("silu: n must be a multiple of 8"); return IMH_ERR_SHAPE; }
            hipLaunchKernelGGL((silu_kernel<T>), dim3(grid_for(p.n >> 3, 256)), dim3(256), 0, stream, p);
            break;
        case EW_ADD:
            if (p.n & 7) { set_error("add: n must be a multiple of 8"); return IMH_ERR_SHAPE; }
            hipLaunchKernelGGL((add_kernel<T>), dim3(grid_for(p.n >> 3, 256)), dim3(256), 0, stream, p);
            break;
        case EW_CONCAT:
            if ((p.i0 & 7) || (p.i1 & 7)) { set_error("concat: channel counts must be multiples of 8"); return IMH_ERR_SHAPE; }
            hipLaunchKernelGGL((concat_kernel<T>), dim3(grid_for(p.n * ((p.i0 + p.i1) >> 3), 256)), dim3(256), 0, stream, p);
            break;
        case EW_CONV_IN: {
            if (p.i3 & 7) { set_error("conv_in: C0 must be a multiple of 8"); return IMH_ERR_SHAPE; }
            const size_t lds = (size_t)(37 * p.i3) * sizeof(float);
            const long long work = (long long)p.i4 * p.i1 * p.i2 * 8;
            // persistent workgroups (3 per CU by LDS): the 46 KB weight transpose is paid 768 times, not once per 1024 outputs
            hipLaunchKernelGGL((conv_in_kernel<T>), dim3(std::min(grid_for(work, 256), 768)), dim3(256), lds, stream, p);
            break;
        }
        case EW_CFG_STEP:
            hipLaunchKernelGGL((cfg_step_kernel<T>), dim3(grid_for((long long)p.i0 * p.i1 * 4, 256)), dim3(256), 0, stream, p);
            break;
        case EW_CFG_RESCALE:
            if (p.i0 <= 0 || p.i1 <= 0) { set_error("cfg_rescale: empty problem"); return IMH_ERR_SHAPE; }
            hipLaunchKernelGGL((cfg_rescale_kernel<T>), dim3(p.i0), dim3(256), 0, stream, p);
            break;
        case EW_SOFTMAX:
            if (p.i0 <= 0 || p.i1 <= 0 || (p.i1 & 3) || (p.i2 & 3) || (p.i3 & 3) || p.f0 <= 0.f) {
                set_error("softmax: rows=%d cols=%d ld=%d/%d (cols and lds must be multiples of 4, scale > 0)", p.i0, p.i1, p.i2, p.i3);
                return IMH_ERR_SHAPE;
            }
            hipLaunchKernelGGL((softmax_rows_kernel<T>), dim3(p.i0), dim3(256), 0, stream, p);
            break;
        case EW_CAST_F32:
            hipLaunchKernelGGL((cast_f32_kernel<T>), dim3(grid_for(p.n, 256)), dim3(256), 0, stream, p);
            break;
        case EW_ROW_STATS:
            if (p.n <= 0 || p.i0 <= 0 || (p.i0 & 7) || (p.i1 & 7) || !p.a) { set_error("row_stats: rows=%lld C=%d ld=%d (C and ld must be multiples of 8)", p.n, p.i0, p.i1); return IMH_ERR_SHAPE; }
            hipLaunchKernelGGL((row_stats_kernel<T>), dim3((unsigned)((p.n + 3) / 4)), dim3(256), 0, stream, p);
            break;
        case EW_STEP_ROW:
            if (p.n <= 0 || (p.n & 7) || !p.a || !p.step) { set_error("step_row: n=%lld must be a positive multiple of 8, a and step non-null", p.n); return IMH_ERR_ARG; }
            hipLaunchKernelGGL((step_row_kernel<T>), dim3(grid_for(p.n >> 3, 256)), dim3(256), 0, stream, p);
            break;
        case EW_STEP_SET:
            hipLaunchKernelGGL(step_set_kernel, dim3(1), dim3(64), 0, stream, (int*)p.y, p.i0, p.i1);
            break;
        default:
            set_error("elementwise: unknown op %d", op);
            return IMH_ERR_ARG;
    }
    return check_launch("elementwise");
}

int ew_launch(int op, const EwParams& p, int dtype, hipStream_t stream) {
    if (dtype == IMH_DT_BF16) return ew_typed<bf16_t>(op, p, stream);
    if (dtype == IMH_DT_F16) return ew_typed<f16_t>(op, p, stream);
    set_error("elementwise: unknown dtype %d", dtype);
    return IMH_ERR_DTYPE;
}

}  // namespace imh
