// Reference-precision (fp32) kernels for the VAE decode tail of the path (ip_adapter/custom_pipelines.py:365-377): the reference
// upcasts the SDXL VAE to fp32 before decoding because it overflows in fp16 (:366-372, `needs_upcasting` / `upcast_vae`), so a decode
// that is to be "like the reference" keeps fp32 activations, fp32 weights and fp32 arithmetic.  Everything here is fp32 in, fp32 out:
//
//   f32_gemm_kernel      Y[M, N] = X[M, K] W[N, K]^T (+ bias[n]) (+ residual[m, n]); conv == 1: 3x3 convolution, padding 1, stride 1,
//                        optional fused nearest x2 upsampling of the input, as an implicit GEMM over NHWC input (K = 9 Cin, weights
//                        packed [Cout][ky][kx][Cin] like the 16-bit kernels').  v_mfma_f32_32x32x2_f32: exact fp32 products, one rounding
//                        per accumulate (bitwise an fmaf chain, cdna_hip_programming.md 3), 64 FLOP / clk / SIMD = 1/16 of the bf16 rate --
//                        the decoder's 10.4 TFLOP take ~0.1 s instead of 18 ms in bf16; it runs once per image.
//                        128 x 128 tile, 4 waves x (64 x 64 = 2 x 2 MFMA blocks), K tile 16, register-staged double-buffered LDS
//                        (k-major tiles: fragment reads are lane-consecutive floats, conflict-free).  MFMA-bound by construction: per K tile a
//                        wave issues 32 MFMAs (2048 cycles) for 16 KB of operands.
//   f32_gn_*             GroupNorm (+ SiLU) in three launches: per (sample, pixel block, group) shifted (sum, M2) -> per (sample, channel)
//                        (scale, shift) merged in double in a fixed order (Chan) -> y = silu(x scale + shift).  No E[x^2] - mean^2.
//   f32_softmax_kernel   row softmax of scale * a (the mid-block's single-head attention, materialised scores).
// Roofline: f32_gemm is MFMA-bound at the fp32 matrix rate (157 TFLOP/s peak); the others are HBM-bound, once per image.
#include "imh_common.h"
#include "imh_kernels.h"

namespace imh {

constexpr int F_BM = 128, F_BN = 128, F_BK = 16, F_LD = F_BM + 4;      // +4 floats: the transposing ds_write_b32 of a loader quad hit four banks

__global__ __launch_bounds__(256) void f32_gemm_kernel(const F32Params p) {
    __shared__ float Xs[2][F_BK][F_LD];
    __shared__ float Ws[2][F_BK][F_LD];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * F_BM, n0 = blockIdx.y * F_BN;
    // loader: thread t fetches 4 consecutive k (16 B) of tile row (t >> 2) + 64 h, h = 0, 1, for both operands
    const int lr = tid >> 2, kq = (tid & 3) * 4;
    const float* xsrc[2];
    const float* wsrc[2];
    int cb[2], coy[2], cox[2];
    bool xok[2], wok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int m = m0 + lr + 64 * h, n = n0 + lr + 64 * h;
        xok[h] = m < p.M;
        wok[h] = n < p.N;
        wsrc[h] = p.W + (size_t)min(n, p.N - 1) * p.ldw + kq;
        if (p.conv) {
            const int hw = p.Ho * p.Wo;
            const int mm = min(m, p.M - 1);
            cb[h] = mm / hw;
            const int rem = mm - cb[h] * hw;
            coy[h] = rem / p.Wo;
            cox[h] = rem - coy[h] * p.Wo;
            xsrc[h] = p.X;
        } else {
            xsrc[h] = p.X + (size_t)min(m, p.M - 1) * p.ldx + kq;
            cb[h] = coy[h] = cox[h] = 0;
        }
    }
    const int Hv = p.H << p.up, Wv = p.Wd << p.up;
    auto load = [&](int kt, f32x4 (&xv)[2], f32x4 (&wv)[2]) {
        const int k0 = kt * F_BK;
        int ky = 0, kx = 0, c0 = 0;
        if (p.conv) {                              // a K tile lies inside one tap (Cin % 16 == 0)
            const int tap = k0 / p.Cin;
            c0 = k0 - tap * p.Cin;
            ky = tap / 3; kx = tap - ky * 3;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (p.conv) {
                const int iy = coy[h] + ky - 1, ix = cox[h] + kx - 1;
                if (xok[h] && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv)
                    v = *(const f32x4*)(p.X + (((size_t)cb[h] * p.H + (iy >> p.up)) * p.Wd + (ix >> p.up)) * p.Cin + c0 + kq);
            } else if (xok[h]) {
                v = *(const f32x4*)(xsrc[h] + k0);
            }
            xv[h] = v;
            wv[h] = wok[h] ? *(const f32x4*)(wsrc[h] + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store = [&](int buf, const f32x4 (&xv)[2], const f32x4 (&wv)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Xs[buf][kq + e][lr + 64 * h] = xv[h][e];
                Ws[buf][kq + e][lr + 64 * h] = wv[h][e];
            }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nkt = p.K / F_BK;
    f32x4 xv[2], wv[2];
    load(0, xv, wv);
    store(0, xv, wv);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load(kt + 1, xv, wv);         // in flight under this tile's 32 MFMAs
#pragma unroll
        for (int kk = 0; kk < F_BK / 2; ++kk) {
            float a[2], bq[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = Xs[buf][2 * kk + hi][wm * 64 + i * 32 + l31];      // A[i = lane & 31][k = lane >> 5]
#pragma unroll
            for (int j = 0; j < 2; ++j) bq[j] = Ws[buf][2 * kk + hi][wn * 64 + j * 32 + l31];     // B[k = lane >> 5][j = lane & 31]
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bq[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nkt) store(buf ^ 1, xv, wv);       // the other buffer: every wave finished reading it before the previous barrier
        __syncthreads();
    }
    // D lane l, register r: row (r & 3) + 8 (r >> 2) + 4 (l >> 5) = pixel, column l & 31 = output channel: 32 lanes store 128 contiguous bytes
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + l31;
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m >= p.M) continue;
                float y = acc[i][j][r] + bv;
                if (p.residual) y += p.residual[(size_t)m * p.ldr + n];
                p.Y[(size_t)m * p.ldy + n] = y;
            }
    }
}

// ---- the same GEMM / conv3x3 with the fp32 operands split into bf16 hi + lo and THREE bf16 MFMAs per product block (round 6) ----
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits; x w ~= hi_x hi_w + hi_x lo_w + lo_x hi_w (the dropped lo lo term is
// 2^-16 of the product), accumulated in fp32 by v_mfma_f32_32x32x16_bf16 -- 3 x 32 cycles per 32 x 32 x 16 block where the exact fp32 MFMA
// takes 8 x 64: the matrix-pipe time of the VAE decoder drops 5.3x and the kernel becomes bound by its loads and the hi / lo conversion.
// Relative error of a product ~2e-5 (random sign), of the decoded image ~1e-5 against the oracle VAE (bound 1e-4; the reference's own fp32
// conv runs in TF32 -- 10 mantissa bits -- on the hardware it was written for).  g_f32_exact (imh_debug_set key 10) = 1 selects the exact
// kernel above for every launch; launches whose K tile does not fit (K % 32, conv Cin % 32) take it anyway.
int g_f32_exact = 0;
constexpr int X3_BK = 32, X3_LD = 40;             // bf16 elements per LDS row: 32 k + 8 of padding (80-B rows: the 32 lanes of a fragment read spread over the banks)

__global__ __launch_bounds__(256, 3) void f32_gemm_x3_kernel(const F32Params p) {
    typedef typename Vec<bf16_t>::v8 v8;
    typedef typename Vec<bf16_t>::v4 v4;
    __shared__ __attribute__((aligned(16))) bf16_t Xh[F_BM][X3_LD], Xl[F_BM][X3_LD], Wh[F_BN][X3_LD], Wl[F_BN][X3_LD];      // 4 x 10 KB
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * F_BM, n0 = blockIdx.y * F_BN;
    // loader: thread t fetches 16 consecutive k (64 B) of tile row t >> 1 for both operands
    const int lr = tid >> 1, kq = (tid & 1) * 16;
    const int m = m0 + lr, n = n0 + lr;
    const bool xok = m < p.M, wok = n < p.N;
    const float* wsrc = p.W + (size_t)min(n, p.N - 1) * p.ldw + kq;
    const float* xsrc = p.X;
    int cb = 0, coy = 0, cox = 0;
    if (p.conv) {
        const int hw = p.Ho * p.Wo;
        const int mm = min(m, p.M - 1);
        cb = mm / hw;
        const int rem = mm - cb * hw;
        coy = rem / p.Wo;
        cox = rem - coy * p.Wo;
    } else {
        xsrc = p.X + (size_t)min(m, p.M - 1) * p.ldx + kq;
    }
    const int Hv = p.H << p.up, Wv = p.Wd << p.up;
    auto load_x = [&](int kt, f32x4 (&xv)[4]) {
        const int k0 = kt * X3_BK;
        const float* xs = nullptr;
        if (p.conv) {                              // a K tile lies inside one tap (Cin % 32 == 0)
            const int tap = k0 / p.Cin;
            const int c0 = k0 - tap * p.Cin;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int iy = coy + ky - 1, ix = cox + kx - 1;
            if (xok && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv)
                xs = p.X + (((size_t)cb * p.H + (iy >> p.up)) * p.Wd + (ix >> p.up)) * p.Cin + c0 + kq;
        } else if (xok) {
            xs = xsrc + k0;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[e] = xs ? *(const f32x4*)(xs + 4 * e) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto load_w = [&](int kt, f32x4 (&wv)[4]) {
        const int k0 = kt * X3_BK;
#pragma unroll
        for (int e = 0; e < 4; ++e) wv[e] = wok ? *(const f32x4*)(wsrc + k0 + 4 * e) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto split_store = [&](const f32x4 (&v)[4], bf16_t (*H)[X3_LD], bf16_t (*Lo)[X3_LD]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v4 h, l;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bf16_t hq = from_f32<bf16_t>(v[e][q]);
                h[q] = hq;
                l[q] = from_f32<bf16_t>(v[e][q] - to_f32(hq));
            }
            *(v4*)&H[lr][kq + 4 * e] = h;
            *(v4*)&Lo[lr][kq + 4 * e] = l;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nkt = p.K / X3_BK;
    f32x4 xv[4], wv[4];
    load_x(0, xv); load_w(0, wv);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();                                  // every wave is past its reads of tile kt - 1
        split_store(xv, Xh, Xl);
        if (kt + 1 < nkt) load_x(kt + 1, xv);             // each operand's next tile is requested as soon as its registers are free: in flight
        split_store(wv, Wh, Wl);                          // under the other operand's conversion, the barrier and this tile's 24 MFMAs
        if (kt + 1 < nkt) load_w(kt + 1, wv);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < X3_BK / 16; ++kk) {
            v8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {                 // A: row lane & 31, eight k from 8 (lane >> 5)
                ah[i] = *(const v8*)&Xh[wm * 64 + i * 32 + l31][kk * 16 + 8 * hi];
                al[i] = *(const v8*)&Xl[wm * 64 + i * 32 + l31][kk * 16 + 8 * hi];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = *(const v8*)&Wh[wn * 64 + j * 32 + l31][kk * 16 + 8 * hi];
                bl[j] = *(const v8*)&Wl[wn * 64 + j * 32 + l31][kk * 16 + 8 * hi];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {             // the small terms first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int nn = n0 + wn * 64 + j * 32 + l31;
        if (nn >= p.N) continue;
        const float bv = p.bias ? p.bias[nn] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (mm >= p.M) continue;
                float y = acc[i][j][r] + bv;
                if (p.residual) y += p.residual[(size_t)mm * p.ldr + nn];
                p.Y[(size_t)mm * p.ldy + nn] = y;
            }
    }
}

// ---- GroupNorm, fp32.  x [B, HW, C]; partial [B, nblk, G, 2] = (mean, M2) of the block's pixels x the group's channels ----
__global__ __launch_bounds__(256) void f32_gn_stats_kernel(const F32Params p) {
    __shared__ float cm[512], cq[512];                   // per channel: mean and M2 over the block's pixels
    __shared__ __attribute__((aligned(16))) float ls[2048], lq[2048];      // [pixel lane][channel] shifted sums (P * C <= 2048)
    const int blk = blockIdx.x, b = blockIdx.y;
    const int ppb = (p.HW + p.nblk - 1) / p.nblk;
    const int p0 = blk * ppb, p1 = min(p.HW, p0 + ppb);
    const int np = p1 - p0;
    const float* x = p.X + ((size_t)b * p.HW + p0) * p.C;
    // round 6: thread (pixel lane pl, channel quad cq) streams pixels pl, pl + P, ... with 16-B loads, four in flight (the round's first
    // form walked the block's pixels one 4-B load at a time per thread: 0.3 TB/s, 39 % of the fp32 decode).  Every lane shifts by the SAME
    // pivot -- the block's first pixel -- so the lanes' (sum, sum of squares) simply add, in a fixed order.
    const int CQ = p.C >> 2;                             // channel quads (C % 4 == 0, C <= 512: CQ <= 128)
    const int P = 256 / CQ;                              // pixel lanes: 8 / 4 / 2 for C = 128 / 256 / 512
    const int cqi = threadIdx.x % CQ, pl = threadIdx.x / CQ;
    if (pl < P) {
        const f32x4 pv = np > 0 ? *(const f32x4*)(x + 4 * cqi) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 sv = {0.f, 0.f, 0.f, 0.f}, qv = {0.f, 0.f, 0.f, 0.f};
        int i = pl;
        for (; i + 3 * P < np; i += 4 * P) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const f32x4*)(x + (size_t)(i + u * P) * p.C + 4 * cqi);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[u][e] - pv[e]; sv[e] += d; qv[e] = __builtin_fmaf(d, d, qv[e]); }
        }
        for (; i < np; i += P) {
            const f32x4 v = *(const f32x4*)(x + (size_t)i * p.C + 4 * cqi);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[e] - pv[e]; sv[e] += d; qv[e] = __builtin_fmaf(d, d, qv[e]); }
        }
        *(f32x4*)&ls[pl * p.C + 4 * cqi] = sv;
        *(f32x4*)&lq[pl * p.C + 4 * cqi] = qv;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < p.C; c += 256) {
        const float pivot = np > 0 ? x[c] : 0.f;
        float s = 0.f, q = 0.f;
        for (int l = 0; l < P; ++l) { s += ls[l * p.C + c]; q += lq[l * p.C + c]; }      // ascending lane order
        const float inv = np > 0 ? 1.0f / (float)np : 0.f;
        cm[c] = pivot + s * inv;
        cq[c] = fmaxf(q - s * s * inv, 0.f);
    }
    __syncthreads();
    const int cg = p.C / p.groups;
    for (int g = threadIdx.x; g < p.groups; g += 256) {        // channels of a group merged in ascending order (equal counts: Chan's formula)
        double mean = 0.0, m2 = 0.0, n = 0.0;
        for (int e = 0; e < cg; ++e) {
            const double mc = cm[g * cg + e], qc = cq[g * cg + e], nc = np;
            const double nn = n + nc;
            if (nn > 0) {
                const double d = mc - mean;
                m2 += qc + d * d * n * nc / nn;
                mean += d * nc / nn;
            }
            n = nn;
        }
        float* o = p.ws + (((size_t)b * p.nblk + blk) * p.groups + g) * 2;
        o[0] = (float)mean; o[1] = (float)m2;
    }
}

// table [B, C, 2] = (gamma rstd, beta - mean gamma rstd): one thread per (sample, group), the pixel blocks merged in ascending order in double
__global__ void f32_gn_table_kernel(const F32Params p) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.B * p.groups) return;
    const int b = t / p.groups, g = t - b * p.groups;
    const int cg = p.C / p.groups;
    const int ppb = (p.HW + p.nblk - 1) / p.nblk;
    double mean = 0.0, m2 = 0.0, n = 0.0;
    for (int blk = 0; blk < p.nblk; ++blk) {
        const int np = max(0, min(p.HW, (blk + 1) * ppb) - blk * ppb);
        const float* s = p.ws + (((size_t)b * p.nblk + blk) * p.groups + g) * 2;
        const double nc = (double)np * cg, nn = n + nc;
        if (nc > 0) {
            const double d = (double)s[0] - mean;
            m2 += (double)s[1] + d * d * n * nc / nn;
            mean += d * nc / nn;
            n = nn;
        }
    }
    const double rstd = 1.0 / sqrt(m2 / n + (double)p.eps);
    float* tab = p.Y + ((size_t)b * p.C + g * cg) * 2;
    for (int e = 0; e < cg; ++e) {
        const double ga = p.gamma[g * cg + e], be = p.beta[g * cg + e];
        tab[2 * e] = (float)(ga * rstd);
        tab[2 * e + 1] = (float)(be - mean * ga * rstd);
    }
}

// y = silu?(x scale + shift); ws = the table; four channels per thread (C % 4 == 0)
__global__ void f32_gn_apply_kernel(const F32Params p) {
    const size_t total = (size_t)p.B * p.HW * p.C / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 4;
        const int c = (int)(e % p.C);
        const int b = (int)(e / ((size_t)p.HW * p.C));
        const f32x4 x = *(const f32x4*)(p.X + e);
        const float* tab = p.ws + ((size_t)b * p.C + c) * 2;
        f32x4 y;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = __builtin_fmaf(x[k], tab[2 * k], tab[2 * k + 1]);
            if (p.silu) v = v / (1.0f + __expf(-v));
            y[k] = v;
        }
        *(f32x4*)(p.Y + e) = y;
    }
}

// y[r, :] = softmax(scale * a[r, :]), one 256-thread workgroup per row
__global__ __launch_bounds__(256) void f32_softmax_kernel(const F32Params p) {
    __shared__ float red[8];
    const int r = blockIdx.x;
    const float* a = p.X + (size_t)r * p.ldx;
    float* y = p.Y + (size_t)r * p.ldy;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float mx = -3.0e38f;
    for (int i = threadIdx.x; i < p.N; i += 256) mx = fmaxf(mx, a[i]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int i = threadIdx.x; i < p.N; i += 256) s += __expf((a[i] - mx) * p.scale);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) red[4 + wave] = s;
    __syncthreads();
    const float inv = 1.0f / (((red[4] + red[5]) + red[6]) + red[7]);
    for (int i = threadIdx.x; i < p.N; i += 256) y[i] = __expf((a[i] - mx) * p.scale) * inv;
}

int f32_launch(int op, const F32Params& p, hipStream_t stream) {
    switch (op) {
    case 0: {                                            // GEMM / conv3x3
        if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.K % F_BK != 0) { set_error("f32 gemm: M, N > 0 and K %% 16 == 0 (M=%d N=%d K=%d)", p.M, p.N, p.K); return IMH_ERR_SHAPE; }
        if (p.conv && (p.Cin % F_BK != 0 || p.K != 9 * p.Cin || p.Ho != (p.H << p.up) || p.Wo != (p.Wd << p.up) || p.M % (p.Ho * p.Wo) != 0)) {
            set_error("f32 conv3x3: stride 1, Cin %% 16 == 0, K == 9 Cin, output = (upsampled) input size (Cin=%d K=%d H=%d W=%d up=%d Ho=%d Wo=%d M=%d)",
                      p.Cin, p.K, p.H, p.Wd, p.up, p.Ho, p.Wo, p.M);
            return IMH_ERR_SHAPE;
        }
        if (!p.conv && (p.ldx & 3)) { set_error("f32 gemm: ldx must be a multiple of 4"); return IMH_ERR_SHAPE; }
        if (p.ldw & 3) { set_error("f32 gemm: ldw must be a multiple of 4"); return IMH_ERR_SHAPE; }
        dim3 grid((p.M + F_BM - 1) / F_BM, (p.N + F_BN - 1) / F_BN);
        if (!g_f32_exact && p.K % X3_BK == 0 && (!p.conv || p.Cin % X3_BK == 0)) {
            hipLaunchKernelGGL(f32_gemm_x3_kernel, grid, dim3(256), 0, stream, p);
            return check_launch("f32_gemm_x3_kernel");
        }
        hipLaunchKernelGGL(f32_gemm_kernel, grid, dim3(256), 0, stream, p);
        return check_launch("f32_gemm_kernel");
    }
    case 1:
        if (p.C > 512 || p.C % 4 || p.C % p.groups || p.nblk <= 0) { set_error("f32 groupnorm: C <= 512, C %% 4 == 0, C %% groups == 0 (C=%d groups=%d)", p.C, p.groups); return IMH_ERR_SHAPE; }
        hipLaunchKernelGGL(f32_gn_stats_kernel, dim3(p.nblk, p.B), dim3(256), 0, stream, p);
        return check_launch("f32_gn_stats_kernel");
    case 2:
        hipLaunchKernelGGL(f32_gn_table_kernel, dim3((p.B * p.groups + 63) / 64), dim3(64), 0, stream, p);
        return check_launch("f32_gn_table_kernel");
    case 3: {
        if (p.C % 4) { set_error("f32 groupnorm apply: C %% 4 == 0"); return IMH_ERR_SHAPE; }
        const size_t total = (size_t)p.B * p.HW * p.C / 4;
        const int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
        hipLaunchKernelGGL(f32_gn_apply_kernel, dim3(blocks), dim3(256), 0, stream, p);
        return check_launch("f32_gn_apply_kernel");
    }
    case 4:
        if (p.M <= 0 || p.N <= 0) { set_error("f32 softmax: empty"); return IMH_ERR_SHAPE; }
        hipLaunchKernelGGL(f32_softmax_kernel, dim3(p.M), dim3(256), 0, stream, p);
        return check_launch("f32_softmax_kernel");
    default:
        set_error("imh_f32: unknown op %d", op);
        return IMH_ERR_ARG;
    }
}

}  // namespace imh
