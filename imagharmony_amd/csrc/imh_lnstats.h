// Row statistics of the transformer blocks' residual stream, handed from the GEMM that WRITES a LayerNorm input
// (to_out + residual, ff.out + residual, proj_in) to the kernels that consume it with the LayerNorm folded in
// (ff.net.0, [Q|K] + V^T, the fused cross-attention's to_q): diffusers BasicTransformerBlock.norm1/2/3
//  without a pass over the tensor and without statistics work inside the consumers' K loops.
//
// Format: stats[token][slot] = (sum, M2) as two floats, one slot per run of SW consecutive channels (SW = the width
// a producing wave owns: 80 for the wave-specialised 160-column tiles, BN / 2 for the plain tiles), M2 = sum of
// squared deviations from the SLOT mean, both taken from the values as rounded to the output dtype (what the consumer
// will read).  Slots are merged with Chan's formula, so the variance never comes from E[x^2] - mean^2 and a large
// common offset of a row costs no precision (torch's LayerNorm, which the reference runs, is Welford-based too).
#pragma once
#include "imh_common.h"

namespace imh {

typedef __attribute__((ext_vector_type(2))) float f32x2s;

// Producer side.  The four 16-lane groups of a wave hold four consecutive NV-column runs of the same output rows
// (lane l, l ^ 16, l ^ 32, l ^ 48 share row l & 15): per lane a two-pass (sum, M2) over its NV values, two equal-count
// Chan merges through v_permlane16_swap / v_permlane32_swap, and lanes 0-15 store the slot's pair.
// nb = first column of the lane's run; the slot is nb / (4 * NV).  Every lane of the wave must call this (the rows of
// one 16-lane group are either all valid or all skipped by the caller's row guard: same row, same m).
template <typename T, int NV>
__device__ __forceinline__ void emit_row_stats(float* stats, const int slots, const int m, const int nb, const float* v,
                                               const int lane) {
    float r[NV];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NV; ++q) { r[q] = to_f32(from_f32<T>(v[q])); s += r[q]; }
    const float mean = s * (1.0f / NV);
    float m2 = 0.f;
#pragma unroll
    for (int q = 0; q < NV; ++q) { const float d = r[q] - mean; m2 = __builtin_fmaf(d, d, m2); }
    {   // rows of 16 lanes: {0,1} and {2,3} (v_permlane16_swap with itself leaves {r0,r0,r2,r2} / {r1,r1,r3,r3})
        auto a = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, s), __builtin_bit_cast(unsigned, s), false, false);
        auto b = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, m2), __builtin_bit_cast(unsigned, m2), false, false);
        const float s0 = __builtin_bit_cast(float, (unsigned)a[0]), s1 = __builtin_bit_cast(float, (unsigned)a[1]);
        const float d = s0 - s1;
        m2 = __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]) + d * d * (0.5f / NV);
        s = s0 + s1;
    }
    {   // halves of the wave
        auto a = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, s), __builtin_bit_cast(unsigned, s), false, false);
        auto b = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, m2), __builtin_bit_cast(unsigned, m2), false, false);
        const float s0 = __builtin_bit_cast(float, (unsigned)a[0]), s1 = __builtin_bit_cast(float, (unsigned)a[1]);
        const float d = s0 - s1;
        m2 = __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]) + d * d * (0.25f / NV);
        s = s0 + s1;
    }
    if (lane < 16) {
        f32x2s o = {s, m2};
        *(f32x2s*)(stats + ((size_t)m * slots + nb / (4 * NV)) * 2) = o;
    }
}

// GroupNorm partials from a GEMM / conv epilogue (format 2, round 4): ONE (sum, M2) pair per (sample, pixel block = one wave's
// rows, sub-run of 10 consecutive output channels -- the finest SDXL group, C = 320 / 32), M2 = sum of squared deviations from
// the PARTIAL's own mean, both taken from the values as stored (rounded to T).  Any consumer grouping (10 / 20 / 30 / 40 / 60 / 80
// channels per group, also across the two producers of a channel concat) is assembled from sub-runs by the finalise kernel
// (norm.hip gn_table_kernel).  No E[x^2] - mean^2 anywhere: a lane accumulates sum(x - p), sum((x - p)^2) about a PIVOT p (the
// first value of its sub-run), converts to (sum, M2) and the 16 lanes of a lane group (= 16 rows) merge by equal-count Chan
// butterflies -- a common offset of the tensor costs no precision (diffusers GroupNorm = torch.nn.GroupNorm, two-pass).
// partial[((b * nblk + blk) * (N / 10) + nb / 10 + k) * 2 + {0, 1}], n = 160 * FM elements each.
template <int NV>
struct GnAcc {
    float s[NV / 10], q[NV / 10], pv[NV / 10];
};
template <int NV>
__device__ __forceinline__ void gn_zero(GnAcc<NV>& g) {
#pragma unroll
    for (int k = 0; k < NV / 10; ++k) { g.s[k] = 0.f; g.q[k] = 0.f; g.pv[k] = 0.f; }
}
// one row of the lane's NV values as stored; first = the lane's first row (sets the pivots)
template <typename T, int NV>
__device__ __forceinline__ void gn_accumulate(const float* v, GnAcc<NV>& g, const bool first) {
    if (first) {
#pragma unroll
        for (int k = 0; k < NV / 10; ++k) g.pv[k] = to_f32(from_f32<T>(v[k * 10]));
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const float r = to_f32(from_f32<T>(v[q])) - g.pv[q / 10];
        g.s[q / 10] += r;
        g.q[q / 10] = __builtin_fmaf(r, r, g.q[q / 10]);
    }
}
// rows = rows accumulated per lane (FM); every lane of the wave must call this (butterflies)
template <int NV>
__device__ __forceinline__ void gn_emit(float* out, const int nblk, const int nsub, const int b, const int blk, const int nb,
                                        const GnAcc<NV>& g, const int rows, const int lane) {
    float* base = out + (((size_t)b * nblk + blk) * nsub + nb / 10) * 2;
#pragma unroll
    for (int k = 0; k < NV / 10; ++k) {
        float n = 10.0f * (float)rows;
        float sum = __builtin_fmaf(n, g.pv[k], g.s[k]);
        float m2 = fmaxf(g.q[k] - g.s[k] * g.s[k] / n, 0.f);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const float so = __shfl_xor(sum, o, 64), mo = __shfl_xor(m2, o, 64);
            const float d = sum - so;
            m2 = m2 + mo + d * d / (2.0f * n);
            sum += so;
            n *= 2.0f;
        }
        if ((lane & 15) == 0) { f32x2s o2 = {sum, m2}; *(f32x2s*)(base + k * 2) = o2; }
    }
}

// Consumer side: (mean, rstd) of one token row from its `slots` partials (equal counts K / slots each).  Equal counts make the
// merge a two-pass over the partials themselves: mean = (sum of the slot sums) / K, M2 = sum_i M2_i + n_i (mean_i - mean)^2 --
// every term a non-negative square of a small difference, never E[x^2] - mean^2 (a row with |mean| >> sigma costs no
// precision), and no division per slot (the sequential Chan chain of round 3 spent ~10 VALU + a v_rcp per slot at the head
// of every consumer).  The loads of a batch are unconditional (index clamped, the duplicate's weight is 0): a guarded load
// would make hipcc branch around and wait for every element.  One thread per row.
__device__ __forceinline__ f32x2s merge_row_stats(const float* stats, const int row, const int slots, const int K,
                                                  const float eps) {
    const f32x2s* st = (const f32x2s*)(stats + (size_t)row * slots * 2);
    const float ni = (float)K / (float)slots;
    const float inv_ni = 1.0f / ni;
    float mean, m2 = 0.f;
    if (slots <= 16) {                      // every SDXL width on the 80-column slots (C / 80 = 4, 8, 16): one batch, kept in registers
        f32x2s t[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) t[j] = st[min(j, slots - 1)];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += (j < slots) ? t[j][0] : 0.f;
        mean = s / (float)K;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float d = __builtin_fmaf(t[j][0], inv_ni, -mean);
            m2 += (j < slots) ? __builtin_fmaf(d * ni, d, t[j][1]) : 0.f;
        }
    } else {
        float s = 0.f;
        for (int i0 = 0; i0 < slots; i0 += 16) {
            f32x2s t[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) t[j] = st[min(i0 + j, slots - 1)];
#pragma unroll
            for (int j = 0; j < 16; ++j) s += (i0 + j < slots) ? t[j][0] : 0.f;
        }
        mean = s / (float)K;
        for (int i0 = 0; i0 < slots; i0 += 16) {      // second pass: the row's partials are L1-resident now
            f32x2s t[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) t[j] = st[min(i0 + j, slots - 1)];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float d = __builtin_fmaf(t[j][0], inv_ni, -mean);
                m2 += (i0 + j < slots) ? __builtin_fmaf(d * ni, d, t[j][1]) : 0.f;
            }
        }
    }
    f32x2s o;
    o[0] = mean;
    o[1] = rsqrtf(m2 / (float)K + eps);
    return o;
}

}  // namespace imh
