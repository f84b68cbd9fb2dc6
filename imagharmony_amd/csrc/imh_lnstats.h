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

// GroupNorm partials from a GEMM / conv epilogue.  A lane accumulates (sum, sum of squares) of its NV = 20 / 40 consecutive output
// channels in sub-runs of 10 (the finest SDXL group: C = 320 / 32) over its rows; here the 16 lanes of a 16-lane group (= 16 rows)
// are added up, sub-runs merged to the group width cpg (10, 20 or 40 channels; at cpg = 40 with NV = 20 two neighbouring lane
// groups form one group), and one lane per group stores the pair.  partial[((b * nblk + blk) * groups + g) * 2 + {0, 1}].
template <int NV>
__device__ __forceinline__ void gn_emit(float* out, const int nblk, const int groups, const int cpg, const int b, const int blk,
                                        const int nb, float (&gs)[NV / 10], float (&gq)[NV / 10], const int lane) {
    constexpr int NS = NV / 10;
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { gs[k] += __shfl_xor(gs[k], o, 64); gq[k] += __shfl_xor(gq[k], o, 64); }
    float* base = out + ((size_t)b * nblk + blk) * groups * 2;
    if (cpg == 10) {
        if ((lane & 15) == 0) {
#pragma unroll
            for (int k = 0; k < NS; ++k) { f32x2s o2 = {gs[k], gq[k]}; *(f32x2s*)(base + (nb / 10 + k) * 2) = o2; }
        }
    } else if (cpg == 20) {
        if ((lane & 15) == 0) {
#pragma unroll
            for (int k = 0; k < NS; k += 2) { f32x2s o2 = {gs[k] + gs[k + 1], gq[k] + gq[k + 1]}; *(f32x2s*)(base + (nb / 20 + k / 2) * 2) = o2; }
        }
    } else {        // cpg == 40
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int k = 0; k < NS; ++k) { s += gs[k]; q += gq[k]; }
        if constexpr (NV == 20) { s = xor16_sum(s); q = xor16_sum(q); }      // lane groups (0,1) and (2,3) hold the two halves
        if ((lane & (NV == 20 ? 31 : 15)) == 0) { f32x2s o2 = {s, q}; *(f32x2s*)(base + (nb / 40) * 2) = o2; }
    }
}
// per-row accumulation of the lane's values as stored (rounded to T)
template <typename T, int NV>
__device__ __forceinline__ void gn_accumulate(const float* v, float (&gs)[NV / 10], float (&gq)[NV / 10]) {
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const float r = to_f32(from_f32<T>(v[q]));
        gs[q / 10] += r;
        gq[q / 10] = __builtin_fmaf(r, r, gq[q / 10]);
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
