// Internal launcher interface between the C ABI (api.hip) and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "imh_gntable.h"

namespace imh {

struct GemmParams {
    const void* X;         // [M, K] tokens (ldx) -- or NHWC conv input [B, H, Wd, Cin]
    const void* W;         // [N, K] weights, K contiguous (ldw)
    void* Y;               // [M, ldy] output (T, or fp32 with GF_OUT_F32)
    float* partial;        // split-K slabs [splits][M][N] fp32
    const void* bias;      // [N] or null
    const void* rowadd;    // [M / rows_per_batch, N] broadcast add (time-embedding projection) or null
    const void* residual;  // [M, ldr] or null (added after activation / GEGLU)
    // LayerNorm folded into the GEMM (GF_LN_ROW / GF_LN_COL): y = rstd*(acc - mean*ln_s) + ln_c, W pre-scaled by gamma;
    // (mean, rstd) of the un-normalised token rows are taken INSIDE the K loop from the MFMA operand fragments
    const float* ln_s;     // sum_k gamma_k W[.,k]   (row form: per n; col form: per m)
    const float* ln_c;     // sum_k beta_k  W[.,k]   (same indexing as ln_s)
    float ln_eps;
    // row statistics hand-over (imh_lnstats.h): ln_stats = precomputed (sum, M2) slot partials of the token rows [tokens][ln_slots][2]
    // (null -> the kernel takes the statistics inside its K loop); ln_stats_out = partials of THIS launch's output rows
    // [M][ln_slots_out][2], written by the epilogue (null -> none)
    const float* ln_stats;
    float* ln_stats_out;
    int ln_slots, ln_slots_out;
    // GroupNorm statistics of THIS launch's output (NHWC rows = pixels), for the GroupNorm that reads it: per (sample, pixel block of
    // one wave's rows, sub-run of 10 channels) the (sum, M2) of the values as stored (imh_lnstats.h gn_emit)
    float* gn_out;
    int gn_nblk, gn_hw;
    // GroupNorm (+ SiLU) of the INPUT applied inside the conv's halo staging (conv_halo.hip): table[b][Cin][2] = (scale, shift)
    const float* gn_tab;
    int gn_silu;
    GnTabSrc gn_src;       // ... or the input's GroupNorm partials: the kernel builds its sample's table in LDS (gn_src.partial != null)
    // channel concat as the conv input: channels [0, Cin1) come from X (pixel stride Cin1), [Cin1, Cin) from X2 (pixel stride Cin - Cin1)
    const void* X2;
    int Cin1;
    // wave-specialised row-form folded-LayerNorm launches: output columns >= yt_col0 are stored TRANSPOSED (V^T layout, 16-token
    // groups permuted) into Yt [N - yt_col0, ldyt] instead of Y -- the V third of the self-attention's one-launch [Q|K|V]
    void* Yt;
    int yt_col0, ldyt;
    int M, N, K;
    int ldx, ldw, ldy, ldr, ldra;
    int rows_per_batch;
    int splits;
    int flags;
    // implicit-GEMM conv3x3 (pad 1): input H x Wd (virtual size << up), output Ho x Wo
    int H, Wd, Cin, Ho, Wo, stride, up;
    // XCD-aware tile placement (set by the launchers): the 8 XCDs form a px x py grid over the tile space
    int px, py, tmx, tny;
    int xcd;               // requested cell shape (imh_gemm_args.xcd): 0 = cost model, 2 .. 5 = (8,1) (4,2) (2,4) (1,8)
    const void* pf_ptr;    // next kernel's weights (tail prefetch), or null
    unsigned pf_bytes;
    int early_res;         // wave-specialised kernels: fetch the residual rows / bias BEFORE the K loop (set by the launcher; g_ws_early)
};

void gemm_pick_config(int M, int N, int K, int* bm, int* bn, int* splits);
size_t gemm_workspace_bytes(int M, int N, int splits);
int gemm_stats_slot_width(int bm, int bn);       // channels per ln_stats_out slot of a tile variant, 0 = no statistics epilogue
int gemm_gn_block_rows(int bm, int bn);         // pixels per gn_out block of a tile variant (one wave's rows), 0 = no GroupNorm epilogue
int gemm_launch(GemmParams p, int dtype, int conv, int bm, int bn, hipStream_t stream);
int gemm_dual_launch(GemmParams a, GemmParams b, int dtype, int bm, int bn, hipStream_t stream);
int gemm_w16_launch(const GemmParams& p, int dtype, hipStream_t stream);      // variant 26256 x 320 (gemm_w16.hip)

struct AttnParams {
    const void* Q;    // [B, Lq, ldq] (+ head*64 columns)
    const void* K;    // [B, Lk_pad, ldk] keys, row-major per key
    const void* Vt;   // [H*64, ldvt] V transposed, keys permuted in 16-groups; batch b at column b*Lk_pad
    const void* K2;   // optional second (image-prompt) key set, same layouts
    const void* Vt2;
    void* O;          // [B, Lq, ldo]
    int B, H, Lq;
    int Lk, Lk_pad;   // valid keys / padded row count per batch (multiple of 64)
    int Lk2, Lk2_pad;
    int ldq, ldk, ldvt, ldk2, ldvt2, ldo;
    float scale;      // softmax scale (1/sqrt(64))
    float scale2;     // weight of the second attention (IP scale)
    const float* scale2_tab;  // optional per-step table of IP scales, indexed by *step
    const int* step;
    const void* pf_ptr;       // next kernel's weights (tail prefetch), or null
    unsigned pf_bytes;
    int split;                // pipelined kernel: items per XCD dealt as four key-quarter workgroups of 32 queries (set by the launcher)
    float defer_log2;         // pipelined key loop: keep the running maximum while no row's maximum grew by more than 2^this (set by the launcher)
};
int attention_launch(const AttnParams& p, int dtype, hipStream_t stream);

// fused to_q (+ folded LayerNorm) + text / image-prompt cross-attention (xattn.hip): `a` carries the key / value caches,
// O and the shapes (a.Q is unused); the K caches hold the head dims of every 16-group in vt_perm16 order
struct XAttnParams {
    AttnParams a;
    const void* X;        // [B*Lq, ldx] token rows (un-normalised when ln_s != null)
    const void* Wq;       // [H*64, ldw] to_q weight, [out, in] (pre-scaled by gamma when ln_s != null)
    const float* ln_s;    // [H*64] sum_k gamma_k Wq[d, k]  or null
    const float* ln_c;    // [H*64] sum_k beta_k  Wq[d, k]
    float ln_eps;
    const float* ln_stats;   // precomputed row statistics of X (imh_lnstats.h) or null
    int ln_slots;
    int C, ldx, ldw;
    int split;            // (set by the launcher) items per XCD dealt as two 64-query halves
};
int xattn_launch(const XAttnParams& p, int dtype, hipStream_t stream);
extern int g_attn_force_nw;
extern int g_xattn_mode;
extern int g_w16_pf;
extern int g_w16_form;
extern int g_f32_exact;
extern int g_attn_mode;
extern int g_xcd_mode;
extern int g_halo_mode;
extern int g_ws_early;

struct SmallAttnParams {
    const void* Q; const void* K; const void* V; void* O;
    int B, H, Lq, Lk, dq, dv;
    int ldq, ldk, ldv, ldo;
    float scale;
};
int attention_small_launch(const SmallAttnParams& p, int dtype, hipStream_t stream);

struct NormParams {
    const void* x;
    void* y;
    const void* gamma;
    const void* beta;
    float* partial;   // group-norm partial sums workspace
    int B, HW, C, groups;
    int rows;         // layer norm: rows
    float eps;
    int silu;
    // group norm (norm.hip): mode 0 statistics + table + apply, 1 statistics only, 2 table from partials, 3 apply a table
    int mode;
    float* table;             // [B][C][2] (scale, shift)
    const float* partial2;    // second producer's partials (channel concat) or null
    int nblk, sub, npart;     // source 1: partial blocks per sample, channels per sub-run, elements per partial (0 = ragged stats-kernel blocks)
    int C1;                   // channels covered by source 1
    int nblk2, sub2, npart2;
    int dtype_f16;            // (set by the launcher)
    const void* pf_ptr;   // next kernel's weights (tail prefetch), or null
    unsigned pf_bytes;
};
int groupnorm_launch(const NormParams& p, int dtype, hipStream_t stream);
size_t groupnorm_workspace_bytes(int B, int HW, int C, int groups);
int groupnorm_stats_blocks(int HW, int C);
int groupnorm_stats_sub(int C, int groups);
int layernorm_launch(const NormParams& p, int dtype, hipStream_t stream);

// fp32 (reference-precision) kernels of the VAE decode tail (f32.hip).  op 0: GEMM / conv3x3 (X, W, Y, bias, residual; conv fields);
// 1: GroupNorm statistics (X [B, HW, C] -> ws [B, nblk, groups, 2]); 2: table (ws, gamma, beta -> Y [B, C, 2]); 3: apply (X, ws = table -> Y);
// 4: row softmax (X [M, ldx] -> Y [M, ldy], N columns, scale)
struct F32Params {
    const float* X; const float* W; float* Y; const float* bias; const float* residual; const float* gamma; const float* beta; float* ws;
    int M, N, K, ldx, ldw, ldy, ldr;
    int conv, H, Wd, Cin, Ho, Wo, up;
    int B, HW, C, groups, nblk, silu;
    float eps, scale;
};
int f32_launch(int op, const F32Params& p, hipStream_t stream);

struct EwParams {
    const void* a;
    const void* b;
    void* y;
    const void* w;
    const void* bias;
    const float* tab;   // optional per-step scalar table
    const int* step;    // device-resident denoise step counter
    long long n;
    int i0, i1, i2, i3, i4, i5;
    float f0, f1, f2, f3;
};
int ew_launch(int op, const EwParams& p, int dtype, hipStream_t stream);

}  // namespace imh
