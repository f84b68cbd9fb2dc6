/*
 * libimh_hip.so -- C ABI of the MI355X (gfx950) kernels behind the IMAGHarmony SDXL
 * denoising hot path.
 *
 * The reference (muzishen/IMAGHarmony) is pure Python and has no FFI of its own; every GPU
 * kernel it runs comes from PyTorch through diffusers.  The entry points below are what a
 * Python binding of THIS path binds with ctypes (INTEGRATION.md shows the stub); each one
 * cites the reference call sites whose arithmetic it replaces.
 *
 * Contract
 *   - plain C: raw device pointers, ints, floats; no torch / C++ types in any signature.
 *   - ownership: the caller owns every buffer (activations, weights, workspaces).  Kernels
 *     never allocate or free; workspace sizes are queried with *_workspace_bytes().
 *   - all work is enqueued on the caller's stream (hipStream_t passed as void*); no internal
 *     synchronisation, no malloc/free -> every entry point is hipGraph-capturable.
 *   - errors: 0 (IMH_OK) or a negative imh_status; imh_last_error() returns a thread-local
 *     message.  Nothing throws across the ABI, nothing calls exit().
 *   - stateless and re-entrant (plans are explicit handles owned by the caller).
 *   - dtype: IMH_DT_BF16 / IMH_DT_F16 activations+weights, fp32 accumulation everywhere.
 *   - layouts: activations are token-major / NHWC ([B, H*W, C] row-major); weights are
 *     [out, in] row-major exactly as torch.nn.Linear stores them; conv3x3 weights are
 *     pre-packed to [Cout][ky][kx][Cin] (imagharmony_amd.unet.Conv2d.packed).
 */
#ifndef IMH_H_
#define IMH_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMH_ABI_VERSION 9

enum imh_status {
    IMH_OK = 0,
    IMH_ERR_ARG = -1,
    IMH_ERR_SHAPE = -2,
    IMH_ERR_DTYPE = -3,
    IMH_ERR_LAUNCH = -4,
    IMH_ERR_WORKSPACE = -5
};

enum imh_dtype { IMH_DT_BF16 = 0, IMH_DT_F16 = 1 };

/* epilogue flags of imh_gemm_args.flags */
enum imh_gemm_flags {
    IMH_GF_GEGLU = 1,     /* diffusers GEGLU: W rows interleaved in quads (value_2k, value_2k+1, gate_2k, gate_2k+1);
                           * out[2k], out[2k+1] = value * gelu(gate); N % 16 == 0, Y is [M, N/2] */
    IMH_GF_ACT_GELU = 2,  /* erf GELU, nn.GELU() (resampler.py:18); |error| <= 1.2e-5 (csrc/imh_common.h gelu_erf_f) */
    IMH_GF_ACT_SILU = 4,  /* SiLU (TimestepEmbedding) */
    IMH_GF_VT_PERM = 8,   /* write the attention V^T key permutation (see imh_attention) */
    IMH_GF_OUT_F32 = 16,  /* fp32 output */
    IMH_GF_LN_ROW = 32,   /* folded LayerNorm, the un-normalised token rows are X's rows (statistics per output row m) */
    IMH_GF_LN_COL = 64    /* folded LayerNorm, the token rows are W's rows (per output column n; swapped-operand V^T form) */
};

/* ---- dense contraction ------------------------------------------------------------------
 * Y[m, n] = epilogue( sum_k X[m, k] * W[n, k] ),  epilogue = (+bias[n]) (+rowadd[m / rows_per_batch, n])
 *           (act) (GEGLU) (+residual[m, n]).
 * conv == 0: torch.nn.Linear.  Replaces attn.to_q/to_k/to_v/to_out[0] and to_k_ip/to_v_ip
 *   (ip_adapter/attention_processor.py:292,299,300,320,396,410,411,432,433,453), diffusers
 *   proj_in/proj_out/FeedForward/TimestepEmbedding/time_emb_proj/conv_shortcut,
 *   Resampler / ImageProjModel / HarmonyAttention linears (resampler.py:13-20,45-47,101-103;
 *   ip_adapter.py:38; train.py:208,239).
 * conv == 1: 3x3 convolution, padding 1, stride 1|2, optional fused nearest x2 upsampling of the
 *   input (up = 1), as an implicit GEMM over NHWC input [B, H, Wd, Cin]; K = 9*Cin; M = B*Ho*Wo.
 *   Replaces diffusers ResnetBlock2D.conv1/conv2, Downsample2D.conv, Upsample2D(+interpolate), conv_out.
 * K must be a multiple of 64; M and N are arbitrary (edge tiles read a zero page).
 * bm/bn/splits = 0 selects the built-in heuristic.  splits > 1 needs `partial`
 * (imh_gemm_workspace_bytes) and runs a second reduce+epilogue kernel.
 */
typedef struct imh_gemm_args {
    const void* X;
    const void* W;
    void* Y;
    float* partial;
    const void* bias;
    const void* rowadd;
    const void* residual;
    /* LayerNorm folded into the contraction (IMH_GF_LN_ROW: X rows are the un-normalised tokens; IMH_GF_LN_COL:
     * W rows are).  With W pre-scaled by gamma:  y = rstd * (acc - mean * ln_s) + ln_c  (exact algebra of
     * LN(x) W^T; BasicTransformerBlock.norm1/2/3 never materialise and cost no launch).  (mean, rstd) of every
     * token row come from `ln_stats` (below; REQUIRED with either flag -- the in-loop sum / sum-of-squares form of ABI
     * versions <= 5 cancelled on rows with |mean| >> sigma and no longer exists; IMH_ERR_ARG without it).
     * ln_s = sum_k gamma_k W[.,k], ln_c = sum_k beta_k W[.,k] (fp32; row form: per output column n, column form: per
     * output row m).  Variants: plain 64/128 tiles (both forms) and the wave-specialised ones 1464 / 2464 / 24128 /
     * 23256 / 22128 (row form).  splits == 1, conv == 0. */
    const float* ln_s;
    const float* ln_c;
    float ln_eps;
    /* Row-statistics hand-over between the GEMM that WRITES a LayerNorm input (attn.to_out + residual,
     * attention_processor.py:320-329,453-462; ff.net.2 + residual; proj_in) and the launches that consume it folded.
     * Format (csrc/imh_lnstats.h): [rows][slots][2] fp32 = (sum, M2 about the slot mean) of `width` consecutive
     * channels per slot, taken from the values as rounded to the output dtype; slots merge by Chan's formula (no
     * E[x^2] - mean^2 cancellation).
     *   ln_stats / ln_slots          input: partials of the token rows of a IMH_GF_LN_ROW / _COL launch (K % slots == 0)
     *   ln_stats_out / ln_slots_out  output: partials of THIS launch's Y rows; ln_slots_out must equal
     *                                N / imh_gemm_stats_slot_width(bm, bn) (N a multiple of that width, plain T output,
     *                                no GEGLU / V^T permutation / fp32 output / split-K / conv)
     * IMH_EW_ROW_STATS writes the same format (one slot per row) for rows no statistics epilogue covers. */
    const float* ln_stats;
    float* ln_stats_out;
    int32_t ln_slots, ln_slots_out;
    /* GroupNorm partials of THIS launch's output, for the GroupNorm that reads it (diffusers ResnetBlock2D.norm2 after conv1, the
     * next block's norm1 / Transformer2DModel.norm after conv2 / conv_shortcut + residual; also through a channel concat): one
     * (sum, M2) pair -- M2 = squared deviations from the partial's own mean, values as rounded to the output dtype -- per
     * (sample, pixel block, sub-run of 10 consecutive output channels): gn_out[((b * gn_nblk + blk) * (N / 10) + n / 10) * 2 + {0, 1}]
     * fp32, one block per imh_gemm_gn_block_rows(bm, bn) consecutive rows (pixels) of a sample (10 * that many elements per
     * partial); gn_hw = rows per sample, gn_nblk = gn_hw / block rows.  Hand the buffer to imh_groupnorm (mode IMH_GN_TABLE:
     * partial / nblk = gn_nblk / sub = 10 / npart = 10 * block rows).  NULL -> none.  Variants with this epilogue: the
     * wave-specialised ones at bn = 160 and the LDS-halo conv3x3; N % 10 == 0; no split-K, GEGLU, V^T permutation, fp32 output, LN. */
    float* gn_out;
    int32_t gn_nblk, gn_hw;
    /* LDS-halo conv3x3 only (variants 7128 / 7564 / 7256 / 73xx / 74xx, stride 1, no upsampling): the ResnetBlock2D front end
     * norm -> SiLU -> conv in one launch.  gn_tab = the (scale, shift) table of the INPUT's GroupNorm, [B][Cin][2] fp32 as written
     * by imh_groupnorm(mode IMH_GN_TABLE): every staged input pixel becomes silu?(x * scale + shift) inside the kernel (padding
     * stays zero: the conv pads the normalised tensor); the normalised activation never exists in memory.  NULL -> plain conv. */
    const float* gn_tab;
    int32_t gn_silu;
    /* ... or (ABI 8) the PARTIALS of the input's GroupNorm instead of a table: every workgroup then builds the (scale, shift) table of its
     * sample in LDS in its prologue (the routine of imh_groupnorm's IMH_GN_TABLE step, bit-identical) and no table launch is needed.
     * gn_part (+ gn_part2 for channels [gn_pC1, Cin) of a channel concat) / gn_pnblk / gn_psub / gn_pnpart (+ ...2) as imh_norm_args.partial /
     * nblk / sub / npart; gn_gamma / gn_beta [Cin] in the activation dtype (NULL = 1 / 0), gn_groups, gn_eps > 0.  Exclusive with gn_tab. */
    const float* gn_part;
    const float* gn_part2;
    const void* gn_gamma;
    const void* gn_beta;
    float gn_eps;
    int32_t gn_groups, gn_pC1;
    int32_t gn_pnblk, gn_psub, gn_pnpart;
    int32_t gn_pnblk2, gn_psub2, gn_pnpart2;
    /* LDS-halo conv3x3 only: the input is the channel concat [X | X2] (torch.cat([hidden, skip], 1) of the up blocks) read from
     * its two producers: channels [0, Cin1) from X (pixel stride Cin1), [Cin1, Cin) from X2 (pixel stride Cin - Cin1); both
     * multiples of 64.  X2 == NULL -> one source. */
    const void* X2;
    int32_t Cin1;
    /* Wave-specialised variants at bn = 160 with flags == IMH_GF_LN_ROW (the self-attention projections, attention_processor.py:
     * 292-300, as ONE launch over W = [Wq; Wk; Wv]): output columns n >= yt_col0 are not written to Y but TRANSPOSED to
     * Yt[(n - yt_col0) * ldyt + m'] with the 16 tokens of every aligned group stored as [0-3, 8-11, 4-7, 12-15] -- the V^T operand
     * layout of imh_attention (what IMH_GF_VT_PERM produces for the swapped-operand form).  Whole tiles only (M a multiple of the
     * variant's rows, N and yt_col0 multiples of 160); Y receives columns [0, yt_col0) with row stride ldy.  NULL -> all of Y. */
    void* Yt;
    int32_t yt_col0, ldyt;
    int32_t M, N, K;
    int32_t ldx, ldw, ldy, ldr, ldra;   /* ldra: row stride of rowadd (0 -> N) */
    int32_t rows_per_batch;
    int32_t splits;
    int32_t flags;
    int32_t H, Wd, Cin, Ho, Wo, stride, up;
    int32_t dtype;
    int32_t conv;
    /* tile variant (0 = heuristic): bm in {64, 128} x bn in {64, 128}: two-stage tiles (gemm.hip); 256 x {128, 256}:
     * 8-wave rings; 3064 x 64 / 3128 x 128: KG2; 4064 / 4128 / 5064: small-tile rings; 5258 x 320, 6128 x 320: the
     * 256 x 320 / 128 x 320 exact tilings; 8256 x 256, 9128 x 320, 9256 x 320: ping-pong kernels (gemm_pp.hip);
     * 1464 / 2464 x 160, 24128 x 160 / 128, 23256 x 160 / 128, 22128 x 160 (two workgroups per CU): wave-specialised kernel
     * (producer + consumer waves, gemm_ring.hip); 7128 / 7564 x 320 / 160, 7256 x 160 and the weight-ring forms
     * 7328 / 7428 / 7356 x 160: LDS-halo conv3x3 (conv_halo.hip, stride 1). */
    int32_t bm, bn;
    /* cache hint: the NEXT launch's weight matrix; exiting workgroups touch it (HBM -> L2 / Infinity Cache) */
    const void* pf_ptr;
    uint32_t pf_bytes;
    /* how the tile grid is dealt to the eight XCDs (workgroup w runs on XCD w % 8; each XCD takes one cell of an M x N grid of
     * cells): 0 = the byte-count model picks the shape, 2 / 3 / 4 / 5 = 8 x 1 / 4 x 2 / 2 x 4 / 1 x 8 cells.  Placement only:
     * results are bit-identical.  Which shape is faster depends on the box, so the host measures. */
    int32_t xcd;
} imh_gemm_args;

int imh_gemm(const imh_gemm_args* a, void* stream);
/* two independent plain GEMMs of one dtype in ONE launch (a's tile config is used for both); e.g. the [Q|K]
 * and V^T projections of a self-attention layer (attention_processor.py:292,299,300), which share x.
 * In a plan: kind IMH_OP_GEMM_DUAL with args = imh_gemm_args[2]. */
int imh_gemm_dual(const imh_gemm_args* a, const imh_gemm_args* b, void* stream);
int imh_gemm_pick_config(int M, int N, int K, int* bm, int* bn, int* splits);
/* channels per statistics slot written by tile variant (bm, bn) through ln_stats_out; 0 = the variant has no
 * statistics epilogue (use IMH_EW_ROW_STATS on its output instead) */
int imh_gemm_stats_slot_width(int bm, int bn);
/* rows per GroupNorm partial block written by tile variant (bm, bn) through gn_out; 0 = no such epilogue */
int imh_gemm_gn_block_rows(int bm, int bn);
size_t imh_gemm_workspace_bytes(int M, int N, int splits);

/* ---- attention, head_dim 64 -------------------------------------------------------------
 * O[b, q, h*64:(h+1)*64] = softmax(Q K^T * scale) V  (+ scale2 * softmax(Q K2^T * scale) V2)
 * Replaces F.scaled_dot_product_attention in AttnProcessor2_0 (attention_processor.py:312) and both
 * SDPA calls + the axpy of IPAttnProcessor2_0 (:423-425, :440-442, :450).
 *   Q  : [B, Lq, ldq], head h at columns h*64..
 *   K  : [B, Lk_pad, ldk] (rows >= Lk are padding and must be finite), head h at columns h*64..
 *   Vt : [H*64, ldvt] = V transposed; batch b occupies columns b*Lk_pad .. ; inside every group of
 *        16 keys the order is [0-3, 8-11, 4-7, 12-15] (what imh_gemm writes with IMH_GF_VT_PERM);
 *        padding columns must be finite (zero).
 *   K2/Vt2 (optional): the image-prompt key set of the IP branch, same layouts.
 * Lk_pad, Lk2_pad multiples of 64.
 */
typedef struct imh_attn_args {
    const void* Q;
    const void* K;
    const void* Vt;
    const void* K2;
    const void* Vt2;
    void* O;
    int32_t B, H, Lq;
    int32_t Lk, Lk_pad;
    int32_t Lk2, Lk2_pad;
    int32_t ldq, ldk, ldvt, ldk2, ldvt2, ldo;
    float scale;
    float scale2;
    const float* scale2_tab; /* optional: scale2 = scale2_tab[*step] (per-step IP-scale gating,
                                custom_pipelines.py:326-329, without re-recording the plan) */
    const int32_t* step;
    int32_t dtype;
    const void* pf_ptr;    /* tail prefetch of the next launch's weights (cache hint) */
    uint32_t pf_bytes;
} imh_attn_args;

int imh_attention(const imh_attn_args* a, void* stream);

/* ---- fused QKV + image-prompt cross-attention ---------------------------------------------
 * The attention side of IPAttnProcessor2_0.__call__ (ip_adapter/attention_processor.py:396-450) in one launch:
 *   q = attn.to_q(LN(x))                      (:396; the BasicTransformerBlock.norm2 in front of it folded in, optional)
 *   O = softmax(q K^T * scale) V  (+ scale2 * softmax(q K2^T * scale) V2)          (:416-425, :432-442, :450)
 * K / Vt (text tokens, attn.to_k / to_v, :410-411) and K2 / Vt2 (image-prompt tokens, to_k_ip / to_v_ip, :432-433)
 * are step-invariant caches in imh_attention's layouts, EXCEPT that K and K2 store the 64 dims of every head with
 * each 16-group ordered [0-3, 8-11, 4-7, 12-15] (write them with IMH_GF_VT_PERM): the projected query leaves the
 * MFMA accumulators in exactly that order and becomes the QK^T operand without touching LDS or memory.
 * attn.to_out (:453) needs all heads of a token and stays a following imh_gemm.
 *   X  : [B*Lq, ldx] token rows, C = H*64 columns; un-normalised when ln_s != NULL
 *   Wq : [H*64, ldw] = attn.to_q.weight ([out, in]); with ln_s != NULL pre-scaled by the LayerNorm gamma,
 *        ln_s[d] = sum_k gamma_k Wq[d,k], ln_c[d] = sum_k beta_k Wq[d,k] (fp32) as for IMH_GF_LN_ROW
 */
typedef struct imh_xattn_args {
    const void* X;
    const void* Wq;
    const float* ln_s;
    const float* ln_c;
    float ln_eps;
    const float* ln_stats;   /* with ln_s: REQUIRED row statistics of X ([B*Lq][ln_slots][2], see imh_gemm_args) */
    int32_t ln_slots;
    const void* K;
    const void* Vt;
    const void* K2;
    const void* Vt2;
    void* O;
    int32_t B, H, Lq, C;
    int32_t Lk, Lk_pad;
    int32_t Lk2, Lk2_pad;
    int32_t ldx, ldw, ldk, ldvt, ldk2, ldvt2, ldo;
    float scale;
    float scale2;
    const float* scale2_tab; /* optional per-step table of IP scales (custom_pipelines.py:326-329), indexed by *step */
    const int32_t* step;
    int32_t dtype;
    const void* pf_ptr;      /* tail prefetch of the next launch's weights (cache hint) */
    uint32_t pf_bytes;
} imh_xattn_args;

int imh_cross_attention(const imh_xattn_args* a, void* stream);

/* small generic attention (arbitrary head dims, short sequences), row-major Q/K/V, one softmax:
 * HarmonyAttention's Cross_Attention (ip_adapter/attention_processor.py:35-56, head_dim 40 / value_dim 64)
 * and the Resampler's PerceiverAttention core (ip_adapter/resampler.py:66-76).  Once per image. */
typedef struct imh_small_attn_args {
    const void* Q;
    const void* K;
    const void* V;
    void* O;
    int32_t B, H, Lq, Lk, dq, dv;
    int32_t ldq, ldk, ldv, ldo;
    float scale;
    int32_t dtype;
} imh_small_attn_args;

int imh_attention_small(const imh_small_attn_args* a, void* stream);

/* ---- normalisation ----------------------------------------------------------------------
 * imh_groupnorm: GroupNorm(groups) over NHWC x[B, HW, C] with optional fused SiLU
 *   (diffusers ResnetBlock2D.norm1/norm2 + nonlinearity, Transformer2DModel.norm, conv_norm_out), in three steps that can run
 *   together or apart (torch.nn.GroupNorm semantics: fp32 statistics, biased variance; Welford / Chan merges, never
 *   E[x^2] - mean^2):
 *     statistics  (sum, M2) partials per (sample, pixel block, sub-run of `sub` consecutive channels) -- from a pass over x
 *                 (IMH_GN_STATS) or from the epilogue of the launch that wrote x (imh_gemm_args.gn_out, sub = 10)
 *     table       (scale, shift)[b][c] = (gamma[c] * rstd, beta[c] - mean * gamma[c] * rstd) from the partials of ONE or TWO producers
 *                 (the second covers channels [C1, C): the other half of a channel concat)                      (IMH_GN_TABLE)
 *     apply       y = silu?(x * scale + shift) as a pass (IMH_GN_APPLY) -- or inside the consuming conv3x3 (imh_gemm_args.gn_tab)
 * imh_layernorm: LayerNorm over the last dim of x[rows, C] (BasicTransformerBlock.norm1/2/3,
 *   ip_adapter.py:39, resampler.py:15,42,43,104, train.py:238).  gamma/beta may be NULL.
 */
enum imh_gn_mode {
    IMH_GN_ALL = 0,     /* statistics + table + apply; `partial` = workspace of imh_groupnorm_workspace_bytes() */
    IMH_GN_STATS = 1,   /* x -> partial[B][imh_groupnorm_stats_blocks(HW, C)][C / sub][2]; sub must divide C / groups of every consumer */
    IMH_GN_TABLE = 2,   /* partial (+ partial2) -> table[B][C][2] */
    IMH_GN_APPLY = 3,   /* x, table -> y */
    IMH_GN_TABLE_APPLY = 4   /* partial (+ partial2), x -> y in ONE launch: every workgroup builds its sample's table in LDS (ABI 8) */
};
typedef struct imh_norm_args {
    const void* x;
    void* y;
    const void* gamma;
    const void* beta;
    float* partial;
    int32_t B, HW, C, groups;
    int32_t rows;
    float eps;
    int32_t silu;
    int32_t dtype;
    /* imh_groupnorm only */
    int32_t mode;            /* enum imh_gn_mode */
    float* table;
    const float* partial2;   /* IMH_GN_TABLE: second producer's partials or NULL */
    int32_t nblk, sub, npart;    /* source 1: partial blocks per sample, channels per sub-run, elements per partial (0 = the ragged
                                  * blocks of IMH_GN_STATS: (pixels of block k) * sub with ceil(HW / nblk) pixels per block) */
    int32_t C1;                  /* channels covered by source 1 (ignored without partial2) */
    int32_t nblk2, sub2, npart2;
    const void* pf_ptr;    /* tail prefetch of the next launch's weights (cache hint) */
    uint32_t pf_bytes;
} imh_norm_args;

int imh_groupnorm(const imh_norm_args* a, void* stream);
size_t imh_groupnorm_workspace_bytes(int B, int HW, int C, int groups);
int imh_groupnorm_stats_blocks(int HW, int C);       /* pixel blocks per sample of IMH_GN_STATS */
int imh_groupnorm_stats_sub(int C, int groups);      /* the sub-run width IMH_GN_ALL uses: 10 when it divides C / groups, else C / groups */
int imh_layernorm(const imh_norm_args* a, void* stream);

/* ---- small fused elementwise kernels (see csrc/elementwise.hip for the field meaning) ---- */
enum imh_ew_op {
    IMH_EW_TIMESTEP = 0,  /* diffusers Timesteps() sinusoid */
    IMH_EW_SILU = 1,
    IMH_EW_CONCAT = 2,    /* NHWC channel concat (up-block skips) */
    IMH_EW_CONV_IN = 3,   /* conv_in + CFG duplication (custom_pipelines.py:332) + scale_model_input (:334) */
    IMH_EW_CFG_STEP = 4,  /* CFG combine (:348-350) + scheduler.step (:357) */
    IMH_EW_CAST_F32 = 5,
    IMH_EW_ADD = 6,
    IMH_EW_STEP_SET = 7,  /* *y(int32) = i1 ? i0 : *y + 1 : the device-side step counter */
    IMH_EW_CFG_RESCALE = 8, /* y[s] = f3 * std(eps_text_s) / std(eps_cfg_s) + 1 - f3 (rescale_noise_cfg, custom_pipelines.py:351-354);
                             * a = noise prediction NHWC [2 i0, i1, 4], f2 = guidance scale; IMH_EW_CFG_STEP reads y through `w` */
    IMH_EW_SOFTMAX = 9,     /* y[r,:] (T) = softmax(f0 * a[r,:]) with a fp32 (VAE mid-block attention); i0 rows, i1 cols, i2 / i3 leading dims */
    IMH_EW_ROW_STATS = 10,  /* y[r] (fp32 pair) = (sum, M2) of a[r, 0:i0] (row stride i1), n rows: LayerNorm statistics in the ln_stats format, one slot */
    IMH_EW_STEP_ROW = 11    /* y[0:n] = a[*step * n + 0:n] (T; n % 8 == 0): row `step` of a per-schedule table (time embeddings of all denoise steps) */
};

typedef struct imh_ew_args {
    const void* a;
    const void* b;
    void* y;
    const void* w;
    const void* bias;
    const float* tab;      /* optional per-step scalar table, indexed by *step */
    const int32_t* step;   /* device-resident denoise-step counter */
    int64_t n;
    int32_t i0, i1, i2, i3, i4, i5;
    float f0, f1, f2, f3;
    int32_t dtype;
} imh_ew_args;

int imh_elementwise(int op, const imh_ew_args* a, void* stream);

/* ---- fp32 (reference-precision) kernels for the VAE decode tail -------------------------------
 * ip_adapter/custom_pipelines.py:365-377 upcasts the SDXL VAE to fp32 before `vae.decode` (it overflows in fp16): this entry keeps
 * fp32 activations, fp32 weights and fp32 arithmetic (v_mfma_f32_32x32x2_f32: exact products, fp32 accumulate).  All pointers fp32.
 *   IMH_F32_GEMM     Y[M, N] = X[M, K] W[N, K]^T (+ bias[n]) (+ residual[m, n]); K % 16 == 0.  conv == 1: 3x3, padding 1, stride 1,
 *                    optional nearest x2 upsampling of the input (up = 1), NHWC input [B, H, Wd, Cin], weights [Cout][ky][kx][Cin],
 *                    K = 9 Cin, Cin % 16 == 0, M = B Ho Wo.  Replaces diffusers AutoencoderKL's Conv2d / Linear / Upsample2D.
 *   IMH_F32_GN_STATS X [B, HW, C] -> ws [B, nblk, groups, 2] = (mean, M2) per pixel block and group   (diffusers GroupNorm(32, eps 1e-6))
 *   IMH_F32_GN_TABLE ws, gamma, beta -> Y [B, C, 2] = (gamma rstd, beta - mean gamma rstd), merged in double in a fixed order
 *   IMH_F32_GN_APPLY Y = silu?(X scale + shift), ws = the table
 *   IMH_F32_SOFTMAX  Y[r, 0:N] = softmax(scale X[r, 0:N]), M rows (the mid-block attention's materialised scores)
 */
enum imh_f32_op { IMH_F32_GEMM = 0, IMH_F32_GN_STATS = 1, IMH_F32_GN_TABLE = 2, IMH_F32_GN_APPLY = 3, IMH_F32_SOFTMAX = 4 };

typedef struct imh_f32_args {
    const float* X;
    const float* W;
    float* Y;
    const float* bias;
    const float* residual;
    const float* gamma;
    const float* beta;
    float* ws;
    int32_t M, N, K, ldx, ldw, ldy, ldr;
    int32_t conv, H, Wd, Cin, Ho, Wo, up;
    int32_t B, HW, C, groups, nblk, silu;
    float eps, scale;
} imh_f32_args;

int imh_f32(int op, const imh_f32_args* a, void* stream);

/* ---- plans: a recorded sequence of the calls above, replayed from C++ (one UNet forward is
 * ~1000 launches; Python would be the bottleneck) and optionally captured into a hipGraph. ---- */
enum imh_op_kind { IMH_OP_GEMM = 0, IMH_OP_ATTN = 1, IMH_OP_GROUPNORM = 2, IMH_OP_LAYERNORM = 3, IMH_OP_EW = 4,
                   IMH_OP_ATTN_SMALL = 5, IMH_OP_GEMM_DUAL = 6, IMH_OP_XATTN = 7 };

typedef struct imh_plan imh_plan;

imh_plan* imh_plan_create(void);
void imh_plan_destroy(imh_plan* p);
/* args points to the matching *_args struct (copied); ew_op only for IMH_OP_EW; tag is a small
 * caller-defined integer carried for profiling (e.g. which layer family). Returns op index or <0. */
int imh_plan_add(imh_plan* p, int kind, const void* args, int ew_op, int tag);
int imh_plan_size(const imh_plan* p);
/* in-place update of one recorded op's argument struct (per-step scalars such as the scheduler
 * coefficients); invalidates a captured graph. */
int imh_plan_update(imh_plan* p, int index, const void* args);
int imh_plan_run(imh_plan* p, void* stream);
/* run ops [first, last) */
int imh_plan_run_range(imh_plan* p, int first, int last, void* stream);
/* capture the whole plan into a hipGraph on `stream` (which must be a non-default stream) */
int imh_plan_capture(imh_plan* p, void* stream);
int imh_plan_replay(imh_plan* p, void* stream);
/* run once with a hipEvent pair around every op on `stream`; ms[i] receives op i's duration.
 * Synchronises the stream (measurement helper, not capturable). */
int imh_plan_time_ops(imh_plan* p, void* stream, float* ms, int n);
int imh_plan_get_tag(const imh_plan* p, int index);
int imh_plan_get_kind(const imh_plan* p, int index);

/* tuning / debugging knobs -- key 0: retired (attention workgroups are always 4 waves; accepted and ignored);
 * key 2: XCD tile placement (0 auto, 1 legacy row-major, 2..5 force the (8,1) (4,2) (2,4) (1,8) partition);
 * keys 3 / 4: cross- / self-attention kernel selection (3: 10 = the wide five-head form, 1 = one head per workgroup, 0 = by shape),
 * key 5: LDS-halo conv form (0 auto: conv_hws.hip for the 160-cout forms; 6: conv_halo.hip's lock-step kernels; 8: the K-split form
 * with the service waves transforming the whole halo), key 6: residual rows fetched before / after the K loop, key 7: ff.net.0's prefetch
 * of the next launch's weights inside (1) / behind (0) its K loop, key 9: the 256 x 320 ff.net.0 tile on eight (1) / sixteen (0) waves,
 * key 1: query -- 1 if the library was built with -DIMH_EXPERIMENTAL,
 * A/B and test use only: the values are process-wide plain ints read at launch time, not meant to change while another
 * thread is launching */
int imh_debug_set(int key, int value);

const char* imh_last_error(void);
int imh_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* IMH_H_ */
