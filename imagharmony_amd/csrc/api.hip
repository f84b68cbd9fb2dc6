// C ABI of libimh_hip.so (include/imh.h): argument validation, dispatch to the kernel
// launchers, and the plan executor (recorded launch sequences + hipGraph capture).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../include/imh.h"
#include "imh_common.h"
#include "imh_kernels.h"

namespace imh {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int experimental_refused(const char* what) {
    set_error("%s is compiled only with -DIMH_EXPERIMENTAL (IMH_EXPERIMENTAL=1 python -m imagharmony_amd.build): measured, not selected by any "
              "tuning.json entry or default mode", what);
    return IMH_ERR_ARG;
}
int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return IMH_ERR_LAUNCH;
    }
    return IMH_OK;
}

static GemmParams to_gemm(const imh_gemm_args* a) {
    GemmParams p;
    p.X = a->X; p.W = a->W; p.Y = a->Y; p.partial = a->partial; p.bias = a->bias; p.rowadd = a->rowadd;
    p.residual = a->residual; p.ln_s = a->ln_s; p.ln_c = a->ln_c; p.ln_eps = a->ln_eps;
    p.ln_stats = a->ln_stats; p.ln_stats_out = a->ln_stats_out; p.ln_slots = a->ln_slots; p.ln_slots_out = a->ln_slots_out;
    p.gn_out = a->gn_out; p.gn_nblk = a->gn_nblk; p.gn_hw = a->gn_hw;
    p.gn_tab = a->gn_tab; p.gn_silu = a->gn_silu; p.X2 = a->X2; p.Cin1 = a->X2 ? a->Cin1 : a->Cin;
    p.gn_src.partial = a->gn_part; p.gn_src.partial2 = a->gn_part2; p.gn_src.gamma = a->gn_gamma; p.gn_src.beta = a->gn_beta;
    p.gn_src.eps = a->gn_eps; p.gn_src.groups = a->gn_groups; p.gn_src.C = a->Cin; p.gn_src.C1 = a->gn_part2 ? a->gn_pC1 : a->Cin;
    p.gn_src.HW = a->H * a->Wd; p.gn_src.nblk = a->gn_pnblk; p.gn_src.sub = a->gn_psub; p.gn_src.npart = a->gn_pnpart;
    p.gn_src.nblk2 = a->gn_pnblk2; p.gn_src.sub2 = a->gn_psub2; p.gn_src.npart2 = a->gn_pnpart2; p.gn_src.dtype_f16 = a->dtype == IMH_DT_F16;
    p.Yt = a->Yt; p.yt_col0 = a->yt_col0; p.ldyt = a->ldyt;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.ldx = a->ldx; p.ldw = a->ldw; p.ldy = a->ldy; p.ldr = a->ldr; p.ldra = a->ldra > 0 ? a->ldra : a->N;
    p.rows_per_batch = a->rows_per_batch; p.splits = a->splits; p.flags = a->flags;
    p.H = a->H; p.Wd = a->Wd; p.Cin = a->Cin; p.Ho = a->Ho; p.Wo = a->Wo; p.stride = a->stride; p.up = a->up;
    p.px = p.py = 1; p.tmx = p.tny = 0; p.xcd = a->xcd;
    p.pf_ptr = a->pf_ptr; p.pf_bytes = a->pf_bytes;
    p.early_res = g_ws_early;
    return p;
}

static int do_gemm_dual(const imh_gemm_args* a, const imh_gemm_args* b, hipStream_t s) {
    if (!a || !b || !a->X || !a->W || !a->Y || !b->X || !b->W || !b->Y) { set_error("gemm_dual: null pointer argument"); return IMH_ERR_ARG; }
    if (a->conv || b->conv || a->dtype != b->dtype) { set_error("gemm_dual: both problems must be plain GEMMs of one dtype"); return IMH_ERR_ARG; }
    int bm = a->bm, bn = a->bn;
    if (bm != 24128 && (bm <= 0 || bn <= 0 || bm > 128)) { bm = 128; bn = 64; }
    for (const imh_gemm_args* g : {a, b}) {
        if ((g->flags & (IMH_GF_LN_ROW | IMH_GF_LN_COL)) && (!g->ln_s || !g->ln_c || !(g->ln_eps > 0.f))) {
            set_error("gemm_dual: folded LayerNorm needs ln_s / ln_c / ln_eps > 0"); return IMH_ERR_ARG;
        }
        if (g->ln_stats_out || g->gn_out) { set_error("gemm_dual: no statistics epilogue"); return IMH_ERR_ARG; }
        if (g->ln_stats && (g->ln_slots <= 0 || g->K % g->ln_slots)) { set_error("gemm_dual: ln_slots=%d must divide K=%d", g->ln_slots, g->K); return IMH_ERR_ARG; }
    }
    if ((a->ln_stats == nullptr) != (b->ln_stats == nullptr) && ((a->flags | b->flags) & (IMH_GF_LN_ROW | IMH_GF_LN_COL))) {
        set_error("gemm_dual: both problems take their LayerNorm statistics the same way"); return IMH_ERR_ARG;
    }
    return gemm_dual_launch(to_gemm(a), to_gemm(b), a->dtype, bm, bn, s);
}

static int do_gemm(const imh_gemm_args* a, hipStream_t s) {
    if (!a || !a->X || !a->W || !a->Y) { set_error("gemm: null pointer argument"); return IMH_ERR_ARG; }
    GemmParams p = to_gemm(a);
    if ((p.flags & (IMH_GF_LN_ROW | IMH_GF_LN_COL)) && (!p.ln_s || !p.ln_c || !(p.ln_eps > 0.f))) { set_error("gemm: folded LayerNorm needs ln_s / ln_c / ln_eps > 0"); return IMH_ERR_ARG; }
    if ((p.flags & IMH_GF_LN_ROW) && (p.flags & IMH_GF_LN_COL)) { set_error("gemm: IMH_GF_LN_ROW and IMH_GF_LN_COL are exclusive"); return IMH_ERR_ARG; }
    if (!(p.flags & (IMH_GF_LN_ROW | IMH_GF_LN_COL))) p.ln_stats = nullptr;
    if (p.ln_stats && (p.ln_slots <= 0 || p.K % p.ln_slots)) { set_error("gemm: ln_slots=%d must divide K=%d", p.ln_slots, p.K); return IMH_ERR_ARG; }
    int bm = a->bm, bn = a->bn;
    if (bm <= 0 || bn <= 0 || p.splits <= 0) {
        int hb, hn, hs;
        gemm_pick_config(p.M, p.N, p.K, &hb, &hn, &hs);
        if (bm <= 0) bm = hb;
        if (bn <= 0) bn = hn;
        if (p.splits <= 0) p.splits = p.partial ? hs : 1;
    }
    if ((p.flags & IMH_GF_VT_PERM) && p.splits == 1) bn = 128;   // the permutation lives in 16-column groups
    if (a->conv) {
        if (p.stride != 1 && p.stride != 2) { set_error("conv3x3: stride must be 1 or 2"); return IMH_ERR_ARG; }
        if (p.up != 0 && p.up != 1) { set_error("conv3x3: up must be 0 or 1"); return IMH_ERR_ARG; }
        const int Hv = p.H << p.up, Wv = p.Wd << p.up;
        if (p.Ho != (Hv + 2 - 3) / p.stride + 1 || p.Wo != (Wv + 2 - 3) / p.stride + 1) {
            set_error("conv3x3: output size %dx%d inconsistent with input %dx%d up=%d stride=%d", p.Ho, p.Wo, p.H, p.Wd, p.up, p.stride);
            return IMH_ERR_SHAPE;
        }
        if (p.M % (p.Ho * p.Wo) != 0) { set_error("conv3x3: M must be B*Ho*Wo"); return IMH_ERR_SHAPE; }
    }
    return gemm_launch(p, a->dtype, a->conv, bm, bn, s);
}

static int do_attn(const imh_attn_args* a, hipStream_t s) {
    if (!a || !a->Q || !a->K || !a->Vt || !a->O) { set_error("attention: null pointer argument"); return IMH_ERR_ARG; }
    if ((a->K2 == nullptr) != (a->Vt2 == nullptr)) { set_error("attention: K2 and Vt2 must come together"); return IMH_ERR_ARG; }
    AttnParams p;
    p.Q = a->Q; p.K = a->K; p.Vt = a->Vt; p.K2 = a->K2; p.Vt2 = a->Vt2; p.O = a->O;
    p.B = a->B; p.H = a->H; p.Lq = a->Lq; p.Lk = a->Lk; p.Lk_pad = a->Lk_pad; p.Lk2 = a->Lk2; p.Lk2_pad = a->Lk2_pad;
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldvt = a->ldvt; p.ldk2 = a->ldk2; p.ldvt2 = a->ldvt2; p.ldo = a->ldo;
    p.scale = a->scale; p.scale2 = a->scale2; p.scale2_tab = a->scale2_tab; p.step = a->step;
    p.pf_ptr = a->pf_ptr; p.pf_bytes = a->pf_bytes; p.split = 0; p.defer_log2 = 0.f;
    return attention_launch(p, a->dtype, s);
}

static int do_xattn(const imh_xattn_args* a, hipStream_t s) {
    if (!a || !a->X || !a->Wq || !a->K || !a->Vt || !a->O) { set_error("cross_attention: null pointer argument"); return IMH_ERR_ARG; }
    if ((a->K2 == nullptr) != (a->Vt2 == nullptr)) { set_error("cross_attention: K2 and Vt2 must come together"); return IMH_ERR_ARG; }
    if ((a->ln_s == nullptr) != (a->ln_c == nullptr) || (a->ln_s && !(a->ln_eps > 0.f))) {
        set_error("cross_attention: folded LayerNorm needs ln_s, ln_c and ln_eps > 0"); return IMH_ERR_ARG;
    }
    if (a->C != a->H * 64) { set_error("cross_attention: C=%d must be H*64 (H=%d)", a->C, a->H); return IMH_ERR_SHAPE; }
    XAttnParams x;
    AttnParams& p = x.a;
    p.Q = nullptr; p.K = a->K; p.Vt = a->Vt; p.K2 = a->K2; p.Vt2 = a->Vt2; p.O = a->O;
    p.B = a->B; p.H = a->H; p.Lq = a->Lq; p.Lk = a->Lk; p.Lk_pad = a->Lk_pad; p.Lk2 = a->Lk2; p.Lk2_pad = a->Lk2_pad;
    p.ldq = 0; p.ldk = a->ldk; p.ldvt = a->ldvt; p.ldk2 = a->ldk2; p.ldvt2 = a->ldvt2; p.ldo = a->ldo;
    p.scale = a->scale; p.scale2 = a->scale2; p.scale2_tab = a->scale2_tab; p.step = a->step;
    p.pf_ptr = a->pf_ptr; p.pf_bytes = a->pf_bytes;
    x.X = a->X; x.Wq = a->Wq; x.ln_s = a->ln_s; x.ln_c = a->ln_c; x.ln_eps = a->ln_eps;
    x.ln_stats = a->ln_s ? a->ln_stats : nullptr; x.ln_slots = a->ln_slots;
    if (x.ln_stats && (a->ln_slots <= 0 || a->C % a->ln_slots)) { set_error("cross_attention: ln_slots=%d must divide C=%d", a->ln_slots, a->C); return IMH_ERR_ARG; }
    x.C = a->C; x.ldx = a->ldx; x.ldw = a->ldw; x.split = 0;
    return xattn_launch(x, a->dtype, s);
}

static int do_attn_small(const imh_small_attn_args* a, hipStream_t s) {
    if (!a || !a->Q || !a->K || !a->V || !a->O) { set_error("attention_small: null pointer argument"); return IMH_ERR_ARG; }
    SmallAttnParams p;
    p.Q = a->Q; p.K = a->K; p.V = a->V; p.O = a->O;
    p.B = a->B; p.H = a->H; p.Lq = a->Lq; p.Lk = a->Lk; p.dq = a->dq; p.dv = a->dv;
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo; p.scale = a->scale;
    return attention_small_launch(p, a->dtype, s);
}

static NormParams to_norm(const imh_norm_args* a) {
    NormParams p;
    p.x = a->x; p.y = a->y; p.gamma = a->gamma; p.beta = a->beta; p.partial = a->partial;
    p.B = a->B; p.HW = a->HW; p.C = a->C; p.groups = a->groups; p.rows = a->rows; p.eps = a->eps; p.silu = a->silu;
    p.mode = a->mode; p.table = a->table; p.partial2 = a->partial2;
    p.nblk = a->nblk; p.sub = a->sub; p.npart = a->npart; p.C1 = a->C1; p.nblk2 = a->nblk2; p.sub2 = a->sub2; p.npart2 = a->npart2;
    p.dtype_f16 = 0;
    p.pf_ptr = a->pf_ptr; p.pf_bytes = a->pf_bytes;
    return p;
}

static int do_ew(int op, const imh_ew_args* a, hipStream_t s) {
    if (!a || !a->y) { set_error("elementwise: null pointer argument"); return IMH_ERR_ARG; }
    EwParams p;
    p.a = a->a; p.b = a->b; p.y = a->y; p.w = a->w; p.bias = a->bias; p.tab = a->tab; p.step = a->step; p.n = a->n;
    p.i0 = a->i0; p.i1 = a->i1; p.i2 = a->i2; p.i3 = a->i3; p.i4 = a->i4; p.i5 = a->i5;
    p.f0 = a->f0; p.f1 = a->f1; p.f2 = a->f2; p.f3 = a->f3;
    return ew_launch(op, p, a->dtype, s);
}

}  // namespace imh

using namespace imh;

struct imh_op {
    int kind;
    int ew_op;
    int tag;
    union {
        imh_gemm_args gemm;
        imh_attn_args attn;
        imh_norm_args norm;
        imh_ew_args ew;
        imh_small_attn_args sattn;
        imh_gemm_args gemm2[2];
        imh_xattn_args xattn;
    } u;
};

struct imh_plan {
    std::vector<imh_op> ops;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

static int run_op(const imh_op& o, hipStream_t s) {
    switch (o.kind) {
        case IMH_OP_GEMM: return do_gemm(&o.u.gemm, s);
        case IMH_OP_ATTN: return do_attn(&o.u.attn, s);
        case IMH_OP_GROUPNORM: return groupnorm_launch(to_norm(&o.u.norm), o.u.norm.dtype, s);
        case IMH_OP_LAYERNORM: {
            if (!o.u.norm.x || !o.u.norm.y) { set_error("layernorm: null pointer argument"); return IMH_ERR_ARG; }
            return layernorm_launch(to_norm(&o.u.norm), o.u.norm.dtype, s);
        }
        case IMH_OP_EW: return do_ew(o.ew_op, &o.u.ew, s);
        case IMH_OP_ATTN_SMALL: return do_attn_small(&o.u.sattn, s);
        case IMH_OP_GEMM_DUAL: return do_gemm_dual(&o.u.gemm2[0], &o.u.gemm2[1], s);
        case IMH_OP_XATTN: return do_xattn(&o.u.xattn, s);
    }
    set_error("plan: unknown op kind %d", o.kind);
    return IMH_ERR_ARG;
}

static void drop_graph(imh_plan* p) {
    if (p->exec) { hipGraphExecDestroy(p->exec); p->exec = nullptr; }
    if (p->graph) { hipGraphDestroy(p->graph); p->graph = nullptr; }
}

static size_t args_size(int kind) {
    switch (kind) {
        case IMH_OP_GEMM: return sizeof(imh_gemm_args);
        case IMH_OP_ATTN: return sizeof(imh_attn_args);
        case IMH_OP_GROUPNORM:
        case IMH_OP_LAYERNORM: return sizeof(imh_norm_args);
        case IMH_OP_EW: return sizeof(imh_ew_args);
        case IMH_OP_ATTN_SMALL: return sizeof(imh_small_attn_args);
        case IMH_OP_GEMM_DUAL: return 2 * sizeof(imh_gemm_args);
        case IMH_OP_XATTN: return sizeof(imh_xattn_args);
    }
    return 0;
}

extern "C" {

int imh_abi_version(void) { return IMH_ABI_VERSION; }
int imh_debug_set(int key, int value) {
    if (key == 0) { g_attn_force_nw = value; return IMH_OK; }
    if (key == 1) {                               // query: was this library built with -DIMH_EXPERIMENTAL?
#ifdef IMH_EXPERIMENTAL
        return 1;
#else
        return 0;
#endif
    }
    if (key == 2) { g_xcd_mode = value; return IMH_OK; }
    if (key == 3) { g_xattn_mode = value; return IMH_OK; }
    if (key == 4) { g_attn_mode = value; return IMH_OK; }
    if (key == 5) { g_halo_mode = value; return IMH_OK; }
    if (key == 7) { g_w16_pf = value; return IMH_OK; }        // sixteen-wave ff.net.0 tile: 1 (default) = the next launch's weights prefetched inside the K loop, 0 = behind the epilogue
    if (key == 9) { g_w16_form = value; return IMH_OK; }      // the 256 x 320 ff.net.0 tile: 0 = sixteen waves of 64 x 80, 1 = eight waves of 128 x 80
    if (key == 10) { g_f32_exact = value; return IMH_OK; }    // imh_f32 GEMM / conv: 1 = the exact fp32 MFMA kernel for every launch, 0 (default) = bf16 hi / lo split where the K tile fits
    if (key == 6) { g_ws_early = value; return IMH_OK; }      // 0: the residual rows of the wave-specialised launches fetched after the K loop (A/B)
    set_error("debug_set: unknown key %d", key);
    return IMH_ERR_ARG;
}
const char* imh_last_error(void) { return g_err; }

int imh_gemm(const imh_gemm_args* a, void* stream) { return do_gemm(a, (hipStream_t)stream); }
int imh_gemm_dual(const imh_gemm_args* a, const imh_gemm_args* b, void* stream) { return do_gemm_dual(a, b, (hipStream_t)stream); }
int imh_gemm_pick_config(int M, int N, int K, int* bm, int* bn, int* splits) {
    if (!bm || !bn || !splits) { set_error("pick_config: null output"); return IMH_ERR_ARG; }
    gemm_pick_config(M, N, K, bm, bn, splits);
    return IMH_OK;
}
size_t imh_gemm_workspace_bytes(int M, int N, int splits) { return gemm_workspace_bytes(M, N, splits); }
int imh_gemm_stats_slot_width(int bm, int bn) { return gemm_stats_slot_width(bm, bn); }
int imh_gemm_gn_block_rows(int bm, int bn) { return gemm_gn_block_rows(bm, bn); }

int imh_attention(const imh_attn_args* a, void* stream) { return do_attn(a, (hipStream_t)stream); }

int imh_cross_attention(const imh_xattn_args* a, void* stream) { return do_xattn(a, (hipStream_t)stream); }

int imh_attention_small(const imh_small_attn_args* a, void* stream) { return do_attn_small(a, (hipStream_t)stream); }

int imh_groupnorm(const imh_norm_args* a, void* stream) {
    if (!a) { set_error("groupnorm: null pointer argument"); return IMH_ERR_ARG; }
    return groupnorm_launch(to_norm(a), a->dtype, (hipStream_t)stream);
}
size_t imh_groupnorm_workspace_bytes(int B, int HW, int C, int groups) { return groupnorm_workspace_bytes(B, HW, C, groups); }
int imh_groupnorm_stats_blocks(int HW, int C) { return groupnorm_stats_blocks(HW, C); }
int imh_groupnorm_stats_sub(int C, int groups) { return groups > 0 && C % groups == 0 ? groupnorm_stats_sub(C, groups) : 0; }
int imh_layernorm(const imh_norm_args* a, void* stream) {
    if (!a || !a->x || !a->y) { set_error("layernorm: null pointer argument"); return IMH_ERR_ARG; }
    return layernorm_launch(to_norm(a), a->dtype, (hipStream_t)stream);
}

int imh_elementwise(int op, const imh_ew_args* a, void* stream) { return do_ew(op, a, (hipStream_t)stream); }

int imh_f32(int op, const imh_f32_args* a, void* stream) {
    if (!a) { set_error("imh_f32: null args"); return IMH_ERR_ARG; }
    F32Params p;
    p.X = a->X; p.W = a->W; p.Y = a->Y; p.bias = a->bias; p.residual = a->residual; p.gamma = a->gamma; p.beta = a->beta; p.ws = a->ws;
    p.M = a->M; p.N = a->N; p.K = a->K; p.ldx = a->ldx; p.ldw = a->ldw; p.ldy = a->ldy; p.ldr = a->ldr;
    p.conv = a->conv; p.H = a->H; p.Wd = a->Wd; p.Cin = a->Cin; p.Ho = a->Ho; p.Wo = a->Wo; p.up = a->up;
    p.B = a->B; p.HW = a->HW; p.C = a->C; p.groups = a->groups; p.nblk = a->nblk; p.silu = a->silu;
    p.eps = a->eps; p.scale = a->scale;
    return f32_launch(op, p, (hipStream_t)stream);
}

imh_plan* imh_plan_create(void) { return new (std::nothrow) imh_plan(); }
void imh_plan_destroy(imh_plan* p) {
    if (!p) return;
    drop_graph(p);
    delete p;
}
int imh_plan_add(imh_plan* p, int kind, const void* args, int ew_op, int tag) {
    if (!p || !args) { set_error("plan_add: null argument"); return IMH_ERR_ARG; }
    const size_t sz = args_size(kind);
    if (!sz) { set_error("plan_add: unknown kind %d", kind); return IMH_ERR_ARG; }
    imh_op o;
    memset(&o, 0, sizeof(o));
    o.kind = kind; o.ew_op = ew_op; o.tag = tag;
    memcpy(&o.u, args, sz);
    p->ops.push_back(o);
    drop_graph(p);
    return (int)p->ops.size() - 1;
}
int imh_plan_size(const imh_plan* p) { return p ? (int)p->ops.size() : 0; }
int imh_plan_update(imh_plan* p, int index, const void* args) {
    if (!p || !args || index < 0 || index >= (int)p->ops.size()) { set_error("plan_update: bad index"); return IMH_ERR_ARG; }
    memcpy(&p->ops[index].u, args, args_size(p->ops[index].kind));
    drop_graph(p);
    return IMH_OK;
}
int imh_plan_run_range(imh_plan* p, int first, int last, void* stream) {
    if (!p || first < 0 || last > (int)p->ops.size() || first > last) { set_error("plan_run: bad range"); return IMH_ERR_ARG; }
    for (int i = first; i < last; ++i) {
        int rc = run_op(p->ops[i], (hipStream_t)stream);
        if (rc != IMH_OK) return rc;
    }
    return IMH_OK;
}
int imh_plan_run(imh_plan* p, void* stream) { return imh_plan_run_range(p, 0, p ? (int)p->ops.size() : 0, stream); }

int imh_plan_capture(imh_plan* p, void* stream) {
    if (!p) { set_error("plan_capture: null plan"); return IMH_ERR_ARG; }
    drop_graph(p);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { set_error("plan_capture: begin: %s", hipGetErrorString(e)); return IMH_ERR_LAUNCH; }
    int rc = imh_plan_run(p, stream);
    e = hipStreamEndCapture(s, &p->graph);
    if (rc != IMH_OK) { drop_graph(p); return rc; }
    if (e != hipSuccess) { set_error("plan_capture: end: %s", hipGetErrorString(e)); drop_graph(p); return IMH_ERR_LAUNCH; }
    e = hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { set_error("plan_capture: instantiate: %s", hipGetErrorString(e)); drop_graph(p); return IMH_ERR_LAUNCH; }
    return IMH_OK;
}
int imh_plan_replay(imh_plan* p, void* stream) {
    if (!p || !p->exec) { set_error("plan_replay: plan not captured"); return IMH_ERR_ARG; }
    hipError_t e = hipGraphLaunch(p->exec, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("plan_replay: %s", hipGetErrorString(e)); return IMH_ERR_LAUNCH; }
    return IMH_OK;
}
int imh_plan_time_ops(imh_plan* p, void* stream, float* ms, int n) {
    if (!p || !ms || n < (int)p->ops.size()) { set_error("plan_time_ops: bad arguments"); return IMH_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const int cnt = (int)p->ops.size();
    std::vector<hipEvent_t> ev(cnt + 1);
    for (auto& e : ev) hipEventCreate(&e);
    int rc = IMH_OK;
    hipEventRecord(ev[0], s);
    for (int i = 0; i < cnt && rc == IMH_OK; ++i) {
        rc = run_op(p->ops[i], s);
        hipEventRecord(ev[i + 1], s);
    }
    hipStreamSynchronize(s);
    if (rc == IMH_OK)
        for (int i = 0; i < cnt; ++i) hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]);
    for (auto& e : ev) hipEventDestroy(e);
    return rc;
}
int imh_plan_get_tag(const imh_plan* p, int index) {
    return (p && index >= 0 && index < (int)p->ops.size()) ? p->ops[index].tag : -1;
}
int imh_plan_get_kind(const imh_plan* p, int index) {
    return (p && index >= 0 && index < (int)p->ops.size()) ? p->ops[index].kind : -1;
}

}  // extern "C"
