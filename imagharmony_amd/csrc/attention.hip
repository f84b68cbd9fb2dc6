// Flash attention for gfx950, head_dim 64: the self-attention of AttnProcessor2_0
// (ip_adapter/attention_processor.py:305-316) and the decoupled text + image-prompt
// cross-attention of IPAttnProcessor2_0 (:416-450):
//
//     O = softmax(Q K^T / 8) V  [ + scale2 * softmax(Q K2^T / 8) V2 ]
//
// Structure (cdna_hip_programming.md Appendix B "8-warp 32x32 ladder", simplified):
// 4 waves x 32 query rows per workgroup, KV tiles of 64 keys double-buffered in LDS by
// global_load_lds (XOR-swizzled on the source side), swapped QK^T (S^T = K Q^T with
// v_mfma_f32_32x32x16) so that every lane owns one query row: the online softmax is
// in-register (one cross-half shuffle per tile), P never touches LDS, and O^T = V^T P^T
// accumulates with the query as the lane-local column.  V arrives TRANSPOSED with keys
// permuted inside 16-groups (imh_layout.h vt_perm16) -- the producing GEMM writes it that
// way -- so the V^T operand is one ds_read_b128 and no LDS transpose is needed.
// The two softmaxes of the IP branch run as two passes over different key sets inside the
// same kernel; Q stays in registers, the combination happens before the single store.
// Roofline: self-attention MFMA-bound; cross-attention (77+T keys) HBM-bound on Q/O.
#include "imh_common.h"
#include "imh_kernels.h"

namespace imh {

int g_attn_force_nw = 0;   // retired tuning knob (imh_debug_set key 0): only the 4-wave workgroup is built

constexpr int ATT_KV = 64;              // keys per LDS tile
constexpr int ATT_TILE_BYTES = 64 * 128;
constexpr int ATT_STAGES = 3;             // K/V^T ring depth: 48 KB -> 3 workgroups per CU (the Q / O staging tile aliases the ring)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;

typedef __attribute__((ext_vector_type(2))) float f32x2;
// three-input max in one VALU op; fmaxf() would add a canonicalising v_max(x, x) per MFMA-produced operand
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// max over the two half-waves (lanes l and l^32) without LDS: v_permlane32_swap leaves {lo,lo} / {hi,hi}
__device__ __forceinline__ float xhalf_max(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
// ATT_ABL (tools/attn_ablate.py only; wrong results by design), bit mask: 1 = no v_exp, 2 = no K/V loads inside the
// loop, 4 = no barrier (with 2), 8 = no max / exponent / sum at all,
// 16 = never wait for the K/V loads, 32 = always load tile 0 (cache-hot)
#ifndef ATT_ABL
#define ATT_ABL 0
#endif
#define ATT_EXP2(x) ((ATT_ABL & 9) ? (x) : __builtin_amdgcn_exp2f(x))
// ATT_TIMING (tools/attn_phase_probe.py only): cycle counter at the phase boundaries of the key loop; work item 0 /
// thread 0 writes the per-phase totals (+ tile count) to p.pf_ptr instead of prefetching
#ifndef ATT_TIMING
#define ATT_TIMING 0
#endif
#if ATT_TIMING
#define ATT_TICK(i) do { asm volatile("s_nop 0" ::: "memory"); const unsigned long long tn_ = __builtin_readcyclecounter(); tacc[i] += tn_ - tm0; tm0 = tn_; } while (0)
#else
#define ATT_TICK(i) do {} while (0)
#endif

// NW waves per workgroup (32 queries each); the launcher uses NW = 4.
// NPASS = 1: single key set (self-attention, text-only cross-attention): no second accumulator, lower register
// pressure -> 3 workgroups per CU instead of 2.  NPASS = 2: text + image-prompt key sets.
template <typename T, int NW, int NPASS>
__global__ __launch_bounds__(64 * NW, (NPASS == 1 && NW == 4) ? 3 : 2) void attn_kernel(const AttnParams p) {
    constexpr int RND = 8 / NW;              // staging rounds: NW*8 rows per round, 64 rows per tile
    typedef typename Vec<T>::v8 v8;
    typedef typename Vec<T>::v4 v4;
    // K / V^T ring, then a wave-private [32 NW][64] Q / O staging tile (coalesced 128-B rows both ways)
    static_assert(32 * NW * 128 <= ATT_STAGES * 2 * ATT_TILE_BYTES, "Q/O staging must fit inside the ring");
    __shared__ __attribute__((aligned(16))) unsigned char smem[ATT_STAGES * 2 * ATT_TILE_BYTES];
    unsigned char* qs = smem;              // used before the first K/V tile is staged and after the last is consumed

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31;
    const int hi = lane >> 5;
    // XCD-aware work-item order (workgroup w runs on XCD w % 8): each XCD takes a CONTIGUOUS range of the
    // head-major item list, so all query blocks of one (batch, head) hit the same 4 MB L2 and K / V^T are fetched
    // from the fabric once instead of once per XCD (FETCH_SIZE 65 MB vs 15 MB algorithmic per launch before).
    const int gx = (p.Lq + 32 * NW - 1) / (32 * NW);
    const int items = gx * p.H * p.B;
    const int per = (items + 7) >> 3;
    const int item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || item >= items) return;
    const int hb = item / gx, qblk = item - hb * gx;
    const int b = hb / p.H, h = hb - b * p.H;
    const int q0 = qblk * (32 * NW);
    const int q = q0 + wave * 32 + l32;

    // Q tile through LDS: each wave DMA-copies its own 32 rows as full 128-B lines (a lane-per-row register load
    // touches 32 different lines per instruction), then reads the Q^T B-operand fragments
    // (lane (col q, half hi) holds d = sd*16 + hi*8 + 0..7) with the K-tile swizzle.
    v8 qf[4];
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 32 + i * 8 + (lane >> 3);
            const int ch = stage_chunk_x(row, lane);
            const T* src = (const T*)p.Q + ((size_t)b * p.Lq + min(q0 + row, p.Lq - 1)) * p.ldq + h * 64 + ch * 8;
            glds16(src, qs + (wave * 32 + i * 8) * 128);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // own rows only: no workgroup barrier needed
        const int row = wave * 32 + l32;
#pragma unroll
        for (int sd = 0; sd < 4; ++sd) qf[sd] = *(const v8*)(qs + tile_off(row, sd * 2 + hi, swz_x(row)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // every wave has its Q fragments: the ring may overwrite the staging rows
    }

    const float c = p.scale * LOG2E;      // > 0


    f32x16 fin[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) fin[dt][r] = 0.f;

#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        const T* Kp = (const T*)(pass == 0 ? p.K : p.K2);
        const T* Vp = (const T*)(pass == 0 ? p.Vt : p.Vt2);
        const int Lk = pass == 0 ? p.Lk : p.Lk2;
        const int Lkp = pass == 0 ? p.Lk_pad : p.Lk2_pad;
        const int ldk = pass == 0 ? p.ldk : p.ldk2;
        const int ldvt = pass == 0 ? p.ldvt : p.ldvt2;
        const float wgt = pass == 0 ? 1.0f : (p.scale2_tab ? p.scale2_tab[*p.step] : p.scale2);
        const int ntiles = (Lk + ATT_KV - 1) / ATT_KV;

        auto stage = [&](int buf, int tile) {
            unsigned char* ks = smem + buf * 2 * ATT_TILE_BYTES;
            unsigned char* vs = ks + ATT_TILE_BYTES;
            const int kbase = tile * ATT_KV;
#pragma unroll
            for (int i = 0; i < RND; ++i) {
                const int row = i * (NW * 8) + wave * 8 + (lane >> 3);
                const int ch = stage_chunk_x(row, lane);
                const T* ksrc = Kp + ((size_t)b * Lkp + kbase + row) * ldk + h * 64 + ch * 8;
                glds16(ksrc, ks + (i * (NW * 8) + wave * 8) * 128);
                const T* vsrc = Vp + ((size_t)h * 64 + row) * ldvt + (size_t)b * Lkp + kbase + ch * 8;
                glds16(vsrc, vs + (i * (NW * 8) + wave * 8) * 128);
            }
        };

        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
        float m_run = NEG_BIG, l_run = 0.f;

        // ring of ATT_STAGES tiles with counted vmcnt: the LDS-DMA queue is never drained inside the loop
        constexpr int LPT = 2 * RND;           // LDS-DMA instructions per thread per tile
#pragma unroll
        for (int s = 0; s < ATT_STAGES - 1; ++s)
            if (s < ntiles) stage(s, s);
        int cur = 0;
#if ATT_TIMING
        unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tm0 = __builtin_readcyclecounter();
#endif
        for (int t = 0; t < ntiles; ++t) {
            if (ATT_ABL & 16) {
            } else if (t + ATT_STAGES - 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ATT_STAGES - 2) * LPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(ATT_ABL & 4)) __builtin_amdgcn_s_barrier();      // tile t landed for every wave; tile t-1 fully consumed
            asm volatile("" ::: "memory");
            ATT_TICK(0);       // vmcnt wait + barrier
            if (!(ATT_ABL & 2) && t + ATT_STAGES - 1 < ntiles) {
                int ns = cur + ATT_STAGES - 1;
                if (ns >= ATT_STAGES) ns -= ATT_STAGES;
                stage(ns, (ATT_ABL & 32) ? 0 : t + ATT_STAGES - 1);
            }
            ATT_TICK(1);       // LDS-DMA issue
            const unsigned char* ks = smem + cur * 2 * ATT_TILE_BYTES;
            const unsigned char* vs = ks + ATT_TILE_BYTES;
            const int kbase = t * ATT_KV;
            const bool ragged = kbase + ATT_KV > Lk;   // only the last tile of a ragged key set needs masking

            // ---- S^T = K Q^T (both 32-key sub-tiles always: padded keys are zero rows, masked below).
            //      MFMA and VALU time add up on a SIMD (tools/attn_ablate.py), so LDS latency is what can be hidden:
            //      all 8 K fragments are requested before the first MFMA, the 8 V^T fragments right after the
            //      QK^T MFMAs so that they land under the softmax ----
            f32x16 st[2];
            {
                v8 kf[2][4];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int sd = 0; sd < 4; ++sd) kf[kt][sd] = *(const v8*)(ks + att_k_off(lane, kt, sd));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
                    for (int sd = 0; sd < 4; ++sd) st[kt] = mfma32(kf[kt][sd], qf[sd], st[kt]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            ATT_TICK(2);       // K fragment reads + QK^T MFMA issue
            v8 vf[2][2][2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) vf[kt][s][dt] = *(const v8*)(vs + att_v_off(lane, dt, kt, s));
            __builtin_amdgcn_sched_barrier(0);
            // ---- online softmax on RAW scores (scale folded into the exponent: p = exp2(s*c - m*c), c > 0);
            //      the row is lane-local, one cross-half exchange per tile; written for v_max3 / v_pk_fma /
            //      v_pk_add (half the VALU instructions of the scalar form) ----
            if (ragged) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kbase + kt * 32 + st_key(r, hi) >= Lk) st[kt][r] = NEG_BIG;
            }
            float mx = max3f(st[0][0], st[1][0], st[0][1]);
            if (!(ATT_ABL & 8)) {
                mx = max3f(mx, st[1][1], st[0][2]);
#pragma unroll
                for (int r = 2; r < 15; ++r) mx = max3f(mx, st[1][r], st[0][r + 1]);
                mx = fmaxf(mx, st[1][15]);
                mx = xhalf_max(mx);
            }
            const float m_new = fmaxf(m_run, mx);
            const f32x2 c2 = {c, c};
            const f32x2 nmc2 = {-m_new * c, -m_new * c};
            f32x2 ps2 = {0.f, 0.f};
            v8 pf[2][2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 s2 = {st[kt][r], st[kt][r + 1]};
                    const f32x2 e2 = (ATT_ABL & 8) ? s2 : __builtin_elementwise_fma(s2, c2, nmc2);
                    const f32x2 p2 = {ATT_EXP2(e2[0]), ATT_EXP2(e2[1])};
                    if (!(ATT_ABL & 8)) ps2 += p2;
                    pf[kt][r >> 3][r & 7] = from_f32<T>(p2[0]);
                    pf[kt][r >> 3][(r & 7) + 1] = from_f32<T>(p2[1]);
                }
            if (__any(m_new != m_run)) {           // wave-uniform: rescale only when some row's max moved
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
                m_run = m_new;
            }
            l_run += ps2[0] + ps2[1];
            // ---- O^T += V^T P^T ----
            __builtin_amdgcn_sched_barrier(0);
            ATT_TICK(3);       // V^T read issue + MFMA drain + softmax
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) o[dt] = mfma32(vf[kt][s][dt], pf[kt][s], o[dt]);
            ATT_TICK(4);       // PV MFMA issue
            asm volatile("" ::: "memory");
            if (++cur == ATT_STAGES) cur = 0;
        }
#if ATT_TIMING
        if (item == 0 && tid == 0 && p.pf_ptr) {
            unsigned long long* dbg = (unsigned long long*)p.pf_ptr;
            for (int i = 0; i < 5; ++i) dbg[i] = tacc[i];
            dbg[5] = ntiles;
        }
#endif
        __builtin_amdgcn_s_barrier();          // the next pass refills the ring from slot 0
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = wgt / l_tot;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) fin[dt][r] += o[dt][r] * inv;
    }

    // ---- store through the wave's staging rows: lane (q, hi) owns d = dt*32 + 8*rg + 4*hi + e, written as 8-B
    //      pieces, read back as full 128-B rows (8 lanes x 16 B) and stored coalesced ----
    {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // Q fragment reads of this wave are done
        const int row = wave * 32 + l32;
        const int sw = swz_x(row);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                v4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = from_f32<T>(fin[dt][rg * 4 + e]);
                *(v4*)(qs + tile_off(row, dt * 4 + rg, sw) + hi * 8) = o4;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // same-wave LDS hand-off (DS ops retire in order)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r2 = wave * 32 + i * 8 + (lane >> 3);
            const int ch = lane & 7;
            const v8 o8 = *(const v8*)(qs + tile_off(r2, ch, swz_x(r2)));
            if (q0 + r2 < p.Lq)
                *(v8*)((T*)p.O + ((size_t)b * p.Lq + q0 + r2) * p.ldo + h * 64 + ch * 8) = o8;
        }
    }
    if (!ATT_TIMING) tail_prefetch(p.pf_ptr, p.pf_bytes, item, items, tid, 64 * NW);
}

// ---- small generic attention (any head dims <= 128, short sequences): one workgroup per (batch, head).
// Used once per image by HarmonyAttention's Cross_Attention (head_dim 40, value_dim 64, 8 queries x 77 keys;
// ip_adapter/attention_processor.py:35-56) and by the Resampler's PerceiverAttention (16 queries x 273 keys,
// fp32 softmax; ip_adapter/resampler.py:66-76).  Latency-bound; not on the per-step path.
template <typename T>
__global__ __launch_bounds__(256) void attn_small_kernel(const SmallAttnParams p) {
    extern __shared__ float sm[];            // [Lk] scores, [dq] query row, [8] reductions
    float* sc = sm;
    float* qrow = sm + p.Lk;
    float* red = qrow + p.dq;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int tid = threadIdx.x;
    const T* Q = (const T*)p.Q + (size_t)b * p.Lq * p.ldq + h * p.dq;
    const T* K = (const T*)p.K + (size_t)b * p.Lk * p.ldk + h * p.dq;
    const T* V = (const T*)p.V + (size_t)b * p.Lk * p.ldv + h * p.dv;
    T* O = (T*)p.O + (size_t)b * p.Lq * p.ldo + h * p.dv;
    for (int i = 0; i < p.Lq; ++i) {
        for (int d = tid; d < p.dq; d += 256) qrow[d] = to_f32(Q[(size_t)i * p.ldq + d]);
        __syncthreads();
        float mx = -1e30f;
        for (int k = tid; k < p.Lk; k += 256) {
            const T* kr = K + (size_t)k * p.ldk;
            float s = 0.f;
            for (int d = 0; d < p.dq; ++d) s += qrow[d] * to_f32(kr[d]);
            s *= p.scale;
            sc[k] = s;
            mx = fmaxf(mx, s);
        }
        mx = wave_max(mx);
        if ((tid & 63) == 0) red[tid >> 6] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float sum = 0.f;
        for (int k = tid; k < p.Lk; k += 256) { const float e = __expf(sc[k] - mx); sc[k] = e; sum += e; }
        sum = wave_sum(sum);
        if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
        __syncthreads();
        sum = red[4] + red[5] + red[6] + red[7];
        for (int d = tid; d < p.dv; d += 256) {
            float o = 0.f;
            for (int k = 0; k < p.Lk; ++k) o += sc[k] * to_f32(V[(size_t)k * p.ldv + d]);
            O[(size_t)i * p.ldo + d] = from_f32<T>(o / sum);
        }
        __syncthreads();
    }
}

int attention_small_launch(const SmallAttnParams& p, int dtype, hipStream_t stream) {
    if (p.B <= 0 || p.H <= 0 || p.Lq <= 0 || p.Lk <= 0 || p.dq <= 0 || p.dv <= 0 || p.Lk > 8192 || p.dq > 1024) {
        set_error("attention_small: unsupported shape"); return IMH_ERR_SHAPE;
    }
    const size_t lds = (size_t)(p.Lk + p.dq + 8) * sizeof(float);
    dim3 grid(p.B * p.H);
    if (dtype == IMH_DT_BF16) hipLaunchKernelGGL((attn_small_kernel<bf16_t>), grid, dim3(256), lds, stream, p);
    else if (dtype == IMH_DT_F16) hipLaunchKernelGGL((attn_small_kernel<f16_t>), grid, dim3(256), lds, stream, p);
    else { set_error("attention_small: unknown dtype %d", dtype); return IMH_ERR_DTYPE; }
    return check_launch("attn_small_kernel");
}

int attention_launch(const AttnParams& p, int dtype, hipStream_t stream) {
    if (p.Lk <= 0 || p.Lk_pad % ATT_KV != 0 || p.Lk_pad < p.Lk) {
        set_error("attention: Lk=%d Lk_pad=%d (pad must be a multiple of 64 and >= Lk)", p.Lk, p.Lk_pad);
        return IMH_ERR_SHAPE;
    }
    if (p.K2 && (p.Lk2 <= 0 || p.Lk2_pad % ATT_KV != 0 || p.Lk2_pad < p.Lk2)) {
        set_error("attention: Lk2=%d Lk2_pad=%d invalid", p.Lk2, p.Lk2_pad);
        return IMH_ERR_SHAPE;
    }
    if ((p.ldq & 7) || (p.ldk & 7) || (p.ldvt & 7) || (p.ldo & 7) || (p.K2 && ((p.ldk2 & 7) || (p.ldvt2 & 7)))) {
        set_error("attention: leading dimensions must be multiples of 8 elements");
        return IMH_ERR_SHAPE;
    }
    if (p.B <= 0 || p.H <= 0 || p.Lq <= 0) { set_error("attention: empty problem"); return IMH_ERR_SHAPE; }
    if (dtype != IMH_DT_BF16 && dtype != IMH_DT_F16) { set_error("attention: unknown dtype %d", dtype); return IMH_ERR_DTYPE; }
    // 4 waves (128 queries) share every K / V^T tile.  Finer workgroups (1 or 2 waves) balance the grid better but
    // re-read K / V^T and measured 25-50 % slower on MI355X, so only this shape is built.
    constexpr int nw = 4;
    const int items = ((p.Lq + 32 * nw - 1) / (32 * nw)) * p.H * p.B;
    dim3 grid(8 * ((items + 7) / 8));
#define IMH_ATT_LAUNCH(TT, NWV) do { if (p.K2) hipLaunchKernelGGL((attn_kernel<TT, NWV, 2>), grid, dim3(64 * NWV), 0, stream, p); \
        else hipLaunchKernelGGL((attn_kernel<TT, NWV, 1>), grid, dim3(64 * NWV), 0, stream, p); } while (0)
    if (dtype == IMH_DT_BF16) IMH_ATT_LAUNCH(bf16_t, 4);
    else IMH_ATT_LAUNCH(f16_t, 4);
#undef IMH_ATT_LAUNCH
    return check_launch("attn_kernel");
}

}  // namespace imh
