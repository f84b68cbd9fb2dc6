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
#include "imh_attn_core.h"

namespace imh {

int g_attn_force_nw = 0;   // retired tuning knob (imh_debug_set key 0): only the 4-wave workgroup is built
int g_attn_mode = 0;       // imh_debug_set key 4 (tests / A-B only; read at launch or capture time, not thread-safe): 0 auto, 1 in-order key loop
                           // (attn_core), 2 software-pipelined key loop (attn_core_pipe; one key set) with the textbook running maximum, 3 the same
                           // with the deferred maximum, 5 key-split workgroups (attn_ks_kernel, deferred maximum), 6 the same, textbook maximum,
                           // 7 = 3 with whole items only (no key-quarter workgroups)
constexpr float ATT_DEFER_LOG2 = 8.0f;

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

    f32x16 fin[2];
    attn_core<T, NW, NPASS>(p, smem, qf, b, h, wave, lane, item, fin);
    attn_store<T, NW>(p, qs, fin, b, h, q0, wave, lane);
    if (!ATT_TIMING) tail_prefetch(p.pf_ptr, p.pf_bytes, item, items, tid, 64 * NW);
}


// The same workgroup shape with the software-pipelined key loop (imh_attn_core.h attn_core_pipe): one key set only (the
// self-attention of AttnProcessor2_0), 4-slot K / V^T ring (64 KB -> two workgroups per CU, 256 registers per lane).
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_pipe_kernel(const AttnParams p) {
    constexpr int NW = 4;
    typedef typename Vec<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // ATT_PIPE_STAGES * 16 KB
    unsigned char* qs = smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31;
    const int hi = lane >> 5;
    const int gx = (p.Lq + 32 * NW - 1) / (32 * NW);
    const int items = gx * p.H * p.B;
    const int per = (items + 7) >> 3;
    // Round 5: when an XCD's item count is not a multiple of its 32 CUs (L = 1024 at UNet batch 2: 40; L = 4096: 80) the left-over items are
    // not run as 4-wave workgroups on a few CUs -- those CUs then carry two waves per SIMD for the whole launch (21.4 us against 14.6,
    // r05_attn_wg_timeline.txt) -- but as 4 * split QUARTER workgroups, one per CU: (item, 32 queries) with the four waves on the four
    // quarters of the keys, merged through LDS.  Every SIMD then carries 1 + 1/4 units of work.
    const int split = p.split;
    const int jx = blockIdx.x >> 3;
    int item, qg = -1;
    if (jx < per - split) item = (blockIdx.x & 7) * per + jx;
    else {
        const int k = jx - (per - split);
        if (k >= 4 * split) return;
        item = (blockIdx.x & 7) * per + (per - split) + (k >> 2);
        qg = k & 3;
    }
    if (item >= items) return;
    const int hb = item / gx, qblk = item - hb * gx;
    const int b = hb / p.H, h = hb - b * p.H;
    const int q0 = qblk * (32 * NW);
#if ATT_TIMING == 2
    const unsigned long long ts_entry_ = __builtin_amdgcn_s_memrealtime();
#endif
    if (qg >= 0) {
        // ---- quarter role: wave w runs the plain in-order tile over keys [w * Lk / 4, +Lk / 4) for queries q0 + 32 qg .. + 31, from a
        //      wave-private one-slot K / V^T tile (no workgroup barrier in the loop: the co-resident whole item's wave on this SIMD fills
        //      the load latency); the launcher guarantees Lk % 256 == 0 and Lq % 128 == 0
        const float c = p.scale * LOG2E;
        unsigned char* my = smem + wave * (2 * ATT_TILE_BYTES);
        v8 qf[4];
        {
            const T* qrow = (const T*)p.Q + ((size_t)b * p.Lq + q0 + qg * 32 + l32) * p.ldq + h * 64 + hi * 8;
#pragma unroll
            for (int sd = 0; sd < 4; ++sd) qf[sd] = *(const v8*)(qrow + sd * 16);
        }
        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
        float m_run = NEG_BIG, l_run = 0.f;
        const int ntq = (p.Lk / ATT_KV) >> 2;
        // staging map of attn_stage_tile with ONE wave doing all four sub-maps: row = 32 i + 8 w + lane / 8; the swizzled chunk depends on
        // (8 (w & 1) + lane / 8) only, so two per-lane sources each (even / odd 8-row groups) + wave-uniform offsets cover the 16 pieces
        const unsigned char* kq[2];
        const unsigned char* vq[2];
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int row = par * 8 + (lane >> 3);
            const int ch = stage_chunk_x(row, lane);
            kq[par] = (const unsigned char*)((const T*)p.K + ((size_t)b * p.Lk_pad + row) * p.ldk + h * 64 + ch * 8);
            vq[par] = (const unsigned char*)((const T*)p.Vt + ((size_t)h * 64 + row) * p.ldvt + (size_t)b * p.Lk_pad + ch * 8);
        }
        const size_t kstep = (size_t)ATT_KV * p.ldk * sizeof(T), vstep = (size_t)ATT_KV * sizeof(T);
        const size_t krow16 = (size_t)16 * p.ldk * sizeof(T), vrow16 = (size_t)16 * p.ldvt * sizeof(T);
        for (int t = 0; t < ntq; ++t) {
            const int tile = wave * ntq + t;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int g16 = i * 2 + (w >> 1);              // 16-row group of the piece
                    glds16(kq[w & 1] + tile * kstep + g16 * krow16, my + (i * 32 + w * 8) * 128);
                    glds16(vq[w & 1] + tile * vstep + g16 * vrow16, my + ATT_TILE_BYTES + (i * 32 + w * 8) * 128);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the Q fragments too, at t = 0)
            attn_tile<T, false>(my, my + ATT_TILE_BYTES, qf, lane, tile * ATT_KV, p.Lk, c, o, m_run, l_run);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's reads of the slot are done before it is staged again
        }
        // merge the four partial (m, l, O) of every query row: waves 1..3 park theirs in their (dead) slots, wave 0 combines in a fixed
        // order -- m = max m_i, a_i = 2^((m_i - m) c), O = sum O_i a_i / sum l_i a_i -- and stores
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        if (wave > 0) {
            float* w = (float*)my;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) w[(dt * 16 + r) * 64 + lane] = o[dt][r];
            w[32 * 64 + lane] = m_run;
            w[33 * 64 + lane] = l_tot;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (wave == 0) {
            float mi[4], li[4];
            mi[0] = m_run; li[0] = l_tot;
#pragma unroll
            for (int g = 1; g < 4; ++g) {
                const float* w = (const float*)(smem + g * (2 * ATT_TILE_BYTES));
                mi[g] = w[32 * 64 + lane]; li[g] = w[33 * 64 + lane];
            }
            const float m = fmaxf(fmaxf(mi[0], mi[1]), fmaxf(mi[2], mi[3]));
            float ai[4], den = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) { ai[g] = __builtin_amdgcn_exp2f((mi[g] - m) * c); den += li[g] * ai[g]; }
            const float inv = 1.0f / den;
            f32x16 fin[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = o[dt][r] * ai[0];
#pragma unroll
                    for (int g = 1; g < 4; ++g) v += ((const float*)(smem + g * (2 * ATT_TILE_BYTES)))[(dt * 16 + r) * 64 + lane] * ai[g];
                    fin[dt][r] = v * inv;
                }
            attn_store<T, 1>(p, smem, fin, b, h, q0 + qg * 32, 0, lane);      // (wave 0's own slot: rows 0..31 as staging rows)
        }
#if ATT_TIMING == 2
        if (tid == 0 && p.pf_ptr) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            unsigned long long* dbg = (unsigned long long*)p.pf_ptr + (size_t)blockIdx.x * 4;
            dbg[0] = ts_entry_; dbg[1] = __builtin_amdgcn_s_memrealtime(); dbg[2] = hwid; dbg[3] = (blockIdx.x & 7) | 0x100;
        }
#else
        if (qg == 0) tail_prefetch(p.pf_ptr, p.pf_bytes, item, items, tid, 64 * NW);
#endif
        return;
    }
    v8 qf[4];
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 32 + i * 8 + (lane >> 3);
            const int ch = stage_chunk_x(row, lane);
            const T* src = (const T*)p.Q + ((size_t)b * p.Lq + min(q0 + row, p.Lq - 1)) * p.ldq + h * 64 + ch * 8;
            glds16(src, qs + (wave * 32 + i * 8) * 128);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int row = wave * 32 + l32;
#pragma unroll
        for (int sd = 0; sd < 4; ++sd) qf[sd] = *(const v8*)(qs + tile_off(row, sd * 2 + hi, swz_x(row)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    f32x16 fin[2];
    attn_core_pipe<T>(p, smem, qf, b, h, wave, lane, fin);
    attn_store<T, NW>(p, qs, fin, b, h, q0, wave, lane);
#if ATT_TIMING == 2     // tools/attn_phase_probe.py wg: every workgroup stamps (entry, exit) on the chip-wide 100 MHz counter + its XCC / CU id
    if (tid == 0 && p.pf_ptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned long long* dbg = (unsigned long long*)p.pf_ptr + (size_t)blockIdx.x * 4;
        dbg[0] = ts_entry_; dbg[1] = __builtin_amdgcn_s_memrealtime(); dbg[2] = hwid; dbg[3] = blockIdx.x & 7;
    }
#else
    tail_prefetch(p.pf_ptr, p.pf_bytes, item, items, tid, 64 * NW);
#endif
}

// Key-split self-attention (imh_attn_core.h attn_core_ks): one workgroup = (batch, head, 128 queries) = 8 waves = 2 key halves x 4
// query groups; 128 registers and 64 KB of LDS -> two workgroups = four waves per SIMD on a CU.  Q fragments come straight from
// memory (lane (q, hi) reads its row's four 16-B pieces: 32 lines per instruction, once per workgroup, in flight beside the
// first K / V^T tile) -- no Q staging tile, no extra barrier.
#ifdef IMH_EXPERIMENTAL
template <typename T>
__global__ __launch_bounds__(512, 4) void attn_ks_kernel(const AttnParams p) {
    typedef typename Vec<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // 2 groups x 2 slots x (K + V^T tile) = 64 KB
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, qg = wave & 3;
    const int hi = lane >> 5;
    const int gx = (p.Lq + 127) / 128;
    const int items = gx * p.H * p.B;
    const int per = (items + 7) >> 3;
    const int item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || item >= items) return;
    const int hb = item / gx, qblk = item - hb * gx;
    const int b = hb / p.H, h = hb - b * p.H;
    const int q0 = qblk * 128;
    v8 qf[4];
    {
        const T* qrow = (const T*)p.Q + ((size_t)b * p.Lq + min(q0 + qg * 32 + (lane & 31), p.Lq - 1)) * p.ldq + h * 64 + hi * 8;
#pragma unroll
        for (int sd = 0; sd < 4; ++sd) qf[sd] = *(const v8*)(qrow + sd * 16);
    }
    f32x16 o[2], fin[2];
    float m_run, l_run;
    attn_core_ks<T>(p, smem + g * (2 * 2 * ATT_TILE_BYTES), qf, b, h, g, qg, lane, o, m_run, l_run);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // nobody reads the rings any more: merge buffer at 0, O staging rows behind it
    attn_ks_merge((float*)smem, p.scale * LOG2E, g, qg, lane, o, m_run, l_run, fin);
    if (g == 0) attn_store<T, 4>(p, smem + 4 * 34 * 64 * 4, fin, b, h, q0, qg, lane);
    else tail_prefetch(p.pf_ptr, p.pf_bytes, item, items, tid - 256, 256);
}
#endif

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

// ---- the same op with the head's keys and values resident in LDS (the Resampler / HarmonyAttention shapes:
// 8-16 queries x 77-273 keys, head dims 40-64).  One workgroup per (batch, head); K is staged TRANSPOSED ([dq][LkP], so a
// wave's lanes = consecutive keys read consecutive LDS addresses), V as is ([Lk][dv], lanes = consecutive head dims);
// every wave then owns every 4th query: scores (lane = key), fp32 softmax across the wave, P V (lane = head dim).
// K and V are read from global memory once per (batch, head) with 16-B loads instead of once per query with
// 2-B loads (attn_small_kernel: 16 x 35 KB uncoalesced per head at the PlusXL Resampler shape).
template <typename T>
__global__ __launch_bounds__(256) void attn_small_lds_kernel(const SmallAttnParams p, const int LkP) {
    typedef typename Vec<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    T* kt = (T*)smraw;                                   // [dq][LkP]
    T* vs = kt + (size_t)p.dq * LkP;                     // [Lk][dv]
    float* pr = (float*)(vs + (size_t)p.Lk * p.dv);      // [4][LkP] scores / probabilities of the wave's current query
    float* qr = pr + 4 * LkP;                            // [4][dq]  the wave's current query row (pre-scaled)
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* Q = (const T*)p.Q + (size_t)b * p.Lq * p.ldq + h * p.dq;
    const T* K = (const T*)p.K + (size_t)b * p.Lk * p.ldk + h * p.dq;
    const T* V = (const T*)p.V + (size_t)b * p.Lk * p.ldv + h * p.dv;
    T* O = (T*)p.O + (size_t)b * p.Lq * p.ldo + h * p.dv;
    const int cq = p.dq >> 3, cv = p.dv >> 3;            // 16-B chunks per row
    for (int i = tid; i < p.Lk * cq; i += 256) {
        const int k = i / cq, c = i - k * cq;
        const v8 t = *(const v8*)(K + (size_t)k * p.ldk + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) kt[(size_t)(c * 8 + e) * LkP + k] = t[e];
    }
    for (int i = tid; i < (LkP - p.Lk) * p.dq; i += 256) {        // padding keys: finite zeros (masked below)
        const int d = i / (LkP - p.Lk), k = p.Lk + i % (LkP - p.Lk);
        kt[(size_t)d * LkP + k] = (T)0.f;
    }
    for (int i = tid; i < p.Lk * cv; i += 256) {
        const int k = i / cv, c = i - k * cv;
        *(v8*)(vs + (size_t)k * p.dv + c * 8) = *(const v8*)(V + (size_t)k * p.ldv + c * 8);
    }
    __syncthreads();
    float* mypr = pr + wave * LkP;
    float* myq = qr + wave * p.dq;
    for (int q = wave; q < p.Lq; q += 4) {
        for (int d = lane; d < p.dq; d += 64) myq[d] = to_f32(Q[(size_t)q * p.ldq + d]) * p.scale;
        __builtin_amdgcn_wave_barrier();
        float mx = -1e30f;
        for (int k0 = 0; k0 < LkP; k0 += 64) {
            const int k = k0 + lane;
            float sc = 0.f;
            for (int d = 0; d < p.dq; ++d) sc += myq[d] * to_f32(kt[(size_t)d * LkP + k]);
            if (k >= p.Lk) sc = -1e30f;
            mypr[k] = sc;
            mx = fmaxf(mx, sc);
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int k0 = 0; k0 < LkP; k0 += 64) {
            const int k = k0 + lane;
            const float e = k < p.Lk ? __expf(mypr[k] - mx) : 0.f;
            mypr[k] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        __builtin_amdgcn_wave_barrier();
        const float inv = 1.0f / sum;
        for (int d = lane; d < p.dv; d += 64) {
            float o = 0.f;
            for (int k = 0; k < p.Lk; ++k) o += mypr[k] * to_f32(vs[(size_t)k * p.dv + d]);
            O[(size_t)q * p.ldo + d] = from_f32<T>(o * inv);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int attention_small_launch(const SmallAttnParams& p, int dtype, hipStream_t stream) {
    if (p.B <= 0 || p.H <= 0 || p.Lq <= 0 || p.Lk <= 0 || p.dq <= 0 || p.dv <= 0 || p.Lk > 8192 || p.dq > 1024) {
        set_error("attention_small: unsupported shape"); return IMH_ERR_SHAPE;
    }
    if (dtype != IMH_DT_BF16 && dtype != IMH_DT_F16) { set_error("attention_small: unknown dtype %d", dtype); return IMH_ERR_DTYPE; }
    {   // LDS-resident form when the head's K^T and V fit and the rows are 16-B addressable
        const int LkP = (p.Lk + 63) & ~63;
        const size_t need = ((size_t)p.dq * LkP + (size_t)p.Lk * p.dv) * 2 + (size_t)(4 * LkP + 4 * p.dq) * sizeof(float);
        const bool aligned = !(p.dq & 7) && !(p.dv & 7) && !(p.ldk & 7) && !(p.ldv & 7) && !(((size_t)p.Lk * p.dv * 2 + (size_t)p.dq * LkP * 2) & 15);
        if (aligned && need <= 150 * 1024) {
            dim3 grid(p.B * p.H);
            if (dtype == IMH_DT_BF16) {
                static DynLdsOnce a0;
                a0.ensure((const void*)attn_small_lds_kernel<bf16_t>, 150 * 1024);
                hipLaunchKernelGGL((attn_small_lds_kernel<bf16_t>), grid, dim3(256), need, stream, p, LkP);
            } else {
                static DynLdsOnce a1;
                a1.ensure((const void*)attn_small_lds_kernel<f16_t>, 150 * 1024);
                hipLaunchKernelGGL((attn_small_lds_kernel<f16_t>), grid, dim3(256), need, stream, p, LkP);
            }
            return check_launch("attn_small_lds_kernel");
        }
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
    // one key set, an even number of whole key tiles, at least two per group: the key-split kernel (every SDXL self-attention)
    // (auto keeps the software-pipelined kernel: the two measure the same in the forward, 38.4 vs 38.7 us per launch,
    // profiles/r04_forward_ab_attn.json)
    if (!p.K2 && p.Lk % (2 * ATT_KV) == 0 && p.Lk >= 4 * ATT_KV && (g_attn_mode == 5 || g_attn_mode == 6)) {
#ifndef IMH_EXPERIMENTAL
        return experimental_refused("the key-split attention kernel (imh_debug_set(4, 5 | 6))");
#else
        const int lds = 2 * 2 * 2 * ATT_TILE_BYTES;
        const int items128 = ((p.Lq + 127) / 128) * p.H * p.B;
        dim3 grid2(8 * ((items128 + 7) / 8));
        AttnParams q = p;
        q.defer_log2 = g_attn_mode == 6 ? 0.0f : ATT_DEFER_LOG2;
        if (dtype == IMH_DT_BF16) { static DynLdsOnce once; once.ensure((const void*)attn_ks_kernel<bf16_t>, lds);
                                    hipLaunchKernelGGL((attn_ks_kernel<bf16_t>), grid2, dim3(512), lds, stream, q); }
        else { static DynLdsOnce once; once.ensure((const void*)attn_ks_kernel<f16_t>, lds);
               hipLaunchKernelGGL((attn_ks_kernel<f16_t>), grid2, dim3(512), lds, stream, q); }
        return check_launch("attn_ks_kernel");
#endif
    }
    if (!p.K2 && p.Lk % ATT_KV == 0 && (g_attn_mode == 2 || g_attn_mode == 3 || g_attn_mode == 7 || (g_attn_mode == 0 && p.Lk >= 4 * ATT_KV))) {
        const int lds = ATT_PIPE_STAGES * 2 * ATT_TILE_BYTES;
        AttnParams q = p;
        q.defer_log2 = g_attn_mode == 2 ? 0.0f : ATT_DEFER_LOG2;
        // items beyond one per CU as key-quarter workgroups (attn_pipe_kernel); imh_debug_set(4, 7) = whole items only (A/B)
        const int per = (items + 7) / 8;
        // (per % 32 items per XCD are left over after whole rounds of one item per CU: 8 of 40 at L = 1024, 16 of 80 at L = 4096 (UNet batch 2);
        // up to 16 -- a quarter workgroup reads its head's whole K / V^T for 32 queries, 4 x the bytes per query of a whole item)
        q.split = (g_attn_mode != 7 && per > 32 && per % 32 != 0 && per % 32 <= 16 && p.Lq % 128 == 0 && p.Lk % (4 * ATT_KV) == 0) ? per % 32 : 0;
        dim3 gridp(8 * (per + 3 * q.split));
        if (dtype == IMH_DT_BF16) { static DynLdsOnce once; once.ensure((const void*)attn_pipe_kernel<bf16_t>, lds);
                                    hipLaunchKernelGGL((attn_pipe_kernel<bf16_t>), gridp, dim3(256), lds, stream, q); }
        else { static DynLdsOnce once; once.ensure((const void*)attn_pipe_kernel<f16_t>, lds);
               hipLaunchKernelGGL((attn_pipe_kernel<f16_t>), gridp, dim3(256), lds, stream, q); }
        return check_launch("attn_pipe_kernel");
    }
#define IMH_ATT_LAUNCH(TT, NWV) do { if (p.K2) hipLaunchKernelGGL((attn_kernel<TT, NWV, 2>), grid, dim3(64 * NWV), 0, stream, p); \
        else hipLaunchKernelGGL((attn_kernel<TT, NWV, 1>), grid, dim3(64 * NWV), 0, stream, p); } while (0)
    if (dtype == IMH_DT_BF16) IMH_ATT_LAUNCH(bf16_t, 4);
    else IMH_ATT_LAUNCH(f16_t, 4);
#undef IMH_ATT_LAUNCH
    return check_launch("attn_kernel");
}

}  // namespace imh
