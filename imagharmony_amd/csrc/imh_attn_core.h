// Shared device code of the flash-attention kernels (attention.hip: Q from memory; xattn.hip: Q projected in the
// kernel's prologue): the online-softmax key loop over LDS-DMA-staged K / V^T tiles and the coalesced O store.
// See attention.hip for the structure notes.
#pragma once
#include "imh_common.h"
#include "imh_kernels.h"
#include <type_traits>

namespace imh {

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
// FIRST read of freshly issued MFMA accumulators: compiler-visible (fmaxf), so that hipcc's hazard recogniser inserts the
// MFMA -> VALU wait states.  max3f() above is inline asm: hipcc pads nothing for an asm consumer (cdna_hip_programming.md 5.7
// item 2), and a v_max3 that issues before the producing MFMA has retired reads stale registers -- the softmax stays
// mathematically valid for any row maximum, so results only wobble in the last bits, run to run (found as a bitwise
// non-determinism of the pipelined key loop).  Once one element of an accumulator has been read this way the whole MFMA has
// retired and the asm reads that follow are safe.
__device__ __forceinline__ float mfma_first_max(float a, float b) { return __builtin_fmaxf(a, b); }
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


template <typename T, bool MASK, int NKT>
__device__ __forceinline__ void attn_tile_n(const unsigned char* ks, const unsigned char* vs, const typename Vec<T>::v8 (&qf)[4],
                                            const int lane, const int kbase, const int Lk, const float c, f32x16 (&o)[2],
                                            float& m_run, float& l_run);

// The key loop: NPASS key sets (text [, image-prompt]) against the Q^T fragments `qf` of this wave's 32 queries;
// returns fin = sum over passes of weight * softmax(QK^T) V, transposed (lane owns one query, att_o_dim head dims).
// smem = the ATT_STAGES * 16 KB ring; every wave of the workgroup must call this (workgroup barriers inside), and no
// wave may still read smem for anything else when it is entered.
template <typename T, int NW, int NPASS>
__device__ __forceinline__ void attn_core(const AttnParams& p, unsigned char* smem, const typename Vec<T>::v8 (&qf)[4],
                                          const int b, const int h, const int wave, const int lane, const int item,
                                          f32x16 (&fin)[2]) {
    constexpr int RND = 8 / NW;              // staging rounds: NW*8 rows per round, 64 rows per tile
    typedef typename Vec<T>::v8 v8;
    const int tid = threadIdx.x;
    const int hi = lane >> 5;
    (void)tid; (void)item;
    const float c = p.scale * LOG2E;      // > 0

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
#if !ATT_ABL && !ATT_TIMING
            // Round 6: a tile whose second 32-key sub-tile is all padding (the 13 keys of the text set's second tile, an image-prompt set
            // of <= 32 tokens: three of the four tiles a fused cross-attention call touches) takes the half-width body -- bit-identical
            // (the skipped scores exponentiate to exactly 0), 8 of 16 MFMAs and 16 of 32 v_exp_f32 fewer
            if (kbase + 32 >= Lk) {                     // wave-uniform
                attn_tile_n<T, true, 1>(ks, vs, qf, lane, kbase, Lk, c, o, m_run, l_run);
                asm volatile("" ::: "memory");
                if (++cur == ATT_STAGES) cur = 0;
                continue;
            }
#endif

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
            float mx = max3f(mfma_first_max(st[0][0], st[1][0]), st[0][1], st[1][1]);
            if (!(ATT_ABL & 8)) {
                mx = fmaxf(mx, st[0][2]);
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

}

// One 64-key tile that is already RESIDENT in LDS (K rows at ks, V^T rows at vs, both landed and visible): S^T = K Q^T,
// online softmax on raw scores, O^T += V^T P^T -- the body of attn_core's loop without the ring (short key sets: the 77 text
// + T image tokens of the cross-attention layers are staged once, ahead of the fused kernel's projection).
// (MASK = false: the caller guarantees whole tiles -- no key masking code at all; with a run-time `ragged` hipcc if-converts the mask into
// 84 compare / select instructions per tile)
// NKT = 32-key sub-tiles computed (2 = the whole tile; 1 = only the first: every key of the second is padding -- the 13 valid keys of the
// text set's second tile, an image-prompt set of <= 32 tokens -- whose scores would be masked to NEG_BIG, exponentiate to exactly 0 and add
// exactly 0 to the row sums and to O: skipping them is bit-identical and saves 8 of the tile's 16 MFMAs and 16 of its 32 v_exp_f32).
template <typename T, bool MASK, int NKT>
__device__ __forceinline__ void attn_tile_n(const unsigned char* ks, const unsigned char* vs, const typename Vec<T>::v8 (&qf)[4],
                                            const int lane, const int kbase, const int Lk, const float c, f32x16 (&o)[2],
                                            float& m_run, float& l_run) {
    typedef typename Vec<T>::v8 v8;
    const int hi = lane >> 5;
    const bool ragged = MASK && kbase + 32 * NKT > Lk;
    f32x16 st[NKT];
    {
        v8 kf[NKT][4];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int sd = 0; sd < 4; ++sd) kf[kt][sd] = *(const v8*)(ks + att_k_off(lane, kt, sd));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
            for (int sd = 0; sd < 4; ++sd) st[kt] = mfma32(kf[kt][sd], qf[sd], st[kt]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    v8 vf[NKT][2][2];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) vf[kt][s][dt] = *(const v8*)(vs + att_v_off(lane, dt, kt, s));
    __builtin_amdgcn_sched_barrier(0);
    if (ragged) {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kbase + kt * 32 + st_key(r, hi) >= Lk) st[kt][r] = NEG_BIG;
    }
    float mx;
    if constexpr (NKT == 2) {
        mx = max3f(mfma_first_max(st[0][0], st[1][0]), st[0][1], st[1][1]);
        mx = fmaxf(mx, st[0][2]);
#pragma unroll
        for (int r = 2; r < 15; ++r) mx = max3f(mx, st[1][r], st[0][r + 1]);
        mx = fmaxf(mx, st[1][15]);
    } else {
        mx = max3f(mfma_first_max(st[0][0], st[0][1]), st[0][2], st[0][3]);
#pragma unroll
        for (int r = 4; r < 16; r += 2) mx = max3f(mx, st[0][r], st[0][r + 1]);
    }
    mx = xhalf_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const f32x2 c2 = {c, c};
    const f32x2 nmc2 = {-m_new * c, -m_new * c};
    f32x2 ps2 = {0.f, 0.f};
    v8 pf[NKT][2];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const f32x2 s2 = {st[kt][r], st[kt][r + 1]};
            const f32x2 e2 = __builtin_elementwise_fma(s2, c2, nmc2);
            const f32x2 p2 = {__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
            ps2 += p2;
            pf[kt][r >> 3][r & 7] = from_f32<T>(p2[0]);
            pf[kt][r >> 3][(r & 7) + 1] = from_f32<T>(p2[1]);
        }
    if (__any(m_new != m_run)) {
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        m_run = m_new;
    }
    l_run += ps2[0] + ps2[1];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) o[dt] = mfma32(vf[kt][s][dt], pf[kt][s], o[dt]);
}

template <typename T, bool MASK = true>
__device__ __forceinline__ void attn_tile(const unsigned char* ks, const unsigned char* vs, const typename Vec<T>::v8 (&qf)[4],
                                          const int lane, const int kbase, const int Lk, const float c, f32x16 (&o)[2],
                                          float& m_run, float& l_run) {
    if (MASK && kbase + 32 >= Lk) attn_tile_n<T, MASK, 1>(ks, vs, qf, lane, kbase, Lk, c, o, m_run, l_run);      // (wave-uniform)
    else attn_tile_n<T, MASK, 2>(ks, vs, qf, lane, kbase, Lk, c, o, m_run, l_run);
}

// LDS-DMA of one 64-key K / V^T tile pair by the four waves of a head group (wave = 0..3 inside the group): the staging map
// of attn_core's ring (rows of 8 per wave instruction, source-side swizzle)
template <typename T>
__device__ __forceinline__ void attn_stage_tile(const T* Kp, const T* Vp, const int Lkp, const int ldk, const int ldvt, const int b,
                                                const int h, const int tile, unsigned char* ks, unsigned char* vs, const int wave,
                                                const int lane) {
    const int kbase = tile * ATT_KV;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = i * 32 + wave * 8 + (lane >> 3);
        const int ch = stage_chunk_x(row, lane);
        glds16(Kp + ((size_t)b * Lkp + kbase + row) * ldk + h * 64 + ch * 8, ks + (i * 32 + wave * 8) * 128);
        glds16(Vp + ((size_t)h * 64 + row) * ldvt + (size_t)b * Lkp + kbase + ch * 8, vs + (i * 32 + wave * 8) * 128);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Software-pipelined key loop (self-attention, one key set).  attn_core above runs  QK^T(t) -> softmax(t) -> PV(t)  strictly in
// order inside a wave: the matrix pipe idles during the softmax (32 v_exp_f32 + ~90 VALU per tile: as long as all 16 MFMAs
// of the tile) and the VALU idles during the MFMAs; a single wave needs ~2100 cycles per 64-key tile for 512 cycles of MFMA
// work (tools/attn_phase_probe.py).  Here the two halves of consecutive tiles overlap inside ONE instruction stream:
//     phase B(t):  S(t+1) = K(t+1) Q^T   (8 MFMAs)   beside   P(t) = exp2(S(t) c - m c), row sums, packing     (VALU)
//     phase A(t):  O^T += V^T(t) P(t)^T  (8 MFMAs)   beside   row max of S(t+1), new running max               (VALU)
// (MFMAs issue in a few cycles and execute for 32; the VALU work is placed in their shadow with sched_group_barrier, the
// recipe of cdna_hip_programming.md "Fused attention prefill" for one wave per SIMD.)  The running-max rescale of O is a
// wave-uniform branch taken only when some row maximum moved.  K / V^T tiles come from a 4-slot LDS-DMA ring with counted
// vmcnt: at the top of iteration t tile t+1 must have landed (its K feeds phase B, V(t) landed one iteration earlier), tiles
// t+2 and t+3 stay in flight; one s_barrier per tile.
constexpr int ATT_PIPE_STAGES = 4;

template <typename T>
__device__ __forceinline__ void attn_core_pipe(const AttnParams& p, unsigned char* smem, const typename Vec<T>::v8 (&qf)[4],
                                               const int b, const int h, const int wave, const int lane, f32x16 (&fin)[2]) {
    typedef typename Vec<T>::v8 v8;
    constexpr int S = ATT_PIPE_STAGES, LPT = 4;
    const int hi = lane >> 5;
    const float c = p.scale * LOG2E;
    const T* Kp = (const T*)p.K;
    const T* Vp = (const T*)p.Vt;
    const int Lk = p.Lk;
    const int ntiles = (Lk + ATT_KV - 1) / ATT_KV;
    auto stage = [&](int tile) {
        unsigned char* ks = smem + (tile & (S - 1)) * 2 * ATT_TILE_BYTES;
        attn_stage_tile<T>(Kp, Vp, p.Lk_pad, p.ldk, p.ldvt, b, h, tile, ks, ks + ATT_TILE_BYTES, wave, lane);
    };
    // the same tile in four single-instruction pieces (q = 2 * round + (0: K, 1: V^T)).  The lane's four source addresses are formed ONCE
    // (for tile 0); a tile step is a wave-uniform byte offset -- the per-piece address arithmetic (64-bit multiplies by ldk / ldvt, ~12
    // integer VALU instructions a piece, four of them quarter-rate) otherwise sits in the key loop beside a softmax that is VALU-bound
    const int srow = wave * 8 + (lane >> 3);
    const unsigned char* ksrc[2];
    const unsigned char* vsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = i * 32 + srow;
        const int ch = stage_chunk_x(row, lane);
        ksrc[i] = (const unsigned char*)(Kp + ((size_t)b * p.Lk_pad + row) * p.ldk + h * 64 + ch * 8);
        vsrc[i] = (const unsigned char*)(Vp + ((size_t)h * 64 + row) * p.ldvt + (size_t)b * p.Lk_pad + ch * 8);
    }
    const size_t kstep = (size_t)ATT_KV * p.ldk * sizeof(T), vstep = (size_t)ATT_KV * sizeof(T);
    auto stage_piece = [&](int tile, int q) {
        unsigned char* ks = smem + (tile & (S - 1)) * 2 * ATT_TILE_BYTES;
        const int i = q >> 1;
        if (q & 1) glds16(vsrc[i] + (size_t)tile * vstep, ks + ATT_TILE_BYTES + (i * 32 + wave * 8) * 128);
        else glds16(ksrc[i] + (size_t)tile * kstep, ks + (i * 32 + wave * 8) * 128);
    };
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;

    // raw scores of one tile: 8 K fragments, 8 MFMAs
    auto qk = [&](f32x16 (&st)[2], const unsigned char* ks) {
        v8 kf[2][4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int sd = 0; sd < 4; ++sd) kf[kt][sd] = *(const v8*)(ks + att_k_off(lane, kt, sd));
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
            for (int sd = 0; sd < 4; ++sd) st[kt] = mfma32(kf[kt][sd], qf[sd], st[kt]);
        }
    };
    // row maximum (lane-local row, one cross-half exchange)
    auto rowmax = [&](f32x16 (&st)[2], const int tile) -> float {
        (void)tile; (void)hi;
        float mx = max3f(mfma_first_max(st[0][0], st[1][0]), st[0][1], st[1][1]);
        mx = fmaxf(mx, st[0][2]);
#pragma unroll
        for (int r = 2; r < 15; ++r) mx = max3f(mx, st[1][r], st[0][r + 1]);
        mx = fmaxf(mx, st[1][15]);
        return xhalf_max(mx);
    };
    // Deferred maximum (cdna_hip_programming.md T13): the running maximum -- and with it the O / l rescale, which has to wait for
    // the PV MFMAs it multiplies -- moves only when some row's maximum grew by more than 2^defer in the exponent domain; until
    // then P = exp2((S - m_old) c) <= 2^defer (256: exact range for bf16 / fp16 operands, fp32 sums).  The decision sits after
    // ALL of tile t's PV MFMAs and before tile t + 1's P is exponentiated, so every term is scaled exactly once.
    // defer = 0 is the textbook rule (rescale whenever a maximum moved).  (s_setprio 1 over phase A -- the other workgroup's wave
    // on the SIMD is mostly in its VALU-dense phase B -- was measured too: 23.1 vs 24.4 us at L = 1024 but 132 vs 128 at L = 4096
    // and 70 vs 63.5 at batch 8, and the two wave-uniform branches alone cost the loop 5 %; not kept.)
    const float defer = p.defer_log2;
    auto rescale = [&](const float m_new) {
        if (__any((m_new - m_run) * c > defer)) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            m_run = m_new;
        }
    };

#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < ntiles) stage(s);
    if (ntiles >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
    else if (ntiles == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // tile 0 has landed for every wave
    asm volatile("" ::: "memory");
    f32x16 stA[2], stB[2];
    qk(stA, smem);
    rescale(fmaxf(m_run, rowmax(stA, 0)));

    // one iteration: softmax + PV of tile t (raw scores in `cur`), QK^T of tile t + 1 (into `nxt`).  The interleave is written
    // out by hand and pinned with sched_barrier(0) fences (sched_group_barrier patterns were not honoured at this register
    // pressure): one MFMA, then the slice of VALU work that issues in its shadow.
    auto iter = [&](auto HN, auto STG, f32x16 (&cur)[2], f32x16 (&nxt)[2], const int t) {
        // has_next and the staging mode (0 none, 1 always, 2 runtime check) are compile-time: with runtime flags hipcc moved the
        // eight MFMAs of phase B into one block and the softmax slices into another (no overlap at all).  No masking code either:
        // the launcher takes this kernel only for key counts that are multiples of 64 (every SDXL self-attention)
        constexpr bool has_next = decltype(HN)::value;
        constexpr int stg = decltype(STG)::value;
        const bool stage_more = stg == 1 || (stg == 2 && has_next && t + 3 < ntiles);   // tile t + 3 -> the slot of tile t - 1
        if (has_next) {
            if (t + 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");   // tile t + 1 landed, t + 2 in flight
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();              // ... for every wave; every wave is past its reads of tile t - 1
            asm volatile("" ::: "memory");
        }
        const unsigned char* vs = smem + (t & (S - 1)) * 2 * ATT_TILE_BYTES + ATT_TILE_BYTES;
        const unsigned char* ksn = smem + ((t + 1) & (S - 1)) * 2 * ATT_TILE_BYTES;
        v8 kf[2][4], vf[2][2][2];
        if (has_next) {
#pragma unroll
            for (int sd = 0; sd < 4; ++sd)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) kf[kt][sd] = *(const v8*)(ksn + att_k_off(lane, kt, sd));
        }
        auto v_reads = [&]() {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) vf[kt][s2][dt] = *(const v8*)(vs + att_v_off(lane, dt, kt, s2));
        };
        if (!has_next) v_reads();
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase B: S(t+1) MFMAs beside P(t) ----
        const f32x2 c2 = {c, c};
        const f32x2 nmc2 = {-m_run * c, -m_run * c};
        f32x2 ps2 = {0.f, 0.f};
        v8 pf[2][2];
        auto p_chunk = [&](auto J) {                   // 4 of the tile's 32 scores per lane: 2 pk_fma, 4 exp, 2 pk_add, 2 cvt_pk
            constexpr int j = decltype(J)::value;
            constexpr int kt = j >> 2, r0 = (j & 3) * 4;
#pragma unroll
            for (int r = r0; r < r0 + 4; r += 2) {
                const f32x2 s2 = {cur[kt][r], cur[kt][r + 1]};
                const f32x2 e2 = __builtin_elementwise_fma(s2, c2, nmc2);
                const f32x2 p2 = {__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
                ps2 += p2;
                pf[kt][r >> 3][r & 7] = from_f32<T>(p2[0]);
                pf[kt][r >> 3][(r & 7) + 1] = from_f32<T>(p2[1]);
            }
        };
        auto b_step = [&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int sd = j >> 1, kt = j & 1;      // alternate the two accumulators: no back-to-back dependent MFMAs
            p_chunk(J);                                 // the slice first: the K fragment reads above land under slice 0
            if constexpr (j == 3) { if (has_next) v_reads(); }     // V^T fragments are for phase A: requested mid-way
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) {
                if constexpr (sd == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) nxt[kt][r] = 0.f;
                }
                nxt[kt] = mfma32(kf[kt][sd], qf[sd], nxt[kt]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        b_step(std::integral_constant<int, 0>{}); b_step(std::integral_constant<int, 1>{});
        b_step(std::integral_constant<int, 2>{}); b_step(std::integral_constant<int, 3>{});
        b_step(std::integral_constant<int, 4>{}); b_step(std::integral_constant<int, 5>{});
        b_step(std::integral_constant<int, 6>{}); b_step(std::integral_constant<int, 7>{});
        l_run += ps2[0] + ps2[1];
        // ---- phase A: PV(t) MFMAs beside the row maximum of S(t+1) ----
        float mx = NEG_BIG;
        __builtin_amdgcn_sched_barrier(0);
        auto a_step = [&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int kt = j >> 2, s2 = (j >> 1) & 1, dt = j & 1;
            o[dt] = mfma32(vf[kt][s2][dt], pf[kt][s2], o[dt]);
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) {                             // 32 scores in 8 slices of 4: two v_max3 each
                constexpr int k2 = j >> 2, r0 = (j & 3) * 4;
                if constexpr ((j & 3) == 0) mx = max3f(mfma_first_max(mx, nxt[k2][r0]), nxt[k2][r0 + 1], nxt[k2][r0 + 2]);   // see mfma_first_max
                else mx = max3f(mx, nxt[k2][r0], nxt[k2][r0 + 1]);
                if constexpr ((j & 3) == 0) mx = fmaxf(mx, nxt[k2][r0 + 3]);
                else mx = max3f(mx, nxt[k2][r0 + 2], nxt[k2][r0 + 3]);
                if constexpr (j >= 2 && j < 6 && stg != 0) {     // ... and one of the four LDS-DMA pieces of tile t + 3
                    if (stage_more) stage_piece(t + 3, j - 2);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        a_step(std::integral_constant<int, 0>{}); a_step(std::integral_constant<int, 1>{});
        a_step(std::integral_constant<int, 2>{}); a_step(std::integral_constant<int, 3>{});
        a_step(std::integral_constant<int, 4>{}); a_step(std::integral_constant<int, 5>{});
        a_step(std::integral_constant<int, 6>{}); a_step(std::integral_constant<int, 7>{});
        if (has_next) rescale(fmaxf(m_run, xhalf_max(mx)));
        asm volatile("" ::: "memory");
    };
    const std::integral_constant<bool, true> YES{};
    const std::integral_constant<bool, false> NO{};
    const std::integral_constant<int, 0> STG_NONE{};
    const std::integral_constant<int, 1> STG_ALWAYS{};
    const std::integral_constant<int, 2> STG_CHECK{};
    int t = 0;
    for (; t + 4 < ntiles; t += 2) {              // steady state: tiles t + 3 and t + 4 exist
        iter(YES, STG_ALWAYS, stA, stB, t);
        iter(YES, STG_ALWAYS, stB, stA, t + 1);
    }
    for (; t + 2 < ntiles; t += 2) {              // the last few tiles: staging by runtime check
        iter(YES, STG_CHECK, stA, stB, t);
        iter(YES, STG_CHECK, stB, stA, t + 1);
    }
    if (t + 1 < ntiles) {
        iter(YES, STG_NONE, stA, stB, t);
        iter(NO, STG_NONE, stB, stA, t + 1);
    } else {
        iter(NO, STG_NONE, stA, stB, t);
    }

    __builtin_amdgcn_s_barrier();          // nobody reads the ring any more (the caller reuses it for the O staging rows)
    const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) fin[dt][r] = o[dt][r] * inv;
}

// ---------------------------------------------------------------------------------------------------------------------
// Key-split key loop (self-attention, one key set): the workgroup is TWO groups of four waves; group g runs the online softmax
// over ITS half of the key tiles (own two-slot K / V^T ring, four waves x 32 queries share every tile as above) and the two
// partial results (m, l, O) of a query row are merged once at the end (attn_ks_merge).  Why: at UNet batch 2 the L = 1024 layers
// are 320 workgroups of 128 queries on 256 CUs -- most SIMDs held ONE wave, and a single wave cannot overlap its softmax VALU
// with its MFMAs (profiles/r03_valu_mfma_rate_microbench.csv: MFMA + 6 VALU 54 cycles with one wave per SIMD, 33 with four).
// Splitting the keys doubles the waves without shrinking the 128 queries that share a K / V^T tile; the body is the plain
// in-order tile (QK^T -> softmax -> PV) held to 128 registers, so two workgroups = FOUR waves per SIMD fit a CU and the
// overlap comes from the hardware's wave interleave instead of a hand-pipelined stream (attn_core_pipe: 222 registers, two
// waves per SIMD).  One s_barrier per tile (both groups: the ring slots are group-private, the barrier is the workgroup's).
// Deferred running maximum as in attn_core_pipe; every term is scaled exactly once (the decision precedes the tile's P).
template <typename T>
__device__ __forceinline__ void attn_core_ks(const AttnParams& p, unsigned char* gs, const typename Vec<T>::v8 (&qf)[4],
                                             const int b, const int h, const int g, const int qg, const int lane,
                                             f32x16 (&o)[2], float& m_run, float& l_run) {
    typedef typename Vec<T>::v8 v8;
    const int hi = lane >> 5;
    (void)hi;
    const float c = p.scale * LOG2E;
    const float defer = p.defer_log2;
    const T* Kp = (const T*)p.K;
    const T* Vp = (const T*)p.Vt;
    const int tg = (p.Lk / ATT_KV) >> 1;           // tiles per group (the launcher guarantees an even tile count, no ragged tile)
    const int t0 = g * tg;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    m_run = NEG_BIG; l_run = 0.f;
    // staging sources of this lane for key tile 0 (rows qg*8 + lane/8 and + 32 of the K tile / the V^T tile); a tile step is a
    // wave-uniform byte offset, so the loop carries no per-lane 64-bit address arithmetic
    const unsigned char* ksrc[2];
    const unsigned char* vsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = i * 32 + qg * 8 + (lane >> 3);
        const int ch = stage_chunk_x(row, lane);
        ksrc[i] = (const unsigned char*)(Kp + ((size_t)b * p.Lk_pad + row) * p.ldk + h * 64 + ch * 8);
        vsrc[i] = (const unsigned char*)(Vp + ((size_t)h * 64 + row) * p.ldvt + (size_t)b * p.Lk_pad + ch * 8);
    }
    const size_t kstep = (size_t)ATT_KV * p.ldk * sizeof(T), vstep = (size_t)ATT_KV * sizeof(T);
    auto stage = [&](int slot, int tile) {
        unsigned char* ks = gs + slot * 2 * ATT_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            glds16(ksrc[i] + tile * kstep, ks + (i * 32 + qg * 8) * 128);
            glds16(vsrc[i] + tile * vstep, ks + ATT_TILE_BYTES + (i * 32 + qg * 8) * 128);
        }
    };
    stage(0, t0);
    for (int t = 0; t < tg; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's part of tile t (and, at t = 0, its Q fragments)
        __builtin_amdgcn_s_barrier();                          // ... everyone's; every wave is past its reads of tile t - 1
        asm volatile("" ::: "memory");
        if (t + 1 < tg) stage((t + 1) & 1, t0 + t + 1);        // a whole tile of compute (x 4 waves per SIMD) to land in
        const unsigned char* ks = gs + (t & 1) * 2 * ATT_TILE_BYTES;
        const unsigned char* vs = ks + ATT_TILE_BYTES;
        f32x16 st[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            v8 kf[4];
#pragma unroll
            for (int sd = 0; sd < 4; ++sd) kf[sd] = *(const v8*)(ks + att_k_off(lane, kt, sd));
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
            for (int sd = 0; sd < 4; ++sd) st[kt] = mfma32(kf[sd], qf[sd], st[kt]);
        }
        float mx = max3f(mfma_first_max(st[0][0], st[1][0]), st[0][1], st[1][1]);
        mx = fmaxf(mx, st[0][2]);
#pragma unroll
        for (int r = 2; r < 15; ++r) mx = max3f(mx, st[1][r], st[0][r + 1]);
        mx = fmaxf(mx, st[1][15]);
        mx = xhalf_max(mx);
        const float m_new = fmaxf(m_run, mx);
        if (__any((m_new - m_run) * c > defer)) {              // wave-uniform; m_run = NEG_BIG at t = 0 always takes it (alpha = 0)
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            m_run = m_new;
        }
        const f32x2 c2 = {c, c};
        const f32x2 nmc2 = {-m_run * c, -m_run * c};
        f32x2 ps2 = {0.f, 0.f};
        v8 pf[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 s2 = {st[kt][r], st[kt][r + 1]};
                const f32x2 e2 = __builtin_elementwise_fma(s2, c2, nmc2);
                const f32x2 p2 = {__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
                ps2 += p2;
                pf[kt][r >> 3][r & 7] = from_f32<T>(p2[0]);
                pf[kt][r >> 3][(r & 7) + 1] = from_f32<T>(p2[1]);
            }
        l_run += ps2[0] + ps2[1];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                v8 vf[2];
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) vf[dt] = *(const v8*)(vs + att_v_off(lane, dt, kt, s2));
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) o[dt] = mfma32(vf[dt], pf[kt][s2], o[dt]);
            }
        asm volatile("" ::: "memory");
    }
}

// Merge of the two key halves: group 1 parks (m, l, O) of its 128 query rows in LDS (lane-linear: register-major, lane-minor,
// conflict-free), group 0 combines them with its own in a fixed order -- m = max(m0, m1), l = l0 a0 + l1 a1, O = O0 a0 + O1 a1
// with a_i = 2^((m_i - m) c) -- and leaves fin = O / l.  `mb` = 4 x 8.5 KB (dead ring space; the caller's barrier before this
// call guarantees nobody reads the rings any more).  Every wave of the workgroup must call it (one barrier inside).
__device__ __forceinline__ void attn_ks_merge(float* mb, const float c, const int g, const int qg, const int lane, f32x16 (&o)[2],
                                              const float m_run, const float l_run, f32x16 (&fin)[2]) {
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    float* w = mb + qg * (34 * 64);
    if (g == 1) {
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
    if (g == 0) {
        const float m1 = w[32 * 64 + lane], l1 = w[33 * 64 + lane];
        const float m = fmaxf(m_run, m1);
        const float a0 = __builtin_amdgcn_exp2f((m_run - m) * c), a1 = __builtin_amdgcn_exp2f((m1 - m) * c);
        const float inv = 1.0f / (l_tot * a0 + l1 * a1);
        const float f0 = a0 * inv, f1 = a1 * inv;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) fin[dt][r] = o[dt][r] * f0 + w[(dt * 16 + r) * 64 + lane] * f1;
    }
}

// O tile through the wave's private staging rows of smem (rows wave*32 ..): lane (q, hi) owns d = dt*32 + 8*rg + 4*hi + e,
// written as 8-B pieces, read back as full 128-B rows (8 lanes x 16 B) and stored coalesced.  The caller guarantees that
// no wave still reads the ring (attn_core ends with a workgroup barrier).
template <typename T, int NW>
__device__ __forceinline__ void attn_store(const AttnParams& p, unsigned char* qs, const f32x16 (&fin)[2], const int b,
                                           const int h, const int q0, const int wave, const int lane) {
    typedef typename Vec<T>::v8 v8;
    typedef typename Vec<T>::v4 v4;
    const int l32 = lane & 31;
    const int hi = lane >> 5;
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
                st16_wt((v8*)((T*)p.O + ((size_t)b * p.Lq + q0 + r2) * p.ldo + h * 64 + ch * 8), o8);      // (8 lanes = one 128-B line)
        }
    }
}

}  // namespace imh
