// Host-side lane-level emulation of the gfx950 GEMM and attention kernels' data movement.
// It re-runs the LDS staging, fragment gathers, MFMA (documented lane<->element maps) and the
// accumulator->output maps with the SAME index functions the kernels use (csrc/imh_layout.h),
// and checks the result against a plain matmul / softmax-attention.  Catches swizzle, fragment
// and C/D-layout mistakes on a CPU.  Build: g++ -O2 -std=c++17 emu_layout.cpp -o emu_layout
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../imagharmony_amd/csrc/imh_layout.h"
using namespace imh;

static float frand() { return (float)((rand() % 17) - 8); }

// ---- MFMA emulation -------------------------------------------------------------------
// 16x16x32: a[l][e], b[l][e] (e<8); acc[l][r] (r<4)
static void mfma16(const float a[64][8], const float b[64][8], float acc[64][4]) {
    float A[16][32], B[32][16];
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 8; ++e) { A[l & 15][8 * (l >> 4) + e] = a[l][e]; B[8 * (l >> 4) + e][l & 15] = b[l][e]; }
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * (l >> 4) + r, col = l & 15;
            float s = 0;
            for (int k = 0; k < 32; ++k) s += A[row][k] * B[k][col];
            acc[l][r] += s;
        }
}
// 32x32x16: acc[l][r] (r<16)
static void mfma32(const float a[64][8], const float b[64][8], float acc[64][16]) {
    float A[32][16], B[16][32];
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 8; ++e) { A[l & 31][8 * (l >> 5) + e] = a[l][e]; B[8 * (l >> 5) + e][l & 31] = b[l][e]; }
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
            float s = 0;
            for (int k = 0; k < 16; ++k) s += A[row][k] * B[k][col];
            acc[l][r] += s;
        }
}

// an LDS tile at element granularity (2-byte elements): byte offset -> element index
struct Lds { std::vector<float> e; Lds(int bytes) : e(bytes / 2, NAN) {} float* at(int byte_off) { return &e[byte_off / 2]; } };

static int test_gemm(int BM, int BN, int K, bool vt_perm) {
    const int FM = BM / 32, FN = BN / 32, RX = BM / 32, RW = BN / 32;
    std::vector<float> X(BM * K), W(BN * K), Y(BM * BN, NAN), R(BM * BN, 0.f);
    for (auto& v : X) v = frand();
    for (auto& v : W) v = frand();
    for (int m = 0; m < BM; ++m) for (int n = 0; n < BN; ++n) { float s = 0; for (int k = 0; k < K; ++k) s += X[m * K + k] * W[n * K + k]; R[m * BN + n] = s; }
    std::vector<std::vector<float>> acc(4 * 64, std::vector<float>(FM * FN * 4, 0.f));
    for (int kt = 0; kt < K / 64; ++kt) {
        Lds xs(BM * 128), ws(BN * 128);
        for (int wave = 0; wave < 4; ++wave) for (int lane = 0; lane < 64; ++lane) {
            for (int i = 0; i < RX; ++i) {
                const int row = stage_row(i, wave, lane), c = stage_chunk_x(row, lane);
                float* d = xs.at(stage_lds_off(i, wave) + lane * 16);
                for (int e = 0; e < 8; ++e) d[e] = X[row * K + kt * 64 + c * 8 + e];
            }
            for (int i = 0; i < RW; ++i) {
                const int row = stage_row(i, wave, lane), c = stage_chunk_w(row, lane, FN);
                float* d = ws.at(stage_lds_off(i, wave) + lane * 16);
                for (int e = 0; e < 8; ++e) d[e] = W[row * K + kt * 64 + c * 8 + e];
            }
        }
        for (int wave = 0; wave < 4; ++wave) {
            const int wm = wave >> 1, wn = wave & 1;
            for (int kk = 0; kk < 2; ++kk)
                for (int i = 0; i < FM; ++i) for (int j = 0; j < FN; ++j) {
                    float a[64][8], b[64][8], c4[64][4];
                    for (int lane = 0; lane < 64; ++lane) {
                        const float* xf = xs.at(xfrag_off(lane, wm, BM, kk) + i * 16 * 128);
                        const float* wf = ws.at(wfrag_off(lane, wn, BN, kk) + j * 4 * 128);
                        for (int e = 0; e < 8; ++e) { a[lane][e] = wf[e]; b[lane][e] = xf[e]; }   // weights = MFMA A operand
                        for (int r = 0; r < 4; ++r) c4[lane][r] = acc[wave * 64 + lane][(i * FN + j) * 4 + r];
                    }
                    mfma16(a, b, c4);
                    for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 4; ++r) acc[wave * 64 + lane][(i * FN + j) * 4 + r] = c4[lane][r];
                }
        }
    }
    for (int wave = 0; wave < 4; ++wave) for (int lane = 0; lane < 64; ++lane) {
        const int wm = wave >> 1, wn = wave & 1;
        for (int i = 0; i < FM; ++i) {
            const int m = out_row(lane, wm, BM, i), nb = out_col(lane, wn, BN);
            float v[16];
            for (int j = 0; j < FN; ++j) for (int r = 0; r < 4; ++r) v[j * 4 + r] = acc[wave * 64 + lane][(i * FN + j) * 4 + r];
            if (vt_perm && FN == 4) for (int r = 0; r < 4; ++r) std::swap(v[4 + r], v[8 + r]);
            for (int q = 0; q < 4 * FN; ++q) Y[m * BN + nb + q] = v[q];
        }
    }
    int bad = 0;
    for (int m = 0; m < BM; ++m) for (int n = 0; n < BN; ++n) {
        const int nn = vt_perm ? (n & ~15) | vt_perm16(n & 15) : n;   // logical column n lands at stored position nn
        if (!(Y[m * BN + nn] == R[m * BN + n])) ++bad;
    }
    printf("gemm %dx%d K=%d vt_perm=%d: %s (%d mismatches)\n", BM, BN, K, (int)vt_perm, bad ? "FAIL" : "ok", bad);
    return bad;
}

// ---- wave-specialised GEMM (gemm_ws_kernel, gemm_ring.hip): NL loader waves stage rows q*8 + (lane>>3) of the concatenated
// [token rows; weight rows] stage (q = k*NL + pw), CM x CN consumer waves read FM + FN fragments per k step; the statistics
// waves read token row r chunk ((c + lane) & 7).  Checks: every stage byte written exactly once, product == matmul, the
// row sums == plain sums, and the ds_read_b128 bank model (lane groups of the guide's table) is conflict-free for the
// fragment reads.
static int ds_read_b128_conflicts(const int addr[64]) {
    static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                      {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                      {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                      {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    int extra = 0;
    for (auto& g : groups) {
        int cnt[64] = {};
        for (int l : g) for (int d = 0; d < 4; ++d) ++cnt[((addr[l] >> 2) + d) & 63];
        int mx = 0;
        for (int b = 0; b < 64; ++b) mx = std::max(mx, cnt[b]);
        extra += mx - 1;
    }
    return extra;
}

static int test_gemm_ws(int BM, int BN, int CM, int CN, int NL, int K) {
    const int TM = BM / CM, TN = BN / CN, FM = TM / 16, FN = TN / 16, NI = (BM + BN) / 8, LP = NI / NL, NC = CM * CN;
    const int XT = BM * 128, STAGE = (BM + BN) * 128;
    std::vector<float> X(BM * K), W(BN * K), Y(BM * BN, NAN), R(BM * BN, 0.f);
    for (auto& v : X) v = frand();
    for (auto& v : W) v = frand();
    for (int m = 0; m < BM; ++m) for (int n = 0; n < BN; ++n) { float s = 0; for (int k = 0; k < K; ++k) s += X[m * K + k] * W[n * K + k]; R[m * BN + n] = s; }
    std::vector<std::vector<float>> acc(NC * 64, std::vector<float>(FM * FN * 4, 0.f));
    std::vector<double> rs(BM, 0.0), rq(BM, 0.0);
    int bad = 0, conflicts = 0;
    for (int kt = 0; kt < K / 64; ++kt) {
        Lds st(STAGE);
        std::vector<int> written(STAGE / 16, 0);
        for (int pw = 0; pw < NL; ++pw) for (int k = 0; k < LP; ++k) for (int lane = 0; lane < 64; ++lane) {
            const int q = k * NL + pw, r = q * 8 + (lane >> 3);
            const int off = q * 8 * 128 + lane * 16;                      // M0 = stage + q*8*128, + lane*16
            ++written[off / 16];
            float* d = st.at(off);
            if (r < BM) { const int c = stage_chunk_x(r, lane); for (int e = 0; e < 8; ++e) d[e] = X[r * K + kt * 64 + c * 8 + e]; }
            else { const int row = r - BM, c = stage_chunk_w(row, lane, FN); for (int e = 0; e < 8; ++e) d[e] = W[row * K + kt * 64 + c * 8 + e]; }
        }
        for (int w : written) if (w != 1) ++bad;
        for (int wave = 0; wave < NC; ++wave) {
            const int wm = wave / CN, wn = wave % CN;
            for (int kk = 0; kk < 2; ++kk) {
                int xo[64], wo[64];
                for (int lane = 0; lane < 64; ++lane) {
                    const int xr = wm * TM + (lane & 15), wr = wn * TN + w_frag_row(lane & 15, 0, FN);
                    xo[lane] = tile_off(xr, kk * 4 + (lane >> 4), swz_x(xr));
                    wo[lane] = XT + tile_off(wr, kk * 4 + (lane >> 4), swz_w(wr, FN));
                }
                for (int i = 0; i < FM; ++i) { int a[64]; for (int l = 0; l < 64; ++l) a[l] = xo[l] + i * 16 * 128; conflicts += ds_read_b128_conflicts(a); }
                for (int j = 0; j < FN; ++j) { int a[64]; for (int l = 0; l < 64; ++l) a[l] = wo[l] + j * 4 * 128; conflicts += ds_read_b128_conflicts(a); }
                for (int i = 0; i < FM; ++i) for (int j = 0; j < FN; ++j) {
                    float a[64][8], b[64][8], c4[64][4];
                    for (int lane = 0; lane < 64; ++lane) {
                        const float* xf = st.at(xo[lane] + i * 16 * 128);
                        const float* wf = st.at(wo[lane] + j * 4 * 128);
                        for (int e = 0; e < 8; ++e) { a[lane][e] = wf[e]; b[lane][e] = xf[e]; }
                        for (int r = 0; r < 4; ++r) c4[lane][r] = acc[wave * 64 + lane][(i * FN + j) * 4 + r];
                    }
                    mfma16(a, b, c4);
                    for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 4; ++r) acc[wave * 64 + lane][(i * FN + j) * 4 + r] = c4[lane][r];
                }
            }
        }
        for (int row = 0; row < BM; ++row) {                              // statistics thread of this row, lane = row & 63
            const int lane = row & 63;
            for (int c = 0; c < 8; ++c) {
                const float* f = st.at(row * 128 + (((c + (lane >> 1)) & 7) << 4));
                for (int e = 0; e < 8; ++e) { rs[row] += f[e]; rq[row] += (double)f[e] * f[e]; }
            }
        }
    }
    for (int wave = 0; wave < NC; ++wave) for (int lane = 0; lane < 64; ++lane) {
        const int wm = wave / CN, wn = wave % CN;
        for (int i = 0; i < FM; ++i) {
            const int m = wm * TM + i * 16 + (lane & 15), nb = wn * TN + (lane >> 4) * 4 * FN;
            for (int j = 0; j < FN; ++j) for (int r = 0; r < 4; ++r) Y[m * BN + nb + j * 4 + r] = acc[wave * 64 + lane][(i * FN + j) * 4 + r];
        }
    }
    for (int i = 0; i < BM * BN; ++i) if (!(Y[i] == R[i])) ++bad;
    for (int m = 0; m < BM; ++m) {
        double s = 0, q = 0;
        for (int k = 0; k < K; ++k) { s += X[m * K + k]; q += (double)X[m * K + k] * X[m * K + k]; }
        if (s != rs[m] || q != rq[m]) ++bad;
    }
    int stat_extra = 0;                                                   // the row-per-lane statistics reads (lane = row & 63)
    for (int c = 0; c < 8; ++c) { int a[64]; for (int l = 0; l < 64; ++l) a[l] = l * 128 + (((c + (l >> 1)) & 7) << 4); stat_extra += ds_read_b128_conflicts(a); }
    printf("gemm_ws %dx%d consumers %dx%d loaders %d K=%d: %s (%d mismatches, %d extra LDS cycles on fragment reads, %d on the "
           "statistics reads)\n", BM, BN, CM, CN, NL, K, (bad || conflicts || stat_extra) ? "FAIL" : "ok", bad, conflicts, stat_extra);
    return bad + conflicts + stat_extra;
}

static int test_attention(int Lk_valid, int Lk_pad) {
    const int D = 64, NQ = 128;
    std::vector<float> Q(NQ * D), Kx(Lk_pad * D, 0.f), V(Lk_pad * D, 0.f), VT(D * Lk_pad, 0.f), O(NQ * D, NAN), R(NQ * D);
    for (auto& v : Q) v = frand() * 0.25f;
    for (int k = 0; k < Lk_valid; ++k) for (int d = 0; d < D; ++d) { Kx[k * D + d] = frand() * 0.25f; V[k * D + d] = frand(); }
    for (int k = 0; k < Lk_pad; ++k) for (int d = 0; d < D; ++d) VT[d * Lk_pad + ((k & ~15) | vt_perm16(k & 15))] = V[k * D + d];
    const float scale = 0.125f;
    for (int q = 0; q < NQ; ++q) {
        std::vector<double> s(Lk_valid); double mx = -1e30, l = 0;
        for (int k = 0; k < Lk_valid; ++k) { double a = 0; for (int d = 0; d < D; ++d) a += Q[q * D + d] * Kx[k * D + d]; s[k] = a * scale; mx = std::max(mx, s[k]); }
        for (int k = 0; k < Lk_valid; ++k) { s[k] = std::exp(s[k] - mx); l += s[k]; }
        for (int d = 0; d < D; ++d) { double a = 0; for (int k = 0; k < Lk_valid; ++k) a += s[k] * V[k * D + d]; R[q * D + d] = (float)(a / l); }
    }
    const float c = scale * 1.4426950408889634f;
    for (int wave = 0; wave < 4; ++wave) {
        float qf[4][64][8];
        for (int lane = 0; lane < 64; ++lane) for (int sd = 0; sd < 4; ++sd) for (int e = 0; e < 8; ++e)
            qf[sd][lane][e] = Q[(wave * 32 + (lane & 31)) * D + sd * 16 + (lane >> 5) * 8 + e];
        float o[2][64][16] = {}; float m_run[64], l_run[64];
        for (int l = 0; l < 64; ++l) { m_run[l] = -1e30f; l_run[l] = 0; }
        const int ntiles = (Lk_valid + 63) / 64;
        for (int t = 0; t < ntiles; ++t) {
            Lds ks(64 * 128), vs(64 * 128);
            const int kbase = t * 64;
            for (int w2 = 0; w2 < 4; ++w2) for (int lane = 0; lane < 64; ++lane) for (int i = 0; i < 2; ++i) {
                const int row = stage_row(i, w2, lane), ch = stage_chunk_x(row, lane);
                float* dk = ks.at(stage_lds_off(i, w2) + lane * 16);
                float* dv = vs.at(stage_lds_off(i, w2) + lane * 16);
                for (int e = 0; e < 8; ++e) { dk[e] = Kx[(kbase + row) * D + ch * 8 + e]; dv[e] = VT[row * Lk_pad + kbase + ch * 8 + e]; }
            }
            const bool second = kbase + 32 < Lk_valid;
            float st[2][64][16] = {};
            for (int kt = 0; kt < 2; ++kt) {
                if (kt == 1 && !second) continue;
                for (int sd = 0; sd < 4; ++sd) {
                    float a[64][8];
                    for (int lane = 0; lane < 64; ++lane) { const float* kf = ks.at(att_k_off(lane, kt, sd)); for (int e = 0; e < 8; ++e) a[lane][e] = kf[e]; }
                    mfma32(a, qf[sd], st[kt]);
                }
            }
            float pf[2][2][64][8];
            for (int lane = 0; lane < 64; ++lane) {
                const int hi = lane >> 5;
                float mx = -1e30f;
                for (int kt = 0; kt < 2; ++kt) for (int r = 0; r < 16; ++r) {
                    const int key = kbase + kt * 32 + st_key(r, hi);
                    float s = st[kt][lane][r] * c; s = key < Lk_valid ? s : -1e30f; st[kt][lane][r] = s; mx = std::max(mx, s);
                }
                m_run[lane] = mx;   // per-lane partial maximum of this tile; combined across halves below
            }
            static float run_m[4][64];
            if (t == 0) for (int l = 0; l < 64; ++l) run_m[wave][l] = -1e30f;
            float tile_m[64];
            for (int lane = 0; lane < 64; ++lane) tile_m[lane] = std::max(m_run[lane], m_run[lane ^ 32]);
            for (int lane = 0; lane < 64; ++lane) {
                const float m_new = std::max(run_m[wave][lane], tile_m[lane]);
                const float alpha = std::exp2(run_m[wave][lane] - m_new);
                run_m[wave][lane] = m_new;
                float psum = 0;
                for (int kt = 0; kt < 2; ++kt) for (int r = 0; r < 16; ++r) { const float pv = std::exp2(st[kt][lane][r] - m_new); psum += pv; pf[kt][r >> 3][lane][r & 7] = pv; }
                l_run[lane] = l_run[lane] * alpha + psum;
                for (int dt = 0; dt < 2; ++dt) for (int r = 0; r < 16; ++r) o[dt][lane][r] *= alpha;
            }
            for (int kt = 0; kt < 2; ++kt) {
                if (kt == 1 && !second) continue;
                for (int s = 0; s < 2; ++s) for (int dt = 0; dt < 2; ++dt) {
                    float a[64][8];
                    for (int lane = 0; lane < 64; ++lane) { const float* vf = vs.at(att_v_off(lane, dt, kt, s)); for (int e = 0; e < 8; ++e) a[lane][e] = vf[e]; }
                    mfma32(a, pf[kt][s], o[dt]);
                }
            }
        }
        for (int lane = 0; lane < 64; ++lane) {
            const float l_tot = l_run[lane] + l_run[lane ^ 32];
            const int q = wave * 32 + (lane & 31), hi = lane >> 5;
            for (int dt = 0; dt < 2; ++dt) for (int r = 0; r < 16; ++r) O[q * D + att_o_dim(dt, r, hi)] = o[dt][lane][r] / l_tot;
        }
    }
    int bad = 0; double worst = 0;
    for (int i = 0; i < NQ * D; ++i) { const double e = std::fabs((double)O[i] - R[i]); if (!(e < 1e-4)) ++bad; if (e > worst) worst = e; }
    printf("attention Lk=%d pad=%d: %s (max err %.2e, %d bad)\n", Lk_valid, Lk_pad, bad ? "FAIL" : "ok", worst, bad);
    return bad;
}

// fused cross-attention prologue (xattn.hip): project Q^T = Wq_h X^T through the kernel's staging / fragment maps,
// hand the accumulators over as QK^T B operands, and form S^T against a K tile stored in vt_perm16 order.
static int test_xattn_handover(int C) {
    const int D = 64, NQ = 128, NK = 64;
    std::vector<float> X(NQ * C), Wq(D * C), Kx(NK * D), KP(NK * D), R(NQ * NK), S(NQ * NK, NAN);
    for (auto& v : X) v = frand() * 0.5f;
    for (auto& v : Wq) v = frand() * 0.5f;
    for (auto& v : Kx) v = frand();
    for (int k = 0; k < NK; ++k) for (int d = 0; d < D; ++d) KP[k * D + ((d & ~15) | vt_perm16(d & 15))] = Kx[k * D + d];
    for (int q = 0; q < NQ; ++q) for (int k = 0; k < NK; ++k) {
        double a = 0;
        for (int d = 0; d < D; ++d) { double qd = 0; for (int c = 0; c < C; ++c) qd += (double)X[q * C + c] * Wq[d * C + c]; a += qd * Kx[k * D + d]; }
        R[q * NK + k] = (float)a;
    }
    std::vector<std::vector<float>> qa(4 * 64, std::vector<float>(32, 0.f));
    for (int kt = 0; kt < C / 64; ++kt) {
        Lds xs(128 * 128), ws(64 * 128);
        for (int wave = 0; wave < 4; ++wave) for (int lane = 0; lane < 64; ++lane) {
            for (int i = 0; i < 4; ++i) {
                const int row = xq_stage_xrow(i, wave, lane), ch = stage_chunk_x(row, lane);
                float* d = xs.at((wave * 32 + i * 8) * 128 + lane * 16);
                for (int e = 0; e < 8; ++e) d[e] = X[row * C + kt * 64 + ch * 8 + e];
            }
            for (int i = 0; i < 2; ++i) {
                const int row = xq_stage_wrow(i, wave, lane), ch = stage_chunk_x(row, lane);
                float* d = ws.at((wave * 16 + i * 8) * 128 + lane * 16);
                for (int e = 0; e < 8; ++e) d[e] = Wq[row * C + kt * 64 + ch * 8 + e];
            }
        }
        for (int wave = 0; wave < 4; ++wave) for (int ks = 0; ks < 4; ++ks) for (int dt = 0; dt < 2; ++dt) {
            float a[64][8], b[64][8], c16[64][16];
            for (int lane = 0; lane < 64; ++lane) {
                const float* xf = xs.at(xq_x_off(wave, lane, ks));
                const float* wf = ws.at(xq_w_off(dt, lane, ks));
                for (int e = 0; e < 8; ++e) { a[lane][e] = wf[e]; b[lane][e] = xf[e]; }
                for (int r = 0; r < 16; ++r) c16[lane][r] = qa[wave * 64 + lane][dt * 16 + r];
            }
            mfma32(a, b, c16);
            for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 16; ++r) qa[wave * 64 + lane][dt * 16 + r] = c16[lane][r];
        }
    }
    // K tile staged exactly as attn_core stages it (from the PERMUTED cache)
    Lds ks(64 * 128);
    for (int w2 = 0; w2 < 4; ++w2) for (int lane = 0; lane < 64; ++lane) for (int i = 0; i < 2; ++i) {
        const int row = stage_row(i, w2, lane), ch = stage_chunk_x(row, lane);
        float* dk = ks.at(stage_lds_off(i, w2) + lane * 16);
        for (int e = 0; e < 8; ++e) dk[e] = KP[row * D + ch * 8 + e];
    }
    for (int wave = 0; wave < 4; ++wave) {
        float qf[4][64][8];
        for (int lane = 0; lane < 64; ++lane) for (int dt = 0; dt < 2; ++dt) for (int r = 0; r < 16; ++r)
            qf[xq_sd(dt, r)][lane][xq_slot(r)] = qa[wave * 64 + lane][dt * 16 + r];
        for (int kt = 0; kt < 2; ++kt) {
            float st[64][16] = {};
            for (int sd = 0; sd < 4; ++sd) {
                float a[64][8];
                for (int lane = 0; lane < 64; ++lane) { const float* kf = ks.at(att_k_off(lane, kt, sd)); for (int e = 0; e < 8; ++e) a[lane][e] = kf[e]; }
                mfma32(a, qf[sd], st);
            }
            for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 16; ++r)
                S[(wave * 32 + (lane & 31)) * NK + kt * 32 + st_key(r, lane >> 5)] = st[lane][r];
        }
    }
    int bad = 0;
    for (int i = 0; i < NQ * NK; ++i) if (!(std::fabs((double)S[i] - R[i]) <= 1e-3 * (1.0 + std::fabs((double)R[i])))) ++bad;
    printf("xattn hand-over C=%d: %s (%d mismatches)\n", C, bad ? "FAIL" : "ok", bad);
    return bad;
}

int main() {
    srand(1234);
    int bad = 0;
    for (int bm : {128, 64}) for (int bn : {128, 64}) bad += test_gemm(bm, bn, 128, false);
    bad += test_gemm(128, 128, 64, true);
    bad += test_gemm(64, 128, 64, true);
    bad += test_gemm_ws(64, 160, 2, 2, 2, 128);     // 1464: two loaders
    bad += test_gemm_ws(64, 160, 2, 2, 4, 128);     // 2464
    bad += test_gemm_ws(128, 160, 2, 2, 4, 128);    // 24128 x 160
    bad += test_gemm_ws(128, 128, 2, 2, 4, 64);     // 24128 x 128
    bad += test_gemm_ws(256, 160, 4, 2, 4, 128);    // 23256
    bad += test_gemm_ws(256, 160, 4, 2, 2, 64);     // 23256 with the folded LayerNorm: two loaders + two statistics waves
    bad += test_attention(64, 64);
    bad += test_attention(128, 128);
    bad += test_attention(77, 128);
    bad += test_attention(4, 64);
    bad += test_attention(33, 64);
    bad += test_xattn_handover(128);
    bad += test_xattn_handover(320);
    return bad ? 1 : 0;
}
