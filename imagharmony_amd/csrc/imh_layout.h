// Index math shared by the gfx950 kernels and the host-side lane-level emulator
// (tests/emu/emu_layout.cpp).  Everything here is plain integer arithmetic, usable
// from host and device, so the LDS swizzles, MFMA fragment coordinates and C/D
// register->(row,col) maps can be checked on a CPU before any GPU time is spent.
//
// MFMA facts used (cdna_hip_programming.md section 3):
//   v_mfma_f32_16x16x32_{bf16,f16}: A lane l: row l&15, k = 8*(l>>4)+0..7
//                                   B lane l: col l&15, k = 8*(l>>4)+0..7
//                                   D lane l reg r: row 4*(l>>4)+r, col l&15
//   v_mfma_f32_32x32x16_{bf16,f16}: A lane l: row l&31, k = 8*(l>>5)+0..7
//                                   B lane l: col l&31, k = 8*(l>>5)+0..7
//                                   D lane l reg r: row (r&3)+8*(r>>2)+4*(l>>5), col l&31
#pragma once

#if defined(__HIPCC__)
#define IMH_HD __host__ __device__ __forceinline__
#else
#define IMH_HD inline
#endif

namespace imh {

// ------------------------------------------------------------------ GEMM tiles
// LDS tile = R rows x 64 elements (128 B per row = 8 chunks of 16 B), filled by
// global_load_lds_dwordx4: a wave instruction writes 64 lanes x 16 B = 8 consecutive
// rows, lane -> (row = lane>>3, physical chunk = lane&7).  The destination is
// lane-linear by construction, so the bank-conflict swizzle is applied on the SOURCE
// chunk (which logical chunk a lane fetches) and again on the fragment READ
// (cdna_hip_programming.md 5.4 rule 21).  phys = logical ^ f(row).
constexpr int GEMM_BK = 64;          // k elements per tile
constexpr int GEMM_ROW_BYTES = 128;  // bf16/f16

// swizzle of the "token" tile (rows are used in natural order by the MFMA B operand)
IMH_HD int swz_x(int row) { return (row >> 1) & 7; }

// The weight tile feeds the MFMA A operand (output rows of D).  To make every lane
// own 4*FN CONSECUTIVE output columns n, MFMA row rho of fragment j is tile row
//     wrow(rho, j) = (rho>>2)*(4*FN) + j*4 + (rho&3)          (within the wave's BN/2 slab)
// so that D reg r of fragment j in lane group g = lane>>4 is column g*4*FN + j*4 + r.
IMH_HD int w_frag_row(int rho, int j, int FN) { return (rho >> 2) * (4 * FN) + j * 4 + (rho & 3); }
// its conflict-free swizzle uses the row bits that vary inside one ds_read_b128 lane group
IMH_HD int swz_w(int row, int FN) { return ((row >> 1) & 1) | (((row / (4 * FN)) & 3) << 1); }

// byte offset inside a tile of (row, logical 16-B chunk c)
IMH_HD int tile_off(int row, int c, int f) { return row * GEMM_ROW_BYTES + ((c ^ f) << 4); }

// --- per-lane coordinates of the GEMM kernel (used verbatim by gemm.hip and by the emulator) ---
// staging: round i, wave w, lane l fills tile row stage_row() at physical chunk l&7 and fetches
// logical chunk stage_chunk_*(); the wave-uniform LDS destination is stage_lds_off() (+ lane*16).
IMH_HD int stage_row(int i, int wave, int lane) { return i * 32 + wave * 8 + (lane >> 3); }
IMH_HD int stage_chunk_x(int row, int lane) { return (lane & 7) ^ swz_x(row); }
IMH_HD int stage_chunk_w(int row, int lane, int FN) { return (lane & 7) ^ swz_w(row, FN); }
IMH_HD int stage_lds_off(int i, int wave) { return (i * 32 + wave * 8) * GEMM_ROW_BYTES; }
// fragment reads: token fragment i adds i*16 rows, weight fragment j adds j*4 rows
IMH_HD int xfrag_off(int lane, int wm, int BM, int kk) {
    const int r = wm * (BM / 2) + (lane & 15);
    return tile_off(r, kk * 4 + (lane >> 4), swz_x(r));
}
IMH_HD int wfrag_off(int lane, int wn, int BN, int kk) {
    const int FN = BN / 32;
    const int r = wn * (BN / 2) + w_frag_row(lane & 15, 0, FN);
    return tile_off(r, kk * 4 + (lane >> 4), swz_w(r, FN));
}
// accumulator acc[i][j][r] of lane l is Y[out_row(l,wm,BM,i)][out_col(l,wn,BN) + j*4 + r]
IMH_HD int out_row(int lane, int wm, int BM, int i) { return wm * (BM / 2) + i * 16 + (lane & 15); }
IMH_HD int out_col(int lane, int wn, int BN) { return wn * (BN / 2) + (lane >> 4) * 4 * (BN / 32); }

// ------------------------------------------------------------ attention tiles
// K tile [64 keys][64 d] and V^T tile [64 d][64 keys(permuted)], same 128-B rows and
// the swz_x swizzle.  V^T keeps keys permuted inside every group of 16 as
// [0-3, 8-11, 4-7, 12-15] so that the 8 k-slots one lane owns after the swapped
// QK^T product (32x32 D layout: rows (r&3)+8*(r>>2)+4*hi) are 16 contiguous bytes.
IMH_HD int vt_perm16(int k) {           // logical key-in-16 -> stored position
    int q = (k >> 2) & 3;               // which run of 4
    int qq = (q == 1) ? 2 : (q == 2 ? 1 : q);
    return (qq << 2) | (k & 3);
}
// key index (within a 32-key sub-tile) held by S^T accumulator register r of lane half hi
IMH_HD int st_key(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// attention per-lane coordinates (attention.hip + emulator); lane l: query column l&31, half l>>5
IMH_HD int att_k_off(int lane, int kt, int sd) {          // K fragment: keys kt*32+(l&31), d = sd*16 + hi*8 ..
    const int row = kt * 32 + (lane & 31);
    return tile_off(row, sd * 2 + (lane >> 5), swz_x(row));
}
IMH_HD int att_v_off(int lane, int dt, int kt, int s) {   // V^T fragment: d = dt*32+(l&31), 8 key slots
    const int row = dt * 32 + (lane & 31);
    return tile_off(row, kt * 4 + 2 * s + (lane >> 5), swz_x(row));
}
// O^T accumulator register r of lane half hi holds head-dim index
IMH_HD int att_o_dim(int dt, int r, int hi) { return dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ------------------------------------------------- fused cross-attention prologue (xattn.hip)
// Q^T[64 d, 32 q] per wave = Wq_h[64, C] X[32 q, C]^T on 32x32x16 MFMAs.  Stage = X tile [128 rows][64 k] (each wave
// fills and reads only its own 32 rows) followed by the Wq tile [64 rows][64 k] (filled by quarters, read by everyone).
IMH_HD int xq_stage_xrow(int i, int wave, int lane) { return wave * 32 + i * 8 + (lane >> 3); }   // i < 4
IMH_HD int xq_stage_wrow(int i, int wave, int lane) { return wave * 16 + i * 8 + (lane >> 3); }   // i < 2
IMH_HD int xq_x_off(int wave, int lane, int ks) {          // B operand: query (lane&31), k = ks*16 + hi*8 ..
    const int row = wave * 32 + (lane & 31);
    return tile_off(row, ks * 2 + (lane >> 5), swz_x(row));
}
IMH_HD int xq_w_off(int dt, int lane, int ks) {            // A operand: head dim dt*32 + (lane&31), same k
    const int row = dt * 32 + (lane & 31);
    return tile_off(row, ks * 2 + (lane >> 5), swz_x(row));
}
// accumulator register r of 32x32 block dt becomes slot xq_slot(r) of the Q^T B-operand fragment of QK^T step
// xq_sd(dt, r): lane half hi then holds head dims {0-3, 8-11} + 4 hi of the step's 16-group, which is the order
// vt_perm16 stores a K row in -- so the matching K fragment is chunk sd*2 + hi of the permuted row (att_k_off)
IMH_HD int xq_sd(int dt, int r) { return dt * 2 + (r >> 3); }
IMH_HD int xq_slot(int r) { return r & 7; }

}  // namespace imh
