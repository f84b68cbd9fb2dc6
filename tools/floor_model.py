"""Per-shape floors of the GEMM / implicit-GEMM launches of the recorded forward against the measured time of the tuned
variant (CPU only; reads a sweep log).  Model per CU, per 64-wide K tile of a BM x BN workgroup tile:
  front end : every 16-B piece of both operand tiles passes the vector-memory address unit at 64 B / clk / CU
              -> (BM + BN) / 8 wave instructions x 16 cycles          (LDS-halo conv: the halo is staged once per 64 channels)
  MFMA      : 2 BM BN 64 flop / (4 SIMDs x 1024 flop / clk)          = BM BN / 32 cycles
A launch cannot be shorter than  tiles-per-CU x K-tiles x max(front end, MFMA)  at 2.4 GHz; the ratio measured / floor says
where the headroom is (launch + prologue + epilogue, ~4-5 us, are NOT in the floor).
usage: python tools/floor_model.py profiles/r02_ws_incremental_sweep.log > profiles/r02_floor_model.md"""
import re
import sys

TILE = {64: (64, None), 128: (128, None), 256: (256, None), 3064: (64, 64), 3128: (128, 128), 4064: (64, None), 4128: (128, None),
        5064: (64, 64), 5256: (256, 320), 5258: (256, 320), 6128: (128, 320), 6064: (64, 160), 7064: (64, 160), 8256: (256, 256),
        9128: (128, 320), 9256: (256, 320), 1464: (64, 160), 2464: (64, 160), 24128: (128, None), 23256: (256, 160)}
HALO = {7128: 128, 7564: 64}            # output pixels per workgroup (8 x 16 / 4 x 16 patch); bn = couts


def floors(M, N, K, conv, bm, bn, splits):
    if bm in HALO:
        px = HALO[bm]
        tiles = -(-M // px) * -(-N // bn)
        halo_rows = (10 * 18) if px == 128 else (6 * 18)
        chunks = K // 9 // 64
        fe = chunks * (halo_rows / 8 * 16 + 9 * bn / 8 * 16)
        mf = chunks * 9 * px * bn / 32
        steps = 1
    else:
        BM, BN = TILE[bm]
        BN = BN or bn
        tiles = -(-M // BM) * -(-N // BN) * splits
        steps = K // 64 / splits
        fe = (BM + BN) / 8 * 16
        mf = BM * BN / 32
    per_cu = -(-tiles // 256)
    return per_cu * steps * fe / 2400.0, per_cu * steps * mf / 2400.0, tiles


def main():
    rows = []
    pat = re.compile(r"^\s+(\S+(?: \S+)?)\s+M=\s*(\d+) N=\s*(\d+) K=\s*(\d+) conv=(\d) best \((\d+), (\d+), (\d+)\)\s+([\d.]+) us")
    for line in open(sys.argv[1]):
        m = pat.match(line)
        if m:
            name = m.group(1)
            M, N, K, conv, bm, bn, sp = (int(v) for v in m.groups()[1:8])
            rows.append((name, M, N, K, conv, bm, bn, sp, float(m.group(9))))
    print("# Floors of the tuned GEMM / conv launches (tools/floor_model.py; model in its docstring)\n")
    print("| op | M x N x K | variant | tiles | measured us | front-end floor us | MFMA floor us | measured / max floor |")
    print("|---|---|---|---|---|---|---|---|")
    for name, M, N, K, conv, bm, bn, sp, us in rows:
        if M < 64:
            continue
        fe, mf, tiles = floors(M, N, K, conv, bm, bn, sp)
        print(f"| {name}{' (conv)' if conv else ''} | {M} x {N} x {K} | {bm} x {bn}{'' if sp == 1 else f' / {sp}'} | {tiles} | {us:.1f} | {fe:.1f} | {mf:.1f} | "
              f"{us / max(fe, mf):.2f} |")


if __name__ == "__main__":
    main()
