"""Per-shape floors of the GEMM / implicit-GEMM launches of the recorded forward against the measured time of the tuned
variant (CPU only; reads a sweep log).  Model per CU, per 64-wide K tile of a BM x BN workgroup tile:
  front end : every 16-B piece of both operand tiles passes the vector-memory address unit at 64 B / clk / CU
              -> (BM + BN) / 8 wave instructions x 16 cycles          (LDS-halo conv: the halo is staged once per 64 channels)
  MFMA      : 2 BM BN 64 flop / (4 SIMDs x 1024 flop / clk)          = BM BN / 32 cycles
A launch cannot be shorter than  tiles-per-CU x K-tiles x max(front end, MFMA)  at 2.4 GHz; the ratio measured / floor says
where the headroom is (launch + prologue + epilogue, ~4-5 us, are NOT in the floor).
usage: python tools/floor_model.py profiles/r02_ws_incremental_sweep.log > profiles/r02_floor_model.md
       python tools/floor_model.py gpurun_out/forward_ab.json [configuration] > profiles/r04_floor_model.md"""
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


def main_json(path, name=None):
    """round 4: input = a tools/forward_ab.py JSON (its by_shape list: per-op HIP-event times of the recorded forward, which carry
    ~2 us of event overhead each); adds the 'recoverable' column the round-3 review asked for"""
    import json
    d = json.loads(open(path).read().strip().splitlines()[-1])
    cfgname = name or next(iter(d))
    rows = d[cfgname]["by_shape"]
    print(f"# Floors of the GEMM / conv launches of one forward (tools/floor_model.py on {path.split('/')[-1]}, configuration `{cfgname}`)\n")
    print("model (per CU, per 64-wide K tile of a BM x BN tile): front end = (BM + BN) / 8 LDS-DMA wave instructions x 16 cycles (64 B / clk / CU "
          "address unit; the LDS-halo conv stages its halo once per 64 channels); MFMA = BM BN / 32 cycles; floor = tiles per CU x K tiles x "
          "max(front end, MFMA) at 2.4 GHz.  Launch, prologue, epilogue and the ~2 us of event overhead in `measured` are NOT in the floor; "
          "`recoverable` = (measured - floor) x launches.\n")
    print("| op | M x N x K | variant | tiles | launches | measured us | front-end floor us | MFMA floor us | measured / floor | recoverable ms per forward |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    tot_meas = tot_floor = 0.0
    out = []
    for r in rows:
        bm, bn, sp = r["cfg"]
        if r["M"] < 64 or (bm not in TILE and bm not in HALO and bm not in (7256, 7356, 7328, 7428, 22128)):
            continue
        if bm in (7256, 7356):
            px, halo_rows = 256, 18 * 18
            tiles = -(-r["M"] // px) * -(-r["N"] // bn)
            chunks = r["K"] // 9 // 64
            fe = chunks * (halo_rows / 8 * 16 + 9 * bn / 8 * 16) * -(-tiles // 256) / 2400.0
            mf = chunks * 9 * px * bn / 32 * -(-tiles // 256) / 2400.0
        else:
            b2 = {7328: 7128, 7428: 7128, 22128: 24128}.get(bm, bm)
            fe, mf, tiles = floors(r["M"], r["N"], r["K"], r["conv"], b2, bn, sp)
        fl = max(fe, mf)
        rec_ms = (r["us_each"] - fl) * r["n"] / 1e3
        tot_meas += r["us_each"] * r["n"] / 1e3; tot_floor += fl * r["n"] / 1e3
        out.append((rec_ms, f"| {r['op']}{' (conv)' if r['conv'] else ''} | {r['M']} x {r['N']} x {r['K']} | {bm} x {bn}{'' if sp == 1 else f' / {sp}'} | {tiles} | {r['n']} | "
                            f"{r['us_each']:.1f} | {fe:.1f} | {mf:.1f} | {r['us_each'] / fl:.2f} | {rec_ms:.3f} |"))
    for _, line in sorted(out, key=lambda t: -t[0]):
        print(line)
    print(f"\nsum over the listed launches: measured {tot_meas:.2f} ms, floors {tot_floor:.2f} ms per forward")


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
    if sys.argv[1].endswith(".json"):
        main_json(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
    else:
        main()
