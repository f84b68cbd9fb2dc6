"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes).  gfx950 correction: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced
streams -> doubled; WRITE_SIZE is uncalibrated (reported as is).  Counter unit: KiB.
Usage: python tools/pmc_summary.py <fetch.db> <write.db> <out.json> [<out.md>]"""
import json
import sqlite3
import sys


def per_kernel(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(counter_value), avg(counter_value), avg(duration) from pmc_events "
                     "group by name").fetchall()
    return {r[0]: dict(n=r[1], total=r[2], avg=r[3], avg_ns=r[4]) for r in rows}


def main():
    f, w = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    out = {}
    for k in f:
        fe = f[k]["avg"] * 1024 * 2.0                    # KiB -> bytes, x2 gfx950 correction
        wr = w.get(k, {"avg": 0.0})["avg"] * 1024
        out[k] = dict(launches=f[k]["n"], fetch_bytes_per_launch=fe, write_bytes_per_launch=wr,
                      hbm_bytes_per_launch=fe + wr, avg_us_profiled=f[k]["avg_ns"] / 1e3)
    gemm = {k: v for k, v in out.items() if any(t in k for t in ("gemm_kernel", "gemm_kg2", "gemm_ring", "gemm_dual", "gemm_pp", "gemm_pq", "gemm_pr", "gemm_ws", "gemm_w16", "gemm_f8", "conv_halo", "conv_hws"))}
    n = sum(v["launches"] for v in gemm.values())
    fam = dict(launches=n,
               fetch_bytes_per_launch=sum(v["fetch_bytes_per_launch"] * v["launches"] for v in gemm.values()) / n,
               write_bytes_per_launch=sum(v["write_bytes_per_launch"] * v["launches"] for v in gemm.values()) / n)
    fam["hbm_bytes_per_launch"] = fam["fetch_bytes_per_launch"] + fam["write_bytes_per_launch"]
    res = dict(note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950), WRITE_SIZE uncalibrated",
               gemm_family=fam, kernels=out)
    json.dump(res, open(sys.argv[3], "w"), indent=1)
    if len(sys.argv) > 4:
        with open(sys.argv[4], "w") as md:
            md.write("# HBM traffic per launch (rocprofv3 PMC, bench.py --denoise-steps 4)\n\n" + res["note"] + "\n\n")
            md.write(f"GEMM family: {fam['launches']} launches, fetch {fam['fetch_bytes_per_launch']/1e6:.2f} MB + write "
                     f"{fam['write_bytes_per_launch']/1e6:.2f} MB = {fam['hbm_bytes_per_launch']/1e6:.2f} MB per launch\n\n")
            md.write("| kernel | launches | fetch MB | write MB | avg us (profiled) |\n|---|---|---|---|---|\n")
            for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:25]:
                md.write(f"| `{k[:90]}` | {v['launches']} | {v['fetch_bytes_per_launch']/1e6:.2f} | "
                         f"{v['write_bytes_per_launch']/1e6:.2f} | {v['avg_us_profiled']:.1f} |\n")
    print(json.dumps(fam))


if __name__ == "__main__":
    main()
