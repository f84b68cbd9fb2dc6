"""Kernel durations and launch gaps grouped by (kernel, predecessor kernel) from a rocprofv3 rocpd results.db of the graph-replayed forward:
does the same launch cost more behind one producer than behind another?   python tools/trace_by_predecessor.py <results.db> [name substring]"""
import sqlite3
import sys
from collections import defaultdict


def short(n):
    for k in ("gemm_ws_kernel", "xattn_kernel", "attn_pipe_kernel", "conv_halo_kernel", "gemm_kernel", "gn_apply", "gemm_ring", "splitk", "gn_stats"):
        if k in n:
            if k == "gemm_ws_kernel":
                i = n.find("Li")
                return "ws " + n[i:i + 12]
            return k
    return n[:24]


def main():
    db = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "Li64ELi160"
    c = sqlite3.connect(db)
    ks = c.execute("select name, start, end from kernels where name like '%imh%' order by start").fetchall()
    g = defaultdict(list)
    for i in range(1, len(ks) - 1):
        n, s, e = ks[i]
        if want not in n or s - ks[i - 1][2] > 50000:
            continue
        g[(short(ks[i - 1][0]), short(ks[i + 1][0]))].append((e - s, s - ks[i - 1][2]))
    print(f"launches of kernels matching '{want}' by (predecessor, successor): count, median duration us, median gap before it us")
    for k, v in sorted(g.items(), key=lambda kv: -len(kv[1])):
        d = sorted(x[0] for x in v); gp = sorted(x[1] for x in v)
        print(f"  after {k[0]:28s} before {k[1]:28s} n={len(v):6d}  dur {d[len(d) // 2] / 1e3:7.2f}  gap {gp[len(gp) // 2] / 1e3:6.2f}")


if __name__ == "__main__":
    main()
