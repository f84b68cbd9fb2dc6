"""Turn a rocprofv3 rocpd results.db (kernel trace) into the per-kernel summary committed under profiles/.
Usage: python tools/rocpd_summary.py <results.db> <out.md> [title]"""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else db
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    span = c.execute("select min(start), max(end) from kernels").fetchone()
    with open(out, "w") as f:
        f.write(f"# {title}\n\nsource: rocprofv3 --kernel-trace --stats (rocpd database), all kernel dispatches of the run\n\n")
        f.write(f"total kernel time {tot / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches; "
                f"first-to-last dispatch span {(span[1] - span[0]) / 1e6:.2f} ms\n\n")
        f.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for name, n, s, a, mn, mx in rows:
            short = name if len(name) < 110 else name[:107] + "..."
            f.write(f"| `{short}` | {n} | {s / 1e6:.3f} | {a / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * s / tot:.2f} |\n")
        # idle time between consecutive dispatches of the library's kernels inside back-to-back regions (graph replays of the
        # forward: every kernel waits for its predecessor): gap = start[i + 1] - end[i], pairs further than 50 us apart are
        # region boundaries (host work), not launch gaps
        ks = c.execute("select start, end from kernels where name like '%imh%' order by start").fetchall()
        gaps = sorted(max(0, ks[i + 1][0] - ks[i][1]) for i in range(len(ks) - 1) if ks[i + 1][0] - ks[i][1] < 50000)
        if gaps:
            busy = sum(e - s0 for s0, e in ks)
            f.write(f"\nlaunch gaps between consecutive imh kernels (pairs < 50 us apart): {len(gaps)} pairs, total {sum(gaps) / 1e6:.2f} ms "
                    f"= {100 * sum(gaps) / (busy + sum(gaps)):.1f} % of busy + gap time; median {gaps[len(gaps) // 2] / 1e3:.2f} us, "
                    f"p10 {gaps[len(gaps) // 10] / 1e3:.2f} us, p90 {gaps[len(gaps) * 9 // 10] / 1e3:.2f} us\n")
    print(open(out).read()[:3000])


if __name__ == "__main__":
    main()
