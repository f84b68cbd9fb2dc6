"""per-kernel average of every counter in a rocprofv3 --pmc rocpd database: python tools/pmc_any_summary.py <results.db>"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
src = "pmc_events" if "pmc_events" in tabs else None
cols = [r[1] for r in c.execute(f"pragma table_info({src})")]
cn = "counter_name" if "counter_name" in cols else ("name" if "name" in cols else cols[0])
rows = c.execute(f"select name, {('counter_name' if 'counter_name' in cols else 'counter_id')}, count(*), avg(counter_value), avg(duration) from {src} group by 1, 2").fetchall()
for r in rows:
    print(f"{r[0][:95]:95s} {str(r[1]):18s} n={r[2]:3d} avg={r[3]:14.1f} dur_us={r[4] / 1e3:8.1f}")
