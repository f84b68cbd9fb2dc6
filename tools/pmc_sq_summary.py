"""Summarise SQ counter passes (rocprofv3 --pmc ... over tools/pmc_gemm.py) into markdown.
Usage: python tools/pmc_sq_summary.py <pass1.db> [<pass2.db> ...] > out.md
Counters come back per shader engine (32 SEs; SQ_INSTS_MFMA x 32 == the analytic MFMA count), cycle counters of
waves are in quad-cycles.  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (32 SIMDs per SE x SQ_BUSY_CYCLES)."""
import sqlite3
import sys

vals = {}
for db in sys.argv[1:]:
    c = sqlite3.connect(db)
    for name, cn, n, avg, dur in c.execute("select name, counter_name, count(*), avg(counter_value), avg(duration) "
                                           "from pmc_events where name like '%imh%' group by name, counter_name"):
        vals.setdefault(name, {})[cn] = avg
        vals[name].setdefault("_us", dur / 1e3)
keys = ["SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_LDS",
        "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"]
print("# SQ counters per launch (rocprofv3 --pmc, two passes, tools/pmc_gemm.py: GEMM 8192x5120x2560 bf16 and "
      "self-attention B=2 H=10 L=4096)\n")
print("| kernel | us | MFMA util | non-MFMA VALU / MFMA | LDS bank-conflict cycles / LDS active | wait-any share of wave time |")
print("|---|---|---|---|---|---|")
for k, v in vals.items():
    busy = v.get("SQ_BUSY_CYCLES", 0)
    util = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (32 * busy) if busy else 0
    mf = v.get("SQ_INSTS_MFMA", 0)
    va = (v.get("SQ_INSTS_VALU", 0) - mf) / mf if mf else 0
    lds = v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1)
    wt = v.get("SQ_WAIT_INST_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1)
    short = k.replace("_ZN3imh", "").replace("NS_10GemmParamsE", "").replace("NS_10AttnParamsE", "")[:60]
    print(f"| `{short}` | {v['_us']:.1f} | {util:.2f} | {va:.2f} | {lds:.4f} | {wt:.2f} |")
print("\nRaw averages per launch:\n")
print("| kernel | " + " | ".join(keys) + " |")
print("|---|" + "---|" * len(keys))
for k, v in vals.items():
    short = k.replace("_ZN3imh", "")[:48]
    print(f"| `{short}` | " + " | ".join(f"{v.get(x, float('nan')):.4g}" for x in keys) + " |")
