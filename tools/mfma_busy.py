"""MFMA utilisation per kernel from a rocprofv3 PMC pass (rocpd sqlite): SQ_VALU_MFMA_BUSY_CYCLES summed over the chip / the MFMA cycles the
launch had available (duration x 1024 SIMDs x clock).  The clock is not in the trace: the fraction is quoted against the 2.4 GHz peak
clock (a LOWER bound of the busy share while the chip runs below it under load).  Also SQ_BUSY_CYCLES where collected.
usage: python tools/mfma_busy.py <results.db> [top_n]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
cur = db.cursor()
rows = cur.execute("select k.name, count(distinct k.dispatch_id), sum(k.duration) from kernels k group by k.name order by sum(k.duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
pm = {}
for name, counter, s in cur.execute("select k.name, p.counter_name, sum(p.value) from counters_collection p join kernels k on k.dispatch_id = p.dispatch_id "
                                    "group by k.name, p.counter_name"):
    pm.setdefault(name, {})[counter] = s
print("MFMA busy share per kernel = SQ_VALU_MFMA_BUSY_CYCLES / (kernel time x 1024 SIMDs x 2.4 GHz); MI355X: 256 CUs x 4 SIMDs")
print(f"{'kernel':72s} {'calls':>6s} {'avg_us':>9s} {'% time':>7s} {'MFMA busy':>10s}")
wsum = 0.0
for name, n, dur in rows[:top]:
    busy = pm.get(name, {}).get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    frac = busy / (dur * 1e-9 * 1024 * 2.4e9) if dur else 0.0
    wsum += busy
    print(f"{name[:72]:72s} {n:6d} {dur / n / 1e3:9.1f} {100 * dur / tot:7.2f} {frac:10.3f}")
allbusy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for v in pm.values())
print(f"whole trace: MFMA busy {allbusy / (tot * 1e-9 * 1024 * 2.4e9):.3f} of the kernel time x 1024 SIMDs x 2.4 GHz")
