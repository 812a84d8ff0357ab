"""Summarise a rocprofv3 rocpd (sqlite) result: per-kernel stats and PMC counter averages -> text.
usage: python tools/rocpd_summary.py <results.db> [more.db ...]"""
import sqlite3
import sys


def summarise(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print(f"== {path}")
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                       "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(grid_y), max(workgroup_x) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}  vgpr sgpr lds scratch grid wg")
    for r in rows:
        print(f"{r[0][:70]:70s} {r[1]:6d} {r[3] / 1e3:10.2f} {r[4] / 1e3:10.2f} {r[5] / 1e3:10.2f} {100 * r[2] / tot:6.2f}  "
              f"{r[6]} {r[7]} {r[8]} {r[9]} {r[10]}x{r[11]} {r[12]}")
    try:
        pm = cur.execute("select k.name, p.counter_name, count(*), avg(p.value), sum(p.value) from counters_collection p "
                         "join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name").fetchall()
    except sqlite3.Error as e:
        pm = []
        try:
            cols = [c[1] for c in cur.execute('pragma table_info("counters_collection")')]
            print("counters_collection columns:", cols)
        except sqlite3.Error:
            pass
    for r in pm:
        print(f"  PMC {r[0][:50]:50s} {r[1]:24s} n={r[2]:4d} avg={r[3]:.6g}")


for p in sys.argv[1:]:
    summarise(p)
