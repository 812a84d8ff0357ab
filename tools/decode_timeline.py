"""Dev tool: timeline of ONE decode step out of a rocprofv3 --kernel-trace rocpd database: for each kernel position inside the
token (gemv / attention / beam update launches between two beam_step kernels) the average duration and the idle gap
before it, over the last TOKENS tokens.    python tools/decode_timeline.py <results.db> [tokens]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tokens = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cur = db.cursor()
cols = [c[1] for c in cur.execute('pragma table_info("kernels")')]
s_col = "start" if "start" in cols else next(c for c in cols if "start" in c)
e_col = "end" if "end" in cols else next(c for c in cols if "end" in c)
rows = cur.execute(f'select name, "{s_col}", "{e_col}", grid_x, workgroup_x from kernels order by "{s_col}"').fetchall()
marks = [i for i, r in enumerate(rows) if "beam_step" in r[0]]
if len(marks) < tokens + 2:
    print("columns:", cols, "beam_step launches:", len(marks))
    sys.exit(1)
steps = [rows[marks[i] + 1: marks[i + 1] + 1] for i in range(len(marks) - tokens - 1, len(marks) - 1)]
n = min(len(s) for s in steps)
if any(len(s) != n for s in steps):
    print("token lengths differ:", sorted(set(len(s) for s in steps)))
steps = [s for s in steps if len(s) == n]
print(f"{len(steps)} tokens x {n} launches; wall per token = {sum(s[-1][2] - s[0][1] for s in steps) / len(steps) / 1e3:.1f} us + gap to the next token")
tot_k = tot_g = 0.0
agg = {}
for p in range(n):
    dur = sum(s[p][2] - s[p][1] for s in steps) / len(steps) / 1e3
    gap = sum((s[p][1] - s[p - 1][2]) for s in steps) / len(steps) / 1e3 if p else 0.0
    tot_k += dur
    tot_g += gap
    name = steps[0][p][0].split("(")[0].replace("void mxvl::", "")[:40]
    a = agg.setdefault((name, round(dur / 2) * 2 if ("gemv" in name or "gemm" in name) else 0), [0, 0.0, 0.0])
    a[0] += 1
    a[1] += dur
    a[2] += gap
    if p < 14:
        print(f"  #{p:3d} {name:40s} {dur:8.2f} us   gap before {gap:6.2f} us")
print(f"kernel time {tot_k:.1f} us + gaps {tot_g:.1f} us per token")
for (name, bucket), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:40s} ~{bucket:4d} us x {a[0]:3d}: {a[1]:8.1f} us total, avg gap before {a[2] / a[0]:5.2f} us")
