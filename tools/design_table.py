"""Dev tool: the rows of DESIGN.md section 5 from the bench lines under profiles/ (python tools/design_table.py [tag=r06])."""
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for f in sorted(glob.glob(os.path.join(root, "profiles", f"{tag}_bench_*.json"))):
    name = os.path.basename(f)[len(tag) + 7:-5]
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:      # a failed run leaves its error text
        print(f"| `{name}` | ERROR {e} |")
        continue
    r, c = d.get("roofline", {}), d.get("cpu_baseline")
    roof = f"{r.get('frac', 0):.3f} of {r.get('bound', '?')}"
    if r.get("traffic"):
        roof += f", traffic {r['traffic'] / 1e6:.0f} MB"
    for extra in ("hbm_kernel", "mfma_kernel"):
        if extra in d:
            e = d[extra]
            roof += f"; {extra} {e.get('kernel', '?')[:24]} {e.get('frac', 0):.3f} of {e.get('bound', '?')}"
    cpu = f"{c['value']:.3g} {c['unit']}" if c else "—"
    print(f"| `{name}` | **{d['value']:.4g} {d['unit']}** | {d['ms_per_step']:.4g} ms | {d.get('dtype')} | {roof} | {cpu} |")
    for k in ("north_star_kernel", "secondary"):
        if k in d:
            o = d[k]
            rr = o.get("roofline", {})
            print(f"|   — its `{k}` | {o.get('value', o.get('kernel_us', ''))} | {o.get('ms_per_step', o.get('kernel_us', ''))} | {o.get('dtype', '')} | {rr.get('frac', 0):.3f} | |")
