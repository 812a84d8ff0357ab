"""Dev tool: instruction histogram of the innermost loop that holds a marker instruction, for one kernel of a hipcc -S listing.
usage: python tools/isa_ops.py file.s <mangled-name-substring> [marker=v_fmac_f32_dpp]"""
import re
import sys
from collections import Counter


def main():
    s = open(sys.argv[1]).read()
    sub = sys.argv[2]
    marker = sys.argv[3] if len(sys.argv) > 3 else "v_fmac_f32_dpp"
    names = [m.group(1) for m in re.finditer(r"^(_Z\S+):", s, re.M) if sub in m.group(1)]
    for name in names:
        i = s.index(name + ":")
        body = s[i:s.index(".Lfunc_end", i)].split("\n")
        idx = [k for k, l in enumerate(body) if marker in l]
        if not idx:
            continue
        labels = [k for k, l in enumerate(body) if re.match(r"\.LBB\d+_\d+:", l)]
        start = max(k for k in labels if k < idx[0])
        end = [k for k, l in enumerate(body) if "s_cbranch" in l and k > idx[-1]][0]
        loop = body[start:end + 1]
        c = Counter(l.split()[0] for l in loop if l.strip() and not l.strip().startswith((".", ";")) and not l.strip().endswith(":"))
        print(name, "loop lines", start, end, "markers", len(idx))
        print("  " + ", ".join(f"{v} {k}" for k, v in c.most_common()), "| total", sum(c.values()))


main()
