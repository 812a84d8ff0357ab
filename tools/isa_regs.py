"""Dev tool: VGPRs / scratch / occupancy of every kernel in a hipcc -S listing.  usage: python tools/isa_regs.py file.s [name-substring]"""
import re
import sys

s = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\S+):\s*; @", s, re.M):
    name = m.group(1)
    if sub not in name:
        continue
    j = s.find("; Kernel info:", m.end())
    tail = s[j:j + 800]
    g = lambda k: int(re.search(r"; %s: (\d+)" % k, tail).group(1))
    print(f"{name[:110]:110s} vgpr {g('NumVgprs'):3d} sgpr {g('TotalNumSgprs'):3d} scratch {g('ScratchSize'):4d} occ {g('Occupancy')}")
