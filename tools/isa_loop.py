"""Dev tool: print the ISA around the fused-DPP scan of one kernel from a -save-temps .s file.
usage: python tools/isa_loop.py file.s mangled_kernel_regex [before] [after]"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
before = int(sys.argv[3]) if len(sys.argv) > 3 else 75
after = int(sys.argv[4]) if len(sys.argv) > 4 else 60
m = re.search(r'^(' + pat + r'):(.*?)\.end_amdhsa_kernel', s, re.S | re.M)
body = m.group(2).split('\n')
idx = [i for i, l in enumerate(body) if 'v_fmac_f32_dpp' in l]
print(m.group(1), idx[0], idx[-1], len(body))
for l in body[max(0, idx[0] - before):idx[-1] + after]:
    l = l.rstrip()
    if l.strip().startswith(';'):
        continue
    print(l)
