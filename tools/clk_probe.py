"""Dev probe: per-workgroup start/end wall ticks + shader clocks of the streaming scan kernel (ablate bit 16)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from medical_image_analysis_amd import _abi
from medical_image_analysis_amd.selective_scan_interface import scan_fwd_raw
dev = torch.device("cuda:0"); B,D,L,N = 8,1536,4096,16
g = torch.Generator(device="cpu").manual_seed(0)
A = (-0.5 * torch.rand(D, N, generator=g)).to(dev)
mk = lambda *s: torch.randn(*s, generator=g).to(dev)
u, z, Bm, Cm = mk(B, D, L), mk(B, D, L), mk(B, 1, N, L), mk(B, 1, N, L)
delta = (0.5 * torch.rand(B, D, L, generator=g)).to(dev)
Dv = torch.randn(D, generator=g).to(dev); bias = (0.5 * torch.rand(D, generator=g)).to(dev)
lib = _abi.load()
nwg = (D // 16) * B
for ab in (24, 24 | 1):
    lib.mxvl_set_scan_variant(10 | (ab << 16))
    for it in range(3):
        out = scan_fwd_raw(u, delta, A, Bm, Cm, Dv, z, bias, True)[0]
        torch.cuda.synchronize()
    t = out.view(-1)[:4 * nwg].view(nwg, 4).double().cpu()
    st, en, clk, hwid = t[:, 0], t[:, 1], t[:, 2], t[:, 3].long()
    t0 = st.min()
    st, en = (st - t0) / 100, (en - t0) / 100
    life = en - st
    print(f"ablate {ab}: kernel span {en.max():.1f} us; WG start min/med/max {st.min():.1f}/{st.median():.1f}/{st.max():.1f} us; "
          f"WG lifetime min/med/max {life.min():.1f}/{life.median():.1f}/{life.max():.1f} us; MHz med {(clk / life).median():.0f}")
    late = (st > 20).sum().item()
    print(f"   WGs starting later than 20 us: {late} of {nwg}")
    # HW_ID: wave_id[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]... (gfx9 layout) ; XCC_ID separate
    cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
    key = se * 32 + sh * 16 + cu
    import collections
    cnt = collections.Counter(key.tolist())
    xcc = (hwid >> 16) & 0xf
    wgid = torch.arange(nwg); by, bx = wgid // (D // 16), wgid % (D // 16)
    for nm, keyv in (("xcc", xcc), ("se", se), ("cu", cu), ("batch", by), ("dtile%8", bx % 8), ("dtile//12", bx // 12)):
        vals = sorted(set(keyv.tolist()))
        print(f"   lifetime by {nm}: " + " ".join(f"{v}:{life[keyv == v].mean():.0f}" for v in vals))
    print("   distinct (se,sh,cu) ids:", len(cnt), " WGs per id histogram:", sorted(collections.Counter(cnt.values()).items()))
