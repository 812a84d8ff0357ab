import sys, os, torch
sys.path.insert(0, os.getcwd())
from medical_image_analysis_amd.report_decoder import _BeamState
dev = "cuda:0"
def mk(): return _BeamState(1, 3, 32000, 128, 0, [2], 128, 2.0, 2.0, False, dev)
logits = torch.randn(3, 32000, device=dev)
for use in (True, False):
    st = mk(); st.use_hip = use
    for _ in range(5): st.advance(logits)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st.reset()
    with torch.cuda.graph(g):
        for _ in range(20): st.advance(logits)
    g.replay(); torch.cuda.synchronize(); st.reset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(("hip  " if use else "torch"), f"advance: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per step (graph replay)")
st = mk()
print("supported:", st._hip_supported(logits), "use_hip:", st.use_hip, "keep", st.keep, "nb", st.nb)
import medical_image_analysis_amd.report_decoder as rd
calls = {"hip": 0, "torch": 0}
oh, ot = rd._BeamState._advance_hip, rd._BeamState.advance_torch
rd._BeamState._advance_hip = lambda self, lg: (calls.__setitem__("hip", calls["hip"] + 1), oh(self, lg))[1]
rd._BeamState.advance_torch = lambda self, lg: (calls.__setitem__("torch", calls["torch"] + 1), ot(self, lg))[1]
st.advance(logits); torch.cuda.synchronize(); print(calls)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st.reset(); e0.record()
for _ in range(20): st.advance(logits)
e1.record(); torch.cuda.synchronize()
print(f"eager hip loop: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per step", calls)
