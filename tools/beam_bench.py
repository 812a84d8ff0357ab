"""Dev tool: one beam-search update (mxvl_beam_step) per graph node, time per step.   python tools/beam_bench.py [batch] [beams]
Measurement build: MXVL_LIB=medical_image_analysis_amd/build/libmxvl_ablate.so with MXVL_BEAM_SHAPE=1 (the 512 x 16 shape) and
MXVL_BEAM_ABLATE=1..3 (stop after the statistics / the candidate sweep / the block arg-max rounds)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medical_image_analysis_amd import _abi
if os.environ.get("MXVL_LIB"):
    _abi.LIB_PATH = os.environ["MXVL_LIB"]
from medical_image_analysis_amd.report_decoder import _BeamState
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 3
st = _BeamState(B, nb, 32000, 128, 0, [2], 128, 2.0, 2.0, False, dev)
logits = torch.randn(B * nb, 32000, device=dev)
assert st.use_hip and st._hip_supported(logits)
for _ in range(5): st.advance(logits)
torch.cuda.synchronize()
st.reset()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): st.advance(logits)
g.replay(); torch.cuda.synchronize(); st.reset()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print(f"batch {B} x beams {nb}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per step (graph replay; shape={os.environ.get('MXVL_BEAM_SHAPE', '-')} ablate={os.environ.get('MXVL_BEAM_ABLATE', '-')})")
