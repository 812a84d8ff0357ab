"""-m gpu: PretrainEngine(use_graph=True) -- the whole training step captured once into a hipGraph and replayed -- against the eager engine:
the same losses step for step (to the run-to-run noise of the fp32 atomics in the scan backward), the cosine schedule's rates reaching the
captured AdamW kernels through device scalars, refusal of a batch of another shape."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(**kw):
    from medical_image_analysis_amd.models_pretrain import VisionMamba
    from medical_image_analysis_amd.pretrain_engine import PretrainEngine
    torch.manual_seed(0)
    m = VisionMamba(img_size=128, patch_size=16, stride=16, embed_dim=128, depth=12, dec_embed_dim=64, rms_norm=True, residual_in_fp32=True,
                    fused_add_norm=True, if_abs_pos_embed=True, bimamba_type="None").to(DEV)
    return m, PretrainEngine(m, lr=1e-3, device=DEV, **kw)


@pytest.mark.parametrize("schedule", [None, dict(min_lr=1e-5, warmup_epochs=0.5, epochs=2.0)])
def test_graph_step_equals_eager_step(schedule):
    g = torch.Generator().manual_seed(7)
    batches = [torch.randn(4, 3, 128, 128, generator=g).to(DEV) for _ in range(3)]
    kw = dict(schedule=schedule, iters_per_epoch=6) if schedule else {}
    me, ee = _build(**kw)
    mg, eg = _build(use_graph=True, graph_warmup=2, **kw)
    le, lg = [], []
    for i in range(10):
        le.append(float(ee.step(batches[i % 3], epoch=i // 6)))
        lg.append(float(eg.step(batches[i % 3], epoch=i // 6)))
    assert eg._graph is not None and eg.data_iter_step == ee.data_iter_step
    for a, b in zip(le, lg):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (le, lg)
    assert le[-1] < le[0]                                       # it trains
    # the parameters moved together: AdamW turns the sign noise of a near-zero gradient (bf16 step, fp32 atomics) into +- lr per step, so two
    # runs of the SAME engine differ by up to 2 lr per step on such entries; the bulk must agree
    for (n, p), q in zip(me.named_parameters(), mg.parameters()):
        d = (p.detach() - q.detach()).abs()
        assert float(d.max()) <= 2 * 1e-3 * 10 + 1e-6, n
        assert float(d.mean()) <= 2e-3, n
    if schedule:
        lr_e = ee.optimizer.param_groups[0]["lr"]
        lr_g = eg.optimizer.param_groups[0]["lr"]
        assert isinstance(lr_g, torch.Tensor) and abs(float(lr_g) - float(lr_e)) <= 1e-9 + 1e-6 * float(lr_e)
    with pytest.raises(RuntimeError, match="captured for batches"):
        eg.step(batches[0][:2])


def test_graph_engine_refuses_what_it_cannot_capture():
    from medical_image_analysis_amd.pretrain_engine import PretrainEngine
    m = torch.nn.Linear(4, 4)
    with pytest.raises(ValueError, match="needs a GPU"):
        PretrainEngine(m, device=None, amp_dtype=None, use_graph=True)
    with pytest.raises(ValueError, match="accum_iter"):
        PretrainEngine(m.to(DEV), device=DEV, use_graph=True, accum_iter=2)


def test_graph_engine_recaptures_after_loading_a_checkpoint():
    """load_checkpoint_state replaces the optimizer / scaler tensors the captured step reads: the engine drops the graph, runs its warm-up
    steps eagerly again and re-captures; training continues from the loaded state exactly as the eager engine does."""
    g = torch.Generator().manual_seed(9)
    batches = [torch.randn(4, 3, 128, 128, generator=g).to(DEV) for _ in range(2)]
    ma, ea = _build(use_graph=True, graph_warmup=1)
    for i in range(4):
        ea.step(batches[i % 2])
    assert ea._graph is not None
    state = ea.checkpoint_state()
    state = {k: (v if not isinstance(v, dict) else __import__("copy").deepcopy(v)) for k, v in state.items()}
    mb, eb = _build(use_graph=True, graph_warmup=1)
    eb.step(batches[0]); eb.step(batches[1])
    assert eb._graph is not None
    eb.load_checkpoint_state(state)
    assert eb._graph is None
    la = [float(ea.step(batches[i % 2])) for i in range(4)]
    lb = [float(eb.step(batches[i % 2])) for i in range(4)]
    assert eb._graph is not None
    for a, b in zip(la, lb):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (la, lb)
