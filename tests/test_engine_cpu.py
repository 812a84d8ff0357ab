"""CPU (gloo, world_size 2 and 8): the data-parallel step of pretrain_engine.py -- gradients are averaged over ranks,
replicas stay bit-identical, and the result equals one process fed the concatenated batch.  The Mamba model
itself needs a GPU, so a small stand-in module with the same `model(imgs) -> per-token loss` contract is used.
World size 8 = one rank per GPU of the MI355X node the driver scales to (main_pretrain.py:109,125-131,167-169); the bucket cap
is set below the model size so the gradient all-reduce runs as SEVERAL buckets (bucket boundaries, gradient_as_bucket_view)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


class TinyLossModel(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = nn.Linear(12, 16)
        self.b = nn.Linear(16, 12)
        self.scale = nn.Parameter(torch.ones(12))

    def no_weight_decay(self):
        return {"scale"}

    def forward(self, x):
        y = self.b(torch.tanh(self.a(x))) * self.scale
        return ((y - x) ** 2).mean(-1).mean(0)  # per-token loss vector, like VisionMamba.forward


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from medical_image_analysis_amd.pretrain_engine import PretrainEngine, init_distributed
    init_distributed("gloo")
    eng = PretrainEngine(TinyLossModel(), lr=1e-2, amp_dtype=None, device=None, bucket_cap_mb=0.0005)   # ~131 floats per bucket
    g = torch.Generator().manual_seed(100 + rank)          # per-rank shard, as seed + rank in main_pretrain.py:109
    losses = []
    for _ in range(3):
        x = torch.randn(4, 7, 12, generator=g)
        losses.append(eng.reduced_loss(eng.step(x)))
    # plain numpy through the queue: tensors travel as file descriptors that die with the worker process
    sd = {k: v.detach().cpu().numpy().copy() for k, v in eng.raw_model.state_dict().items()}
    out.put((rank, [float(l) for l in losses], sd))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_ddp_step_matches_single_process_on_the_concatenated_batch(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, l0, sd0) = results[0]
    for (_, l1, sd1) in results[1:]:
        assert l0 == l1, "all_reduce_mean(loss) must agree on every rank"
        for k in sd0:
            assert (sd0[k] == sd1[k]).all(), f"replicas diverged at {k}"
    # single process, global batch = concat of the shards: mean loss over 4 * world samples == mean of rank means
    from medical_image_analysis_amd.pretrain_engine import PretrainEngine
    eng = PretrainEngine(TinyLossModel(), lr=1e-2, amp_dtype=None, device=None)
    gens = [torch.Generator().manual_seed(100 + r) for r in range(world)]
    for step in range(3):
        x = torch.cat([torch.randn(4, 7, 12, generator=g) for g in gens], dim=0)
        loss = float(eng.step(x))
        assert abs(loss - l0[step]) < 1e-5
    for k, v in eng.raw_model.state_dict().items():
        assert torch.allclose(v, torch.from_numpy(sd0[k]), atol=1e-5, rtol=1e-4), k


def test_weight_decay_groups_follow_timm_rule():
    from medical_image_analysis_amd.pretrain_engine import param_groups_weight_decay
    m = TinyLossModel()
    no_decay, decay = param_groups_weight_decay(m, 0.05, m.no_weight_decay())
    assert no_decay["weight_decay"] == 0.0 and decay["weight_decay"] == 0.05
    assert len(decay["params"]) == 2 and len(no_decay["params"]) == 3  # two weight matrices | two biases + scale


def test_lr_schedule_matches_the_reference_function():
    """cosine_lr / adjust_learning_rate == CXPMRG_Bench_MambaXray_VL/pretrain/utils/lr_sched.py on the committed grid
    (tests/golden/lr_sched.npz, produced by executing the reference's function), including the lr_scale param group."""
    import types
    import numpy as np
    from conftest import GOLDEN
    from medical_image_analysis_amd.pretrain_engine import adjust_learning_rate
    g = np.load(os.path.join(GOLDEN, "lr_sched.npz"))
    for tag in ("a", "b"):
        lr, min_lr, warm, epochs = (float(v) for v in g[f"{tag}_args"])
        opt = types.SimpleNamespace(param_groups=[{"lr": 0.0}, {"lr": 0.0, "lr_scale": 0.65}])
        for e, want, want_s in zip(g[f"{tag}_epoch"], g[f"{tag}_lr"], g[f"{tag}_lr_scaled"]):
            adjust_learning_rate(opt, float(e), lr, min_lr, warm, epochs)
            assert opt.param_groups[0]["lr"] == want and opt.param_groups[1]["lr"] == want_s, (tag, e)


def test_engine_applies_the_schedule_on_update_boundaries_only():
    """engine_pretrain.py:36-37: the schedule is evaluated at data_iter_step / len(loader) + epoch when data_iter_step % accum_iter == 0."""
    from medical_image_analysis_amd.pretrain_engine import PretrainEngine, cosine_lr
    sched = dict(min_lr=1e-5, warmup_epochs=1, epochs=4)
    eng = PretrainEngine(TinyLossModel(), lr=1e-2, amp_dtype=None, device=None, accum_iter=2, schedule=sched, iters_per_epoch=6)
    g = torch.Generator().manual_seed(0)
    for epoch in range(2):
        eng.start_epoch()
        for it in range(6):
            eng.step(torch.randn(4, 7, 12, generator=g), epoch=epoch)
            boundary = it - it % 2
            want = cosine_lr(boundary / 6 + epoch, 1e-2, **sched)
            assert all(abs(pg["lr"] - want) < 1e-15 for pg in eng.optimizer.param_groups), (epoch, it)


def _accum_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from medical_image_analysis_amd.pretrain_engine import PretrainEngine, init_distributed
    init_distributed("gloo")
    eng = PretrainEngine(TinyLossModel(), lr=1e-2, amp_dtype=None, device=None, accum_iter=2, bucket_cap_mb=0.0005)
    g = torch.Generator().manual_seed(100 + rank)
    eng.start_epoch()
    synced = []
    for _ in range(4):                                   # two accumulation windows of two micro-batches
        x = torch.randn(4, 7, 12, generator=g)
        eng.step(x[:2])
        # after a micro-step that does not update, the gradients are this rank's own (no_sync): they differ between ranks
        grad = eng.raw_model.a.weight.grad.clone()
        gl = [torch.zeros_like(grad) for _ in range(world)]
        dist.all_gather(gl, grad)
        synced.append(bool(all(torch.equal(gl[0], t) for t in gl)))
        eng.step(x[2:])
    sd = {k: v.detach().cpu().numpy().copy() for k, v in eng.raw_model.state_dict().items()}
    out.put((rank, synced, sd))
    dist.barrier()
    dist.destroy_process_group()


def test_accumulated_micro_batches_equal_one_step_of_the_full_batch_under_ddp():
    """accum_iter = 2 under gloo world-2: two half-batches per rank == one engine step on the full batch (the reference's
    `loss /= accum_iter`, update on the last micro-batch, engine_pretrain.py:49-53), the gradient all-reduce only on the updating
    micro-step (DDP no_sync on the other)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_accum_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, synced0, sd0), (_, synced1, sd1) = results
    assert not any(synced0) and not any(synced1), "a non-updating micro-step must not all-reduce its gradients"
    for k in sd0:
        assert (sd0[k] == sd1[k]).all(), f"replicas diverged at {k}"
    from medical_image_analysis_amd.pretrain_engine import PretrainEngine
    eng = PretrainEngine(TinyLossModel(), lr=1e-2, amp_dtype=None, device=None)
    gens = [torch.Generator().manual_seed(100 + r) for r in range(world)]
    for _ in range(4):
        eng.step(torch.cat([torch.randn(4, 7, 12, generator=g) for g in gens], dim=0))
    for k, v in eng.raw_model.state_dict().items():
        assert torch.allclose(v, torch.from_numpy(sd0[k]), atol=1e-5, rtol=1e-4), k


def test_fp16_engine_uses_a_grad_scaler_like_the_mae_recipe():
    """HD_Xray_Pretrain_MAE/pretrain/main.py:211-213,317: fp16 autocast + NativeScaler.  CPU autocast(fp16) of the tiny model: the
    engine scales the loss, unscales before clipping, skips nothing on finite gradients, and lands near the fp32 step."""
    from medical_image_analysis_amd.pretrain_engine import PretrainEngine
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(4, 7, 12, generator=g) for _ in range(3)]
    e16 = PretrainEngine(TinyLossModel(), lr=1e-2, amp_dtype=torch.float16, device=None)
    e32 = PretrainEngine(TinyLossModel(), lr=1e-2, amp_dtype=None, device=None)
    assert e16.scaler is not None and e32.scaler is None
    for x in xs:
        l16, l32 = float(e16.step(x)), float(e32.step(x))
        assert abs(l16 - l32) < 5e-3 * max(1.0, abs(l32))
    for (k, a), (_, b) in zip(e16.raw_model.state_dict().items(), e32.raw_model.state_dict().items()):
        assert torch.allclose(a, b, atol=5e-3), k
