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


class _TinyMaeLoss(nn.Module):
    """bench.py's MaeLoss around a tiny MaskedAutoencoderViT: forward(imgs) -> the masked mean loss (main.py:319-323)."""

    def __init__(self):
        super().__init__()
        from medical_image_analysis_amd.mae import MaskedAutoencoderViT
        torch.manual_seed(0)
        self.net = MaskedAutoencoderViT(img_size=64, patch_size=16, in_chans=1, embed_dim=32, depth=2, num_heads=4, decoder_embed_dim=32,
                                        decoder_depth=1, decoder_num_heads=4)

    def forward(self, imgs):
        noise = imgs[:, 0, 0, :16].contiguous()       # the masking noise as a function of the sample: sharding does not change it
        loss, mask = self.net(imgs, 0, 0.75, 0.0, noise)
        return ((loss * mask).sum() / mask.sum()).reshape(1)


def _mae_worker(rank, world, port, out, find_unused):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from medical_image_analysis_amd.pretrain_engine import PretrainEngine, init_distributed
    init_distributed("gloo")
    try:
        eng = PretrainEngine(_TinyMaeLoss(), lr=1e-3, amp_dtype=None, device=None, bucket_cap_mb=0.01, find_unused_parameters=find_unused)
        g = torch.Generator().manual_seed(100 + rank)
        losses = [float(eng.step(torch.randn(2, 1, 64, 64, generator=g))) for _ in range(4)]
        sd = {k: v.detach().cpu().numpy().copy() for k, v in eng.raw_model.state_dict().items()}
        unused = eng.raw_model.net.decoder_image.weight.grad
        out.put((rank, losses, sd, eng.find_unused_parameters, unused is None or float(unused.abs().max()) == 0.0))
    except RuntimeError as e:
        out.put((rank, "RuntimeError: " + str(e)[:200], None, None, None))
    dist.barrier()
    dist.destroy_process_group()


def _run_mae_world2(find_unused):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mae_worker, args=(r, world, port, q, find_unused)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


def test_mae_with_its_unused_decoder_image_trains_under_ddp():
    """HD_Xray_Pretrain_MAE/pretrain/main.py:183 wraps the MAE with find_unused_parameters=True, broadcast_buffers=False because
    `decoder_image` (models/mae.py:91) never gets a gradient.  The mirror says so itself (ddp_find_unused_parameters): a tiny
    MaskedAutoencoderViT steps four times through PretrainEngine under gloo world-2 with the DEFAULT arguments, the replicas stay
    bit-identical, the unused parameter never moves, and the run equals one process on the concatenated batch.  Plain wrapping
    (find_unused_parameters=False) is the failure the flag exists for: it must raise on the second step, not hang."""
    results = _run_mae_world2(None)
    (_, l0, sd0, fu0, idle0), (_, l1, sd1, fu1, idle1) = results
    assert isinstance(l0, list), l0
    assert fu0 and fu1 and idle0 and idle1
    assert l0 == l1
    for k in sd0:
        assert (sd0[k] == sd1[k]).all(), f"replicas diverged at {k}"
    from medical_image_analysis_amd.pretrain_engine import PretrainEngine
    eng = PretrainEngine(_TinyMaeLoss(), lr=1e-3, amp_dtype=None, device=None)
    start = {k: v.clone() for k, v in eng.raw_model.state_dict().items()}
    gens = [torch.Generator().manual_seed(100 + r) for r in range(2)]
    for s in range(4):
        # the masked mean is per rank (sum(loss*mask)/sum(mask) with the same mask count on both): mean of rank means == global mean
        loss = float(eng.step(torch.cat([torch.randn(2, 1, 64, 64, generator=g) for g in gens], dim=0)))
        assert abs(loss - l0[s]) <= 1e-4 * max(1.0, abs(loss)), (s, loss, l0[s])
    moved = 0
    for k, v in eng.raw_model.state_dict().items():
        assert torch.allclose(v, torch.from_numpy(sd0[k]), atol=2e-4, rtol=1e-3), k
        moved += int(not torch.equal(v, start[k]))
    assert torch.equal(eng.raw_model.net.decoder_image.weight, start["net.decoder_image.weight"])
    assert moved > 10
    bad = _run_mae_world2(False)
    assert all(isinstance(r[1], str) and "Expected to have finished reduction" in r[1] for r in bad), bad


def test_bf16_engine_default_scaler_follows_the_reference_and_equals_the_plain_step():
    """engine_pretrain.py:49-50 + misc.py:236-256: the stage-1 loop scales / unscales / inf-checks every step under bf16 too.
    With use_scaler=True (the default on a GPU) the step equals the plain one while gradients are finite (power-of-two scaling
    of an fp32 backward), SKIPS the update on a non-finite gradient and halves the scale, and the scale state is part of
    checkpoint_state() under the reference's key ('scaler', misc.py:290)."""
    from medical_image_analysis_amd.pretrain_engine import PretrainEngine
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(4, 7, 12, generator=g) for _ in range(3)]
    es = PretrainEngine(TinyLossModel(), lr=1e-2, amp_dtype=None, device=None, use_scaler=True)
    ep = PretrainEngine(TinyLossModel(), lr=1e-2, amp_dtype=None, device=None, use_scaler=False)
    assert es.scaler is not None and ep.scaler is None
    assert PretrainEngine(TinyLossModel(), amp_dtype=torch.bfloat16, device=None).scaler is None      # CPU default: plain step
    for x in xs:
        assert float(es.step(x)) == float(ep.step(x))
    for (k, a), (_, b) in zip(es.raw_model.state_dict().items(), ep.raw_model.state_dict().items()):
        assert torch.equal(a, b), k                    # 65536 * loss is exact in fp32; the unscale gives the same gradients back
    scale0 = es.scaler.get_scale()
    before = {k: v.clone() for k, v in es.raw_model.state_dict().items()}
    bad = xs[0].clone()
    bad[0, 0, 0] = float("inf")
    es.step(bad)
    for k, v in es.raw_model.state_dict().items():
        assert torch.equal(v, before[k]), f"{k} moved on a non-finite step"
    assert es.scaler.get_scale() == scale0 * 0.5
    st = es.checkpoint_state(epoch=3)
    assert set(st) == {"model", "optimizer", "epoch", "scaler"} and st["epoch"] == 3 and st["scaler"]["scale"] == scale0 * 0.5
    e2 = PretrainEngine(TinyLossModel(), lr=1e-2, amp_dtype=None, device=None, use_scaler=True)
    e2.load_checkpoint_state(st)
    assert e2.scaler.get_scale() == scale0 * 0.5 and e2._epoch == 3
    for (k, a), (_, b) in zip(es.raw_model.state_dict().items(), e2.raw_model.state_dict().items()):
        assert torch.equal(a, b), k
    la, lb = float(es.step(xs[1])), float(e2.step(xs[1]))
    assert la == lb


def test_epoch_change_always_restarts_the_iteration_counter():
    """an epoch that ended early (shorter loader) must not carry its counter into the next epoch's schedule."""
    from medical_image_analysis_amd.pretrain_engine import PretrainEngine, cosine_lr
    sched = dict(min_lr=1e-5, warmup_epochs=1, epochs=4)
    eng = PretrainEngine(TinyLossModel(), lr=1e-2, amp_dtype=None, device=None, schedule=sched, iters_per_epoch=6)
    g = torch.Generator().manual_seed(1)
    for _ in range(4):                                 # epoch 0 stops after 4 of its 6 iterations
        eng.step(torch.randn(4, 7, 12, generator=g), epoch=0)
    eng.step(torch.randn(4, 7, 12, generator=g), epoch=1)
    assert eng.data_iter_step == 1
    assert all(abs(pg["lr"] - cosine_lr(0 / 6 + 1, 1e-2, **sched)) < 1e-15 for pg in eng.optimizer.param_groups)
