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
