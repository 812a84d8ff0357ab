"""Stage-1 pre-training step of MambaXray-VL on MI355X (one process per GPU, RCCL gradient all-reduce).

Host-side mirror of the reference's hand-rolled loop:
  CXPMRG_Bench_MambaXray_VL/pretrain/main_pretrain.py:150-173  model -> DDP -> AdamW(betas=(0.9, 0.95)) with timm's
                                                               add_weight_decay grouping (no decay on 1-D params)
  CXPMRG_Bench_MambaXray_VL/pretrain/engine_pretrain.py:37-62  autocast(bf16) forward -> loss.mean() -> backward ->
                                                               clip_grad_norm_(3.0) -> step -> all_reduce_mean(loss)
  CXPMRG_Bench_MambaXray_VL/pretrain/utils/misc.py:211-233     env:// NCCL(=RCCL) init
The reference runs its GradScaler (misc.py:236-256 NativeScalerWithGradNormCount: scale -> backward -> unscale_ -> clip ->
step -> update) on EVERY step, also under bf16 autocast where no scaling is needed (engine_pretrain.py:40,49-50) -- so does
this engine on a GPU by default (`use_scaler=None`): the same inf / nan skip of the optimizer step, the same scale state in the
checkpoint (`checkpoint_state()["scaler"]`, misc.py:286-292), the same extra pass over the gradients in the timed step.  With
finite gradients the scaled step equals the unscaled one up to the power-of-two scaling of the backward (tests/test_engine_cpu.py).
`use_scaler=False` is the plain bf16 step.  The ViT-MAE stage (HD_Xray_Pretrain_MAE/pretrain/main.py:211-213,317) trains under
fp16 autocast, where the scaler is needed: amp_dtype=torch.float16 always has one.  Gradients stay fp32 (params are fp32
under autocast) exactly as in the reference.
DDP wrapping follows the two call sites: main_pretrain.py:167-169 (plain) and HD_Xray_Pretrain_MAE/pretrain/main.py:183
(`broadcast_buffers=False, find_unused_parameters=True`: MaskedAutoencoderViT.decoder_image never sees a gradient) -- a model
says so through `ddp_find_unused_parameters = True` (mae.MaskedAutoencoderViT does) or the caller through the argument.
Gradient accumulation and the per-iteration schedule of engine_pretrain.py:28-52 / utils/lr_sched.py are part of step():
accum_iter micro-batches per optimizer update (loss / accum_iter, DDP no_sync() on the micro-steps that do not update),
half-cycle cosine with linear warm-up evaluated at data_iter_step / iters_per_epoch + epoch on every update boundary.
Gradient exchange: torch DDP over RCCL with 256 MiB buckets (ARM-large = 1.16 GiB of fp32 grads -> 5 large
reduce-scatter+all-gather rounds that use all 7 xGMI links per GPU, overlapped with backward) and
gradient_as_bucket_view (no extra copy).
"""
from __future__ import annotations

import contextlib
import os

import torch
import torch.distributed as dist
import torch.nn as nn


TUNED_GEMMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "tunableop_gfx950.csv")


def enable_tuned_gemms(path: str | None = None) -> bool:
    """Library GEMMs (in/out/x/dt projections, SwiGLU, decoder) keep going to hipBLASLt / rocBLAS, but with the
    per-shape solution picked offline on an MI355X by PyTorch TunableOp (tools/tune_gemms.py; e.g. the merged
    SwiGLU w1|w2 GEMM 32640x5460x1024: 0.55 ms default -> 0.27 ms tuned).  Read-only at run time: no tuning, no file
    writes; shapes that are not in the file use the library default."""
    path = path or TUNED_GEMMS
    if not torch.cuda.is_available() or not os.path.exists(path):
        return False
    if os.environ.get("PYTORCH_TUNABLEOP_TUNING") == "1":      # an explicit tuning session (tools/tune_gemms.py) owns the settings
        return False
    tun = torch.cuda.tunable
    tun.enable(True)
    tun.tuning_enable(False)
    return bool(tun.read_file(path))


def init_distributed(backend: str | None = None):
    """env:// rendezvous (RANK / WORLD_SIZE / LOCAL_RANK from torchrun), as misc.init_distributed_mode."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def wrap_ddp(model: nn.Module, device=None, bucket_cap_mb: float = 256, find_unused_parameters: bool | None = None):
    """DistributedDataParallel the way this package's steps use it (one process per GPU, RCCL over xGMI): 256 MiB buckets,
    gradients as bucket views, no buffer broadcasts (main.py:183), `find_unused_parameters` from the argument or from any
    sub-module's `ddp_find_unused_parameters` flag.  Parameters that do not require a gradient (the frozen LLM / encoder of the
    fine-tuning stages: 13.5 GB for Llama-2-7B) are left out of DDP altogether -- no rank-0 broadcast of them at construction;
    every rank builds them from the same seed or checkpoint, as the reference's Lightning strategies assume too."""
    on_gpu = device is not None and torch.device(device).type == "cuda"
    if find_unused_parameters is None:
        find_unused_parameters = any(getattr(m, "ddp_find_unused_parameters", False) for m in model.modules())
    frozen = [n for n, p in model.named_parameters() if not p.requires_grad]
    if frozen:
        model._ddp_params_and_buffers_to_ignore = list(getattr(model, "_ddp_params_and_buffers_to_ignore", [])) + frozen
    return nn.parallel.DistributedDataParallel(model, device_ids=[torch.device(device).index] if on_gpu else None,
                                               bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True, broadcast_buffers=False,
                                               find_unused_parameters=bool(find_unused_parameters))


def param_groups_weight_decay(model: nn.Module, weight_decay: float = 0.05, skip=()):
    """timm.optim.optim_factory.add_weight_decay: 1-D tensors, biases and `skip` names get no decay."""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if (p.ndim <= 1 or name.endswith(".bias") or name in skip) else decay).append(p)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


def cosine_lr(epoch: float, lr: float, min_lr: float, warmup_epochs: float, epochs: float) -> float:
    """utils/lr_sched.py adjust_learning_rate: linear warm-up, then half-cycle cosine down to min_lr (epoch is fractional)."""
    import math
    if epoch < warmup_epochs:
        return lr * epoch / warmup_epochs
    return min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * (epoch - warmup_epochs) / (epochs - warmup_epochs)))


def adjust_learning_rate(optimizer, epoch: float, lr: float, min_lr: float, warmup_epochs: float, epochs: float) -> float:
    """Same side effect as the reference's: every param group gets lr (x its "lr_scale" when it has one)."""
    v = cosine_lr(epoch, lr, min_lr, warmup_epochs, epochs)
    for g in optimizer.param_groups:
        g["lr"] = v * g["lr_scale"] if "lr_scale" in g else v
    return v


class PretrainEngine:
    """model(imgs) -> per-token loss; one call of step() = forward + backward + clip + AdamW step."""

    def __init__(self, model: nn.Module, lr: float = 1.5e-4, weight_decay: float = 0.05, clip_grad: float | None = 3.0,
                 amp_dtype: torch.dtype | None = torch.bfloat16, bucket_cap_mb: int = 256, device=None, accum_iter: int = 1,
                 schedule: dict | None = None, iters_per_epoch: int | None = None, use_scaler: bool | None = None,
                 find_unused_parameters: bool | None = None, use_graph: bool = False, graph_warmup: int = 3):
        """schedule = dict(min_lr=, warmup_epochs=, epochs=) switches the per-iteration cosine schedule on (peak = lr); it needs
        iters_per_epoch = len(data_loader).  accum_iter micro-batches feed one optimizer update.
        use_scaler: None = the reference's recipe (a GradScaler whenever the step autocasts on a GPU, and for fp16 anywhere).
        find_unused_parameters: None = any(m.ddp_find_unused_parameters for m in model.modules()).
        use_graph: after `graph_warmup` ordinary steps the whole step -- forward, backward, unscale, clip, AdamW, the scaler's update, the
        low-precision weight refresh -- is captured ONCE into a hipGraph and every later step() is a copy of the batch into the
        captured input + one graph launch: the ~1 000 .. 2 000 kernel launches of a step whose kernels are short (192 x 192 / 224 x 224
        images) no longer wait for the host (arm_pretrain_base_192: 24.5 -> 18.8 ms per step).  Single process, accum_iter 1, fixed batch
        shape; the learning rate lives in a device scalar the schedule writes before each launch.  Same kernels, same arithmetic."""
        self.device = device
        self.accum_iter, self.lr, self.schedule, self.iters_per_epoch = int(accum_iter), lr, schedule, iters_per_epoch
        if schedule is not None and not iters_per_epoch:
            raise ValueError("a schedule needs iters_per_epoch (the reference evaluates it at data_iter_step / len(data_loader) + epoch)")
        self.data_iter_step = 0
        self._epoch = 0
        on_gpu = device is not None and torch.device(device).type == "cuda"
        self.use_graph, self.graph_warmup = bool(use_graph), int(graph_warmup)
        self._graph = self._graph_in = self._graph_out = self._graph_local = None
        self._eager_steps = 0
        if self.use_graph:
            if not on_gpu:
                raise ValueError("use_graph: a hipGraph needs a GPU step")
            if int(accum_iter) != 1:
                raise ValueError("use_graph: one optimizer update per captured step (accum_iter == 1)")
            if dist.is_initialized() and dist.get_world_size() > 1:
                raise ValueError("use_graph: single-process steps only (DDP's bucket hooks are not captured)")
        # fp16 autocast needs dynamic loss scaling (MAE: main.py:317 NativeScaler); the stage-1 loop runs the same scaler under
        # bf16 (engine_pretrain.py:49-50): the default follows it on a GPU, the CPU test models stay on the plain step
        if use_scaler is None:
            use_scaler = amp_dtype == torch.float16 or (on_gpu and amp_dtype == torch.bfloat16)
        self.scaler = torch.amp.GradScaler("cuda" if on_gpu else "cpu") if use_scaler else None
        self.tuned_gemms = enable_tuned_gemms() if (device is not None and torch.device(device).type == "cuda") else False
        self.amp_dtype = amp_dtype
        self.clip_grad = clip_grad
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.raw_model = model
        skip = model.no_weight_decay() if hasattr(model, "no_weight_decay") else ()
        self.optimizer = torch.optim.AdamW(param_groups_weight_decay(model, weight_decay, skip), lr=lr, betas=(0.9, 0.95),
                                           fused=bool(device is not None and torch.device(device).type == "cuda"),
                                           **({"capturable": True} if self.use_graph else {}))
        self._cast_params = [p for p in model.parameters() if p.requires_grad and p.ndim >= 1 and p.is_floating_point() and p.is_cuda]
        self._cast_shadow = None
        if self.world > 1:
            self.model = wrap_ddp(model, device, bucket_cap_mb, find_unused_parameters)
            self.find_unused_parameters = self.model.find_unused_parameters
        else:
            self.model = model

    @torch.no_grad()
    def _refresh_casts(self):
        """Low-precision copies of every parameter (the GEMM weights and the biases added to their outputs) in ONE multi-tensor launch, right after the optimizer
        step; the projections pick them up through autograd_util.cast_param (one cast kernel per weight and forward before).
        Modules that keep their own kernel-side parameter forms (models_mamba.SwiGLU._fused_params) get the optimizer-step count as
        their stamp: their next forward rebuilds, later forwards of the same step (accumulation, evaluation) are served."""
        self._param_epoch = getattr(self, "_param_epoch", 0) + 1
        for m in self.raw_model.modules():
            if hasattr(m, "_fused_params"):
                m.__dict__["_mxvl_epoch"] = self._param_epoch
        if self.amp_dtype not in (torch.bfloat16, torch.float16) or not self._cast_params:
            return
        if self._cast_shadow is None:
            self._cast_shadow = [torch.empty_like(p, dtype=self.amp_dtype) for p in self._cast_params]
        torch._foreach_copy_(self._cast_shadow, self._cast_params)
        for p, s_ in zip(self._cast_params, self._cast_shadow):
            p._mxvl_lp = ((p._version, p.data_ptr(), p.device), s_)

    def drop_casts(self):
        """Forget the low-precision copies (after writing parameters through `.data`, which no stamp can see; before evaluating the
        model outside the engine with weights changed that way)."""
        for p in self._cast_params:
            if hasattr(p, "_mxvl_lp"):
                del p._mxvl_lp
        for m in self.raw_model.modules():
            m.__dict__.pop("_mxvl_fused", None)          # models_mamba.SwiGLU._fused_params
            m.__dict__.pop("_mxvl_epoch", None)

    def start_epoch(self):
        """engine_pretrain.py:31 `optimizer.zero_grad()` + the iteration counter the schedule and the accumulation window read."""
        self.data_iter_step = 0
        self.optimizer.zero_grad(set_to_none=True)

    def _set_lr(self, value: float):
        """adjust_learning_rate's side effect; under use_graph the rates are device scalars the captured AdamW kernels read."""
        for g in self.optimizer.param_groups:
            v = value * g["lr_scale"] if "lr_scale" in g else value
            if isinstance(g["lr"], torch.Tensor):
                g["lr"].fill_(v)
            else:
                g["lr"] = v

    def _graph_step(self, imgs: torch.Tensor, epoch: int) -> torch.Tensor:
        if self._graph is None:
            # capture: the rates become device scalars first (a float would be baked into the launch parameters)
            for g in self.optimizer.param_groups:
                if not isinstance(g["lr"], torch.Tensor):
                    g["lr"] = torch.tensor(float(g["lr"]), dtype=torch.float32, device=imgs.device)
            self._graph_in = imgs.clone()
            torch.cuda.synchronize(imgs.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._graph_out = self._eager_step(self._graph_in, epoch, adjust_lr=False)
                self._graph_local = self.last_local_loss
            self._graph = graph
            self.data_iter_step -= 1          # the capture ran the bookkeeping of a step that has not executed yet
        if imgs.shape != self._graph_in.shape or imgs.dtype != self._graph_in.dtype:
            raise RuntimeError(f"use_graph: the step was captured for batches of {tuple(self._graph_in.shape)} {self._graph_in.dtype}, "
                               f"got {tuple(imgs.shape)} {imgs.dtype}")
        if epoch != self._epoch:
            self._epoch, self.data_iter_step = epoch, 0
        if self.schedule is not None:
            self._set_lr(cosine_lr(self.data_iter_step / self.iters_per_epoch + epoch, self.lr, **self.schedule))
        self._graph_in.copy_(imgs, non_blocking=True)
        self._graph.replay()
        self.last_local_loss = self._graph_local
        self.data_iter_step += 1
        return self._graph_out.clone()

    def step(self, imgs: torch.Tensor, epoch: int = 0) -> torch.Tensor:
        """One iteration of train_one_epoch (engine_pretrain.py:36-62): a micro-batch forward + backward; on the last micro-batch of
        an accumulation window also clip, optimizer step and zero_grad.  Returns misc.all_reduce_mean(loss) of this micro-batch."""
        if self.use_graph:
            if self._eager_steps >= self.graph_warmup:
                return self._graph_step(imgs, epoch)
            self._eager_steps += 1
        return self._eager_step(imgs, epoch)

    def _eager_step(self, imgs: torch.Tensor, epoch: int = 0, adjust_lr: bool = True) -> torch.Tensor:
        dev_type = imgs.device.type
        if epoch != self._epoch:          # a caller that passes `epoch` without start_epoch(): the per-epoch iteration restarts with it
            self._epoch = epoch           # (unconditionally: an epoch that ended early -- shorter loader, drop_last -- must not carry its
            self.data_iter_step = 0       # counter into the next one's schedule and accumulation window)
        it, acc = self.data_iter_step, self.accum_iter
        if self.schedule is not None and it % acc == 0 and adjust_lr:
            self._set_lr(cosine_lr(it / self.iters_per_epoch + epoch, self.lr, **self.schedule))
        update = (it + 1) % acc == 0
        ddp = self.world > 1
        # micro-steps that do not update keep their gradients local: the all-reduce of the window rides on its last backward
        sync = self.model.no_sync() if (ddp and not update) else contextlib.nullcontext()
        with sync:
            with torch.autocast(device_type=dev_type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
                loss = self.model(imgs)
            loss = loss.mean()
            # misc.all_reduce_mean(loss_value) (engine_pretrain.py:62) is part of every reference step: issued here, on the
            # collective stream, so the 4-byte all-reduce rides under the backward pass instead of costing a host round trip
            reduced, work = loss.detach().clone(), None
            if ddp:
                work = dist.all_reduce(reduced, async_op=True)
            if acc == 1:
                self.optimizer.zero_grad(set_to_none=True)
            back = loss / acc if acc > 1 else loss
            (self.scaler.scale(back) if self.scaler is not None else back).backward()
        if update:
            if self.scaler is not None:
                if self.clip_grad is not None:
                    self.scaler.unscale_(self.optimizer)
                    nn.utils.clip_grad_norm_(self.raw_model.parameters(), self.clip_grad)
                self.scaler.step(self.optimizer)
                self.scaler.update()
            else:
                if self.clip_grad is not None:
                    nn.utils.clip_grad_norm_(self.raw_model.parameters(), self.clip_grad)
                self.optimizer.step()
            self._refresh_casts()
            if acc > 1:
                self.optimizer.zero_grad(set_to_none=True)
        if work is not None:
            work.wait()
            reduced /= self.world
        self.last_local_loss = loss.detach()
        self.data_iter_step = it + 1
        return reduced

    def checkpoint_state(self, epoch: int | None = None) -> dict:
        """misc.save_model's dictionary (misc.py:286-292): 'model' (without the DDP prefix), 'optimizer', 'epoch', 'scaler'."""
        out = {"model": self.raw_model.state_dict(), "optimizer": self.optimizer.state_dict(),
               "epoch": self._epoch if epoch is None else epoch}
        if self.scaler is not None:
            out["scaler"] = self.scaler.state_dict()
        return out

    def load_checkpoint_state(self, state: dict) -> None:
        """misc.load_model (misc.py:323-340): model, then optimizer + epoch + scaler when the file has them."""
        self.raw_model.load_state_dict(state["model"])
        self.drop_casts()
        # a captured step reads the optimizer / scaler state tensors it was captured with: loading replaces them, so the next steps run
        # eagerly again and the step is re-captured behind them
        self._graph = self._graph_in = self._graph_out = self._graph_local = None
        self._eager_steps = 0
        if "optimizer" in state:
            self.optimizer.load_state_dict(state["optimizer"])
        if "epoch" in state:
            self._epoch = int(state["epoch"])
        if self.scaler is not None and "scaler" in state:
            self.scaler.load_state_dict(state["scaler"])

    def reduced_loss(self, loss: torch.Tensor) -> float:
        """step() already returns misc.all_reduce_mean(loss) (engine_pretrain.py:62); this is the host read of it."""
        return float(loss)
