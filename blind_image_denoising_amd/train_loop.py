"""
The training drop-in boundary: the closures `train_loop` creates in the reference
(bfcnn/train_loop.py:249-321) -- train_step, test_step, train_step_single_gpu, apply_grads --
plus the 8-GPU data-parallel step the reference does not have (it is single-device).

One process per GPU.  The image batch is sharded on N across ranks; each rank runs bf_train_step
on its shard and the flat fp32 gradient buffer (84,272 floats = 337 KB for resnet 1x18) goes
through ONE all-reduce (torch.distributed backend "nccl" = RCCL over xGMI), after which every
rank applies the identical fused clip + Adam update.  Batch-norm statistics stay local to a
rank (what tf.distribute.MirroredStrategy does by default).
"""
import os
import time
from collections import namedtuple
from typing import Callable, Dict, Iterable, List, Optional

import torch

from . import _native as N
from .constants import *
from .custom_logger import logger
from .loss import loss_function_builder
from .model import HydraModel, model_builder, save_model
from .optimizer import optimizer_builder
from .utilities import load_config

TrainFunctions = namedtuple("TrainFunctions", ["train_step", "test_step", "train_step_single_gpu", "apply_grads"])


def build_train_functions(model: HydraModel, loss_fn_map: Dict[str, Callable]) -> TrainFunctions:
    """The four closures of bfcnn/train_loop.py:249-321 for a single-output hydra.

    Losses come back as 0-d views of one device buffer (no host synchronisation); call
    `.item()` when a Python float is wanted."""
    denoiser_loss_fn = loss_fn_map[DENOISER_LOSS_FN_STR]
    state = {"grads": None, "losses": None}

    def _buffers():
        if state["grads"] is None or state["grads"].device != model.params.device:
            state["grads"] = torch.zeros(model.n_params, dtype=torch.float32, device=model.device)
            state["losses"] = torch.zeros(N.BF_LOSS_COUNT, dtype=torch.float32, device=model.device)
        return state["grads"], state["losses"]

    def train_step(n: List[torch.Tensor]):
        """bfcnn/train_loop.py:249-251: ckpt.model(n, training=True)."""
        return model(n, training=True)

    def test_step(n: List[torch.Tensor]):
        """bfcnn/train_loop.py:253-257."""
        return model(n, training=False)

    def train_step_single_gpu(p_input_image_batch, p_noisy_image_batch, p_depth_weight=(1.0,),
                              p_percentage_done=0.0, p_trainable_variables=None):
        """bfcnn/train_loop.py:259-312.  Returns (total_loss, model_loss, [denoiser_loss],
        predictions, grads) with grads = ONE flat tensor laid out like model.params (the
        per-variable gradients are views of it: grads[v.offset : v.offset + numel])."""
        grads, losses = _buffers()
        gt = p_input_image_batch.to(device=model.device, dtype=torch.float32).contiguous()
        noisy = p_noisy_image_batch.to(device=model.device, dtype=torch.float32).contiguous()
        if gt.shape != noisy.shape:
            raise ValueError(f"gt {tuple(gt.shape)} and noisy {tuple(noisy.shape)} batches differ in shape")
        dw = p_depth_weight[0] if hasattr(p_depth_weight, "__len__") else p_depth_weight
        predictions = model.train_forward_backward(gt, noisy, denoiser_loss_fn.desc(float(dw)), grads, losses, True)
        model_loss = {REGULARIZATION_LOSS_STR: losses[N.BF_LOSS_REGULARIZATION], TOTAL_LOSS_STR: losses[N.BF_LOSS_MODEL_TOTAL]}
        denoiser_loss = {TOTAL_LOSS_STR: losses[N.BF_LOSS_DENOISER_TOTAL], MSE_LOSS_STR: losses[N.BF_LOSS_MSE],
                         MAE_LOSS_STR: losses[N.BF_LOSS_MAE], SSIM_LOSS_STR: losses[N.BF_LOSS_SSIM]}
        return losses[N.BF_LOSS_TOTAL], model_loss, [denoiser_loss], predictions, grads

    def apply_grads(internal_optimizer, internal_gradients, internal_trainable_variables=None, grad_scale: float = 1.0):
        """bfcnn/train_loop.py:314-321."""
        internal_optimizer.apply_gradients(internal_gradients, model, grad_scale=grad_scale, losses=state["losses"])

    return TrainFunctions(train_step, test_step, train_step_single_gpu, apply_grads)


# ---- data parallel ---------------------------------------------------------------------------

def shard_batch(batch: torch.Tensor, rank: int, world_size: int) -> torch.Tensor:
    """contiguous N-slice of a global batch for `rank` (global batch must divide evenly)."""
    n = batch.shape[0]
    if n % world_size != 0:
        raise ValueError(f"global batch {n} is not divisible by world size {world_size}")
    per = n // world_size
    return batch[rank * per:(rank + 1) * per]


def allreduce_gradients(grads: torch.Tensor, group=None, async_op: bool = False):
    """the ONE collective of a training step: sum-all-reduce of the flat gradient buffer
    (the 1/world_size scale is folded into bf_adam_step's grad_scale)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return None
    return dist.all_reduce(grads, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


class DataParallelTrainer:
    """train_step_single_gpu + all-reduce + apply_grads for one rank of an N-GPU job."""

    def __init__(self, model: HydraModel, loss_fn_map, optimizer, group=None):
        import torch.distributed as dist
        self.model, self.optimizer, self.group = model, optimizer, group
        self.fns = build_train_functions(model, loss_fn_map)
        self.world_size = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1

    def broadcast_parameters(self, src: int = 0):
        import torch.distributed as dist
        if self.world_size > 1:
            dist.broadcast(self.model.params, src=src, group=self.group)
            dist.broadcast(self.model.state, src=src, group=self.group)
            self.model.mark_dirty()

    def step(self, gt_shard: torch.Tensor, noisy_shard: torch.Tensor, depth_weight: float = 1.0):
        total, model_loss, denoiser_loss, predictions, grads = self.fns.train_step_single_gpu(
            gt_shard, noisy_shard, (depth_weight,), 0.0, None)
        work = allreduce_gradients(grads, self.group, async_op=True)
        if work is not None:
            work.wait()          # stream-ordered on RCCL: no host block, the Adam kernels queue behind it
        self.fns.apply_grads(self.optimizer, grads, None, grad_scale=1.0 / self.world_size)
        return total, model_loss, denoiser_loss, predictions


# ---- outer loop --------------------------------------------------------------------------------

def train_loop(pipeline_config_path, model_dir: str, dataset: Iterable = None, weights_dir: str = None,
               device=None, max_steps: Optional[int] = None):
    """Outer loop of bfcnn/train_loop.py:40-601 reduced to what surrounds the hot path: config ->
    loss / optimizer / model builders -> epochs over `dataset` (an iterable yielding
    (input_image_batch, noisy_image_batch) float tensors in value range) with gradient
    accumulation over `gpu_batches_per_step` micro-batches -> model directory per epoch.
    The tf.data pipeline, TensorBoard summaries and TF checkpoints are out of scope."""
    config = load_config(pipeline_config_path)
    train_config = config["train"]
    epochs = train_config["epochs"]
    gpu_batches_per_step = int(train_config.get("gpu_batches_per_step", 1))
    if gpu_batches_per_step <= 0:
        raise ValueError("gpu_batches_per_step must be > 0")               # train_loop.py:114-115
    if dataset is None:
        raise ValueError("dataset must be an iterable of (input_image_batch, noisy_image_batch)")
    loss_fn_map = loss_function_builder(config=config["loss"])
    optimizer, lr_schedule = optimizer_builder(config=train_config["optimizer"])
    if weights_dir:
        from .model import load_hydra
        model = load_hydra(weights_dir, device=device)
    else:
        model = model_builder(config[MODEL_STR], device=device).hydra
    fns = build_train_functions(model, loss_fn_map)
    accumulated = torch.zeros(model.n_params, dtype=torch.float32, device=model.device)
    step, history = 0, []
    for epoch in range(int(epochs)):
        counter = 0
        t0 = time.time()
        for input_image_batch, noisy_image_batch in dataset:
            total, _, denoiser_loss, _, grads = fns.train_step_single_gpu(input_image_batch, noisy_image_batch, (1.0,), 0.0, None)
            accumulated.add_(grads)
            counter += 1
            if counter >= gpu_batches_per_step:
                fns.apply_grads(optimizer, accumulated, None, grad_scale=1.0 / counter)
                accumulated.zero_()
                counter = 0
                step += 1
                history.append(float(total.item()))
                if max_steps is not None and step >= max_steps:
                    break
        logger.info(f"epoch {epoch}: step {step}, {time.time() - t0:.1f}s")
        save_model(model, os.path.join(model_dir, f"epoch_{epoch}"), config)
        if max_steps is not None and step >= max_steps:
            break
    save_model(model, os.path.join(model_dir, "final"), config)
    return model, history
