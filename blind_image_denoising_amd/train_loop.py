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
from .optimizer import optimizer_builder, deep_supervision_schedule_builder
from .utilities import load_config

TrainFunctions = namedtuple("TrainFunctions", ["train_step", "test_step", "train_step_single_gpu", "apply_grads"])


def build_train_functions(model: HydraModel, loss_fn_map: Dict[str, Callable]) -> TrainFunctions:
    """The four closures of bfcnn/train_loop.py:249-321 for a single-output hydra.

    Losses come back as 0-d views of one device buffer (no host synchronisation); call
    `.item()` when a Python float is wanted."""
    denoiser_loss_fn = loss_fn_map[DENOISER_LOSS_FN_STR]
    if getattr(model, "multi_output", False):
        return _build_multi_output_train_functions(model, denoiser_loss_fn)
    if type(model).__name__ == "GenericResnetHydra":
        return _build_generic_resnet_train_functions(model, denoiser_loss_fn)
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


def _build_generic_resnet_train_functions(model, denoiser_loss_fn, seed: int = 0) -> TrainFunctions:
    """the four closures for the resnet configs outside the 16-filter 3x3 engine (GenericResnetHydra), through
    resnet_generic_train.GenericResnetTrainGraph (explicit forward / backward over the operator library).
    `dropout_rate` (RandomOnOff on every block's branch, backbone_blocks.py:223-225): the per-sample keep factors are drawn here per
    step from a NumPy generator; `train_step_single_gpu.randomness = False` switches them off, `.drop_scale = {block: [B] tensor}`
    pins them (parity tests)."""
    import numpy as np
    from .resnet_generic_train import GenericResnetTrainGraph
    rng = np.random.default_rng(seed)
    d = denoiser_loss_fn.desc(1.0)
    loss_config = {"hinge": d.hinge, "cutoff": d.cutoff, "mae_multiplier": d.mae_multiplier, "mse_multiplier": d.mse_multiplier,
                   "ssim_multiplier": d.ssim_multiplier, "regularization": d.regularization}
    graph = GenericResnetTrainGraph(model, loss_config)
    state = {"grads": None}

    def train_step(n):
        raise NotImplementedError("hydra(n, training=True) on its own is not built for this model; use train_step_single_gpu")

    def test_step(n):
        return model(n, training=False)

    def train_step_single_gpu(p_input_image_batch, p_noisy_image_batch, p_depth_weight=(1.0,), p_percentage_done=0.0,
                              p_trainable_variables=None):
        if state["grads"] is None or state["grads"].device != model.params.device:
            state["grads"] = torch.zeros(model.n_params, dtype=torch.float32, device=model.device)
        grads = state["grads"]
        dw = p_depth_weight[0] if hasattr(p_depth_weight, "__len__") else p_depth_weight
        dw = 1.0 if dw is None else dw
        ds = train_step_single_gpu.drop_scale
        rate = getattr(model, "dropout_rate", -1)
        if ds is None and train_step_single_gpu.randomness and rate > 0.0:
            B = int(p_noisy_image_batch.shape[0])
            ds = {i: torch.from_numpy((rng.uniform(size=B) >= rate).astype(np.float32) / np.float32(1.0 - rate)).to(model.device)
                  for i in range(model.no_layers)}
        pred, sl, totals = graph.step(p_input_image_batch, p_noisy_image_batch, grads, float(dw), ds)
        model_loss = {REGULARIZATION_LOSS_STR: totals[1], TOTAL_LOSS_STR: totals[2]}
        denoiser_loss = {TOTAL_LOSS_STR: sl[N.BF_LOSS_DENOISER_TOTAL], MSE_LOSS_STR: sl[N.BF_LOSS_MSE], MAE_LOSS_STR: sl[N.BF_LOSS_MAE],
                         SSIM_LOSS_STR: sl[N.BF_LOSS_SSIM]}
        return totals[0], model_loss, [denoiser_loss], pred, grads

    train_step_single_gpu.randomness = True
    train_step_single_gpu.drop_scale = None

    def apply_grads(internal_optimizer, internal_gradients, internal_trainable_variables=None, grad_scale: float = 1.0):
        internal_optimizer.apply_gradients(internal_gradients, model, grad_scale=grad_scale, losses=None)

    return TrainFunctions(train_step, test_step, train_step_single_gpu, apply_grads)


def _build_multi_output_train_functions(model, denoiser_loss_fn, seed: int = 0) -> TrainFunctions:
    """the same four closures for a multi-output hydra (unet_laplacian): one denoiser loss per output scale against the
    ground-truth pyramid, times its depth weight (bfcnn/train_loop.py:273-294), through unet_train.UnetTrainGraph.

    Training-mode randomness (StochasticDepth on the blocks' branches, dropout on the attention weights:
    backbone_unet_laplacian.py:176-177, 333, 351-352) is drawn here per step from a NumPy generator and handed to the graph as
    explicit scale tensors; `train_step_single_gpu.randomness = False` switches it off (deterministic steps for parity tests)."""
    import numpy as np
    from .unet_train import UnetTrainGraph
    d = denoiser_loss_fn.desc(1.0)
    loss_config = {"hinge": d.hinge, "cutoff": d.cutoff, "mae_multiplier": d.mae_multiplier, "mse_multiplier": d.mse_multiplier,
                   "ssim_multiplier": d.ssim_multiplier, "regularization": d.regularization}
    graph = UnetTrainGraph(model, loss_config)
    bb = model.config["backbone"]
    depth_drop = [float(r) for r in np.linspace(0.0, max(0.0, float(bb.get("depth_drop_rate", 0.0))), model.width)]
    attn_drop = float(bb.get("convolutional_self_attention_dropout_rate", 0.0))
    rng = np.random.default_rng(seed)
    state = {"grads": None}

    def train_step(n):
        raise NotImplementedError("hydra(n, training=True) on its own is not built for unet_laplacian; use train_step_single_gpu")

    def test_step(n):
        """bfcnn/train_loop.py:253-257: the first (full-resolution) output"""
        return model(n, training=False)[0]

    def _randomness(B, H=0, W=0):
        ds, at = {}, {}
        if not train_step_single_gpu.randomness:
            return ds, at
        blocks = [f"enc{dl}_{w}" for dl in range(model.depth) for w in range(model.width)] + \
                 [f"dec{dl}_{w}" for dl in range(model.depth - 1) for w in range(model.width)]
        for prefix in blocks:
            rate = depth_drop[int(prefix.rsplit("_", 1)[1])]
            if rate > 0.0:            # StochasticDepth: the branch of a whole sample is dropped, kept ones scaled by 1 / (1 - rate)
                keep = (rng.uniform(size=B) >= rate).astype(np.float32) / np.float32(1.0 - rate)
                ds[prefix] = torch.from_numpy(keep).to(model.device)
            if attn_drop > 0.0 and prefix.startswith(f"enc{model.depth - 1}_") and model._is_attention(model.depth - 1):
                if model.attention_rows:          # one sequence per image row of the deepest level
                    NB, T = B * (H >> (model.depth - 1)), W >> (model.depth - 1)
                else:
                    NB, T = B, model.attention_resolution[0] * model.attention_resolution[1]
                keep = (rng.uniform(size=(NB, T, T)) >= attn_drop).astype(np.float32) / np.float32(1.0 - attn_drop)
                at[prefix] = torch.from_numpy(keep).to(model.device)
        return ds, at

    def train_step_single_gpu(p_input_image_batch, p_noisy_image_batch, p_depth_weight=None, p_percentage_done=0.0,
                              p_trainable_variables=None):
        if state["grads"] is None or state["grads"].device != model.params.device:
            state["grads"] = torch.zeros(model.n_params, dtype=torch.float32, device=model.device)
        grads = state["grads"]
        dw = [1.0] * model.depth if p_depth_weight is None else [float(v) for v in p_depth_weight]
        if len(dw) < model.depth:
            raise ValueError(f"{model.depth} output scales need {model.depth} depth weights, got {len(dw)}")
        ds, at = _randomness(*(int(v) for v in p_noisy_image_batch.shape[:3]))
        preds, scale_losses, totals = graph.step(p_input_image_batch, p_noisy_image_batch, dw, grads, ds, at)
        model_loss = {REGULARIZATION_LOSS_STR: totals[1], TOTAL_LOSS_STR: totals[2]}
        all_denoiser_loss = [{TOTAL_LOSS_STR: sl[N.BF_LOSS_DENOISER_TOTAL], MSE_LOSS_STR: sl[N.BF_LOSS_MSE], MAE_LOSS_STR: sl[N.BF_LOSS_MAE],
                              SSIM_LOSS_STR: sl[N.BF_LOSS_SSIM]} for sl in scale_losses]
        return totals[0], model_loss, all_denoiser_loss, preds, grads

    train_step_single_gpu.randomness = True

    def apply_grads(internal_optimizer, internal_gradients, internal_trainable_variables=None, grad_scale: float = 1.0):
        internal_optimizer.apply_gradients(internal_gradients, model, grad_scale=grad_scale, losses=None)

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


class NativeCommunicator:
    """The gradient exchange through the C ABI (bf_comm_* / bf_allreduce_grads: RCCL bound inside libbfcnn_hip.so), for a
    host that does not want PyTorch on the data path.  The 128-byte rendezvous id travels over whatever channel the host has;
    here torch.distributed's (already initialised) process group carries it once."""

    def __init__(self, device, group=None):
        import ctypes as C
        import torch.distributed as dist
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        lib = N.lib()
        # Every rank first proves that it CAN talk to RCCL (bf_comm_unique_id binds librccl at run time), and the ranks agree on it
        # before anything blocks: bf_comm_init_rank is itself a collective, so one rank failing alone would leave its peers waiting
        # inside it.  After this point all ranks raise together or none does.
        buf = C.create_string_buffer(128)
        rc = lib.bf_comm_unique_id(buf)
        note = "" if rc == N.BF_OK else lib.bf_comm_last_error().decode()
        flag = torch.tensor([1 if rc == N.BF_OK else 0], dtype=torch.int32,
                            device="cpu" if dist.get_backend(group) == "gloo" else device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) != 1:
            raise RuntimeError("bf_comm_unique_id failed on " + ("this rank: " + note if note else "another rank"))
        box = [buf.raw if self.rank == 0 else None]
        # `src` of a torch broadcast is a GLOBAL rank: the group's rank 0 is not global rank 0 in a sub-group
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
        comm = C.c_void_p()
        with torch.cuda.device(device):
            rc = lib.bf_comm_init_rank(C.byref(comm), self.world, self.rank, box[0])
        if rc != N.BF_OK:
            raise RuntimeError(f"bf_comm_init_rank: {lib.bf_comm_last_error().decode()}")
        self._comm, self._lib = comm, lib
        # the collective runs on its OWN stream, ordered against the compute stream by events, so that whatever the caller queues
        # on the compute stream between launch() and wait() really overlaps with it
        self._stream = torch.cuda.Stream(device=device)

    def launch(self, grads: torch.Tensor):
        """all-reduce of `grads` on the communicator's stream, behind everything the compute stream has queued so far"""
        cur = torch.cuda.current_stream(grads.device)
        self._stream.wait_stream(cur)
        rc = self._lib.bf_allreduce_grads(None, N.ptr(grads), grads.numel(), self._comm, self._stream.cuda_stream)
        if rc != N.BF_OK:
            raise RuntimeError(f"bf_allreduce_grads: {self._lib.bf_comm_last_error().decode()}")
        grads.record_stream(self._stream)

    def wait(self, grads: torch.Tensor):
        """the compute stream waits (stream-ordered, no host block) for the all-reduce launch() started"""
        torch.cuda.current_stream(grads.device).wait_stream(self._stream)

    def allreduce(self, grads: torch.Tensor):
        self.launch(grads)
        self.wait(grads)

    def close(self):
        if self._comm:
            self._lib.bf_comm_destroy(self._comm)
            self._comm = None


class DataParallelTrainer:
    """train_step_single_gpu + all-reduce + apply_grads for one rank of an N-GPU job.

    The step has ONE collective, on one flat buffer whose consumer (the global-norm clip of Adam) needs all of it: the
    all-reduce cannot overlap with the optimizer kernels themselves.  What it does overlap with is the work that does not
    depend on it: `step(..., overlap=fn)` runs `fn()` -- typically the on-device corruption of the NEXT batch
    (PrepareData / bf_noise_augment) -- between the launch of the all-reduce (asynchronous, on RCCL's stream) and the point
    where the compute stream waits for it."""

    def __init__(self, model: HydraModel, loss_fn_map, optimizer, group=None, native_collective: bool = False):
        import torch.distributed as dist
        self.model, self.optimizer, self.group = model, optimizer, group
        self.fns = build_train_functions(model, loss_fn_map)
        self.world_size = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # native_collective="force": the C ABI's collective even in a world of one (a one-rank RCCL group still launches its kernel on the
        # communicator's stream: bench.py --force-collective, the two-stream trace of a step under profiles/)
        self.comm = NativeCommunicator(model.device, group) if (native_collective == "force" or (native_collective and self.world_size > 1)) else None

    def broadcast_parameters(self, src: int = 0):
        import torch.distributed as dist
        if self.world_size > 1:
            dist.broadcast(self.model.params, src=src, group=self.group)
            state = getattr(self.model, "state", None)      # BatchNorm statistics: unet_laplacian graphs have none
            if state is not None and state.numel() > 0:
                dist.broadcast(state, src=src, group=self.group)
            self.model.mark_dirty()

    def _depth_weights(self, depth_weight):
        """a scalar stands for every output scale of a multi-output (unet_laplacian) model; a sequence is passed through"""
        if isinstance(depth_weight, (tuple, list)):
            return tuple(float(v) for v in depth_weight)
        n = int(getattr(self.model, "depth", 1)) if getattr(self.model, "multi_output", False) else 1
        return (float(depth_weight),) * n

    def step(self, gt_shard: torch.Tensor, noisy_shard: torch.Tensor, depth_weight=1.0, overlap: Optional[Callable] = None):
        """one data-parallel step on this rank's shard.  Always returns (total, model_loss, denoiser_loss, predictions, side):
        side = overlap()'s result, None without an overlap function."""
        total, model_loss, denoiser_loss, predictions, grads = self.fns.train_step_single_gpu(
            gt_shard, noisy_shard, self._depth_weights(depth_weight), 0.0, None)
        side = None
        if self.comm is not None:
            self.comm.launch(grads)         # RCCL kernel on the communicator's own stream, behind the backward pass
            if overlap is not None:
                side = overlap()            # runs on the compute stream while the all-reduce is in flight
            self.comm.wait(grads)           # stream-ordered: no host block, the Adam kernels queue behind it
        else:
            work = allreduce_gradients(grads, self.group, async_op=True)
            if overlap is not None:
                side = overlap()            # runs on the compute stream while RCCL's stream carries the all-reduce
            if work is not None:
                work.wait()                 # stream-ordered on RCCL: no host block, the Adam kernels queue behind it
        self.fns.apply_grads(self.optimizer, grads, None, grad_scale=1.0 / self.world_size)
        return total, model_loss, denoiser_loss, predictions, side


# ---- outer loop --------------------------------------------------------------------------------

def train_loop(pipeline_config_path, checkpoint_directory: str = None, weights_dir: str = None, dataset: Iterable = None,
               device=None, max_steps: Optional[int] = None, model_dir: str = None):
    """bfcnn/train_loop.py:40-601 with the reference's positional order (pipeline_config_path, checkpoint_directory, weights_dir);
    `model_dir` is this package's older keyword for the same directory.  Returns (model, loss history).  See `_train_loop`."""
    model_dir = checkpoint_directory if checkpoint_directory is not None else model_dir
    if model_dir is None:
        raise ValueError("checkpoint_directory must be given")
    return _train_loop(pipeline_config_path, str(model_dir), dataset, None if weights_dir is None else str(weights_dir), device, max_steps)


def _train_loop(pipeline_config_path, model_dir: str, dataset: Iterable = None, weights_dir: str = None,
                device=None, max_steps: Optional[int] = None):
    """Outer loop of bfcnn/train_loop.py:40-601 reduced to what surrounds the hot path: config ->
    loss / optimizer / model builders -> checkpoint manager (restore the latest checkpoint of `model_dir` if there is one:
    weights, BN statistics, Adam slots, step, epoch -- train_loop.py:158-181) -> epochs over `dataset` (an iterable
    yielding (input_image_batch, noisy_image_batch) float tensors in value range) with gradient accumulation over
    `gpu_batches_per_step` micro-batches; `dataset=None`: the configuration's own `dataset` section, image directories and all), a checkpoint every `checkpoint_every` steps and at the end of every epoch
    (train_loop.py:563-566, 597) -> model directory per epoch.
    The tf.data pipeline, TensorBoard summaries and TF's checkpoint format are out of scope."""
    from .checkpoint import Checkpoint, CheckpointManager
    config = load_config(pipeline_config_path)
    train_config = config["train"]
    epochs = int(train_config["epochs"])
    total_steps = int(train_config.get("total_steps", -1))                 # train_loop.py:103
    checkpoint_every = int(train_config.get("checkpoint_every", -1))       # train_loop.py:107-108
    checkpoints_to_keep = int(train_config.get("checkpoints_to_keep", 3))  # train_loop.py:105-106
    gpu_batches_per_step = int(train_config.get("gpu_batches_per_step", 1))
    if gpu_batches_per_step <= 0:
        raise ValueError("gpu_batches_per_step must be > 0")               # train_loop.py:114-115
    if dataset is None:
        # bfcnn/train_loop.py:81-85: the dataset of the configuration's own `dataset` section (image directories)
        if DATASET_STR not in config:
            raise ValueError("no dataset: pass an iterable of (input_image_batch, noisy_image_batch) or a configuration with a dataset section")
        from .dataset import dataset_builder
        dataset = dataset_builder(config[DATASET_STR], device=device, seed=train_config.get("seed")).training
    loss_fn_map = loss_function_builder(config=config["loss"])
    optimizer, lr_schedule = optimizer_builder(config=train_config["optimizer"])
    # `seed` in the train section (an extension; absent = keras-like non-deterministic initialisation): initial weights and the
    # dataset's draws reproducible
    model = model_builder(config[MODEL_STR], device=device, seed=train_config.get("seed")).hydra
    ckpt = Checkpoint(model=model, optimizer=optimizer)
    manager = CheckpointManager(checkpoint=ckpt, directory=model_dir, max_to_keep=checkpoints_to_keep)
    if not manager.restore_latest() and weights_dir:
        from .model import load_hydra
        logger.info(f"loading weights from [{weights_dir}]")               # train_loop.py:183-210
        w = load_hydra(weights_dir, device=device).get_weights()
        model.set_weights(*w) if isinstance(w, tuple) else model.set_weights(w)
    fns = build_train_functions(model, loss_fn_map)
    # per-output loss weights over the course of training (bfcnn/train_loop.py:350-381, optimizer.py:21-78)
    no_outputs = int(getattr(model, "depth", 1)) if getattr(model, "multi_output", False) else 1
    deep_supervision_schedule = deep_supervision_schedule_builder(
        config=train_config.get("deep_supervision", {TYPE_STR: "linear_low_to_high"}), no_outputs=no_outputs)
    accumulated = torch.empty(model.n_params, dtype=torch.float32, device=model.device)
    history = []
    finished = 0 < total_steps <= ckpt.step          # a restored run may already be complete
    # bfcnn/train_loop.py:359: epochs == -1 trains until total_steps
    while not finished and (epochs == -1 or ckpt.epoch < epochs):
        counter = 0
        t0 = time.time()
        if epochs > 0:                               # train_loop.py:366-371
            percentage_done = float(ckpt.epoch) / float(epochs)
        elif total_steps > 0:
            percentage_done = float(ckpt.step) / float(total_steps)
        else:
            percentage_done = 0.0
        depth_weight = tuple(float(v) for v in deep_supervision_schedule(percentage_done=percentage_done))
        logger.info("percentage done [{:.2f}], weight per output index: {}".format(percentage_done, ["{0:.2f}".format(d) for d in depth_weight]))
        for input_image_batch, noisy_image_batch in dataset:
            total, _, denoiser_loss, _, grads = fns.train_step_single_gpu(input_image_batch, noisy_image_batch, depth_weight,
                                                                          percentage_done, None)
            # accumulated (+)= grads on the engine's stream (first micro-batch of a step: overwrite)
            N.check(N.lib().bf_op_axpy(N.ptr(accumulated), N.ptr(grads), 1.0, int(counter == 0), accumulated.numel(),
                                       N.stream_ptr(accumulated)), None, "bf_op_axpy")
            counter += 1
            if counter >= gpu_batches_per_step:
                fns.apply_grads(optimizer, accumulated, None, grad_scale=1.0 / counter)
                counter = 0
                history.append(float(total.item()))
                if checkpoint_every > 0 and ckpt.step > 0 and ckpt.step % checkpoint_every == 0:
                    manager.save()
                ckpt.step += 1
                if (0 < total_steps <= ckpt.step) or (max_steps is not None and len(history) >= max_steps):
                    finished = True
                    break
        logger.info(f"end of epoch [{ckpt.epoch}], step {ckpt.step}, took [{time.time() - t0:.1f}] seconds")
        save_model(model, os.path.join(model_dir, f"epoch_{ckpt.epoch}"), config)
        if not finished:
            ckpt.epoch += 1
        manager.save()
    save_model(model, os.path.join(model_dir, "final"), config)
    return model, history
