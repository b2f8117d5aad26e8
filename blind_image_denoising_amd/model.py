"""
model_builder / hydra: the host-side mirror of bfcnn/model.py for the resnet backbone.

The reference builds a Keras graph (normalize -> backbone -> denoiser head -> denormalize,
bfcnn/model.py:58-162); here `model_builder` parses the same configuration into the C-ABI
description (include/bfcnn_hip.h: bf_resnet_desc), and the returned `hydra` is a thin object
that owns the flat parameter / BN-state / packed / workspace buffers (torch tensors used purely
as device-memory containers) and enqueues the gfx950 kernels through the C ABI.
There is no CPU execution path.
"""
import copy
import ctypes as C
import json
import math
import os
from collections import namedtuple
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import _native as N
from .constants import *
from .custom_logger import logger
from .utilities import input_shape_fixer, load_config

# bfcnn/model.py:25-34
BuilderResults = namedtuple(
    "BuilderResults", ["backbone", "normalizer", "denormalizer", "denoiser", "hydra", "options"])

_ACT = {"linear": N.BF_ACT_LINEAR, "relu": N.BF_ACT_RELU}
_LEAKY = {"leakyrelu": 0.3, "leaky_relu": 0.3, "leakyrelu_01": 0.1, "leaky_relu_01": 0.1,
          "leaky_relu_001": 0.01, "leakyrelu_001": 0.01}     # bfcnn/utilities.py:240-251
_REG = {None: N.BF_REG_NONE, "l1": N.BF_REG_L1, "l2": N.BF_REG_L2}


def _act_code(name: Optional[str]):
    name = (name or "linear").lower().strip()
    if name in _ACT:
        return _ACT[name], 0.0
    if name in _LEAKY:
        return N.BF_ACT_LEAKY_RELU, _LEAKY[name]
    raise NotImplementedError(f"activation [{name}] is outside the hot path")


def _reg_code(reg):
    if isinstance(reg, str):
        reg = reg.lower().strip()
    if reg not in _REG:
        raise NotImplementedError(
            f"regularizer [{reg}]: only the keras strings 'l1' / 'l2' are on the hot path")
    return _REG[reg]


def describe_resnet(config: Dict, strict_snapshot: bool = False) -> N.ResnetDesc:
    """config = the `model` section of a pipeline JSON.  Follows the argument handling of
    bfcnn/model.py:168-245 (model_backbone_builder), bfcnn/backbone_resnet.py:19-128 (builder
    defaults and checks) and bfcnn/model.py:266-275 (denoiser defaults)."""
    config_backbone = copy.deepcopy(config[BACKBONE_STR])
    config_denoiser = copy.deepcopy(config[DENOISER_STR])
    model_type = config_backbone[TYPE_STR].strip().lower()
    if model_type in ("unet", "convnext"):
        raise NotImplementedError(
            f"backbone [{model_type}] is outside the MI355X hot path (resnet and unet_laplacian only)")
    if model_type == "unet_laplacian":
        raise ValueError("unet_laplacian models are described by blind_image_denoising_amd.unet_laplacian")
    if model_type == "efficientnet":
        raise NotImplementedError("efficientnet not implemented yet")          # model.py:213
    if model_type != "resnet":
        raise ValueError("don't know how to build model [{0}]".format(model_type))   # model.py:215

    value_range = config_backbone.get("value_range", (0, 255))
    input_shape = input_shape_fixer(config_backbone.get(INPUT_SHAPE_STR, (None, None, 1)))
    block_kernels = list(config_backbone.get("block_kernels", [3, 3]))
    block_filters = list(config_backbone.get("block_filters", [32, 32]))
    kernel_regularizer = config_backbone.get("kernel_regularizer", "l1")
    block_regularizer = config_backbone.get("block_regularizer") or [kernel_regularizer] * len(block_kernels)
    activation = config_backbone.get("activation", "relu")
    block_activation = config_backbone.get("block_activation") or [activation] * len(block_kernels)
    block_depthwise = config_backbone.get("block_depthwise") or [-1] * len(block_kernels)
    block_groups = config_backbone.get("block_groups") or [1] * len(block_kernels)

    # --- argument checking (bfcnn/backbone_resnet.py:110-127)
    if len(block_kernels) <= 0:
        raise ValueError("len(block_kernels) must be >= 0 ")
    if len(block_kernels) > 3:
        raise ValueError("len(block_kernels) must be <= 3")
    if len(block_filters) <= 0:
        raise ValueError("len(block_filters) must be >= 0 ")
    if len(block_kernels) != len(block_filters):
        raise ValueError("len(block_filters) must == len(block_kernels)")
    if len(block_kernels) != len(block_groups):
        raise ValueError("len(block_filters) must == len(block_groups)")
    if len(block_regularizer) != len(block_groups):
        raise ValueError("len(block_regularizer) must == len(block_groups)")
    if len(block_activation) != len(block_groups):
        raise ValueError("len(block_activation) must == len(block_groups)")
    if len(block_depthwise) != len(block_kernels):
        raise ValueError("len(block_depthwise) must == len(block_kernels)")

    unsupported = [k for k in ("add_gelu", "add_gates", "add_final_bn", "add_initial_bn",
                               "add_concat_input", "add_gradient_dropout", "add_channelwise_scaling",
                               "add_learnable_multiplier", "add_mean_sigma_normalization")
                   if config_backbone.get(k, False)]
    if config_backbone.get("selector_params") is not None:
        unsupported.append("selector_params")
    if config_backbone.get("dropout_rate", -1) != -1:
        unsupported.append("dropout_rate")
    if any(d != -1 for d in block_depthwise):
        unsupported.append("block_depthwise")
    if any(g != 1 for g in block_groups):
        unsupported.append("block_groups")
    if config_backbone.get(USE_BIAS, False) or config_denoiser.get(USE_BIAS, False):
        unsupported.append("use_bias")
    if config_denoiser.get("use_bn", False) or config_denoiser.get("use_ln", False):
        unsupported.append("denoiser use_bn/use_ln")
    if unsupported:
        raise NotImplementedError(f"resnet options outside the hot path: {unsupported}")
    if len(set(block_kernels)) != 1 or len(set(block_filters)) != 1 or \
            block_filters[0] != config_backbone["filters"]:
        raise NotImplementedError("block_kernels / block_filters must be uniform and equal to `filters`")
    if len(set(block_regularizer)) != 1:
        raise NotImplementedError("block_regularizer must be uniform")

    d = N.ResnetDesc()
    d.struct_size = C.sizeof(N.ResnetDesc)
    d.in_channels = int(input_shape[-1])
    d.filters = int(config_backbone["filters"])
    d.kernel_size = int(config_backbone["kernel_size"])
    d.no_layers = int(config_backbone["no_layers"])
    d.block_convs = len(block_kernels)
    d.block_kernel = int(block_kernels[0])
    # conv i of a block uses block_activation[i]; the last is forced to base_activation
    # (bfcnn/backbone_resnet.py:178)
    d.activation, alpha0 = _act_code(block_activation[0])
    d.base_activation, _ = _act_code(config_backbone.get("base_activation", "linear"))
    d.use_bn = 1 if config_backbone.get("use_bn", True) else 0
    d.head_filters = int(config_denoiser.get("filters", 32))
    d.head_activation, alpha1 = _act_code(config_denoiser.get("activation", "linear"))
    d.out_channels = int(config_denoiser.get("output_channels", 3))
    d.denormalize = 0 if strict_snapshot else 1
    d.reg_base = _reg_code(kernel_regularizer)
    d.reg_block = _reg_code(block_regularizer[0])
    d.reg_head = _reg_code(config_denoiser.get(KERNEL_REGULARIZER, "l2"))
    d.v_min, d.v_max = float(value_range[0]), float(value_range[1])
    d.bn_eps, d.bn_momentum = DEFAULT_BN_EPSILON, DEFAULT_BN_MOMENTUM
    d.leaky_alpha = alpha1 or alpha0
    return d


def glorot_normal(shape: Sequence[int], rng: np.random.Generator) -> np.ndarray:
    """keras "glorot_normal" initializer (bfcnn/backbone_resnet.py:36, model.py:273): truncated
    normal, stddev = sqrt(2/(fan_in+fan_out)) / 0.87962566103423978, resampled outside 2 sigma.
    Host-side, runs once at model creation."""
    kh, kw, ci, co = shape
    std = math.sqrt(2.0 / (kh * kw * ci + kh * kw * co)) / 0.87962566103423978
    w = rng.standard_normal(shape)
    bad = np.abs(w) > 2.0
    while bad.any():
        w[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(w) > 2.0
    return (w * std).astype(np.float32)


class Variable:
    """A named view into the flat parameter buffer (stands in for a keras variable)."""

    def __init__(self, name, tensor, kind, regularizer, offset):
        self.name, self.tensor, self.kind, self.regularizer, self.offset = name, tensor, kind, regularizer, offset

    @property
    def shape(self):
        return tuple(self.tensor.shape)

    def numpy(self):
        return self.tensor.detach().cpu().numpy()

    def __repr__(self):
        return f"<Variable {self.name} {self.shape}>"


class HydraModel:
    """`hydra` of bfcnn/model.py:145-151 executed by libbfcnn_hip.so.

    model(x) / model(x, training=False): float32 [B,H,W,C] in value_range -> denoised float32
    (test_step, bfcnn/train_loop.py:253-257); model(x, training=True): training-mode forward
    with batch statistics, moving stats updated (train_step, train_loop.py:249-251)."""

    name = "hydra"

    def __init__(self, config: Dict, device: Optional[Union[str, torch.device]] = None,
                 strict_snapshot: bool = False, seed: Optional[int] = None):
        self.config = copy.deepcopy(config)
        self.desc = describe_resnet(config, strict_snapshot)
        self._lib = N.lib()
        handle = C.c_void_p()
        N.check(self._lib.bf_create(C.byref(self.desc), C.byref(handle)), None, "model_builder")
        self._h = handle
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() \
                else torch.device("cpu")
        self.device = torch.device(device)
        self.n_params = int(self._lib.bf_param_count(self._h))
        self.n_state = int(self._lib.bf_state_count(self._h))
        self._infos, self._state_infos = self._tensor_infos(0), self._tensor_infos(1)
        params, state = self._initial_values(seed)
        self.params = torch.from_numpy(params).to(self.device)
        self.state = torch.from_numpy(state).to(self.device)
        self._packed = None
        self._packed_dirty = True
        self.version = 0                 # bumped whenever weights, state or an option change (GraphedDenoiserModule keys on it)
        self._workspace = None
        self.outputs = [None]            # single-output hydra (len(model.outputs) == 1)
        self.inputs = [None]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.bf_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- parameter inventory -----------------------------------------------------------
    def _tensor_infos(self, state: int):
        out = []
        ti = N.TensorInfo()
        for i in range(self._lib.bf_tensor_count(self._h, state)):
            N.check(self._lib.bf_tensor_at(self._h, state, i, C.byref(ti)), self._h)
            out.append((ti.name.decode(), int(ti.offset), tuple(ti.shape[:ti.rank]), int(ti.kind),
                        int(ti.regularizer)))
        return out

    def _initial_values(self, seed):
        rng = np.random.default_rng(seed)
        params = np.empty(self.n_params, np.float32)
        for name, off, shape, kind, _ in self._infos:
            n = int(np.prod(shape))
            params[off:off + n] = glorot_normal(shape, rng).ravel() if kind == 0 else 1.0   # gamma = 1
        state = np.empty(self.n_state, np.float32)
        for name, off, shape, kind, _ in self._state_infos:
            state[off:off + int(np.prod(shape))] = 0.0 if kind == 2 else 1.0   # moving_mean 0, moving_var 1
        return params, state

    @property
    def trainable_variables(self) -> List[Variable]:
        return [Variable(n, self.params[o:o + int(np.prod(s))].view(*s), k, r, o)
                for n, o, s, k, r in self._infos]

    @property
    def non_trainable_variables(self) -> List[Variable]:
        return [Variable(n, self.state[o:o + int(np.prod(s))].view(*s), k, r, o)
                for n, o, s, k, r in self._state_infos]

    def count_params(self) -> int:
        return self.n_params + self.n_state

    def get_weights(self):
        return self.params.detach().cpu().numpy().copy(), self.state.detach().cpu().numpy().copy()

    def set_weights(self, params: np.ndarray, state: Optional[np.ndarray] = None):
        params = np.ascontiguousarray(params, np.float32).ravel()
        if params.size != self.n_params:
            raise ValueError(f"expected {self.n_params} parameters, got {params.size}")
        self.params.copy_(torch.from_numpy(params))
        if state is not None:
            state = np.ascontiguousarray(state, np.float32).ravel()
            if state.size != self.n_state:
                raise ValueError(f"expected {self.n_state} state values, got {state.size}")
            self.state.copy_(torch.from_numpy(state))
        self.mark_dirty()

    def mark_dirty(self):
        """call after params / state changed: inference re-packs the weights on next use."""
        self._packed_dirty = True
        self.version = getattr(self, "version", 0) + 1

    # ---- buffers -------------------------------------------------------------------------
    def _require_gpu(self):
        if self.device.type != "cuda":
            raise RuntimeError("this model lives on the CPU: the engine has no CPU execution path "
                               "(construct it with a cuda device on an MI355X)")

    def workspace(self, mode: int, batch: int, height: int, width: int) -> torch.Tensor:
        need = int(self._lib.bf_workspace_bytes(self._h, mode, batch, height, width))
        if need < 0:
            raise ValueError("invalid workspace query")
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = None
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._workspace

    def packed(self) -> torch.Tensor:
        if self._packed is None:
            self._packed = torch.empty(int(self._lib.bf_packed_bytes(self._h)), dtype=torch.uint8, device=self.device)
        if self._packed_dirty:
            N.check(self._lib.bf_pack_inference(self._h, N.ptr(self.params), N.ptr(self.state), N.ptr(self._packed),
                                                N.stream_ptr(self.params)), self._h, "bf_pack_inference")
            self._packed_dirty = False
        return self._packed

    def set_option(self, key: str, value: int):
        self.version = getattr(self, "version", 0) + 1
        if self.device.type == "cuda":
            with torch.cuda.device(self.device):       # ("timing" creates its events: on THIS model's device)
                N.check(self._lib.bf_set_option(self._h, key.encode(), int(value)), self._h)
        else:
            N.check(self._lib.bf_set_option(self._h, key.encode(), int(value)), self._h)

    def block_kernel(self):
        """(name of the kernel that ran most of the residual-block launches of the last forward, block launches of that forward),
        as the library reports them (bf_get_block_kernel): what a profile of the run must show."""
        import ctypes as C
        n = C.c_int()
        name = self._lib.bf_get_block_kernel(self._h, C.byref(n))
        return (name.decode() if name else ""), int(n.value)

    # ---- execution -----------------------------------------------------------------------
    def _as_device(self, x, dtype):
        was_numpy = isinstance(x, np.ndarray)
        if was_numpy:
            x = torch.from_numpy(np.ascontiguousarray(x))
        if not isinstance(x, torch.Tensor):
            raise ValueError("input must be a torch.Tensor or numpy array")
        if x.dim() != 4:
            raise ValueError(f"input must be rank 4 [B,H,W,C], got shape {tuple(x.shape)}")
        if x.shape[-1] != self.desc.in_channels:
            raise ValueError(f"expected {self.desc.in_channels} channels, got {x.shape[-1]}")
        x = x.to(device=self.device, dtype=dtype).contiguous()
        return x, was_numpy

    def __call__(self, x, training: bool = False):
        if isinstance(x, (list, tuple)):
            if len(x) != 1:
                raise ValueError("hydra takes one input tensor")
            x = x[0]
        self._require_gpu()
        x, was_numpy = self._as_device(x, torch.float32)
        B, H, W, _ = x.shape
        if training:
            out = self.training_forward(x)
        else:
            out = torch.empty((B, H, W, self.desc.out_channels), dtype=torch.float32, device=self.device)
            ws = self.workspace(N.BF_MODE_INFERENCE, B, H, W)
            N.check(self._lib.bf_forward_f32(self._h, N.ptr(self.packed()), N.ptr(x), N.ptr(out), B, H, W,
                                             N.ptr(ws), ws.numel(), N.stream_ptr(x)), self._h, "bf_forward_f32")
            if was_numpy and not self.check_status(raise_on_overflow=not self.auto_exact_fallback):
                # an activation left the f16 range: these weights need the exact-fp32 kernels -- switch for good, re-run
                logger.warning("activations exceed the f16 range: switching this model to the exact-fp32 kernels (arith = 0)")
                self.set_option("arith", 0)
                return self(x, training=False).cpu().numpy()
        return out.cpu().numpy() if was_numpy else out

    # True: when host arrays are handed back and the status word reports an f16-range overflow, the model switches to the
    # exact-fp32 kernels and the forward is repeated (a drop-in must not fail on weights the reference handles);
    # False: raise FloatingPointError instead.
    auto_exact_fallback = True

    def status_tensor(self) -> Optional[torch.Tensor]:
        """int32 view of the status word of the last inference forward (device memory, workspace tail), or None."""
        ws = self._workspace
        if ws is None:
            return None
        off = (ws.numel() - N.BF_STATUS_BYTES) // 4 * 4
        return ws[off:off + 4].view(torch.int32)

    def check_status(self, raise_on_overflow: bool = True) -> bool:
        """Reads the status word of the last inference forward (synchronises the stream).  Returns True when it is
        clean; when an activation left the f16 range inside the split-f16 blocks -- the result is then not trustworthy
        and the exact-fp32 kernels (`set_option("arith", 0)`) are the ones to use for these weights -- raises
        FloatingPointError or returns False."""
        ws = self._workspace
        if ws is None:
            return True
        n = ws.numel()
        off = (n - N.BF_STATUS_BYTES) // 4 * 4
        status = int(ws[off:off + 4].view(torch.int32).item())
        if status & N.BF_STATUS_F16_RANGE:
            if raise_on_overflow:
                raise FloatingPointError("an activation left the f16 range (|x| >= 65504) inside the split-f16 residual "
                                         "blocks; call set_option('arith', 0) to run the exact-fp32 kernels")
            return False
        return True

    def predict(self, x):
        return self(x, training=False)

    def infer_u8(self, image: torch.Tensor, cast_to_uint8: bool = True) -> torch.Tensor:
        """the fused DenoiserModule path: uint8 in -> uint8 out on device (cast_to_uint8=False: float32 out, not rounded)."""
        self._require_gpu()
        B, H, W, _ = image.shape
        out = torch.empty((B, H, W, self.desc.out_channels), dtype=torch.uint8 if cast_to_uint8 else torch.float32, device=self.device)
        ws = self.workspace(N.BF_MODE_INFERENCE, B, H, W)
        fn = self._lib.bf_forward_u8 if cast_to_uint8 else self._lib.bf_forward_u8_f32
        N.check(fn(self._h, N.ptr(self.packed()), N.ptr(image), N.ptr(out), B, H, W,
                                        N.ptr(ws), ws.numel(), N.stream_ptr(image)), self._h, "bf_forward_u8")
        return out

    def train_forward_backward(self, gt: torch.Tensor, noisy: torch.Tensor, loss_desc: N.LossDesc,
                               grads: torch.Tensor, losses: torch.Tensor, want_predictions: bool = True):
        """bf_train_step: training forward + loss + gradients (see train_loop.train_step_single_gpu)."""
        self._require_gpu()
        B, H, W, _ = noisy.shape
        pred = torch.empty((B, H, W, self.desc.out_channels), dtype=torch.float32, device=self.device) \
            if want_predictions else None
        ws = self.workspace(N.BF_MODE_TRAIN, B, H, W)
        N.check(self._lib.bf_train_step(self._h, N.ptr(self.params), N.ptr(self.state), N.ptr(gt), N.ptr(noisy),
                                        B, H, W, C.byref(loss_desc), N.ptr(pred), N.ptr(grads), N.ptr(losses),
                                        N.ptr(ws), ws.numel(), N.stream_ptr(noisy)), self._h, "bf_train_step")
        self.mark_dirty()        # BN moving statistics changed
        return pred

    def training_forward(self, x: torch.Tensor) -> torch.Tensor:
        """hydra(x, training=True) on its own (train_step of bfcnn/train_loop.py:249-251)."""
        ld = N.LossDesc()
        ld.struct_size = C.sizeof(N.LossDesc)
        ld.hinge, ld.cutoff, ld.mae_multiplier, ld.regularization, ld.depth_weight = 0.0, 255.0, 1.0, 0.0, 1.0
        grads = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
        losses = torch.empty(N.BF_LOSS_COUNT, dtype=torch.float32, device=self.device)
        if self.desc.in_channels != self.desc.out_channels:
            raise NotImplementedError("training needs in_channels == output_channels")
        return self.train_forward_backward(x, x, ld, grads, losses, True)

    @property
    def losses(self):
        """keras model.losses: one regularisation scalar per regularised kernel (bfcnn/loss.py:181-187), computed by the
        regulariser kernel of the operator library (0-d views of one device buffer)."""
        self._require_gpu()
        regs = [v for v in self.trainable_variables if v.regularizer in (N.BF_REG_L1, N.BF_REG_L2)]
        vals = torch.zeros(max(len(regs), 1), dtype=torch.float32, device=self.device)
        for k, v in enumerate(regs):
            N.check(self._lib.bf_op_reg_elementwise(N.ptr(v.tensor.reshape(-1)), None, v.tensor.numel(), v.regularizer, 0.01, 0.0,
                                                    N.ptr(vals[k:k + 1]), N.stream_ptr(vals)), None, "bf_op_reg_elementwise")
        return [vals[k] for k in range(len(regs))]


def _normalize_op(x, lo: float, hi: float, inverse: int):
    """the standalone (de)normalisation layer through the C ABI (bf_op_normalize); GPU tensors only"""
    if not isinstance(x, torch.Tensor) or x.device.type != "cuda":
        raise RuntimeError("normalize / denormalize need a GPU tensor: there is no CPU execution path")
    x = x.to(torch.float32).contiguous()
    out = torch.empty_like(x)
    N.check(N.lib().bf_op_normalize(N.ptr(x), N.ptr(out), x.numel(), lo, hi, inverse, N.stream_ptr(x)), None, "bf_op_normalize")
    return out


def build_normalize_model(input_dims=None, min_value: float = 0.0, max_value: float = 255.0, name: str = "normalize"):
    """bfcnn/model.py:364-394: [min,max] -> [-0.5,+0.5].  Standalone helper only; inside the
    hydra this is fused into the base-convolution kernel."""
    lo, hi = float(min_value), float(max_value)

    def normalize(x, training=False):
        return _normalize_op(x, lo, hi, 0)
    normalize.name = name
    return normalize


def build_denormalize_model(input_dims=None, min_value: float = 0.0, max_value: float = 255.0, name: str = "denormalize"):
    """bfcnn/model.py:399-430: [-0.5,+0.5] -> [min,max].  Fused into the head kernel in the hydra."""
    lo, hi = float(min_value), float(max_value)

    def denormalize(y, training=False):
        return _normalize_op(y, lo, hi, 1)
    denormalize.name = name
    return denormalize


class _SubModelView:
    """Names the backbone / denoiser sub-graphs of the hydra (keras sub-models in the reference).
    They are not separately executable: the engine runs the fused hydra."""

    def __init__(self, name: str, hydra: HydraModel, prefixes):
        self.name, self._hydra, self._prefixes = name, hydra, prefixes
        self.outputs = [None]

    @property
    def trainable_variables(self):
        return [v for v in self._hydra.trainable_variables if v.name.startswith(self._prefixes)]


def _build_hydra(config: Dict, device=None, strict_snapshot: bool = False, seed: Optional[int] = None):
    """backbone type -> model class: unet_laplacian; the 16-filter 3x3 resnet family (fused engine, trainable); any other
    resnet the operator library covers (per-position kernels / filters, depthwise multipliers, groups; inference)."""
    btype = str(config[BACKBONE_STR].get(TYPE_STR, "")).strip().lower()
    if btype == "unet_laplacian":
        from .unet_laplacian import UnetLaplacianHydra
        return UnetLaplacianHydra(config, device=device, seed=seed)
    try:
        return HydraModel(config, device=device, strict_snapshot=strict_snapshot, seed=seed)
    except NotImplementedError as first:
        if btype != "resnet":
            raise
        from .resnet_generic import GenericResnetHydra
        try:
            return GenericResnetHydra(config, device=device, seed=seed)
        except NotImplementedError as second:
            raise NotImplementedError(f"{first}; generic resnet path: {second}") from None


def model_builder(config: Dict, device=None, strict_snapshot: bool = False, seed: Optional[int] = None) -> BuilderResults:
    """bfcnn/model.py:58-162.  `config` is the `model` section ({"backbone":…, "denoiser":…})."""
    hydra = _build_hydra(config, device, strict_snapshot, seed)
    if not isinstance(hydra, HydraModel):
        outs = getattr(hydra, "depth", 1) if getattr(hydra, "multi_output", False) else 1
        logger.warning(f"Backbone model has [{outs}] outputs, probably of different scale or depth")
        return BuilderResults(
            backbone=None, denoiser=None, hydra=hydra, options={},
            normalizer=build_normalize_model(min_value=hydra.v_min, max_value=hydra.v_max),
            denormalizer=build_denormalize_model(min_value=hydra.v_min, max_value=hydra.v_max))
    vr = config[BACKBONE_STR].get("value_range", (0, 255))
    logger.warning(f"Backbone model has [1] outputs, probably of different scale or depth")
    return BuilderResults(
        backbone=_SubModelView(f"resnet_backbone", hydra, ("base/", "block")),
        normalizer=build_normalize_model(min_value=vr[0], max_value=vr[1]),
        denormalizer=build_denormalize_model(min_value=vr[0], max_value=vr[1]),
        denoiser=_SubModelView("denoiser_head", hydra, ("head/",)),
        hydra=hydra,
        options={})


# ---- model directories (this package's own on-disk format) ---------------------------------

def save_model(hydra: HydraModel, directory: str, pipeline_config: Optional[Dict] = None) -> None:
    """Writes `pipeline.json` (same schema as the reference pipelines) + `weights.npz`
    (flat params / state and the per-tensor table).  Stands in for export_model's SavedModel
    (bfcnn/export_model.py:106-140); TF serialisation formats are out of scope."""
    os.makedirs(directory, exist_ok=True)
    cfg = copy.deepcopy(pipeline_config) if pipeline_config else {MODEL_STR: hydra.config}
    cfg.setdefault(MODEL_STR, hydra.config)
    if not isinstance(hydra, HydraModel):                        # unet_laplacian / generic resnet: flat vectors + tensor table
        with open(os.path.join(directory, PIPELINE_FILE_STR), "w") as f:
            json.dump(cfg, f, indent=4)
        tv = hydra.trainable_variables
        w = hydra.get_weights()
        params, state = w if isinstance(w, tuple) else (w, np.zeros(0, np.float32))
        np.savez(os.path.join(directory, WEIGHTS_FILE_STR), params=params, state=state,
                 names=np.array([v[0] for v in tv]), offsets=np.array([v[3] for v in tv]))
        return
    cfg["strict_snapshot"] = not bool(hydra.desc.denormalize)
    with open(os.path.join(directory, PIPELINE_FILE_STR), "w") as f:
        json.dump(cfg, f, indent=4)
    params, state = hydra.get_weights()
    np.savez(os.path.join(directory, WEIGHTS_FILE_STR), params=params, state=state,
             names=np.array([i[0] for i in hydra._infos]), offsets=np.array([i[1] for i in hydra._infos]))


def load_hydra(directory: str, device=None) -> HydraModel:
    cfg_path = os.path.join(directory, PIPELINE_FILE_STR)
    w_path = os.path.join(directory, WEIGHTS_FILE_STR)
    if not os.path.isfile(cfg_path) or not os.path.isfile(w_path):
        raise ValueError(f"model_path [{directory}] does not hold {PIPELINE_FILE_STR} + {WEIGHTS_FILE_STR}")
    cfg = load_config(cfg_path)
    hydra = _build_hydra(cfg[MODEL_STR], device, bool(cfg.get("strict_snapshot", False)))
    if not isinstance(hydra, HydraModel):
        with np.load(w_path) as z:
            if getattr(hydra, "n_state", 0):
                hydra.set_weights(z["params"], z["state"])
            else:
                hydra.set_weights(z["params"])
        return hydra
    with np.load(w_path) as z:
        hydra.set_weights(z["params"], z["state"])
    return hydra
