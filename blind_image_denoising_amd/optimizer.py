"""optimizer_builder / schedule_builder / deep_supervision_schedule_builder (bfcnn/optimizer.py).

Learning-rate schedules are host scalars; the update itself is the fused global-norm + Adam
kernel pair behind bf_adam_step."""
import math
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch

from . import _native as N
from .constants import *
from .custom_logger import logger


def deep_supervision_schedule_builder(config: Dict, no_outputs: int) -> Callable[[float], np.ndarray]:
    """bfcnn/optimizer.py:21-78 (host-side per-output loss weights)."""
    if not isinstance(config, dict):
        raise ValueError("config must be a dictionary")
    if no_outputs <= 0:
        raise ValueError("no_outputs must be positive integer")
    schedule_type = config.get(TYPE_STR, None)
    if schedule_type is None:
        raise ValueError("schedule_type cannot be None")
    if not isinstance(schedule_type, str):
        raise ValueError("schedule_type must be a string")
    schedule_type = schedule_type.strip().lower()
    base = np.array(list(range(1, no_outputs + 1))).astype(np.float32)
    base = base / np.sum(base)
    if schedule_type == "constant_equal":
        return lambda percentage_done=0.0: np.array([1.0] * no_outputs) / no_outputs
    if schedule_type == "constant_low_to_high":
        return lambda percentage_done=0.0: base.copy()
    if schedule_type == "constant_high_to_low":
        return lambda percentage_done=0.0: base[::-1].copy()
    if schedule_type == "linear_low_to_high":
        return lambda percentage_done=0.0: base * (1.0 - percentage_done) + base[::-1] * percentage_done
    if schedule_type == "non_linear_low_to_high":
        def schedule(percentage_done: float = 0.0):
            x = np.clip(np.tanh(2.5 * percentage_done), a_min=0.0, a_max=1.0)
            return base * (1.0 - x) + base[::-1] * x
        return schedule
    raise ValueError(f"don't know how to handle deep supervision schedule_type [{schedule_type}]")


class LearningRateSchedule:
    """callable step -> learning rate (keras LearningRateSchedule)."""

    def __init__(self, fn: Callable[[float], float], config: Dict):
        self._fn, self.config = fn, config

    def __call__(self, step) -> float:
        return float(self._fn(float(step)))


def schedule_builder(config: Dict) -> LearningRateSchedule:
    """bfcnn/optimizer.py:83-139 with the keras-2.13 schedule formulas."""
    if not isinstance(config, dict):
        raise ValueError("config must be a dictionary")
    schedule_type = config.get(TYPE_STR, None)
    if schedule_type is None:
        raise ValueError("schedule_type cannot be None")
    if not isinstance(schedule_type, str):
        raise ValueError("schedule_type must be a string")
    params = config.get(CONFIG_STR, {})
    schedule_type = schedule_type.strip().lower()
    logger.info(f"building schedule: {schedule_type}, with params: {params}")
    if schedule_type == "exponential_decay":
        decay_rate, decay_steps, lr0 = params["decay_rate"], params["decay_steps"], params["learning_rate"]
        # keras ExponentialDecay, staircase=False
        return LearningRateSchedule(lambda step: lr0 * decay_rate ** (step / decay_steps), config)
    if schedule_type == "cosine_decay":
        decay_steps, lr0 = params["decay_steps"], params["learning_rate"]
        alpha = params.get("alpha", 0.0001)

        def cosine(step):
            s = min(step, decay_steps)
            c = 0.5 * (1.0 + math.cos(math.pi * s / decay_steps))
            return lr0 * ((1.0 - alpha) * c + alpha)
        return LearningRateSchedule(cosine, config)
    if schedule_type == "cosine_decay_restarts":
        first, lr0 = params["decay_steps"], params["learning_rate"]
        t_mul, m_mul, alpha = params.get("t_mul", 2.0), params.get("m_mul", 0.9), params.get("alpha", 0.001)

        def restarts(step):
            completed = step / first
            if t_mul == 1.0:
                i = math.floor(completed)
                frac = completed - i
            else:
                i = math.floor(math.log(1.0 - completed * (1.0 - t_mul)) / math.log(t_mul))
                sum_r = (1.0 - t_mul ** i) / (1.0 - t_mul)
                frac = (completed - sum_r) / t_mul ** i
            c = 0.5 * (m_mul ** i) * (1.0 + math.cos(math.pi * frac))
            return lr0 * ((1.0 - alpha) * c + alpha)
        return LearningRateSchedule(restarts, config)
    raise ValueError(f"don't know how to handle learning_rate schedule_type [{schedule_type}]")


class Adam:
    """keras-2.13 Adam (bfcnn/optimizer.py:190-206): slots m, v as flat device buffers, the
    update is bf_adam_step.  `iterations` counts applied steps like optimizer.iterations."""

    name = "Adam"

    def __init__(self, learning_rate, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False,
                 clipvalue=None, clipnorm=None, global_clipnorm=None):
        if amsgrad:
            raise NotImplementedError("amsgrad is outside the hot path")
        if clipnorm is not None and global_clipnorm is not None:            # keras 2.13 optimizer.__init__
            raise ValueError("At most one of `clipnorm` and `global_clipnorm` can be set")
        self.learning_rate = learning_rate
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
        self.global_clipnorm, self.clipnorm, self.clipvalue = global_clipnorm, clipnorm, clipvalue
        self.iterations = 0
        self.m = self.v = self._scratch = self._offsets = self._tensor_scratch = None

    def lr(self) -> float:
        lr = self.learning_rate
        return float(lr(self.iterations)) if callable(lr) else float(lr)

    def _slots(self, model):
        if self.m is None or self.m.numel() != model.n_params or self.m.device != model.params.device:
            self.m = torch.zeros_like(model.params)
            self.v = torch.zeros_like(model.params)
            self._scratch = torch.zeros(4, dtype=torch.float32, device=model.params.device)
            # offsets of the trainable tensors in the flat vector: per-tensor clipping (clipnorm / clipvalue) walks them
            infos = getattr(model, "_infos", None)                    # engine models: (name, offset, ...) ; operator-library
            starts = [int(i[1]) for i in infos] if infos is not None else [int(v[3]) for v in model.trainable_variables]
            offs = sorted(starts) + [int(model.n_params)]
            self._offsets = torch.tensor(offs, dtype=torch.int64, device=model.params.device)
            self._tensor_scratch = torch.zeros(len(offs), dtype=torch.float32, device=model.params.device)

    def apply_gradients(self, grads: torch.Tensor, model, grad_scale: float = 1.0, losses: Optional[torch.Tensor] = None):
        """grads: flat tensor laid out like model.params (what train_step_single_gpu returns)."""
        model._require_gpu()
        if grads.numel() != model.n_params:
            raise ValueError("gradient / variable size mismatch")
        self._slots(model)
        clip = float(self.global_clipnorm) if self.global_clipnorm else 0.0
        local = float(self.clipnorm) if self.clipnorm else 0.0
        value = float(self.clipvalue) if self.clipvalue else 0.0
        if getattr(model, "_h", None) is None:
            # a model assembled from the operator library (unet_laplacian): the same kernels on its flat vector
            N.check(N.lib().bf_op_adam_step(N.ptr(model.params), N.ptr(grads), N.ptr(self.m), N.ptr(self.v), int(model.n_params),
                                            int(self.iterations), self.lr(), self.beta_1, self.beta_2, self.epsilon, clip, local, value,
                                            N.ptr(self._offsets), int(self._offsets.numel() - 1), N.ptr(self._tensor_scratch),
                                            float(grad_scale), N.ptr(losses), N.ptr(self._scratch), N.stream_ptr(grads)), None,
                    "bf_op_adam_step")
        elif local > 0.0 or value > 0.0:      # keras precedence: clipnorm, else global_clipnorm, else clipvalue (in the library)
            N.check(N.lib().bf_adam_step_ex(model._h, N.ptr(model.params), N.ptr(grads), N.ptr(self.m), N.ptr(self.v),
                                            int(self.iterations), self.lr(), self.beta_1, self.beta_2, self.epsilon, clip, local,
                                            value, N.ptr(self._offsets), int(self._offsets.numel() - 1),
                                            N.ptr(self._tensor_scratch), float(grad_scale), N.ptr(losses), N.ptr(self._scratch),
                                            N.stream_ptr(grads)), model._h, "bf_adam_step_ex")
        else:
            N.check(N.lib().bf_adam_step(model._h, N.ptr(model.params), N.ptr(grads), N.ptr(self.m), N.ptr(self.v),
                                         int(self.iterations), self.lr(), self.beta_1, self.beta_2, self.epsilon, clip,
                                         float(grad_scale), N.ptr(losses), N.ptr(self._scratch), N.stream_ptr(grads)),
                    model._h, "bf_adam_step")
        self.iterations += 1
        model.mark_dirty()


def optimizer_builder(config: Dict) -> Tuple[Adam, LearningRateSchedule]:
    """bfcnn/optimizer.py:145-224."""
    if not isinstance(config, dict):
        raise ValueError("config must be a dictionary")
    lr_schedule = schedule_builder(config=config["schedule"])
    gradient_clipvalue = config.get("gradient_clipping_by_value", None)
    gradient_clipnorm = config.get("gradient_clipping_by_norm_local", None)
    gradient_global_clipnorm = config.get("gradient_clipping_by_norm", None)
    optimizer_type = config.get("type", "RMSprop").strip().upper()
    if optimizer_type == "ADAM":
        optimizer = Adam(learning_rate=lr_schedule, beta_1=config.get("beta_1", 0.9), beta_2=config.get("beta_2", 0.999),
                         epsilon=config.get("epsilon", 1e-07), amsgrad=config.get("amsgrad", False),
                         clipvalue=gradient_clipvalue, clipnorm=gradient_clipnorm,
                         global_clipnorm=gradient_global_clipnorm)
    elif optimizer_type in ("RMSPROP", "ADADELTA"):
        raise NotImplementedError(f"optimizer [{optimizer_type}] is outside the hot path (ADAM only)")
    else:
        raise ValueError(f"don't know how to handle optimizer_type: [{optimizer_type}]")
    return optimizer, lr_schedule
