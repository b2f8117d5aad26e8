"""loss_function_builder (bfcnn/loss.py:152-253).

Inside a training step the loss is computed by the fused head kernel (bf_train_step); the
callables returned here carry the same configuration (`.desc` -> bf_loss_desc) and also evaluate
the same quantities on arbitrary GPU batches for monitoring / evaluation through the C ABI
(bf_op_denoiser_loss: the same kernels, csrc/loss_terms.hip)."""
import ctypes as C
from typing import Callable, Dict

import torch

from . import _native as N
from .constants import *
from .custom_logger import logger


def _denoiser_loss_slots(gt_batch: torch.Tensor, predicted_batch: torch.Tensor, desc: "N.LossDesc") -> torch.Tensor:
    """the loss slots of one output scale through the C ABI (bf_op_denoiser_loss: the kernels of csrc/loss_terms.hip that the
    training step itself uses); both tensors on the GPU.  There is no CPU execution path."""
    if gt_batch.device.type != "cuda" or predicted_batch.device.type != "cuda":
        raise RuntimeError("denoiser_loss needs its tensors on the GPU: there is no CPU execution path")
    if gt_batch.shape != predicted_batch.shape or gt_batch.dim() != 4:
        raise ValueError(f"expected two [B,H,W,C] batches of one shape, got {tuple(gt_batch.shape)} and {tuple(predicted_batch.shape)}")
    gt = gt_batch.to(torch.float32).contiguous()
    pr = predicted_batch.to(device=gt.device, dtype=torch.float32).contiguous()
    B, H, W, Cc = gt.shape
    lib = N.lib()
    losses = torch.zeros(N.BF_LOSS_COUNT, dtype=torch.float32, device=gt.device)
    dpred = torch.empty_like(pr)
    scratch = torch.empty(int(lib.bf_op_denoiser_loss_scratch_floats(B, H, W, Cc)) + 64, dtype=torch.float32, device=gt.device)
    N.check(lib.bf_op_denoiser_loss(N.ptr(pr), N.ptr(gt), B, H, W, Cc, C.byref(desc), N.ptr(dpred), N.ptr(losses), N.ptr(scratch),
                                    scratch.numel(), N.stream_ptr(gt)), None, "bf_op_denoiser_loss")
    return losses


def loss_function_builder(config: Dict) -> Dict[str, Callable]:
    """bfcnn/loss.py:152-253.  Returns {"model": model_loss, "denoiser": denoiser_loss}; both
    callables expose `.desc(depth_weight)` = the bf_loss_desc for bf_train_step."""
    logger.info("building loss_function with config [{0}]".format(config))
    hinge = config.get("hinge", 0.0)
    cutoff = config.get("cutoff", 255.0)
    mae_multiplier = config.get("mae_multiplier", 1.0)
    use_mae = mae_multiplier > 0.0
    ssim_multiplier = config.get("ssim_multiplier", 1.0)     # default 1.0 as the reference (:171)
    use_ssim = ssim_multiplier > 0.0
    mse_multiplier = config.get("mse_multiplier", 0.0)
    use_mse = mse_multiplier > 0.0
    regularization_multiplier = config.get("regularization", 1.0)

    def desc(depth_weight: float = 1.0) -> N.LossDesc:
        d = N.LossDesc()
        d.struct_size = C.sizeof(N.LossDesc)
        d.hinge, d.cutoff = float(hinge), float(cutoff)
        d.mae_multiplier, d.mse_multiplier = float(mae_multiplier), float(mse_multiplier)
        d.ssim_multiplier = float(ssim_multiplier)
        d.regularization, d.depth_weight = float(regularization_multiplier), float(depth_weight)
        return d

    def model_loss(model):
        # tf.add_n(model.losses): the regulariser kernel accumulates every tensor's term into ONE device scalar
        terms = list(model.losses)
        if not terms:
            z = torch.zeros(())
            return {REGULARIZATION_LOSS_STR: z, TOTAL_LOSS_STR: z}
        out = terms[0].new_zeros(2)                      # [0] sum of the terms, [1] the sum times `regularization`
        for t in terms:
            N.check(N.lib().bf_op_axpy(N.ptr(out[0:1]), N.ptr(t), 1.0, 0, 1, N.stream_ptr(t)), None, "bf_op_axpy")
        N.check(N.lib().bf_op_axpy(N.ptr(out[1:2]), N.ptr(out[0:1]), float(regularization_multiplier), 0, 1, N.stream_ptr(out)), None,
                "bf_op_axpy")
        return {REGULARIZATION_LOSS_STR: out[0], TOTAL_LOSS_STR: out[1]}

    def denoiser_loss(gt_batch: torch.Tensor, predicted_batch: torch.Tensor) -> Dict[str, torch.Tensor]:
        """bfcnn/loss.py:190-247 on arbitrary batches (monitoring / evaluation): 0-d views of one device buffer"""
        sl = _denoiser_loss_slots(gt_batch, predicted_batch, desc(1.0))
        return {TOTAL_LOSS_STR: sl[N.BF_LOSS_DENOISER_TOTAL], MSE_LOSS_STR: sl[N.BF_LOSS_MSE], MAE_LOSS_STR: sl[N.BF_LOSS_MAE],
                SSIM_LOSS_STR: sl[N.BF_LOSS_SSIM]}

    model_loss.desc = desc
    denoiser_loss.desc = desc
    return {MODEL_LOSS_FN_STR: model_loss, DENOISER_LOSS_FN_STR: denoiser_loss}
