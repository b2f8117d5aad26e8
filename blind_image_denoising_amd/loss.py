"""loss_function_builder (bfcnn/loss.py:152-253).

Inside a training step the loss is computed by the fused head kernel (bf_train_step); the
callables returned here carry the same configuration (`.desc` -> bf_loss_desc) and also evaluate
the same quantities on arbitrary tensors for monitoring / evaluation, using torch elementwise
ops on whatever device the tensors live on (not part of the timed hot path)."""
import ctypes as C
from typing import Callable, Dict

import torch

from . import _native as N
from .constants import *
from .custom_logger import logger


def mae_diff(error: torch.Tensor, hinge: float = 0.0, cutoff: float = 255.0) -> torch.Tensor:
    """bfcnn/loss.py:40-65: keras relu(|e|, threshold=hinge, max_value=cutoff), global mean."""
    a = error.abs()
    d = torch.where(a > hinge, a, torch.zeros_like(a)).clamp(max=cutoff)
    return d.mean(dim=(1, 2, 3)).mean()


def mae(original, prediction, **kwargs):
    """bfcnn/loss.py:71-86."""
    return mae_diff(error=(original - prediction), **kwargs)


def rmse_diff(error: torch.Tensor, hinge: float = 0.0, cutoff: float = 255.0 * 255.0) -> torch.Tensor:
    """bfcnn/loss.py:92-113 (relu on the signed error, sqrt(mean + DEFAULT_EPSILON))."""
    d = torch.where(error > hinge, error, torch.zeros_like(error)).clamp(max=cutoff) ** 2
    return torch.sqrt(d.mean(dim=(1, 2, 3)) + DEFAULT_EPSILON).mean()


def rmse(original, prediction, **kwargs):
    """bfcnn/loss.py:119-134."""
    return rmse_diff(error=(original - prediction), **kwargs)


def ssim_mean(original, prediction, max_val: float = 255.0, filter_size: int = 7, filter_sigma: float = 1.5):
    """tf.reduce_mean(tf.image.ssim(original, prediction, filter_size=7, max_val=255)) as bfcnn/loss.py:219-226 calls it
    (VALID Gaussian windows, k1 0.01, k2 0.03), for the host-side metric dict; the training step computes the same value
    and its gradient in csrc/loss_terms.hip."""
    x, y = original.to(torch.float64).permute(0, 3, 1, 2), prediction.to(torch.float64).permute(0, 3, 1, 2)
    c = torch.arange(filter_size, dtype=torch.float64, device=x.device) - (filter_size - 1) / 2.0
    g = -0.5 * c * c / (filter_sigma * filter_sigma)
    g = torch.softmax((g[None, :] + g[:, None]).reshape(-1), dim=0).reshape(1, 1, filter_size, filter_size)
    ch = x.shape[1]
    red = lambda t: torch.nn.functional.conv2d(t, g.repeat(ch, 1, 1, 1), groups=ch)
    c1, c2 = (0.01 * max_val) ** 2, (0.03 * max_val) ** 2
    a, b = red(x), red(y)
    lum = (2.0 * a * b + c1) / (a * a + b * b + c1)
    cs = (2.0 * red(x * y) - 2.0 * a * b + c2) / (red(x * x + y * y) - a * a - b * b + c2)
    return (lum * cs).mean().to(torch.float32)


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
        regularization_loss = torch.stack(list(model.losses)).sum() if model.losses else torch.zeros(())
        return {REGULARIZATION_LOSS_STR: regularization_loss,
                TOTAL_LOSS_STR: regularization_loss * regularization_multiplier}

    def denoiser_loss(gt_batch: torch.Tensor, predicted_batch: torch.Tensor) -> Dict[str, torch.Tensor]:
        mae_actual = mae(gt_batch, predicted_batch, hinge=0.0, cutoff=255.0)
        mse_actual = rmse(gt_batch, predicted_batch, hinge=0.0, cutoff=255.0)
        zero = torch.zeros((), dtype=torch.float32, device=gt_batch.device)
        mae_prediction_loss = mae(gt_batch, predicted_batch, hinge=hinge, cutoff=cutoff) if use_mae else zero
        mse_prediction_loss = rmse(gt_batch, predicted_batch, hinge=hinge, cutoff=cutoff * cutoff) if use_mse else zero
        ssim_loss = 1.0 - ssim_mean(gt_batch, predicted_batch) if use_ssim else zero          # :217-227
        return {TOTAL_LOSS_STR: mae_prediction_loss * mae_multiplier + mse_prediction_loss * mse_multiplier
                                + ssim_loss * ssim_multiplier,
                MSE_LOSS_STR: mse_actual, MAE_LOSS_STR: mae_actual, SSIM_LOSS_STR: ssim_loss}

    model_loss.desc = desc
    denoiser_loss.desc = desc
    return {MODEL_LOSS_FN_STR: model_loss, DENOISER_LOSS_FN_STR: denoiser_loss}
