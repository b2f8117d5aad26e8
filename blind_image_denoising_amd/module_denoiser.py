"""DenoiserModule: the inference drop-in boundary (bfcnn/module_denoiser.py:15-75)."""
import numpy as np
import torch

from .constants import DENOISER_STR
from .utilities import next_power_of_2


class DenoiserModule:
    """denoising inference module: uint8 [B,H,W,C] -> uint8 [B,H,W,C].

    Reference: cast -> pad_to_power_of_2 -> hydra -> [take output 0] -> remove_padding ->
    tf.round -> uint8 (module_denoiser.py:46-75).  Here the whole chain is ONE C-ABI call
    (bf_forward_u8): the cast/normalise is fused into the base-convolution kernel, the padding
    is virtual, and denormalise + crop + round-half-even + cast are fused into the head kernel."""

    def __init__(self, model_hydra, cast_to_uint8: bool = True):
        # --- argument checking (module_denoiser.py:31-33)
        if model_hydra is None:
            raise ValueError("model_denoise should not be None")
        self.name = DENOISER_STR
        self._cast_to_uint8 = cast_to_uint8
        self._model_hydra = model_hydra

    @property
    def model_hydra(self):
        return self._model_hydra

    def __call__(self, image):
        """image: uint8 tensor of rank 4 (the input_signature of module_denoiser.py:43-45)."""
        was_numpy = isinstance(image, np.ndarray)
        if was_numpy:
            image = torch.from_numpy(np.ascontiguousarray(image))
        if not isinstance(image, torch.Tensor):
            raise ValueError("image must be a torch.Tensor or numpy array")
        if image.dtype != torch.uint8 or image.dim() != 4:
            raise ValueError(f"input must be a uint8 tensor of shape [B,H,W,C], "
                             f"got {image.dtype} {tuple(image.shape)}")
        hydra = self._model_hydra
        if image.shape[-1] != hydra.desc.in_channels:
            raise ValueError(f"expected {hydra.desc.in_channels} channels, got {image.shape[-1]}")
        if image.shape[0] == 0:
            out = torch.empty((0,) + tuple(image.shape[1:3]) + (hydra.desc.out_channels,), dtype=torch.uint8)
            return out.numpy() if was_numpy else out
        hydra._require_gpu()
        image = image.to(hydra.device).contiguous()
        if getattr(hydra, "multi_output", False):
            # several outputs (one per scale): the module keeps the first, full-resolution one (module_denoiser.py:62-65)
            out = hydra.infer_u8(image, self._cast_to_uint8)
        elif self._cast_to_uint8:
            out = hydra.infer_u8(image)
        else:
            # float output: explicit pad -> hydra -> crop (no rounding)
            B, H, W, C = image.shape
            Hp, Wp = next_power_of_2(H), next_power_of_2(W)
            x = torch.zeros((B, Hp, Wp, C), dtype=torch.float32, device=hydra.device)
            x[:, :H, :W, :] = image.to(torch.float32)
            out = hydra(x, training=False)[:, :H, :W, :].contiguous()
        if was_numpy and not hydra.check_status(raise_on_overflow=not hydra.auto_exact_fallback):
            # host arrays are handed back (the stream is synchronised anyway) and an activation left the f16 range:
            # switch this model to the exact-fp32 kernels for good and repeat the call
            hydra.set_option("arith", 0)
            return self(image.cpu().numpy())
        return out.cpu().numpy() if was_numpy else out
