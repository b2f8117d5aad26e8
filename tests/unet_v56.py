"""Shared by the CPU and GPU tests of the trained unet_laplacian_v5.6 network: the committed fixture
(tests/golden/unet_v56.npz, written by tests/golden/make_unet_v56_fixture.py from the reference's DATA files) and the
measures of the reference's own tests/bfcnn/test_pretrained.py (PSNR, SSIM, MAE of noisy vs denoised against the clean
frame; truncated-normal noise of a given standard deviation, clipped and rounded to uint8)."""
import json
import os

import numpy as np
from scipy.ndimage import gaussian_filter

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "unet_v56.npz")


def load():
    z = np.load(FIXTURE)
    return z, json.loads(bytes(z["config"]).decode())


def corrupt(image_u8: np.ndarray, std: float, seed: int = 0) -> np.ndarray:
    """test_pretrained.py:41-56 with numpy's generator in place of tf.random.truncated_normal (resample beyond 2 sigma)."""
    rng = np.random.default_rng(seed)
    n = rng.normal(0.0, std, image_u8.shape)
    bad = np.abs(n) > 2 * std
    while bad.any():
        n[bad] = rng.normal(0.0, std, int(bad.sum()))
        bad = np.abs(n) > 2 * std
    return np.clip(np.round(image_u8.astype(np.float64) + n), 0, 255).astype(np.uint8)


def psnr(a, b) -> float:
    return float(10.0 * np.log10(255.0 ** 2 / np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def mae(a, b) -> float:
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).mean())


def ssim(a, b) -> float:
    """mean SSIM of one [H,W,C] pair, Gaussian window sigma 1.5 (what tf.image.ssim measures, up to the window's support)."""
    a, b = a.astype(np.float64), b.astype(np.float64)
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    f = lambda x: gaussian_filter(x, sigma=(1.5, 1.5, 0), truncate=3.5)
    ma, mb = f(a), f(b)
    va, vb, cab = f(a * a) - ma * ma, f(b * b) - mb * mb, f(a * b) - ma * mb
    return float(np.mean(((2 * ma * mb + c1) * (2 * cab + c2)) / ((ma * ma + mb * mb + c1) * (va + vb + c2))))


def assert_denoised(clean, noisy, denoised, what=""):
    """the three inequalities of test_pretrained.py:62-78."""
    assert denoised.shape == noisy.shape == clean.shape and denoised.dtype == np.uint8
    assert psnr(clean, noisy) < psnr(clean, denoised), f"{what}: PSNR {psnr(clean, noisy):.2f} -> {psnr(clean, denoised):.2f}"
    for i in range(clean.shape[0]):
        assert ssim(clean[i], noisy[i]) < ssim(clean[i], denoised[i]), f"{what}: SSIM"
    assert mae(clean, denoised) < mae(clean, noisy), f"{what}: MAE {mae(clean, noisy):.2f} -> {mae(clean, denoised):.2f}"
