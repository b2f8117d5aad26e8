"""GPU parity of the individual HIP kernels against the fp64 oracle (called through the C ABI's
diagnostic entry points).  Tolerance: fp32 accumulation, |err| <= 2e-5 * max(1, max|ref|)."""
import numpy as np
import pytest
import torch

from oracle import bfcnn_oracle as O
from blind_image_denoising_amd import _native as N
from helpers import (bwd3x3_h3_gpu, bwd_block_h3t_gpu, conv3x3_h3_pre_gpu, fwd_block_h3t_gpu, assert_close, conv3x3_gpu, conv3x3_h3_gpu, dev, fused_block_gpu, fused_block_h3_gpu,
                     fused_block2_h3_gpu, host, wgrad_gpu)

pytestmark = pytest.mark.gpu

SHAPES = [(2, 16, 32), (1, 19, 37), (3, 5, 3), (1, 64, 96), (2, 33, 65)]


def _rand(shape, seed):
    return np.random.default_rng(seed).standard_normal(shape).astype(np.float32)


def test_mfma_operand_layout():
    """D = A(16x4) B(4x16) with asymmetric integer data: pins the lane maps the conv kernels assume."""
    a = np.arange(64, dtype=np.float32).reshape(16, 4) - 20
    b = (np.arange(64, dtype=np.float32).reshape(4, 16) * 3 + 1) % 17
    d = torch.zeros(256, dtype=torch.float32, device="cuda")
    ad, bd = dev(a), dev(b)
    assert N.lib().bf_debug_mfma_probe(N.ptr(ad), N.ptr(bd), N.ptr(d), N.stream_ptr(d)) == 0
    assert np.array_equal(host(d).reshape(16, 16), a @ b)


@pytest.mark.parametrize("shape", SHAPES)
def test_conv3x3_plain(shape):
    B, H, W = shape
    x, w = _rand((B, H, W, 16), 1), _rand((3, 3, 16, 16), 2) * 0.1
    assert_close(conv3x3_gpu(x, w), O.conv2d_same(x.astype(np.float64), w.astype(np.float64)), what=f"conv {shape}")


def test_conv3x3_is_exact_on_integers():
    """small-integer data: fp32 MFMA accumulation is exact -> bit-equal to the oracle."""
    rng = np.random.default_rng(3)
    x = rng.integers(-4, 5, (2, 21, 40, 16)).astype(np.float32)
    w = rng.integers(-3, 4, (3, 3, 16, 16)).astype(np.float32)
    assert np.array_equal(conv3x3_gpu(x, w), O.conv2d_same(x.astype(np.float64), w.astype(np.float64)))


def test_conv3x3_asymmetric_kernel_orientation():
    """a one-hot kernel must shift the image the right way (no tap flip / cin-cout transpose)."""
    x = _rand((1, 12, 20, 16), 4)
    for (ky, kx, ci, co) in [(0, 2, 3, 7), (2, 1, 15, 0), (1, 0, 8, 8)]:
        w = np.zeros((3, 3, 16, 16), np.float32)
        w[ky, kx, ci, co] = 1.0
        assert np.array_equal(conv3x3_gpu(x, w), O.conv2d_same(x.astype(np.float64), w.astype(np.float64)).astype(np.float32))


@pytest.mark.parametrize("shape", SHAPES[:3])
def test_conv3x3_epilogues(shape):
    B, H, W = shape
    x, w = _rand((B, H, W, 16), 5), _rand((3, 3, 16, 16), 6) * 0.1
    sc, sh = _rand(16, 7), _rand(16, 8)
    res, mask = _rand((B, H, W, 16), 9), _rand((B, H, W, 16), 10)
    c = O.conv2d_same(x.astype(np.float64), w.astype(np.float64))
    assert_close(conv3x3_gpu(x, w, N.EPI_RELU), np.maximum(c, 0), what="relu")
    assert_close(conv3x3_gpu(x, w, N.EPI_AFFINE, sc, sh), c * sc + sh, what="affine")
    assert_close(conv3x3_gpu(x, w, N.EPI_AFFINE | N.EPI_RELU, sc, sh), np.maximum(c * sc + sh, 0), what="affine+relu")
    assert_close(conv3x3_gpu(x, w, N.EPI_AFFINE | N.EPI_RES, sc, sh, res), c * sc + sh + res, what="affine+res")
    assert_close(conv3x3_gpu(x, w, N.EPI_RES, res=res), c + res, what="res")
    assert_close(conv3x3_gpu(x, w, N.EPI_MASK, mask=mask), c * (mask > 0), what="mask")


@pytest.mark.parametrize("shape", SHAPES[:4])
def test_conv3x3_stats(shape):
    B, H, W = shape
    x, w = _rand((B, H, W, 16), 11), _rand((3, 3, 16, 16), 12) * 0.1
    out, stats = conv3x3_gpu(x, w, N.EPI_STATS, want_stats=True)
    c = O.conv2d_same(x.astype(np.float64), w.astype(np.float64))
    assert_close(out, c, what="raw output")
    tot = stats.astype(np.float64).sum(axis=0)
    assert_close(tot[:16], c.sum(axis=(0, 1, 2)), rel=1e-5 * np.sqrt(c.size), what="sum")
    assert_close(tot[16:], (c * c).sum(axis=(0, 1, 2)), rel=2e-5, what="sumsq")


@pytest.mark.parametrize("shape", SHAPES[:3])
def test_conv3x3_data_gradient_form(shape):
    B, H, W = shape
    dy, w = _rand((B, H, W, 16), 13), _rand((3, 3, 16, 16), 14) * 0.1
    assert_close(conv3x3_gpu(dy, w, transpose_flip=1),
                 O.conv2d_same_grad_input(dy.astype(np.float64), w.astype(np.float64)), what="dgrad")


# ---- the split-f16 single convolution of the training step: same oracle, same bars as conv3x3_c16_kernel -------
@pytest.mark.parametrize("shape", SHAPES + [(2, 40, 70)])
def test_conv3x3_h3_plain_and_dgrad(shape):
    B, H, W = shape
    x, w = _rand((B, H, W, 16), 1), _rand((3, 3, 16, 16), 2) * 0.1
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    assert_close(conv3x3_h3_gpu(x, w), O.conv2d_same(x64, w64), what=f"h3 conv {shape}")
    assert_close(conv3x3_h3_gpu(x, w, transpose_flip=1), O.conv2d_same_grad_input(x64, w64), what=f"h3 dgrad {shape}")


def test_conv3x3_h3_is_exact_on_integers():
    rng = np.random.default_rng(3)
    x = rng.integers(-4, 5, (2, 21, 40, 16)).astype(np.float32)
    w = rng.integers(-3, 4, (3, 3, 16, 16)).astype(np.float32)
    assert np.array_equal(conv3x3_h3_gpu(x, w), O.conv2d_same(x.astype(np.float64), w.astype(np.float64)))


@pytest.mark.parametrize("shape", SHAPES[:4])
def test_conv3x3_h3_epilogues_and_stats(shape):
    B, H, W = shape
    x, w = _rand((B, H, W, 16), 5), _rand((3, 3, 16, 16), 6) * 0.1
    res, mask = _rand((B, H, W, 16), 9), _rand((B, H, W, 16), 10)
    c = O.conv2d_same(x.astype(np.float64), w.astype(np.float64))
    assert_close(conv3x3_h3_gpu(x, w, N.EPI_RELU), np.maximum(c, 0), what="relu")
    assert_close(conv3x3_h3_gpu(x, w, N.EPI_RES, res=res), c + res, what="res")
    assert_close(conv3x3_h3_gpu(x, w, N.EPI_MASK, mask=mask), c * (mask > 0), what="mask")
    out, stats = conv3x3_h3_gpu(x, w, N.EPI_STATS, want_stats=True)
    assert_close(out, c, what="raw output")
    tot = stats.astype(np.float64).sum(axis=0)
    assert_close(tot[:16], c.sum(axis=(0, 1, 2)), rel=1e-5 * np.sqrt(c.size), what="sum")
    assert_close(tot[16:], (c * c).sum(axis=(0, 1, 2)), rel=2e-5, what="sumsq")


@pytest.fixture(params=[0, 1, 2, 3, 4], ids=["tile14x32x4", "tile32x32x8", "tile16x64x8", "dma14x32x4", "v4_14x32x4"])
def fused_tile(request):
    """every fused-block tile geometry compiled into the library must pass the same parity tests."""
    import blind_image_denoising_amd as bf
    m = bf.model_builder(O.canonical_config(no_layers=0)["model"], device="cuda").hydra
    m.set_option("fused_tile", request.param)
    yield request.param
    m.set_option("fused_tile", -1)        # back to the default variant


@pytest.mark.parametrize("shape", SHAPES + [(1, 14, 32), (2, 28, 64), (1, 15, 33), (4, 70, 40)])
@pytest.mark.parametrize("relu", [1, 0])
def test_fused_block(shape, relu, fused_tile):
    B, H, W = shape
    x = _rand((B, H, W, 16), 15)
    w1, w2 = _rand((3, 3, 16, 16), 16) * 0.1, _rand((3, 3, 16, 16), 17) * 0.1
    sc, sh = _rand(16, 18), _rand(16, 19)
    x64 = x.astype(np.float64)
    t = O.conv2d_same(x64, w1.astype(np.float64))
    if relu:
        t = np.maximum(t, 0)
    ref = x64 + O.conv2d_same(t, w2.astype(np.float64)) * sc + sh
    assert_close(fused_block_gpu(x, w1, w2, sc, sh, relu), ref, what=f"fused {shape}")


def test_fused_block_many_tiles_persistent_schedule(fused_tile):
    """more tiles than resident workgroups (grid-stride + XCD chunking must cover every tile once)."""
    B, H, W = 24, 128, 160          # 24 * 10 * 5 = 1200 tiles > 512 workgroups
    x = _rand((B, H, W, 16), 20)
    w1, w2 = _rand((3, 3, 16, 16), 21) * 0.1, _rand((3, 3, 16, 16), 22) * 0.1
    sc, sh = np.ones(16, np.float32), np.zeros(16, np.float32)
    got = fused_block_gpu(x, w1, w2, sc, sh, 1)
    # unfused GPU path as the comparator at this size (itself pinned against the oracle above)
    t = conv3x3_gpu(x, w1, N.EPI_RELU)
    ref = conv3x3_gpu(t, w2, N.EPI_RES, res=x)
    assert_close(got, ref, rel=1e-5, what="fused vs unfused")
    sub = slice(5, 7)
    t64 = np.maximum(O.conv2d_same(x[sub].astype(np.float64), w1.astype(np.float64)), 0)
    assert_close(got[sub], x[sub] + O.conv2d_same(t64, w2.astype(np.float64)), what="fused vs oracle (2 images)")


# ---- split-f16 ("f16x3") fused block: same oracle, same bar as the exact-fp32 kernels ---------------------
@pytest.fixture(params=[4, 260, 1, 0, 2, 3], ids=["fullrow", "fullrow_up", "rows", "groups", "rows16x16", "specialised"], autouse=False)
def h3_variant(request):
    """every split-f16 kernel (full-row streaming, row-streaming tiles, group-per-pass, ...) must pass the same parity tests."""
    N.lib().bf_debug_set_h3_variant(request.param)
    yield request.param
    N.lib().bf_debug_set_h3_variant(-1)


@pytest.mark.parametrize("shape", SHAPES + [(1, 16, 32), (2, 32, 64), (1, 17, 33), (4, 70, 40), (1, 2, 2), (3, 48, 100)])
@pytest.mark.parametrize("relu", [1, 0])
def test_fused_block_h3(shape, relu, h3_variant):
    B, H, W = shape
    x = _rand((B, H, W, 16), 15)
    w1, w2 = _rand((3, 3, 16, 16), 16) * 0.1, _rand((3, 3, 16, 16), 17) * 0.1
    sc, sh = _rand(16, 18), _rand(16, 19)
    x64 = x.astype(np.float64)
    t = O.conv2d_same(x64, w1.astype(np.float64))
    if relu:
        t = np.maximum(t, 0)
    ref = x64 + O.conv2d_same(t, w2.astype(np.float64)) * sc + sh
    assert_close(fused_block_h3_gpu(x, w1, w2, sc, sh, relu), ref, what=f"fused h3 {shape}")


def test_fused_block_h3_is_exact_on_small_integers(h3_variant):
    """integers that f16 holds exactly: every product and sum is exact, so the result must equal the oracle
    bit for bit (pins the K packing / tap pairing / lane maps of the f16 MFMA path)."""
    rng = np.random.default_rng(5)
    x = rng.integers(-4, 5, (2, 24, 40, 16)).astype(np.float32)
    w1 = rng.integers(-2, 3, (3, 3, 16, 16)).astype(np.float32)
    w2 = rng.integers(-2, 3, (3, 3, 16, 16)).astype(np.float32)
    sc, sh = np.ones(16, np.float32), rng.integers(-3, 4, 16).astype(np.float32)
    t = np.maximum(O.conv2d_same(x.astype(np.float64), w1.astype(np.float64)), 0)
    ref = x + O.conv2d_same(t, w2.astype(np.float64)) + sh
    assert np.abs(t).max() < 2048 and np.abs(ref).max() < 2 ** 22        # exactly representable as hi + lo
    assert np.array_equal(fused_block_h3_gpu(x, w1, w2, sc, sh, 1).astype(np.float64), ref)


def test_fused_block_h3_wide_dynamic_range(h3_variant):
    """weights of very different magnitudes (power-of-two pre-scale + lo part) and activations spanning
    1e-3 .. 1e2: the split must hold ~22 bits relative to the largest term."""
    rng = np.random.default_rng(6)
    x = (rng.standard_normal((1, 40, 72, 16)) * np.exp(rng.uniform(-7, 4.5, (1, 40, 72, 16)))).astype(np.float32)
    w1 = (rng.standard_normal((3, 3, 16, 16)) * np.exp(rng.uniform(-6, 0, (3, 3, 16, 16)))).astype(np.float32)
    w2 = (rng.standard_normal((3, 3, 16, 16)) * 3e-3).astype(np.float32)
    sc, sh = _rand(16, 18), _rand(16, 19)
    t = np.maximum(O.conv2d_same(x.astype(np.float64), w1.astype(np.float64)), 0)
    ref = x + O.conv2d_same(t, w2.astype(np.float64)) * sc + sh
    assert_close(fused_block_h3_gpu(x, w1, w2, sc, sh, 1), ref, what="fused h3 wide range")


def test_fused_block_h3_many_tiles_persistent_schedule(h3_variant):
    """more tiles than persistent workgroups: the double-buffered DMA pipeline, the XCD chunking and every
    chunk length (1, 2, 3+ tiles per workgroup) must cover each tile exactly once."""
    for B, H, W in [(24, 128, 160), (3, 100, 70), (9, 64, 96)]:
        x = _rand((B, H, W, 16), 20)
        w1, w2 = _rand((3, 3, 16, 16), 21) * 0.1, _rand((3, 3, 16, 16), 22) * 0.1
        sc, sh = np.ones(16, np.float32), np.zeros(16, np.float32)
        got = fused_block_h3_gpu(x, w1, w2, sc, sh, 1)
        t = conv3x3_gpu(x, w1, N.EPI_RELU)                  # exact-fp32 GPU path as the comparator at this size
        ref = conv3x3_gpu(t, w2, N.EPI_RES, res=x)
        assert_close(got, ref, rel=1e-5, what=f"h3 vs unfused {(B, H, W)}")
    sub = slice(1, 2)
    t64 = np.maximum(O.conv2d_same(x[sub].astype(np.float64), w1.astype(np.float64)), 0)
    assert_close(got[sub], x[sub] + O.conv2d_same(t64, w2.astype(np.float64)), what="h3 vs oracle")


# ---- two blocks per launch on 128-column strips (fused_h3w.hip): same oracle, same bar -------------------------------------
def _two_blocks_oracle(x, w4, sc2, sh2, relu):
    y = x.astype(np.float64)
    for b in range(2):
        t = O.conv2d_same(y, w4[2 * b].astype(np.float64))
        if relu:
            t = np.maximum(t, 0)
        y = y + O.conv2d_same(t, w4[2 * b + 1].astype(np.float64)) * sc2[b] + sh2[b]
    return y


# one strip (W <= 144), two strips with the second grid clamped to the right edge (145 .. 256), interior strips (> 256), ragged
# heights, bands shorter than the 12-step pipeline
PAIR_SHAPES = SHAPES + [(1, 16, 256), (2, 40, 256), (1, 33, 144), (1, 20, 145), (1, 24, 150), (2, 17, 200), (1, 9, 129), (1, 2, 2),
                        (1, 1, 1), (1, 12, 300), (1, 10, 512), (1, 14, 400), (3, 48, 100)]


@pytest.mark.parametrize("shape", PAIR_SHAPES)
@pytest.mark.parametrize("relu", [1, 0])
@pytest.mark.parametrize("reverse", [0, 1], ids=["down", "up"])
def test_fused_block2_h3(shape, relu, reverse):
    B, H, W = shape
    x = _rand((B, H, W, 16), 15)
    w4 = _rand((4, 3, 3, 16, 16), 16) * 0.1
    sc2, sh2 = _rand((2, 16), 18), _rand((2, 16), 19)
    ref = _two_blocks_oracle(x, w4, sc2, sh2, relu)
    assert_close(fused_block2_h3_gpu(x, w4, sc2, sh2, relu, reverse), ref, what=f"two fused h3 blocks {shape}")


@pytest.mark.parametrize("reverse", [0, 1], ids=["down", "up"])
def test_fused_block2_h3_is_exact_on_small_integers(reverse):
    """integers that hi + lo hold exactly through BOTH blocks: the result must equal the oracle bit for bit (pins the ring
    bookkeeping of the four chained convolutions: slots, halo columns, which row each accumulator belongs to)."""
    rng = np.random.default_rng(7)
    x = rng.integers(-3, 4, (2, 20, 150, 16)).astype(np.float32)

    def sparse(p, lo, hi):
        w = rng.integers(lo, hi + 1, (3, 3, 16, 16)).astype(np.float32)
        return w * (rng.random((3, 3, 16, 16)) < p)

    w4 = np.stack([sparse(0.15, -2, 2), sparse(0.15, -2, 2), sparse(0.06, -1, 1), sparse(0.06, -1, 1)])
    sc2, sh2 = np.ones((2, 16), np.float32), rng.integers(-3, 4, (2, 16)).astype(np.float32)
    ref = _two_blocks_oracle(x, w4, sc2, sh2, 1)
    assert np.abs(ref).max() < 2 ** 20
    assert np.array_equal(fused_block2_h3_gpu(x, w4, sc2, sh2, 1, reverse).astype(np.float64), ref)


def test_fused_block2_h3_many_units_persistent_schedule():
    """more (image, band, strip) units than workgroups and several bands per image: every unit exactly once, ring state of one
    unit must not leak into the next (left strip after right strip and the other way round)."""
    for B, H, W in [(70, 40, 256), (9, 300, 200), (5, 64, 520)]:
        x = _rand((B, H, W, 16), 20)
        w4 = _rand((4, 3, 3, 16, 16), 21) * 0.1
        sc2, sh2 = np.ones((2, 16), np.float32), np.zeros((2, 16), np.float32)
        got = fused_block2_h3_gpu(x, w4, sc2, sh2, 1, 0)
        y = x                                           # exact-fp32 GPU path as the comparator at this size
        for b in range(2):
            t = conv3x3_gpu(y, w4[2 * b], N.EPI_RELU)
            y = conv3x3_gpu(t, w4[2 * b + 1], N.EPI_RES, res=y)
        assert_close(got, y, rel=1e-5, what=f"two blocks vs unfused {(B, H, W)}")
        sub = slice(B - 1, B)
        assert_close(got[sub], _two_blocks_oracle(x[sub], w4, sc2, sh2, 1), what="two blocks vs oracle")


@pytest.mark.parametrize("shape", SHAPES + [(8, 64, 64)])
@pytest.mark.parametrize("h3", [False, True], ids=["f32", "f16x3"])
def test_wgrad3x3(shape, h3):
    B, H, W = shape
    x, dy = _rand((B, H, W, 16), 23), _rand((B, H, W, 16), 24)
    ref = O.conv2d_same_grad_kernel(x.astype(np.float64), dy.astype(np.float64), 3, 3)
    assert_close(wgrad_gpu(x, dy, h3), ref, rel=3e-6 * np.sqrt(B * H * W), what=f"wgrad {shape}")


# ---- the fused training kernels of the split-f16 step (train_bwd_h3.hip, conv3x3_h3_kernel<.., PRE>) ---------------------
BWD_SHAPES = SHAPES + [(24, 128, 160)]          # the last one: more tiles (1200) than persistent workgroups (512)


@pytest.mark.parametrize("dbuf", [0, 1], ids=["2wg", "dbuf"])
@pytest.mark.parametrize("shape", BWD_SHAPES)
@pytest.mark.parametrize("reverse", [0, 1])
def test_bwd3x3_h3_second_convolution(shape, reverse, dbuf):
    """BatchNorm-backward apply on load + weight gradient + masked data gradient (a block's convolution j >= 1)."""
    B, H, W = shape
    x = np.maximum(_rand((B, H, W, 16), 40), 0)                       # activated input: also the mask
    g, c = _rand((B, H, W, 16), 41), _rand((B, H, W, 16), 42)
    w = _rand((3, 3, 16, 16), 43) * 0.1
    coef = np.concatenate([1 + 0.3 * _rand(16, 44), 0.2 * _rand(16, 45), 0.1 * _rand(16, 46)]).astype(np.float32)
    gp = coef[:16].astype(np.float64) * g + coef[16:32].astype(np.float64) * c + coef[32:].astype(np.float64)
    ref_dw = O.conv2d_same_grad_kernel(x.astype(np.float64), gp, 3, 3)
    ref_dx = O.conv2d_same_grad_input(gp, w.astype(np.float64)) * (x > 0)
    dx, dw = bwd3x3_h3_gpu(x, g, w, N.EPI_MASK, c=c, coef=coef, reverse=reverse, dbuf=dbuf)
    assert_close(dx, ref_dx, what=f"dx {shape}")
    assert_close(dw, ref_dw, rel=3e-6 * np.sqrt(B * H * W), what=f"dw {shape}")
    dx0, dw0 = bwd3x3_h3_gpu(x, gp.astype(np.float32), w, 0, reverse=reverse, dbuf=dbuf)           # no BatchNorm, linear activation
    assert_close(dx0, O.conv2d_same_grad_input(gp, w.astype(np.float64)), what=f"dx plain {shape}")
    assert_close(dw0, ref_dw, rel=3e-6 * np.sqrt(B * H * W), what=f"dw plain {shape}")


@pytest.mark.parametrize("dbuf", [0, 1], ids=["2wg", "dbuf"])
@pytest.mark.parametrize("shape", BWD_SHAPES)
def test_bwd3x3_h3_first_convolution(shape, dbuf):
    """weight gradient + data gradient + skip gradient, and the sums the BatchNorm backward of the block in front needs."""
    B, H, W = shape
    x, g = _rand((B, H, W, 16), 50), _rand((B, H, W, 16), 51)
    res, bnc = _rand((B, H, W, 16), 52), _rand((B, H, W, 16), 53)
    w = _rand((3, 3, 16, 16), 54) * 0.1
    g64 = g.astype(np.float64)
    ref_dx = O.conv2d_same_grad_input(g64, w.astype(np.float64)) + res
    ref_dw = O.conv2d_same_grad_kernel(x.astype(np.float64), g64, 3, 3)
    dx, dw, stats = bwd3x3_h3_gpu(x, g, w, N.EPI_RES | N.EPI_BNBWD, res=res, bnc=bnc, dbuf=dbuf)
    assert_close(dx, ref_dx, what=f"dx {shape}")
    assert_close(dw, ref_dw, rel=3e-6 * np.sqrt(B * H * W), what=f"dw {shape}")
    tot = stats.astype(np.float64).sum(axis=0)
    assert_close(tot[:16], ref_dx.sum(axis=(0, 1, 2)), rel=1e-5 * np.sqrt(ref_dx.size), what="sum dx")
    assert_close(tot[16:], (ref_dx * bnc).sum(axis=(0, 1, 2)), rel=1e-5 * np.sqrt(ref_dx.size), what="sum dx * c")
    dx1, dw1 = bwd3x3_h3_gpu(x, g, w, N.EPI_RES, res=res, dbuf=dbuf)
    assert np.array_equal(dx1, dx) and np.array_equal(dw1, dw)


def test_bwd3x3_h3_exact_on_integers_and_deterministic():
    rng = np.random.default_rng(55)
    x = rng.integers(0, 4, (3, 40, 70, 16)).astype(np.float32)
    g = rng.integers(-3, 4, (3, 40, 70, 16)).astype(np.float32)
    w = rng.integers(-3, 4, (3, 3, 16, 16)).astype(np.float32)
    dx, dw = bwd3x3_h3_gpu(x, g, w, N.EPI_MASK)
    assert np.array_equal(dw, O.conv2d_same_grad_kernel(x.astype(np.float64), g.astype(np.float64), 3, 3))
    assert np.array_equal(dx, O.conv2d_same_grad_input(g.astype(np.float64), w.astype(np.float64)) * (x > 0))
    xr, gr, wr = np.maximum(_rand((3, 40, 70, 16), 56), 0), _rand((3, 40, 70, 16), 57), _rand((3, 3, 16, 16), 58)
    a, b = bwd3x3_h3_gpu(xr, gr, wr, N.EPI_MASK), bwd3x3_h3_gpu(xr, gr, wr, N.EPI_MASK)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])          # fixed-order reductions


@pytest.mark.parametrize("shape", SHAPES + [(4, 70, 40)])
@pytest.mark.parametrize("relu", [1, 0])
def test_conv3x3_h3_affine_add_on_load(shape, relu):
    B, H, W = shape
    x, c = _rand((B, H, W, 16), 60), _rand((B, H, W, 16), 61)
    sc, sh = (1 + 0.3 * _rand(16, 62)).astype(np.float32), _rand(16, 63)
    w = _rand((3, 3, 16, 16), 64) * 0.1
    y, out = conv3x3_h3_pre_gpu(x, c, sc, sh, w, relu=relu, reverse=shape[0] & 1)
    y_ref = x + (sc * c + sh)                                     # fp32, the rounding of affine_add_kernel up to the fma
    assert np.abs(y - y_ref).max() <= 4e-7 * np.abs(y_ref).max()
    ref = O.conv2d_same(y.astype(np.float64), w.astype(np.float64))
    assert_close(out, np.maximum(ref, 0) if relu else ref, what=f"conv of the formed input {shape}")


# the whole training forward of a block in one row-streaming kernel (train_fwd_h3t.hip): full 256-column rows and narrower ones,
# bands shorter than the pipeline (1 and 2 rows), more bands than workgroups (300 x 8 rows), both walking directions
FWD_BLOCK_SHAPES = [(1, 1, 1), (2, 2, 5), (1, 5, 256), (3, 33, 47), (2, 64, 256), (1, 130, 255), (2, 40, 70), (300, 8, 16)]


@pytest.mark.parametrize("shape", FWD_BLOCK_SHAPES)
@pytest.mark.parametrize("pre", [1, 0])
def test_fwd_block_h3t_matches_oracle(shape, pre):
    """A_i = x + scale * c + shift on load, T_i = relu(conv_0 A_i), C_i = conv_1 T_i and the batch statistics of C_i, one kernel,
    against the fp64 oracle (bfcnn/backbone_blocks.py:174-246 under training=True)."""
    B, H, W = shape
    x, c = _rand((B, H, W, 16), 70), _rand((B, H, W, 16), 71)
    sc, sh = (1 + 0.3 * _rand(16, 72)).astype(np.float32), _rand(16, 73)
    w0, w1 = _rand((3, 3, 16, 16), 74) * 0.1, _rand((3, 3, 16, 16), 75) * 0.1
    relu = 1 if (B + H) % 3 else 0
    a, t, cout, stats = fwd_block_h3t_gpu(x, w0, w1, c if pre else None, sc, sh, relu=relu, reverse=(H + W) & 1)
    if pre:
        a_ref = x + (sc * c + sh)                                 # fp32, the rounding of affine_add_kernel up to the fma
        assert np.abs(a - a_ref).max() <= 6e-7 * max(np.abs(a_ref).max(), 1.0)     # stored as the hi + lo pair the convolution saw: 22 bits
    else:
        assert a is None
        a = x
    t_ref = O.conv2d_same(a.astype(np.float64), w0.astype(np.float64))
    t_ref = np.maximum(t_ref, 0) if relu else t_ref
    assert_close(t, t_ref, what=f"T {shape}")
    c_ref = O.conv2d_same(t_ref, w1.astype(np.float64))
    assert_close(cout, c_ref, what=f"C {shape}")
    n = np.sqrt(B * H * W)
    assert_close(stats[:16], cout.astype(np.float64).sum(axis=(0, 1, 2)), rel=1e-5 * n, what="sum C")
    assert_close(stats[16:], (cout.astype(np.float64) ** 2).sum(axis=(0, 1, 2)), rel=1e-5 * n, what="sum C^2")
    # without the T output the same C and sums, bit for bit; twice the same bits
    _, t2, cout2, stats2 = fwd_block_h3t_gpu(x, w0, w1, c if pre else None, sc, sh, relu=relu, reverse=(H + W) & 1, want_t=False)
    assert t2 is None and np.array_equal(cout2, cout) and np.array_equal(stats2, stats)


def test_fwd_block_h3t_exact_on_integers():
    """small integers through both convolutions: bit-exact (pins the K packing, the tap pairing, the ring bookkeeping and the
    rows / columns outside the image, which must be conv_1's ZERO padding and not values computed from padded input)"""
    rng = np.random.default_rng(76)
    for shape in [(2, 37, 256), (3, 19, 100)]:
        x = rng.integers(-2, 3, shape + (16,)).astype(np.float32)
        w0 = rng.integers(-2, 3, (3, 3, 16, 16)).astype(np.float32)
        w1 = rng.integers(-2, 3, (3, 3, 16, 16)).astype(np.float32)
        for reverse in (0, 1):
            _, t, cout, stats = fwd_block_h3t_gpu(x, w0, w1, relu=1, reverse=reverse)
            t_ref = np.maximum(O.conv2d_same(x.astype(np.float64), w0.astype(np.float64)), 0)
            c_ref = O.conv2d_same(t_ref, w1.astype(np.float64))
            assert np.array_equal(t, t_ref) and np.array_equal(cout, c_ref), (shape, reverse)
            assert_close(stats[:16], c_ref.sum(axis=(0, 1, 2)), rel=1e-6 * np.sqrt(c_ref.size), what="sum C")


def _block_backward_oracle(a, g, c, coef, w0, w1, relu):
    """fp64 restatement of what bwd_block_h3t_kernel computes, from the oracle's convolution primitives"""
    a64, g64, c64, w0_64, w1_64 = (v.astype(np.float64) for v in (a, g, c, w0, w1))
    k1, k2, k3 = coef[:16].astype(np.float64), coef[16:32].astype(np.float64), coef[32:].astype(np.float64)
    dc = k1 * g64 + k2 * c64 + k3
    pre = O.conv2d_same(a64, w0_64)
    t = np.maximum(pre, 0) if relu else pre
    dw1 = O.conv2d_same_grad_kernel(t, dc, 3, 3)
    dt = O.conv2d_same_grad_input(dc, w1_64) * ((t > 0) if relu else 1.0)
    dw0 = O.conv2d_same_grad_kernel(a64, dt, 3, 3)
    out = O.conv2d_same_grad_input(dt, w0_64) + g64
    return out, dw1, dw0, pre


# one to three strips of 128 columns, grids clamped to the left edge, partial last strips, bands shorter than the pipeline,
# more units than workgroups (70 x 40 x 256 = 280 strips x bands), both walking directions
BWD_BLOCK_SHAPES = [(1, 1, 1), (2, 3, 7), (1, 12, 128), (3, 33, 47), (2, 40, 200), (1, 70, 256), (2, 24, 300), (70, 40, 256)]


@pytest.mark.parametrize("shape", BWD_BLOCK_SHAPES)
@pytest.mark.parametrize("bn_in_front", [1, 0])
def test_bwd_block_h3t_matches_oracle(shape, bn_in_front):
    """BatchNorm backward on load, T recomputed, both weight gradients, the masked data gradient of conv_1 kept in LDS, the data
    gradient of conv_0 + the skip's gradient and the sums of the next BatchNorm backward: one kernel against the fp64 oracle
    (tape.gradient of bfcnn/train_loop.py:273-294 through one block of bfcnn/backbone_blocks.py:174-246)."""
    B, H, W = shape
    a, g, c = _rand((B, H, W, 16), 80), _rand((B, H, W, 16), 81) * 0.1, _rand((B, H, W, 16), 82)
    bnc = _rand((B, H, W, 16), 83) if bn_in_front else None
    coef = np.concatenate([1 + 0.2 * _rand(16, 84), 0.1 * _rand(16, 85), 0.01 * _rand(16, 86)]).astype(np.float32)
    w0, w1 = _rand((3, 3, 16, 16), 87) * 0.1, _rand((3, 3, 16, 16), 88) * 0.1
    relu = 1 if (B + W) % 4 else 0
    got = bwd_block_h3t_gpu(a, g, c, coef, w0, w1, bnc=bnc, relu=relu, reverse=(H + W) & 1)
    out, dw1, dw0, pre = _block_backward_oracle(a, g, c, coef, w0, w1, relu)
    # an element of conv_0's output within fp32 rounding of the ReLU's kink may take either side (its whole gradient flips): the
    # comparison of the data gradient is skipped when the oracle itself names such an element (rare: |pre| < 1e-6)
    near = np.abs(pre) < 1e-6 if relu else np.zeros_like(pre, bool)
    assert near.mean() < 1e-4
    n = np.sqrt(B * H * W)
    if not near.any():
        assert_close(got[0], out, what=f"dA' {shape}")
    assert_close(got[1], dw1, rel=3e-6 * n, what=f"dW1 {shape}")
    assert_close(got[2], dw0, rel=3e-6 * n, what=f"dW0 {shape}")
    if bn_in_front:
        o64 = got[0].astype(np.float64)
        assert_close(got[3][:16], o64.sum(axis=(0, 1, 2)), rel=1e-5 * np.sqrt(o64.size), what="sum dA'")
        assert_close(got[3][16:], (o64 * bnc).sum(axis=(0, 1, 2)), rel=1e-5 * np.sqrt(o64.size), what="sum dA' * c")
    again = bwd_block_h3t_gpu(a, g, c, coef, w0, w1, bnc=bnc, relu=relu, reverse=(H + W) & 1)
    assert all(np.array_equal(x, y) for x, y in zip(got, again))               # fixed-order reductions: bitwise reproducible


def test_bwd_block_h3t_exact_on_integers():
    """small integers through all five operators: bit-exact (pins the K packing of both kinds of operand, the ring bookkeeping, the
    own-rows / own-columns rule of the weight gradients and the zero padding of every intermediate)"""
    rng = np.random.default_rng(89)
    coef = np.concatenate([np.ones(16), np.zeros(16), np.zeros(16)]).astype(np.float32)
    for shape in [(2, 37, 256), (3, 19, 100), (1, 9, 300)]:
        a = rng.integers(-2, 3, shape + (16,)).astype(np.float32)
        g = rng.integers(-2, 3, shape + (16,)).astype(np.float32)
        c = rng.integers(-2, 3, shape + (16,)).astype(np.float32)
        w0 = rng.integers(-1, 2, (3, 3, 16, 16)).astype(np.float32)
        w1 = rng.integers(-1, 2, (3, 3, 16, 16)).astype(np.float32)
        for reverse in (0, 1):
            got = bwd_block_h3t_gpu(a, g, c, coef, w0, w1, relu=1, reverse=reverse)
            out, dw1, dw0, _ = _block_backward_oracle(a, g, c, coef, w0, w1, 1)
            assert np.array_equal(got[0], out), (shape, reverse)
            assert_close(got[1], dw1, rel=1e-6, what="dW1")
            assert_close(got[2], dw0, rel=1e-6, what="dW0")


@pytest.mark.parametrize("h3", [False, True], ids=["f32", "f16x3"])
def test_wgrad3x3_exact_on_integers_and_deterministic(h3):
    rng = np.random.default_rng(25)
    x = rng.integers(-3, 4, (3, 40, 70, 16)).astype(np.float32)
    dy = rng.integers(-3, 4, (3, 40, 70, 16)).astype(np.float32)
    a = wgrad_gpu(x, dy, h3)
    assert np.array_equal(a, O.conv2d_same_grad_kernel(x.astype(np.float64), dy.astype(np.float64), 3, 3))
    xr, dr = _rand((3, 40, 70, 16), 26), _rand((3, 40, 70, 16), 27)
    assert np.array_equal(wgrad_gpu(xr, dr, h3), wgrad_gpu(xr, dr, h3))      # fixed-order reduction
