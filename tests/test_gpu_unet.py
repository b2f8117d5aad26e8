"""GPU parity of the unet_laplacian operators (csrc/unet_ops.hip, through the C ABI) and of the whole
hydra / DenoiserModule path against the fp64 oracle (oracle/unet_oracle.py).
Bars as for the resnet path: f32 outputs MAE <= 1e-4 on the normalised scale, uint8 within +-1 LSB."""
import os

import numpy as np
import pytest
import torch

import blind_image_denoising_amd as bf
from blind_image_denoising_amd import unet_laplacian as UL
from oracle import bfcnn_oracle as O
from oracle import unet_oracle as U
from helpers import dev, host, assert_close

pytestmark = pytest.mark.gpu


def _rng(seed):
    return np.random.default_rng(seed)


@pytest.mark.parametrize("cin,cout", [(32, 32), (32, 64), (64, 128), (128, 64), (64, 32), (128, 32), (32, 128), (128, 128), (64, 64)])
@pytest.mark.parametrize("npix", [1, 63, 64, 1000])
def test_pointwise_gemm(cin, cout, npix):
    r = _rng(cin + cout + npix)
    x, w = r.normal(size=(1, 1, npix, cin)), r.normal(size=(1, 1, cin, cout)) / np.sqrt(cin)
    mult, res = r.uniform(0.2, 1.0, cout), r.normal(size=(1, 1, npix, cout))
    wp = UL.pack_pointwise(dev(w))
    for act, alpha in (("linear", None), ("leaky_relu_01", None), ("gelu", None), ("leaky_relu", 0.2)):
        ref = O.conv2d_same(x, w)
        ref = U.leaky(ref, alpha) if alpha is not None else U.act(ref, act)
        got = UL.pointwise(dev(x), wp, cout, act, alpha=alpha)
        assert_close(host(got), ref, what=f"pointwise {act}")
        got = UL.pointwise(dev(x), wp, cout, act, mult=dev(mult), res=dev(res), alpha=alpha)
        assert_close(host(got), res + mult * ref, what=f"pointwise {act} mult res")


def test_pointwise_small_integers_exact():
    r = _rng(5)
    x = r.integers(-8, 9, size=(1, 1, 200, 64)).astype(np.float64)
    w = r.integers(-4, 5, size=(1, 1, 64, 128)).astype(np.float64)
    got = host(UL.pointwise(dev(x), UL.pack_pointwise(dev(w)), 128))
    assert np.array_equal(got, O.conv2d_same(x, w))


@pytest.mark.parametrize("C", [32, 64, 128])
@pytest.mark.parametrize("npix", [1, 17, 64, 129, 777])
def test_convnext_mlp_fused(C, npix):
    r = _rng(C + npix)
    x, skip = r.normal(size=(1, 1, npix, C)), r.normal(size=(1, 1, npix, C))
    w1, w2 = r.normal(size=(1, 1, C, 4 * C)) / np.sqrt(C), r.normal(size=(1, 1, 4 * C, C)) / np.sqrt(4 * C)
    mult = r.uniform(0.2, 1.0, C)
    w1p, w2p = UL.pack_pointwise(dev(w1)), UL.pack_pointwise(dev(w2))
    for act in ("leaky_relu_01", "gelu", "linear"):
        ref = skip + mult * O.conv2d_same(U.act(O.conv2d_same(x, w1), act), w2)
        got = UL.convnext_mlp(dev(x), dev(skip), w1p, w2p, dev(mult), act)
        assert_close(host(got), ref, what=f"mlp C={C} {act}")
    got = UL.convnext_mlp(dev(x), None, w1p, w2p, None, "relu")
    assert_close(host(got), O.conv2d_same(np.maximum(O.conv2d_same(x, w1), 0), w2), what="mlp plain")


@pytest.mark.parametrize("C", [32, 64])
@pytest.mark.parametrize("npix", [1, 17, 64, 129, 777, 5000])
def test_convnext_mlp_split_f16(C, npix):
    """the f16x3 kernel (csrc/unet_h3.hip) against the same fp64 reference and the same tolerance as the fp32 kernel."""
    r = _rng(C + npix + 1)
    x, skip = r.normal(size=(1, 1, npix, C)) * 1.5, r.normal(size=(1, 1, npix, C))
    w1, w2 = r.normal(size=(1, 1, C, 4 * C)) / np.sqrt(C), r.normal(size=(1, 1, 4 * C, C)) / np.sqrt(4 * C)
    mult = r.uniform(0.2, 1.0, C)
    pk = UL.pack_mlp_h3(dev(w1.reshape(C, 4 * C)), dev(w2.reshape(4 * C, C)))
    for act in ("leaky_relu_01", "gelu", "linear", "relu"):
        ref = skip + mult * O.conv2d_same(U.act(O.conv2d_same(x, w1), act), w2)
        got = UL.convnext_mlp_h3(dev(x), dev(skip), pk, dev(mult), act)
        assert_close(host(got), ref, what=f"mlp h3 C={C} {act}")
    got = UL.convnext_mlp_h3(dev(x), None, pk, None, "linear")
    assert_close(host(got), O.conv2d_same(O.conv2d_same(x, w1), w2), what="mlp h3 plain")


@pytest.mark.parametrize("C", [32, 64])
@pytest.mark.parametrize("npix", [1, 100, 2049])
@pytest.mark.parametrize("use_ln", [True, False])
def test_convnext_block_with_1x1_depthwise_in_one_kernel(C, npix, use_ln):
    r = _rng(C + npix + 2)
    x = r.normal(size=(1, 1, npix, C)) * 2 + 0.3
    dw, g = r.normal(size=(1, 1, C, 1)), r.uniform(0.5, 1.5, C)
    w1, w2 = r.normal(size=(1, 1, C, 4 * C)) / np.sqrt(C), r.normal(size=(1, 1, 4 * C, C)) / np.sqrt(4 * C)
    mult = r.uniform(0.2, 1.0, C)
    pk = UL.pack_mlp_h3(dev(w1.reshape(C, 4 * C)), dev(w2.reshape(4 * C, C)))
    t = U.depthwise_same(x, dw)
    if use_ln:
        t = U.layer_norm(t, g)
    ref = x + mult * O.conv2d_same(U.act(O.conv2d_same(t, w1), "leaky_relu_01"), w2)
    got = UL.convnext_block1_h3(dev(x), dev(dw.reshape(C)), dev(g) if use_ln else None, pk, dev(mult), "leaky_relu_01")
    assert_close(host(got), ref, rel=5e-5, what="block1 h3")


@pytest.mark.parametrize("shape", [(1, 2, 2), (1, 4, 6), (2, 10, 14), (1, 34, 62), (3, 64, 64), (1, 2, 130)])
@pytest.mark.parametrize("act_up", ["linear", "relu", "leaky_relu"])
@pytest.mark.parametrize("use_ln", [True, False])
def test_first_decoder_block_forms_its_node_on_load(shape, act_up, use_ln):
    """bf_op_convnext_block1_up_h3 = bf_op_upsample_act_add then bf_op_convnext_block1_h3, bit for bit (the same expressions in the same
    order), and both against the fp64 restatement"""
    B, OH, OW = shape
    C = 32
    r = _rng(OH * 7 + OW + B)
    enc = r.normal(size=(B, OH, OW, C)) * 1.5
    low = r.normal(size=(B, OH // 2, OW // 2, C)) * 2 - 0.2
    dw, g = r.normal(size=(1, 1, C, 1)), r.uniform(0.5, 1.5, C)
    w1, w2 = r.normal(size=(1, 1, C, 4 * C)) / np.sqrt(C), r.normal(size=(1, 1, 4 * C, C)) / np.sqrt(4 * C)
    mult = r.uniform(0.2, 1.0, C)
    pk = UL.pack_mlp_h3(dev(w1.reshape(C, 4 * C)), dev(w2.reshape(4 * C, C)))
    gd = dev(g) if use_ln else None
    node = UL.upsample_act_add(dev(low), dev(enc), act_up)
    two = UL.convnext_block1_h3(node, dev(dw.reshape(C)), gd, pk, dev(mult), "leaky_relu_01")
    one = UL.convnext_block1_up_h3(dev(enc), dev(low), dev(dw.reshape(C)), gd, pk, dev(mult), "leaky_relu_01", act_up)
    assert np.array_equal(host(one), host(two)), f"max |fused - two kernels| = {np.abs(host(one) - host(two)).max():.3e}"
    x = enc + U.act(U.resize_bilinear(low, OH, OW), act_up)
    t = U.depthwise_same(x, dw)
    if use_ln:
        t = U.layer_norm(t, g)
    ref = x + mult * O.conv2d_same(U.act(O.conv2d_same(t, w1), "leaky_relu_01"), w2)
    assert_close(host(one), ref, rel=5e-5, what="fused first decoder block")


@pytest.mark.parametrize("shape", [(1, 2, 2), (2, 10, 14), (1, 34, 62), (3, 64, 64)])
@pytest.mark.parametrize("nblocks", [1, 2, 3])
@pytest.mark.parametrize("up,use_ln,act", [(True, True, "leaky_relu_01"), (False, True, "gelu"), (True, False, "relu"), (False, False, "leaky_relu_01")])
def test_decoder_chain_kernel(shape, nblocks, up, use_ln, act):
    """bf_op_convnext_chain32_h3: 1..3 pixel-wise ConvNext blocks (and the node in front of them) in one launch, against the fp64
    restatement and against the same blocks launched one by one (equal to rounding: the LayerNorm sums run in another order)"""
    B, OH, OW = shape
    C = 32
    r = _rng(OH * 5 + OW + nblocks)
    enc = r.normal(size=(B, OH, OW, C)) * 1.5
    low = r.normal(size=(B, OH // 2, OW // 2, C)) * 2 - 0.2
    blocks, dev_blocks, one_by_one = [], [], []
    for b in range(nblocks):
        dw, g = r.normal(size=(1, 1, C, 1)), r.uniform(0.5, 1.5, C)
        w1, w2 = r.normal(size=(1, 1, C, 4 * C)) / np.sqrt(C), r.normal(size=(1, 1, 4 * C, C)) / np.sqrt(4 * C)
        mult = r.uniform(0.2, 1.0, C) if b != 1 else None
        blocks.append((dw, g, w1, w2, mult))
        gd = dev(g) if use_ln else None
        md = None if mult is None else dev(mult)
        dev_blocks.append((UL.pack_mlp_h3_chain(dev(w1.reshape(C, 4 * C)), dev(w2.reshape(4 * C, C))), dev(dw.reshape(C)), gd, md))
        one_by_one.append((UL.pack_mlp_h3(dev(w1.reshape(C, 4 * C)), dev(w2.reshape(4 * C, C))), dev(dw.reshape(C)), gd, md))
    x = enc + U.act(U.resize_bilinear(low, OH, OW), "relu") if up else enc
    ref = x
    for dw, g, w1, w2, mult in blocks:
        t = U.depthwise_same(ref, dw)
        if use_ln:
            t = U.layer_norm(t, g)
        y = O.conv2d_same(U.act(O.conv2d_same(t, w1), act), w2)
        ref = ref + (y if mult is None else mult * y)
    got = UL.convnext_chain32_h3(dev(enc), dev(low) if up else None, dev_blocks, act, "relu")
    assert_close(host(got), ref, rel=8e-5, what=f"chain of {nblocks}")
    seq = UL.upsample_act_add(dev(low), dev(enc), "relu") if up else dev(enc)
    for pk, dwd, gd, md in one_by_one:
        seq = UL.convnext_block1_h3(seq, dwd, gd, pk, md, act)
    assert_close(host(got), host(seq).astype(np.float64), rel=2e-5, what="chain vs one launch per block")


def test_v5_hydra_with_and_without_the_decoder_chain():
    cfg, spec, params, m = _model(arith=1)
    _, noisy = O.synthetic_batch(2, 64, 96, seed=6)
    x = noisy.astype(np.float32)
    assert m.fuse_chain == 1 and any(k.endswith("mlp_h3c") for k in m._pack())
    chained = [np.array(t) for t in m(x)]
    m.set_option("fuse_chain", 0)
    single = [np.array(t) for t in m(x)]
    for a, b in zip(chained, single):
        assert np.abs(a - b).max() <= 2e-3                                       # 0..255 scale
    assert not np.array_equal(chained[0], single[0])                            # level 0 is the 32-channel one: two summation orders


def test_first_decoder_block_operator_refuses_what_it_was_not_built_for():
    x = torch.zeros((1, 4, 4, 64), device="cuda")
    low = torch.zeros((1, 2, 2, 64), device="cuda")
    pk = torch.zeros(UL.N.lib().bf_op_mlp_h3_pack_bytes(64), dtype=torch.uint8, device="cuda")
    with pytest.raises(Exception):
        UL.convnext_block1_up_h3(x, low, torch.ones(64, device="cuda"), None, pk, None, "relu")          # 64 channels
    x32, pk32 = torch.zeros((1, 3, 4, 32), device="cuda"), torch.zeros(UL.N.lib().bf_op_mlp_h3_pack_bytes(32), dtype=torch.uint8, device="cuda")
    with pytest.raises(ValueError):
        UL.convnext_block1_up_h3(x32, torch.zeros((1, 1, 2, 32), device="cuda"), torch.ones(32, device="cuda"), None, pk32, None, "relu")


@pytest.mark.parametrize("variant", [4, 3, 2, 1, 0], ids=["two-consumers-per-simd", "two-workgroups-per-cu", "wave-specialised-unrolled", "wave-specialised", "single-role"])
@pytest.mark.parametrize("k", [3, 5])
@pytest.mark.parametrize("shape", [(1, 1, 1), (1, 16, 32), (2, 20, 37), (1, 7, 70), (1, 64, 64), (1, 33, 8), (3, 48, 96)])
@pytest.mark.parametrize("use_ln", [True, False])
def test_encoder_convnext_block_in_one_kernel(k, shape, use_ln, variant):
    from blind_image_denoising_amd import _native as N
    N.check(N.lib().bf_op_set_variant(b"enc32", variant), None, "bf_op_set_variant")
    try:
        _encoder_block_case(k, shape, use_ln)
    finally:
        N.lib().bf_op_set_variant(b"enc32", -1)               # back to the default


@pytest.mark.parametrize("shape", [(7, 112, 192), (8, 128, 128), (9, 100, 90)])
def test_encoder_convnext_block_tile_order_over_the_xcds(shape):
    """more tiles than workgroups: every XCD walks its own contiguous range of tiles (294 tiles: 37 per XCD, 35 on the last; 256: one per
    workgroup; 9 x 7 x 3 = 189 < 256: the round-robin order) -- every pixel written once, against the fp64 restatement"""
    _encoder_block_case(5, shape, True)


def _encoder_block_case(k, shape, use_ln):
    C = 32
    r = _rng(k + shape[1] + shape[2])
    x = r.normal(size=shape + (C,)) * 2 + 0.3
    dw, g = r.normal(size=(k, k, C, 1)) * 0.3, r.uniform(0.5, 1.5, C)
    w1, w2 = r.normal(size=(1, 1, C, 4 * C)) / np.sqrt(C), r.normal(size=(1, 1, 4 * C, C)) / np.sqrt(4 * C)
    mult = r.uniform(0.2, 1.0, C)
    pk = UL.pack_mlp_h3(dev(w1.reshape(C, 4 * C)), dev(w2.reshape(4 * C, C)))
    t = U.depthwise_same(x, dw)
    if use_ln:
        t = U.layer_norm(t, g)
    ref = x + mult * O.conv2d_same(U.act(O.conv2d_same(t, w1), "leaky_relu_01"), w2)
    got = UL.convnext_block_h3(dev(x), dev(dw.reshape(k, k, C)), dev(g) if use_ln else None, pk, dev(mult), "leaky_relu_01")
    assert_close(host(got), ref, rel=5e-5, what="encoder block h3")


def test_convnext_mlp_split_f16_small_integers_exact_and_tiny_weights():
    r = _rng(77)
    C = 32
    x = r.integers(-8, 9, size=(1, 1, 300, C)).astype(np.float64)
    w1 = r.integers(-3, 4, size=(1, 1, C, 4 * C)).astype(np.float64)
    w2 = r.integers(-3, 4, size=(1, 1, 4 * C, C)).astype(np.float64)
    pk = UL.pack_mlp_h3(dev(w1.reshape(C, 4 * C)), dev(w2.reshape(4 * C, C)))
    got = host(UL.convnext_mlp_h3(dev(x), None, pk, None, "relu"))
    assert np.array_equal(got, O.conv2d_same(np.maximum(O.conv2d_same(x, w1), 0), w2))
    # weights far below the f16 normal range (and large inputs): the power-of-two pre-scale keeps their lo parts exact.
    # (the ACTIVATIONS are not rescaled: an activation below 2^-14 keeps an absolute rounding floor of 2^-25, the
    # documented domain of the split-f16 arithmetic -- here x w1 stays O(1..100))
    xs = x * 2048.0
    w1s, w2s = w1 * 2.0 ** -14 * r.uniform(0.5, 1, w1.shape), w2 * 1e-7 * r.uniform(0.5, 1, w2.shape)
    pk = UL.pack_mlp_h3(dev(w1s.reshape(C, 4 * C)), dev(w2s.reshape(4 * C, C)))
    ref = O.conv2d_same(np.maximum(O.conv2d_same(xs, w1s), 0), w2s)
    got = host(UL.convnext_mlp_h3(dev(xs), None, pk, None, "relu")).astype(np.float64)
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()


@pytest.mark.parametrize("C", [32, 64, 128])
@pytest.mark.parametrize("k", [0, 1, 3, 5])
@pytest.mark.parametrize("shape", [(1, 1, 1), (2, 7, 9), (1, 16, 40)])
def test_dwconv_layernorm(C, k, shape):
    r = _rng(C * 10 + k)
    x = r.normal(size=shape + (C,)) * 2 + 0.5
    w = r.normal(size=(k, k, C, 1)) if k else None
    g = r.uniform(0.5, 1.5, C)
    t = U.depthwise_same(x, w) if k else x
    wd = dev(w.reshape(k, k, C)) if k else None
    assert_close(host(UL.dwconv_ln(dev(x), wd, dev(g))), U.layer_norm(t, g), rel=5e-5, what="dw+ln")
    assert_close(host(UL.dwconv_ln(dev(x), wd, None, "leaky_relu_01")), U.act(t, "leaky_relu_01"), what="dw+act")
    assert_close(host(UL.dwconv_ln(dev(x), wd, dev(g), "leaky_relu_01")), U.act(U.layer_norm(t, g), "leaky_relu_01"),
                 rel=5e-5, what="dw+ln+act")


@pytest.mark.parametrize("shape", [(1, 2, 2, 32), (2, 8, 12, 32), (1, 16, 16, 64), (1, 6, 10, 128)])
def test_smooth_split_average_and_gaussian(shape):
    x = _rng(3).normal(size=shape)
    for k in (2, 4):                                       # even kernels: TF SAME puts the extra pad after
        smooth = O.avg_pool_same(x, (k, k), 1)
        lap, down = UL.smooth_split(dev(x), k)
        assert_close(host(lap), x - smooth, what="lap avg even")
        assert_close(host(down), smooth[:, ::2, ::2], what="down avg even")
        lap, full = UL.smooth_split(dev(x), k, down_stride=1)
        assert_close(host(full), smooth, what="smooth full")
    smooth = O.avg_pool_same(x, (3, 3), 1)
    lap, down = UL.smooth_split(dev(x), 3)
    assert_close(host(lap), x - smooth, what="lap avg")
    assert_close(host(down), smooth[:, ::2, ::2], what="down avg")
    g = U.gaussian_kernel_3()
    assert np.array_equal(g.astype(np.float32), UL.gaussian_kernel((3, 3)))
    smooth = U.depthwise_same(x, np.repeat(g[:, :, None, None], shape[-1], axis=2))
    lap, down = UL.smooth_split(dev(x), 3, dev(g))
    assert_close(host(lap), x - smooth, what="lap gauss")
    assert_close(host(down), smooth[:, ::2, ::2], what="down gauss")


@pytest.mark.parametrize("shape", [(1, 2, 2, 32), (2, 8, 12, 32), (1, 17, 35, 64), (1, 6, 10, 128), (1, 40, 33, 32)])
@pytest.mark.parametrize("k", [3, 5])
def test_norm_smooth_split_fused(shape, k):
    r = _rng(13 + k)
    x, g = r.normal(size=shape) * 2 + 0.4, r.uniform(0.5, 1.5, shape[-1])
    for use_ln in (True, False):
        y = U.act(U.layer_norm(x, g) if use_ln else x, "leaky_relu_01")
        smooth = O.avg_pool_same(y, (k, k), 1)
        lap, down = UL.norm_smooth_split(dev(x), dev(g) if use_ln else None, "leaky_relu_01", k)
        assert_close(host(lap), y - smooth, rel=5e-5, what="lap")
        assert_close(host(down), smooth[:, ::2, ::2], rel=5e-5, what="down")
    gk = U.gaussian_kernel_3((k, k))
    y = U.act(U.layer_norm(x, g), "leaky_relu_01")
    smooth = U.depthwise_same(y, np.repeat(gk[:, :, None, None], shape[-1], axis=2))
    lap, down = UL.norm_smooth_split(dev(x), dev(g), "leaky_relu_01", k, dev(gk))
    assert_close(host(lap), y - smooth, rel=5e-5, what="lap gauss")
    assert_close(host(down), smooth[:, ::2, ::2], rel=5e-5, what="down gauss")


@pytest.mark.parametrize("shape,out", [((1, 128, 128, 32), (16, 16)), ((2, 16, 16, 32), (128, 128)), ((1, 20, 36, 64), (16, 16)),
                                       ((1, 16, 16, 32), (20, 36)), ((1, 4, 4, 128), (16, 16)), ((1, 16, 16, 32), (16, 16))])
def test_resize_bilinear(shape, out):
    x = _rng(7).normal(size=shape)
    assert_close(host(UL.resize_bilinear(dev(x), *out)), U.resize_bilinear(x, *out), what="resize")


def test_upsample_act_add():
    r = _rng(8)
    x, other = r.normal(size=(2, 5, 7, 32)), r.normal(size=(2, 10, 14, 32))
    assert_close(host(UL.upsample_act_add(dev(x), dev(other), "leaky_relu_01")),
                 other + U.act(O.upsample_bilinear_2x(x), "leaky_relu_01"), what="up+act+add")
    assert_close(host(UL.upsample_act_add(dev(x), None)), O.upsample_bilinear_2x(x), what="up")


@pytest.mark.parametrize("B,T", [(1, 256), (3, 256), (2, 64), (5, 100), (2, 512), (3, 513), (2, 1100), (300, 33)])
def test_attention(B, T):
    r = _rng(9)
    q, v, k = (r.normal(size=(B, T, 32)) for _ in range(3))
    assert_close(host(UL.attention(dev(q), dev(v), dev(k))), U.dot_attention(q, v, k), rel=5e-5, what="attention")


@pytest.mark.parametrize("u8", [True, False])
def test_first_conv_normalise_and_virtual_padding(u8):
    r = _rng(10)
    img = r.integers(0, 256, size=(2, 13, 21, 3)).astype(np.uint8)
    w = r.normal(size=(5, 5, 3, 32)) * 0.2
    padded = np.zeros((2, 16, 32, 3)); padded[:, :13, :21] = img
    ref = U.act(O.conv2d_same(O.layer_normalize(padded, 0.0, 255.0), w), "leaky_relu_01")
    x = torch.from_numpy(img).cuda() if u8 else dev(img)
    got = UL.first_conv(x, dev(w), 16, 32, "leaky_relu_01", True, 0.0, 255.0)
    assert_close(host(got), ref, what="first conv")


@pytest.mark.parametrize("u8", [True, False])
@pytest.mark.parametrize("ks", [5, 7, 3])
@pytest.mark.parametrize("shape,padded_hw", [((2, 13, 21), (16, 32)), ((1, 70, 130), (128, 256)), ((3, 64, 64), (64, 64))])
def test_first_conv_split_f16(u8, shape, padded_hw, ks):
    """csrc/unet_h3_first.hip against the same reference and tolerance as the fp32 kernel; all four activations; tiles that
    straddle the source image, the padded image and the launch grid."""
    r = _rng(12 + shape[1])
    img = r.integers(0, 256, size=shape + (3,)).astype(np.uint8)
    w = r.normal(size=(ks, ks, 3, 32)) * 0.2
    H, W = padded_hw
    padded = np.zeros((shape[0], H, W, 3)); padded[:, :shape[1], :shape[2]] = img
    pre = O.conv2d_same(O.layer_normalize(padded, 0.0, 255.0), w)
    x = torch.from_numpy(img).cuda() if u8 else dev(img)
    for act in ("leaky_relu_01", "linear", "relu", "gelu"):
        got = UL.first_conv(x, dev(w), H, W, act, True, 0.0, 255.0, arith=1)
        assert_close(host(got), U.act(pre, act), what=f"first conv f16x3 {act}")
    ints = r.integers(-3, 4, size=(ks, ks, 3, 32)).astype(np.float64)          # exact on small integers without normalisation
    got = UL.first_conv(x, dev(ints), H, W, "linear", False, 0.0, 255.0, arith=1)
    assert np.array_equal(host(got), O.conv2d_same(padded, ints))


def test_head_out_f32_u8_and_crop():
    r = _rng(11)
    x, w = r.normal(size=(2, 8, 8, 32)), r.normal(size=(1, 1, 32, 3)) * 0.3
    ref = O.layer_denormalize(np.tanh(2 * O.conv2d_same(x, w)) * 0.51, 0.0, 255.0)
    assert_close(host(UL.head_out(dev(x), dev(w), 8, 8, False, True, 0.0, 255.0)), ref, rel=2e-6 * 255, what="head f32")
    got = host(UL.head_out(dev(x), dev(w), 5, 7, True, True, 0.0, 255.0))
    want = np.clip(np.rint(ref[:, :5, :7]), 0, 255)
    assert got.dtype == np.uint8 and np.abs(got.astype(int) - want).max() <= 1 and (got != want).mean() < 0.01


@pytest.mark.parametrize("C", [32, 64, 128])
@pytest.mark.parametrize("use_ln", [True, False])
@pytest.mark.parametrize("arith", [0, 1], ids=["f32", "f16x3"])
def test_head_fused_layernorm_two_convs_tanh(C, use_ln, arith):
    r = _rng(C + 12)
    x = r.normal(size=(2, 9, 11, C)) * 2 + 0.2
    g = r.uniform(0.5, 1.5, C)
    w0, w1 = r.normal(size=(1, 1, C, 32)) / np.sqrt(C), r.normal(size=(1, 1, 32, 3)) * 0.3
    t = U.layer_norm(x, g) if use_ln else x
    ref = O.layer_denormalize(np.tanh(2 * O.conv2d_same(U.act(O.conv2d_same(t, w0), "leaky_relu_01"), w1)) * 0.51, 0.0, 255.0)
    w0p = UL.pack_pointwise(dev(w0))
    got = UL.head_fused(dev(x), dev(g) if use_ln else None, w0p, "leaky_relu_01", dev(w1), 9, 11, False, True, 0.0, 255.0, arith=arith)
    assert_close(host(got), ref, rel=3e-5 * 255 / max(1.0, np.abs(ref).max()), what="fused head f32")
    got = host(UL.head_fused(dev(x), dev(g) if use_ln else None, w0p, "leaky_relu_01", dev(w1), 6, 7, True, True, 0.0, 255.0, arith=arith))
    want = np.clip(np.rint(ref[:, :6, :7]), 0, 255)
    assert got.dtype == np.uint8 and np.abs(got.astype(int) - want).max() <= 1 and (got != want).mean() < 0.02


def test_head_fused_split_f16_on_a_large_ragged_map_and_exact_on_integers():
    """bf_op_head_fused_h3 on a map that spans many wave iterations with a ragged tail and a crop, both channel counts; and with integer
    operands that fit f16 (nothing to round in the first 1x1) the result equals the fp32 kernel's"""
    r = _rng(77)
    for C in (32, 64):
        x = r.normal(size=(3, 70, 61, C)) * 1.5
        g = r.uniform(0.5, 1.5, C)
        w0, w1 = r.normal(size=(1, 1, C, 32)) / np.sqrt(C), r.normal(size=(1, 1, 32, 3)) * 0.3
        w0p = UL.pack_pointwise(dev(w0))
        ref = O.layer_denormalize(np.tanh(2 * O.conv2d_same(U.act(O.conv2d_same(U.layer_norm(x, g), w0), "relu"), w1)) * 0.51, 0.0, 255.0)
        got = UL.head_fused(dev(x), dev(g), w0p, "relu", dev(w1), 67, 59, False, True, 0.0, 255.0, arith=1)
        assert_close(host(got), ref[:, :67, :59], rel=3e-5 * 255 / max(1.0, np.abs(ref).max()), what="fused head f16x3, large")
        xi = r.integers(-8, 9, size=(1, 33, 47, C)).astype(np.float64)
        wi = r.integers(-4, 5, size=(1, 1, C, 32)).astype(np.float64)
        wip = UL.pack_pointwise(dev(wi))
        a = host(UL.head_fused(dev(xi), None, wip, "linear", dev(w1 * 0.01), 33, 47, False, True, 0.0, 255.0, arith=1))
        b = host(UL.head_fused(dev(xi), None, wip, "linear", dev(w1 * 0.01), 33, 47, False, True, 0.0, 255.0, arith=0))
        assert np.array_equal(a, b)


@pytest.mark.parametrize("cin,cout", [(32, 32), (32, 64), (64, 128), (128, 64), (128, 128)])
@pytest.mark.parametrize("k,stride,shape", [(3, 1, (2, 9, 11)), (2, 2, (1, 10, 14)), (2, 2, (1, 7, 9)), (3, 2, (1, 8, 8)), (5, 1, (1, 6, 6)),
                                            (1, 1, (1, 5, 5))])
def test_conv2d_strided_same(cin, cout, k, stride, shape):
    r = _rng(cin + cout + k)
    x = r.normal(size=shape + (cin,))
    w = r.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin)
    ref = O.conv2d_same(x, w, stride=stride)
    res = r.normal(size=ref.shape)
    wp = UL.pack_conv(dev(w))
    assert_close(host(UL.conv2d(dev(x), wp, cout, k, stride)), ref, what="conv2d")
    assert_close(host(UL.conv2d(dev(x), wp, cout, k, stride, "leaky_relu_01", dev(res))), res + U.act(ref, "leaky_relu_01"),
                 what="conv2d act res")


@pytest.mark.parametrize("shape", [(1, 2, 2, 32), (2, 7, 9, 32), (1, 16, 16, 64)])
def test_maxpool2_same(shape):
    x = _rng(14).normal(size=shape)
    assert np.array_equal(host(UL.maxpool2(dev(x))), U.max_pool_2x2_same(x).astype(np.float32))


def test_channel_multiplier():
    w = np.linspace(-2.0, 1.0, 64)
    assert_close(host(UL.channel_multiplier(dev(w))), np.tanh(np.maximum(1 + w, 0)), what="multiplier")


# ---- whole model ---------------------------------------------------------------------------------

def _model(depth=3, width=3, seed=42, arith=1, **bb):
    cfg = U.canonical_config(depth=depth, width=width)
    cfg["model"]["backbone"].update(bb)
    spec = U.UnetLaplacianSpec.from_config(cfg["model"])
    params = U.init_params(spec, seed=seed)
    m = bf.model_builder(cfg["model"], device="cuda").hydra
    assert [(v[0], tuple(v[1])) for v in m.trainable_variables] == [(n, tuple(s)) for n, s, _ in spec.tensors()]
    m.set_weights(params)
    m.set_option("arith", arith)
    return cfg, spec, params, m


def _check_f32(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape and np.isfinite(got).all()
    mae = np.abs(got - ref).mean() / 255.0
    assert mae <= 1e-4, f"normalised MAE {mae:.3e}"
    assert np.abs(got - ref).max() <= 0.05, f"max err {np.abs(got - ref).max():.3e} (0..255 scale)"


def _check_u8(got, ref):
    assert got.dtype == np.uint8 and got.shape == ref.shape
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.01, (d.max(), (d > 0).mean())


@pytest.mark.parametrize("arith", [1, 0], ids=["f16x3", "f32"])
@pytest.mark.parametrize("shape", [(1, 64, 64), (2, 32, 96), (1, 128, 128)])
def test_v5_hydra_all_scales_match_oracle(shape, arith):
    cfg, spec, params, m = _model(arith=arith)
    _, noisy = O.synthetic_batch(*shape, seed=shape[1])
    x = noisy.astype(np.float32)
    got, ref = m(x), U.hydra_forward(spec, params, x.astype(np.float64))
    assert len(got) == 3
    for g, r in zip(got, ref):
        _check_f32(g, r)


def test_v5_hydra_same_bits_with_and_without_the_fused_first_decoder_block():
    cfg, spec, params, m = _model(arith=1)
    _, noisy = O.synthetic_batch(2, 64, 96, seed=5)
    x = noisy.astype(np.float32)
    assert m.fuse_up_block == 1
    fused = [np.array(t) for t in m(x)]
    m.set_option("fuse_up_block", 0)
    plain = [np.array(t) for t in m(x)]
    for a, b in zip(fused, plain):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("bb", [dict(use_self_attention=False, depth=2), dict(use_laplacian_averaging=False),
                                dict(use_mix_project=True, width=1), dict(upsample_type="bilinear", filters=32, depth=1),
                                dict(use_output_normalization=False, use_gamma=False, width=2),
                                dict(decoder_kernel_size=3, encoder_kernel_size=3, width=1),
                                dict(use_self_attention=False, width=1),
                                dict(downsample_type="conv2d", upsample_type="upsample_bilinear_conv2d", width=1),
                                dict(downsample_type="maxpool", upsample_type="upsample_nearest_conv2d", width=1, depth=2),
                                dict(downsample_type="conv2d", upsample_type="nearest", filters_level_multiplier=1.0, width=1,
                                     use_self_attention=False)],
                         ids=["no-attn-d2", "gaussian", "mix-project", "depth1", "no-outnorm-no-gamma", "k3", "convnext-c128",
                              "conv2d-down-bilinear-conv-up", "maxpool-down-nearest-conv-up", "nearest-up-flat-filters"])
def test_builder_variants_match_oracle(bb):
    depth, width = bb.pop("depth", 3), bb.pop("width", 2)
    cfg, spec, params, m = _model(depth=depth, width=width, seed=7, **bb)
    _, noisy = O.synthetic_batch(1, 64, 64, seed=3)
    x = noisy.astype(np.float32)
    for g, r in zip(m(x), U.hydra_forward(spec, params, x.astype(np.float64))):
        _check_f32(g, r)


def test_shipped_v6_config_matches_oracle():
    """configs/unet_laplacian_v6.json: 2x2 averaging, 5x5 decoder depthwise, nearest + 3x3 conv upsample, 2x2 stride-2 conv
    downsample."""
    cfg, spec, params, m = _model(seed=6, gaussian_kernel_size=2, decoder_kernel_size=5, use_laplacian_averaging=True,
                                  upsample_type="upsample_nearest_conv2d", downsample_type="conv2d")
    _, noisy = O.synthetic_batch(1, 64, 96, seed=6)
    x = noisy.astype(np.float32)
    for g, r in zip(m(x), U.hydra_forward(spec, params, x.astype(np.float64))):
        _check_f32(g, r)
    _check_u8(bf.DenoiserModule(m)(noisy), U.denoiser_module_call(spec, params, noisy))


@pytest.mark.parametrize("up", ["upsample_nearest_conv2d", "upsample_laplacian_conv2d"], ids=["v3", "v4"])
def test_shipped_v3_v4_configs_match_oracle(up):
    """configs/unet_laplacian_v3.json / v4.json: 4 levels (32/64/128/256, self-attention on the 256-channel level) and
    AdditiveAttentionGate in front of every decoder Add."""
    cfg, spec, params, m = _model(depth=4, width=3, seed=8, use_attention_gates=True, upsample_type=up)
    _, noisy = O.synthetic_batch(1, 64, 128, seed=8)
    x = noisy.astype(np.float32)
    got, ref = m(x), U.hydra_forward(spec, params, x.astype(np.float64))
    assert len(got) == 4
    for g, r in zip(got, ref):
        _check_f32(g, r)
    _check_u8(bf.DenoiserModule(m)(noisy), U.denoiser_module_call(spec, params, noisy))


def test_pointwise_attention_gate_epilogues():
    r = _rng(21)
    C = 64
    x, res, add = r.normal(size=(1, 1, 333, C)), r.normal(size=(1, 1, 333, C)), r.normal(size=(1, 1, 333, C))
    w = r.normal(size=(1, 1, C, C)) / np.sqrt(C)
    mult = r.uniform(0.2, 1.0, C)
    wp = UL.pack_pointwise(dev(w))
    acc = O.conv2d_same(x, w)
    assert_close(host(UL.pointwise_ex(dev(x), wp, C, 1, "leaky_relu_01", res=dev(res))), U.act(acc + res, "leaky_relu_01"), what="mode 1")
    ref = res / (1.0 + np.exp(-4.0 * mult * acc)) + add
    assert_close(host(UL.pointwise_ex(dev(x), wp, C, 2, mult=dev(mult), res=dev(res), add=dev(add))), ref, what="mode 2")


@pytest.mark.parametrize("arith", [1, 0], ids=["f16x3", "f32"])
@pytest.mark.parametrize("hw", [(64, 64), (40, 50), (17, 100), (128, 128)])
def test_denoiser_module_u8(hw, arith):
    cfg, spec, params, m = _model(seed=11, arith=arith)
    _, noisy = O.synthetic_batch(1, hw[0], hw[1], seed=hw[0] + hw[1])
    got = bf.DenoiserModule(m)(noisy)
    assert got.shape == noisy.shape and got.dtype == np.uint8
    _check_u8(got, U.denoiser_module_call(spec, params, noisy))
    f = bf.DenoiserModule(m, cast_to_uint8=False)(noisy)
    _check_f32(f, U.denoiser_module_call(spec, params, noisy, cast_to_uint8=False))


def test_midgrey_fixed_point_and_determinism():
    cfg, spec, params, m = _model(seed=2)
    y = m(np.full((1, 64, 64, 3), 127.5, np.float32))
    assert all(np.array_equal(o, np.full_like(o, 127.5)) for o in y)
    _, noisy = O.synthetic_batch(3, 64, 64, seed=1)
    mod = bf.DenoiserModule(m)
    a = mod(noisy)
    assert np.array_equal(a, mod(noisy)) and np.array_equal(a[1:2], mod(noisy[1:2]))


def test_full_size_config_512_batch_properties_and_256_oracle():
    """BASELINE.json configs[4] sizes: a 256x256 image against the oracle (the fp64 NumPy oracle needs minutes for 512x512),
    and at 512x512 the size-independent properties: determinism and batch independence."""
    cfg, spec, params, m = _model(seed=21)
    mod = bf.DenoiserModule(m)
    _, noisy = O.synthetic_batch(1, 256, 256, seed=9)
    _check_u8(mod(noisy), U.denoiser_module_call(spec, params, noisy))
    _, big = O.synthetic_batch(3, 512, 512, seed=10)
    full = mod(big)
    assert full.shape == big.shape and np.array_equal(full, mod(big))
    assert np.array_equal(full[2:3], mod(big[2:3]))
    # the batch of the config itself: 32 x 512 x 512 (the three images above tiled; device tensors in, device tensor out)
    import torch
    b32 = torch.from_numpy(np.concatenate([big] * 11)[:32]).cuda()
    out32 = mod(b32)
    assert out32.shape == b32.shape and out32.dtype == torch.uint8 and torch.equal(out32, mod(b32))
    got = out32.cpu().numpy()
    for i in (0, 1, 2, 17, 31):                                   # image i of the batch = image i % 3 on its own
        assert np.array_equal(got[i], full[i % 3]), i
    assert m.check_status()


def test_shape_and_config_errors():
    cfg, spec, params, m = _model(seed=2)
    with pytest.raises(ValueError, match="multiples of 4"):
        m(np.zeros((1, 30, 64, 3), np.float32))
    bad = U.canonical_config()["model"]; bad["backbone"]["use_complex_base"] = True
    with pytest.raises(NotImplementedError):
        bf.model_builder(bad, device="cuda")


def test_save_and_load_model_roundtrip(tmp_path):
    cfg, spec, params, m = _model(depth=2, width=1, seed=5)
    bf.save_model(m, str(tmp_path / "unet"))
    mod = bf.load_model(str(tmp_path / "unet"))
    _, noisy = O.synthetic_batch(1, 32, 32, seed=4)
    assert np.array_equal(mod(noisy), bf.DenoiserModule(m)(noisy))


@pytest.mark.parametrize("k,s,shape", [(2, 2, (2, 9, 7, 5, 8)), (3, 2, (1, 16, 16, 32, 32)), (4, 2, (2, 5, 6, 3, 4)),
                                       (5, 2, (1, 8, 12, 16, 8)), (3, 1, (1, 7, 9, 4, 4)), (3, 3, (1, 6, 5, 2, 3))])
def test_conv2d_transpose_matches_oracle(k, s, shape):
    """upsample_type "conv2d_transpose" (bfcnn/upsampling.py:37-48): Conv2DTranspose padding same."""
    from blind_image_denoising_amd import unet_laplacian as UL
    B, H, W, cin, cout = shape
    rng = np.random.default_rng(k + 7 * s)
    x = rng.standard_normal((B, H, W, cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, cout, cin)) * 0.2).astype(np.float32)
    for act in ("linear", "leaky_relu_01"):
        got = UL.conv2d_transpose(torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda(), s, act).cpu().numpy()
        ref = U.act(U.conv2d_transpose_same(x.astype(np.float64), w.astype(np.float64), s), act)
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("mix", [True, False], ids=["mix_project", "no_mix"])
@pytest.mark.parametrize("depth,width", [(2, 1), (3, 2), (2, 3)])
@pytest.mark.parametrize("arith", [1, 0], ids=["f16x3", "f32"])
def test_use_concat_decoder(mix, depth, width, arith):
    """use_concat = True, the reference builder's DEFAULT (backbone_unet_laplacian.py:52, 516-517): the decoder nodes Concatenate the
    encoder feature and the up-sampled map instead of adding them.  With use_mix_project the 1x1 behind the Concatenate runs as two
    half-matrices (the 2 C map is never formed); without it the first decoder block of a level works on 2 C channels (depthwise,
    LayerNorm, 1x1 2C -> 4C -> C, no skip).  Every output scale against the oracle, u8 through the module."""
    cfg, spec, params, m = _model(depth=depth, width=width, seed=5 + depth, arith=arith, use_concat=True, use_mix_project=mix)
    assert spec.use_concat and m.use_concat
    S = 16 * 2 ** (depth - 2)
    _, noisy = O.synthetic_batch(2, 2 * S, 3 * S, seed=depth)
    x = noisy.astype(np.float32)
    got = m(x)
    ref = U.hydra_forward(spec, params, x.astype(np.float64))
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        _check_f32(g, r)
    _check_u8(bf.DenoiserModule(m)(noisy), U.denoiser_module_call(spec, params, noisy))


@pytest.mark.parametrize("width,enc_k,dec_k", [(1, 5, 3), (2, 3, 1)])
@pytest.mark.parametrize("arith", [1, 0], ids=["f16x3", "f32"])
def test_256_channel_convnext_level(width, enc_k, dec_k, arith):
    """a four-level model WITHOUT the self-attention bottleneck: its deepest level is ConvNext blocks on 256 channels (depthwise +
    LayerNorm over 256, the 256 -> 1024 -> 256 MLP in eight 128-channel slices of the hidden layer over the library's 1x1 operators).
    Every output scale against the oracle."""
    cfg, spec, params, m = _model(depth=4, width=width, seed=9, arith=arith, use_self_attention=False,
                                  encoder_kernel_size=enc_k, decoder_kernel_size=dec_k)
    assert spec.level_filters(3) == 256
    _, noisy = O.synthetic_batch(2, 32, 48, seed=4)
    x = noisy.astype(np.float32)
    got = m(x)
    ref = U.hydra_forward(spec, params, x.astype(np.float64))
    for g, r in zip(got, ref):
        _check_f32(g, r)
    _check_u8(bf.DenoiserModule(m)(noisy), U.denoiser_module_call(spec, params, noisy))


def test_use_concat_is_refused_where_it_is_not_built():
    cfg = U.canonical_config(depth=3, width=1)
    cfg["model"]["backbone"].update(use_concat=True, use_attention_gates=True)
    with pytest.raises(NotImplementedError):
        bf.model_builder(cfg["model"], device="cuda")
    cfg["model"]["backbone"].update(use_attention_gates=False, use_mix_project=False, depth=4)      # level 2: 2 x 128 channels
    with pytest.raises(NotImplementedError):
        bf.model_builder(cfg["model"], device="cuda")


def _random_unet_backbone(rng):
    """a random point of the unet_laplacian builder's option space (backbone_unet_laplacian.py:35-130), filters 32"""
    depth = int(rng.integers(2, 5))
    bb = dict(depth=depth, width=int(rng.integers(1, 4)), filters=32,
              encoder_kernel_size=int(rng.choice([3, 5])), decoder_kernel_size=int(rng.choice([1, 3, 5])),
              activation=str(rng.choice(["leaky_relu_01", "leaky_relu", "relu"])),
              use_ln=bool(rng.random() < 0.8), use_gamma=bool(rng.random() < 0.8),
              use_mix_project=bool(rng.random() < 0.3), use_self_attention=bool(rng.random() < 0.6),
              use_attention_gates=bool(rng.random() < 0.3), use_output_normalization=bool(rng.random() < 0.7),
              use_concat=bool(rng.random() < 0.3),
              use_laplacian=True, use_laplacian_averaging=bool(rng.random() < 0.5), gaussian_kernel_size=int(rng.choice([2, 3, 5])),
              downsample_type=str(rng.choice(["strides", "conv2d", "maxpool"])),
              upsample_type=str(rng.choice(["upsample_laplacian_conv2d", "upsample_bilinear_conv2d", "upsample_nearest_conv2d", "bilinear", "nn"])))
    if bb["upsample_type"] in ("bilinear", "nn"):
        bb["filters_level_multiplier"] = 1.0
    if rng.random() < 0.3:                         # the trained archive's graph revision, option by option
        bb.update(convnext_activation="gelu", encoder_level_activation=bool(rng.integers(2)),
                  output_normalization_at_heads=bool(rng.integers(2)), upsample_linear=bool(rng.integers(2)),
                  attention_full_resolution=bool(rng.integers(2)), attention_activation=str(rng.choice(["", "gelu"])))
    return bb


@pytest.mark.parametrize("seed", range(int(os.environ.get("BF_SWEEP_N", 48))))     # BF_SWEEP_N=300: a longer hunt
def test_random_unet_configurations_match_oracle(seed):
    """a seeded sweep over the unet_laplacian builder's options (depth 2-4, width, kernel sizes, LayerNorm / multiplier / mix projection /
    attention / attention gates / output normalisation on or off, both Laplacian splits, all three down-samplers, five up-samplers, the
    archive revision's switches): what builds must match the oracle on every output scale; what the operators do not cover must refuse"""
    rng = np.random.default_rng(3000 + seed)
    bb = _random_unet_backbone(rng)
    try:
        rest = {k: v for k, v in bb.items() if k not in ("depth", "width")}
        cfg, spec, params, m = _model(depth=bb["depth"], width=bb["width"], seed=seed, arith=int(rng.integers(2)), **rest)
    except NotImplementedError as e:
        pytest.skip(f"outside the built graph (refused): {e}")
    S = 16 * 2 ** (bb["depth"] - 2)
    _, noisy = O.synthetic_batch(2, 2 * S, 3 * S, seed=seed)
    x = noisy.astype(np.float32)
    try:
        got = m(x)
    except NotImplementedError as e:
        pytest.skip(f"outside the built operators (refused at run time): {e}")
    ref = U.hydra_forward(spec, params, x.astype(np.float64))
    assert len(got) == len(ref) == bb["depth"]
    for g, r in zip(got, ref):
        _check_f32(g, r)
    if seed % 3 == 0:                                          # and the uint8 module on a ragged frame (padded to a power of two inside)
        ragged = noisy[:1, :2 * S - 3, :3 * S - 5]
        _check_u8(bf.DenoiserModule(m)(ragged), U.denoiser_module_call(spec, params, ragged))
