"""GPU parity of the whole inference path (DenoiserModule.__call__ / hydra) through the public
host API -> C ABI, against the fp64 oracle and the committed golden fixtures.
Bars (BASELINE.json north_star): uint8 outputs within +-1 LSB; f32 hydra output MAE <= 1e-4 on
the normalised [-0.5, 0.5] scale (= 0.0255 on the 0..255 scale)."""
import os
import pathlib

import numpy as np
import pytest
import torch

import blind_image_denoising_amd as bf
from oracle import bfcnn_oracle as O

pytestmark = pytest.mark.gpu
G = pathlib.Path(__file__).resolve().parent / "golden"


def _model(no_layers, seed=42, nontrivial_bn=True, kernel_size=3, **kw):
    cfg = O.canonical_config(no_layers=no_layers, kernel_size=kernel_size)
    spec = O.ResnetSpec.from_config(cfg["model"], **kw)
    params, state = O.init_params(spec, seed=seed, nontrivial_bn=nontrivial_bn)
    m = bf.model_builder(cfg["model"], device="cuda", **kw).hydra
    m.set_weights(params, state)
    return cfg, spec, params, state, m


def _check_f32(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape and np.isfinite(got).all()
    mae_norm = np.abs(got - ref).mean() / 255.0
    assert mae_norm <= 1e-4, f"normalised MAE {mae_norm:.3e} > 1e-4"
    assert np.abs(got - ref).max() <= 0.05, f"max err {np.abs(got - ref).max():.3e} (0..255 scale)"


def _check_u8(got, ref):
    assert got.dtype == np.uint8 and got.shape == ref.shape
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1, f"max LSB diff {d.max()}"
    assert (d > 0).mean() < 0.01, f"{(d > 0).mean():.4f} of the pixels differ by 1 LSB"


def test_golden_net_f32_and_u8():
    z = np.load(G / "net_2blocks.npz")
    m = bf.model_builder(O.canonical_config(no_layers=2)["model"], device="cuda").hydra
    m.set_weights(z["params"], z["state"])
    _check_f32(m(z["noisy"].astype(np.float32)), z["hydra_f32"])
    mod = bf.DenoiserModule(m)
    _check_u8(mod(z["noisy"]), z["out_u8"])
    _check_u8(mod(z["ragged"]), z["ragged_u8"])          # 40x50 -> padded to 64x64 -> cropped


# (fused_blocks, arith): fused split-f16 blocks on the f16 matrix cores (default), fused exact-fp32 blocks,
# one exact-fp32 kernel per convolution -- all three against the same oracle with the same bars
@pytest.mark.parametrize("fused,arith", [(1, 1), (1, 0), (0, 0)], ids=["fused-f16x3", "fused-f32", "unfused-f32"])
@pytest.mark.parametrize("no_layers,shape", [(0, (1, 16, 16)), (1, (2, 32, 48)), (3, (2, 30, 45)), (6, (1, 64, 64))])
def test_hydra_f32_matches_oracle(no_layers, shape, fused, arith):
    cfg, spec, params, state, m = _model(no_layers)
    m.set_option("fused_blocks", fused)
    m.set_option("arith", arith)
    _, noisy = O.synthetic_batch(shape[0], shape[1], shape[2], seed=no_layers + 3)
    x = noisy.astype(np.float32)
    _check_f32(m(x), O.hydra_forward(spec, params, state, x.astype(np.float64)))


@pytest.mark.parametrize("nb", [1, 3])
@pytest.mark.parametrize("no_layers,shape", [(1, (1, 16, 16)), (3, (2, 30, 45)), (4, (1, 64, 96))])
def test_block_variants_match_oracle(nb, no_layers, shape):
    """block_kernels of length 1 and 3 (backbone_resnet.py:111-113 allows 1..3; backbone_blocks.py:174-214:
    first conv without BN, BN on the second and third, last activation = base_activation) -- inference only."""
    cfg = O.canonical_config(no_layers=no_layers)
    cfg["model"]["backbone"].update(block_kernels=[3] * nb, block_filters=[16] * nb)
    spec = O.ResnetSpec.from_config(cfg["model"])
    params, state = O.init_params(spec, seed=nb * 10 + no_layers, nontrivial_bn=True)
    m = bf.model_builder(cfg["model"], device="cuda").hydra
    assert [v.name for v in m.trainable_variables] == [t[0] for t in spec.tensors()]
    m.set_weights(params, state)
    _, noisy = O.synthetic_batch(shape[0], shape[1], shape[2], seed=nb + no_layers)
    x = noisy.astype(np.float32)
    _check_f32(m(x), O.hydra_forward(spec, params, state, x.astype(np.float64)))
    _check_u8(bf.DenoiserModule(m)(noisy), O.denoiser_module_call(spec, params, state, noisy))


@pytest.mark.parametrize("arith", [1, 0], ids=["f16x3", "f32"])
@pytest.mark.parametrize("hw", [(32, 32), (64, 64), (128, 128), (256, 256), (17, 23), (1, 1), (5, 130)])
def test_denoiser_module_u8_shapes_and_values(hw, arith):
    """tests/bfcnn/test_model_denoiser.py:61-70 (same shape, uint8) + value parity, incl. ragged
    sizes that need pad_to_power_of_2."""
    cfg, spec, params, state, m = _model(2, seed=7)
    m.set_option("arith", arith)
    _, noisy = O.synthetic_batch(1, hw[0], hw[1], seed=hw[0] * 7 + hw[1])
    got = bf.DenoiserModule(m)(noisy)
    assert got.shape == noisy.shape and got.dtype == np.uint8
    _check_u8(got, O.denoiser_module_call(spec, params, state, noisy))


@pytest.mark.parametrize("hw", [(64, 64), (256, 256), (17, 23), (1, 1), (40, 130)])
@pytest.mark.parametrize("no_layers", [1, 3])
def test_head_folded_into_the_last_block(hw, no_layers):
    """option fused_head = 1: the linear head runs in the epilogue of the last split-f16 block (u8 and f32 outputs, ragged
    sizes with the virtual power-of-two padding, crop)."""
    cfg, spec, params, state, m = _model(no_layers, seed=13)
    m.set_option("fused_head", 1)
    _, noisy = O.synthetic_batch(2, hw[0], hw[1], seed=hw[0] + hw[1])
    _check_u8(bf.DenoiserModule(m)(noisy), O.denoiser_module_call(spec, params, state, noisy))
    f = bf.DenoiserModule(m, cast_to_uint8=False)(noisy)
    _check_f32(f, O.denoiser_module_call(spec, params, state, noisy, cast_to_uint8=False))
    m.set_option("fused_head", 0)
    assert np.abs(bf.DenoiserModule(m)(noisy).astype(int) - O.denoiser_module_call(spec, params, state, noisy).astype(int)).max() <= 1


@pytest.mark.parametrize("hw", [(64, 64), (40, 256), (17, 23), (33, 144), (20, 150), (12, 200)])
@pytest.mark.parametrize("no_layers", [2, 3, 6])
@pytest.mark.parametrize("pair_head", [0, 1])
def test_two_blocks_per_launch(hw, no_layers, pair_head):
    """h3_pair (default on wherever the full-row streaming kernel runs; forced here with h3_variant = 4 at small sizes): consecutive
    residual blocks run two per launch (fused_block2_h3w_kernel), an odd count runs its single block first; with h3_pair_head the
    last pair launch also carries the linear head (u8 and f32 outputs, ragged sizes, virtual power-of-two padding, crop).  Same
    oracle, same bars as one block per launch, and the two schedules agree with each other to the last bit of the u8 output
    except where a value sits on a rounding boundary."""
    cfg, spec, params, state, m = _model(no_layers, seed=21)
    m.set_option("h3_variant", 4)
    m.set_option("h3_pair_head", pair_head)
    _, noisy = O.synthetic_batch(3, hw[0], hw[1], seed=hw[0] + 3 * hw[1])
    ref8 = O.denoiser_module_call(spec, params, state, noisy)
    got = bf.DenoiserModule(m)(noisy)
    _check_u8(got, ref8)
    _check_f32(bf.DenoiserModule(m, cast_to_uint8=False)(noisy), O.denoiser_module_call(spec, params, state, noisy, cast_to_uint8=False))
    m.set_option("h3_pair", 0)
    one = bf.DenoiserModule(m)(noisy)
    _check_u8(one, ref8)
    assert np.abs(one.astype(int) - got.astype(int)).max() <= 1
    if not pair_head:
        # same MFMA sequence per pixel, same hi / lo rounding between the blocks: with every block walking its rows in the same
        # direction (h3_zigzag 0) the float outputs are bit-for-bit those of the one-block kernel (the head inside the launch uses
        # exp / rcp where the head kernel uses tanhf: not compared bitwise)
        m.set_option("h3_zigzag", 0)
        single_f = np.asarray(bf.DenoiserModule(m, cast_to_uint8=False)(noisy))
        m.set_option("h3_pair", 1)
        pair_f = np.asarray(bf.DenoiserModule(m, cast_to_uint8=False)(noisy))
        m.set_option("h3_zigzag", 1)
        assert np.array_equal(pair_f, single_f)
    m.set_option("h3_pair", 1)
    m.set_option("h3_pair_head", 0)
    m.set_option("h3_variant", -1)


@pytest.mark.parametrize("no_layers", [2, 3])
def test_wide_images_run_two_blocks_per_launch_by_default(no_layers):
    """images wider than the 256 columns of the one-block streaming kernel: with enough rows the default selection still runs the
    blocks two per launch (128-column strips of fused_block2_h3w_kernel; an odd count runs its single block on the tile kernel).
    Same oracle and bars; the library reports the kernel it chose; h3_pair = 0 (tile kernel throughout) agrees to one grey level."""
    cfg, spec, params, state, m = _model(no_layers, seed=31)
    _, noisy = O.synthetic_batch(6, 352, 300, seed=12)                   # 6 x 352 rows x 3 strips = 6 336 strip rows
    mod = bf.DenoiserModule(m)
    got = mod(noisy)
    assert m.block_kernel() == ("fused_block2_h3w_kernel", 1 + (no_layers & 1))
    _check_u8(got[:1], O.denoiser_module_call(spec, params, state, noisy[:1]))
    _check_u8(got[5:], O.denoiser_module_call(spec, params, state, noisy[5:]))
    m.set_option("h3_pair", 0)
    try:
        tiles = mod(noisy)
        assert m.block_kernel()[0] == "fused_block_h3r_kernel"
    finally:
        m.set_option("h3_pair", 1)
    assert np.abs(tiles.astype(int) - got.astype(int)).max() <= 1
    _, few = O.synthetic_batch(1, 64, 300, seed=13)                      # too few rows: tiles
    mod(few)
    assert m.block_kernel()[0] == "fused_block_h3r_kernel"


@pytest.mark.parametrize("seed", range(max(4, int(os.environ.get("BF_SWEEP_N", 24)) // 4)))
def test_random_large_shapes_default_selection_matches_tiles_and_oracle(seed):
    """seeded draws of shapes that the default selection sends to two blocks per launch (>= 4 096 rows of 128-column strips: any width,
    ragged sizes, odd and even block counts): the kernel the library reports, the oracle on the first and last image, and the tile
    kernel (h3_pair = 0, h3_variant = 1) on the whole batch within one grey level."""
    rng = np.random.default_rng(5000 + seed)
    no_layers = int(rng.integers(2, 6))
    W = int(rng.choice([rng.integers(24, 129), rng.integers(129, 257), rng.integers(257, 900)]))
    H = int(rng.integers(24, 300))
    Hp, Wp = 1 << (H - 1).bit_length(), 1 << (W - 1).bit_length()           # the module pads to powers of two (virtually)
    nstrips = (Wp + 127) // 128
    B = -(-4096 // (Hp * nstrips)) + int(rng.integers(0, 3))
    if B * Hp * Wp > 6_000_000:
        pytest.skip("too large for a sweep draw")
    cfg, spec, params, state, m = _model(no_layers, seed=seed)
    _, noisy = O.synthetic_batch(B, H, W, seed=seed)
    mod = bf.DenoiserModule(m)
    got = mod(noisy)
    assert m.block_kernel()[0] == "fused_block2_h3w_kernel", (B, H, W, m.block_kernel())
    _check_u8(got[:1], O.denoiser_module_call(spec, params, state, noisy[:1]))
    _check_u8(got[-1:], O.denoiser_module_call(spec, params, state, noisy[-1:]))
    m.set_option("h3_pair", 0)
    m.set_option("h3_variant", 1)
    try:
        tiles = mod(noisy)
    finally:
        m.set_option("h3_pair", 1)
        m.set_option("h3_variant", -1)
    assert np.abs(tiles.astype(int) - got.astype(int)).max() <= 1


def test_two_blocks_per_launch_status_word_with_the_head_in_the_launch():
    """activations beyond the f16 range must still reach the status word when the head runs inside the last pair launch"""
    cfg, spec, params, state, m = _model(2, seed=5)
    big = params.copy()
    off = spec.offsets()
    for name in ("base/kernel", "block0/conv0/kernel"):
        o, shape = off[name]
        big[o:o + int(np.prod(shape))] *= 3000.0
    m.set_weights(big, state)
    m.set_option("h3_variant", 4)
    m.set_option("h3_pair_head", 1)
    m.auto_exact_fallback = False
    _, noisy = O.synthetic_batch(2, 64, 64, seed=3)
    with pytest.raises(FloatingPointError):
        bf.DenoiserModule(m)(noisy)


def test_denoiser_module_device_tensors_and_float_output():
    cfg, spec, params, state, m = _model(1, seed=9)
    _, noisy = O.synthetic_batch(2, 24, 40, seed=1)
    t = torch.from_numpy(noisy).cuda()
    out = bf.DenoiserModule(m)(t)
    assert out.is_cuda and out.dtype == torch.uint8 and tuple(out.shape) == noisy.shape
    f = bf.DenoiserModule(m, cast_to_uint8=False)(t)
    ref = O.denoiser_module_call(spec, params, state, noisy, cast_to_uint8=False)
    _check_f32(f.cpu().numpy(), ref)


@pytest.mark.parametrize("k", [1, 5, 7])
def test_base_kernel_sizes(k):
    cfg, spec, params, state, m = _model(1, seed=k, kernel_size=k)
    _, noisy = O.synthetic_batch(1, 20, 28, seed=k)
    _check_u8(bf.DenoiserModule(m)(noisy), O.denoiser_module_call(spec, params, state, noisy))


def test_strict_snapshot_graph_has_no_denormalise():
    """model.py:110-116: the literal single-output graph returns the raw +-0.51 tensor."""
    cfg, spec, params, state, m = _model(1, seed=4, strict_snapshot=True)
    _, noisy = O.synthetic_batch(1, 16, 16, seed=2)
    x = noisy.astype(np.float32)
    got = m(x)
    ref = O.hydra_forward(spec, params, state, x.astype(np.float64))
    assert np.abs(ref).max() <= 0.51 and np.abs(got - ref).max() < 1e-5


def test_kat_mid_grey_and_homogeneity_on_gpu():
    cfg, spec, params, state, m = _model(3, seed=1, nontrivial_bn=False)
    y = m(np.full((1, 32, 32, 3), 127.5, np.float32))
    assert np.array_equal(y, np.full_like(y, 127.5))


def test_clipping_of_out_of_range_float_input():
    """normalise clips to value_range (utilities.py:455-458)."""
    cfg, spec, params, state, m = _model(1, seed=3)
    x = np.random.default_rng(0).uniform(-100, 400, (1, 16, 16, 3)).astype(np.float32)
    _check_f32(m(x), O.hydra_forward(spec, params, state, x.astype(np.float64)))
    assert np.array_equal(m(x), m(np.clip(x, 0, 255)))


def test_batch_is_independent_and_deterministic():
    cfg, spec, params, state, m = _model(2, seed=5)
    _, noisy = O.synthetic_batch(5, 48, 48, seed=11)
    mod = bf.DenoiserModule(m)
    full = mod(noisy)
    assert np.array_equal(full, mod(noisy))
    for i in (0, 3):
        assert np.array_equal(full[i:i + 1], mod(noisy[i:i + 1]))


def test_full_size_config_properties():
    """BASELINE.json configs[2] at its real size (1x18, batch 128, 256x256: two blocks per launch, which the default
    picks from 4 096 rows of 128-column strips per forward on): size-independent properties instead of the (slow) oracle on the whole batch --
    the oracle on ONE image, position independence inside the batch (the same 16 images at eight places; the reversed batch),
    fused == unfused path within one LSB -- and the same properties at batch 8 (the smallest batch that still runs two blocks per launch)."""
    cfg, spec, params, state, m = _model(18, seed=42)
    _, noisy16 = O.synthetic_batch(16, 256, 256, seed=1234)
    mod = bf.DenoiserModule(m)
    want = O.denoiser_module_call(spec, params, state, noisy16[:1])
    for reps in (8, 0):                                   # batch 128, then batch 8
        noisy = np.concatenate([noisy16] * reps) if reps else noisy16[:8]
        out = mod(noisy)
        _check_u8(out[:1], want)
        assert out.min() >= 0 and out.max() <= 255
        assert np.array_equal(mod(noisy[::-1].copy())[::-1], out)                  # an image's result does not depend on its place
        if reps:
            for r in range(1, reps):
                assert np.array_equal(out[16 * r:16 * (r + 1)], out[:16])
        m.set_option("fused_blocks", 0)
        out2 = mod(noisy)
        m.set_option("fused_blocks", 1)
        assert np.abs(out.astype(int) - out2.astype(int)).max() <= 1
        # two blocks per launch (batch 128: the default) against one block per launch: the same MFMA sequence per pixel and the same
        # hi / lo rounding of the activation between the two blocks (in LDS there, in memory here) -- with every block walking its
        # rows in the same direction (h3_zigzag 0: a bottom-up walk sums the vertical taps in the other order) the outputs are IDENTICAL
        m.set_option("h3_zigzag", 0)
        out_p = mod(noisy)
        assert m.block_kernel()[0] == "fused_block2_h3w_kernel"            # both batches hold >= 4 096 rows of strips
        m.set_option("h3_pair", 0)
        m.set_option("h3_variant", 4)                                       # the one-block streaming kernel (batch 8: forced)
        out_s = mod(noisy)
        assert m.block_kernel()[0] == "fused_block_h3v_kernel"
        m.set_option("h3_variant", -1)
        out_t = mod(noisy)                                                  # one block per launch, the library's choice (batch 8: tiles)
        m.set_option("h3_pair", 1)
        m.set_option("h3_zigzag", 1)
        assert np.array_equal(out_p, out_s)
        assert np.abs(out_p.astype(int) - out.astype(int)).max() <= 1
        assert np.abs(out_t.astype(int) - out.astype(int)).max() <= 1


@pytest.mark.parametrize("arith", [1, 0], ids=["f16x3", "f32"])
def test_baseline_config2_shape_f32_hydra(arith):
    """BASELINE.json configs[1] at its real shape: resnet 1x6, batch 64, 256 x 256 x 3 float32 through bf_forward_f32 (the
    hydra-level entry: floats in value range in, denormalised floats out).  The oracle on one image at each end of the batch
    (normalised MAE <= 1e-4), and at the full size: finite, inside the value range, position independence (the reversed batch;
    the same 16 images at four places), the two block arithmetics within the parity bar of each other."""
    cfg, spec, params, state, m = _model(6, seed=42)
    m.set_option("arith", arith)
    _, noisy16 = O.synthetic_batch(16, 256, 256, seed=4321)
    x = np.concatenate([noisy16] * 4).astype(np.float32)
    assert x.shape == (64, 256, 256, 3)
    got = np.asarray(m(x))
    assert got.shape == x.shape and got.dtype == np.float32 and np.isfinite(got).all()
    assert got.min() >= spec.v_min and got.max() <= spec.v_max
    for i in (0, 63):
        ref = O.hydra_forward(spec, params, state, x[i:i + 1].astype(np.float64))
        _check_f32(got[i:i + 1], ref)
    assert np.array_equal(np.asarray(m(x[::-1].copy()))[::-1], got)
    for r in range(1, 4):
        assert np.array_equal(got[16 * r:16 * (r + 1)], got[:16])
    m.set_option("arith", 1 - arith)
    other = np.asarray(m(x))
    assert np.abs(other.astype(np.float64) - got).mean() / 255.0 <= 1e-4
    m.set_option("arith", 1)


@pytest.mark.parametrize("shape,no_layers", [((12, 256, 256), 18), ((16, 192, 200), 6), ((13, 240, 199), 3), ((3100, 1, 7), 2)],
                         ids=["full-width", "narrower", "odd-width", "one-row-images"])
def test_compact_activation_layout(shape, no_layers):
    """set_option("h3_compact", 1): between the launches of the full-row streaming kernel the lo planes travel as fp8 (e4m3 of
    lo * 2^12, 48 instead of 64 bytes per pixel; csrc/bf_common.h).  Same bars as the default layout against the oracle (north star:
    +-1 LSB, normalised MAE <= 1e-4; the emulation of tools/exp/emulate_f16x3.py predicts 6e-6), all columns / rows of ragged
    sizes, bottom-up bands on alternate blocks, values beyond the fp8 range of lo (|x| > 224) degrading to f16 precision, and the
    f16-range status still raised through the hi planes."""
    cfg, spec, params, state, m = _model(no_layers, seed=31)
    B, H, W = shape
    if H < 24:
        m.set_option("h3_variant", 4)          # short, narrow images run the tile kernel by default: the streaming kernel is asked for
    _, noisy = O.synthetic_batch(B, H, W, seed=77)
    x = noisy.astype(np.float32)
    ref = O.hydra_forward(spec, params, state, x[:2].astype(np.float64))
    plain = np.asarray(m(x), np.float64)
    m.set_option("h3_compact", 1)
    got = np.asarray(m(x), np.float64)
    assert got.shape == plain.shape and np.isfinite(got).all()
    assert not np.array_equal(got, plain)                                     # the other layout did run
    assert np.abs(got[:2] - ref).mean() / 255.0 <= 1e-4 and np.abs(got[:2] - ref).max() / 255.0 <= 2e-3
    assert np.abs(got - plain).mean() / 255.0 <= 5e-5
    u8 = bf.DenoiserModule(m)(noisy)
    _check_u8(u8[:2], O.denoiser_module_call(spec, params, state, noisy[:2]))
    assert np.array_equal(bf.DenoiserModule(m)(noisy[::-1].copy())[::-1], u8)  # deterministic, position independent
    m.set_option("h3_compact", 0)
    assert np.array_equal(np.asarray(m(x), np.float64), plain)


def test_compact_activation_layout_large_values_and_status():
    cfg, spec, params, state, m = _model(3, seed=9)
    big = params.copy()
    o, shape = spec.offsets()["base/kernel"]
    big[o:o + int(np.prod(shape))] *= 400.0                                    # activations in the hundreds: lo beyond the fp8 range
    m.set_weights(big, state)
    m.set_option("h3_compact", 1)
    _, noisy = O.synthetic_batch(16, 192, 64, seed=4)
    x = noisy.astype(np.float32)
    ref = O.hydra_forward(spec, big, state, x[:1].astype(np.float64))
    got = np.asarray(m(x), np.float64)
    assert np.isfinite(got).all() and np.abs(got[:1] - ref).mean() / 255.0 <= 1e-3
    for name in ("base/kernel", "block0/conv0/kernel"):
        o, shape = spec.offsets()[name]
        big[o:o + int(np.prod(shape))] *= 3000.0
    m.set_weights(big, state)
    m.auto_exact_fallback = False
    with pytest.raises(FloatingPointError):
        m(x)


def test_f16_range_guard():
    """split-f16 blocks need |activation| < 65504: weights that blow the activations up must be reported (status
    word -> FloatingPointError when host arrays are handed back), and the exact-fp32 kernels must still be right."""
    cfg, spec, params, state, m = _model(3, seed=9)
    big = params.copy()
    off = spec.offsets()
    for name in ("base/kernel", "block0/conv0/kernel"):
        o, shape = off[name]
        big[o:o + int(np.prod(shape))] *= 3000.0
    m.set_weights(big, state)
    _, noisy = O.synthetic_batch(1, 32, 32, seed=4)
    x = noisy.astype(np.float32)
    ref = O.hydra_forward(spec, big, state, x.astype(np.float64))
    assert np.isfinite(ref).all()
    m.auto_exact_fallback = False
    with pytest.raises(FloatingPointError):
        m(x)
    with pytest.raises(FloatingPointError):
        bf.DenoiserModule(m)(noisy)
    m.auto_exact_fallback = True                      # default: switch to the exact-fp32 kernels and repeat the forward
    got_auto = m(x)
    assert np.isfinite(got_auto).all() and np.abs(got_auto - ref).max() <= 0.6
    u8 = bf.DenoiserModule(m)(noisy)
    assert u8.dtype == np.uint8 and u8.shape == noisy.shape
    m.set_option("arith", 0)
    got = m(x)                                        # exact fp32: no range limit
    assert np.array_equal(got, got_auto)
    assert np.isfinite(got).all() and np.abs(got - ref).max() <= 0.6     # saturated tanh head: coarse bar
    m.set_option("arith", 1)
    m.set_weights(params, state)
    _check_f32(m(x), O.hydra_forward(spec, params, state, x.astype(np.float64)))    # status word is cleared by the next forward


def test_device_tensor_calls_in_a_pipelined_loop_do_not_lose_an_overflow():
    """device tensors in, device tensors out, nothing synchronises: the status word of EVERY call must be seen (the engine clears
    the device word at the start of each forward, and the host runs ahead of the GPU), also when a clean call follows the bad one."""
    cfg, spec, params, state, m = _model(3, seed=9)
    big = params.copy()
    off = spec.offsets()
    for name in ("base/kernel", "block0/conv0/kernel"):
        o, shape = off[name]
        big[o:o + int(np.prod(shape))] *= 3000.0
    m.auto_exact_fallback = False
    _, noisy = O.synthetic_batch(4, 64, 64, seed=4)
    t = torch.from_numpy(noisy).cuda()
    mod = bf.DenoiserModule(m)
    m.set_weights(big, state)
    mod(t)                                            # overflows
    m.set_weights(params, state)
    raised, out = 0, None
    for _ in range(24):                               # clean calls queue behind it (more than the status slots)
        try:
            out = mod(t)                              # a call looks at the words that have arrived by now: may raise here ...
        except FloatingPointError:
            raised += 1
    try:
        mod.check_status(wait=True)                   # ... or here, at the latest
    except FloatingPointError:
        raised += 1
    assert raised == 1                                # seen, and reported once
    assert mod.check_status(wait=True)
    ref = O.denoiser_module_call(spec, params, state, noisy)
    _check_u8(mod(t).cpu().numpy(), ref)


def test_errors_surface_as_python_exceptions():
    cfg, spec, params, state, m = _model(1)
    with pytest.raises(ValueError):
        m(np.zeros((1, 8, 8, 1), np.float32))
    with pytest.raises(ValueError):
        m(np.zeros((8, 8, 3), np.float32))
    from blind_image_denoising_amd import _native as N
    ws = torch.empty(16, dtype=torch.uint8, device="cuda")
    x = torch.zeros((1, 8, 8, 3), dtype=torch.float32, device="cuda")
    rc = N.lib().bf_forward_f32(m._h, N.ptr(m.packed()), N.ptr(x), N.ptr(x), 1, 8, 8, N.ptr(ws), 16, None)
    assert rc == N.BF_EWORKSPACE and "workspace too small" in N.last_error(m._h)


@pytest.mark.gpu
def test_model_on_a_device_that_is_not_the_current_one():
    """a model built with device='cuda:1' while cuda:0 is current must launch on GPU 1 (HIP launches go to the
    current device of the calling thread; _native._device_guard selects the device of the stream argument)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    import blind_image_denoising_amd as bf
    torch.cuda.set_device(0)
    cfg = O.canonical_config(no_layers=2)
    spec = O.ResnetSpec.from_config(cfg["model"])
    params, state = O.init_params(spec, seed=3, nontrivial_bn=True)
    model = bf.model_builder(cfg["model"], device="cuda:1").hydra
    model.set_weights(params, state)
    _, noisy = O.synthetic_batch(2, 32, 32, seed=5)
    out = bf.DenoiserModule(model)(torch.from_numpy(noisy).to("cuda:1"))
    assert out.device.index == 1 and torch.cuda.current_device() == 0
    ref = O.denoiser_module_call(spec, params, state, noisy)
    assert np.abs(out.cpu().numpy().astype(np.int32) - ref.astype(np.int32)).max() <= 1


def test_standalone_normalize_and_denormalize_layers():
    """build_normalize_model / build_denormalize_model (bfcnn/model.py:364-430) through bf_op_normalize"""
    x = np.random.default_rng(0).uniform(-20, 280, (2, 5, 7, 3)).astype(np.float32)
    n = bf.model.build_normalize_model(None, 0.0, 255.0)(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.array_equal(n, (np.clip(x, 0, 255) - 0.0) / np.float32(255.0) - np.float32(0.5))
    y = np.random.default_rng(1).uniform(-0.7, 0.7, (2, 5, 7, 3)).astype(np.float32)
    d = bf.model.build_denormalize_model(None, 0.0, 255.0)(torch.from_numpy(y).cuda()).cpu().numpy()
    assert np.abs(d - ((np.clip(y, -0.5, 0.5) + 0.5) * 255.0)).max() <= 2e-5
    with pytest.raises(RuntimeError, match="GPU"):
        bf.model.build_normalize_model()(torch.zeros(1))


@pytest.mark.parametrize("seed", range(int(os.environ.get("BF_SWEEP_N", 40))))     # BF_SWEEP_N=300: a longer hunt
def test_random_engine_configurations_and_options_match_oracle(seed):
    """a seeded sweep over the 16-filter engine's configuration and option space: depth, base kernel size, 1 / 2 / 3 convolutions per
    block, activations, BatchNorm on / off, head activation, ragged image sizes and batches on both sides of the row count where the
    default moves from the tile kernel to the full-row streaming kernel, every set_option switch of the inference path (arithmetic,
    fused blocks, kernel variant, band order, head folding, compact layout).  Whatever combination is asked for must match the oracle."""
    rng = np.random.default_rng(9000 + seed)
    nb = int(rng.choice([1, 2, 2, 2, 3]))
    cfg = O.canonical_config(no_layers=int(rng.integers(1, 7)), kernel_size=int(rng.choice([1, 3, 5, 7])))
    cfg["model"]["backbone"].update(block_kernels=[3] * nb, block_filters=[16] * nb, activation=str(rng.choice(["relu", "relu", "linear"])),
                                    use_bn=bool(rng.random() < 0.8))
    cfg["model"]["denoiser"]["activation"] = str(rng.choice(["linear", "relu", "leaky_relu"]))
    spec = O.ResnetSpec.from_config(cfg["model"])
    params, state = O.init_params(spec, seed=seed, nontrivial_bn=True)
    try:
        m = bf.model_builder(cfg["model"], device="cuda").hydra
    except NotImplementedError as e:
        pytest.skip(f"refused: {e}")
    m.set_weights(params, state)
    big = rng.random() < 0.5                                   # big: >= 3072 image rows in the batch
    H, W = (int(rng.integers(96, 200)), int(rng.integers(8, 257))) if big else (int(rng.integers(1, 70)), int(rng.integers(1, 300)))
    B = -(-3072 // H) + int(rng.integers(0, 3)) if big else int(rng.integers(1, 4))
    opts = {"arith": int(rng.random() < 0.75), "fused_blocks": int(rng.random() < 0.85), "h3_zigzag": int(rng.integers(2)),
            "fused_head": int(rng.random() < 0.3), "h3_compact": int(rng.random() < 0.4),
            "h3_variant": int(rng.choice([-1, -1, 1, 4]))}
    for k, v in opts.items():
        m.set_option(k, v)
    _, noisy = O.synthetic_batch(B, H, W, seed=seed)
    x = noisy.astype(np.float32)
    _check_f32(np.asarray(m(x))[:2], O.hydra_forward(spec, params, state, x[:2].astype(np.float64)))
    _check_u8(bf.DenoiserModule(m)(noisy)[:2], O.denoiser_module_call(spec, params, state, noisy[:2]))


@pytest.mark.parametrize("hw", [(64, 64), (17, 23), (40, 300), (1, 1), (33, 257), (20, 516), (5, 4)])
def test_row_streaming_base_convolution_matches_the_vector_kernel_and_the_oracle(hw):
    """base_conv_rows_kernel (u8 in, 3x3x3 -> 16 on the f16 matrix cores: exact u8 operands, the normalisation offset and the zero /
    power-of-two padding carried by a fourth 'inside' channel) against the vector kernel (option base_rows = 0) and the oracle,
    ragged sizes with the virtual padding on, more than one 256-column chunk."""
    cfg, spec, params, state, m = _model(2, seed=17)
    m.set_option("base_rows", 2)               # the row kernel wherever it can run (by default only from ~8 192 rows of chunks on)
    _, noisy = O.synthetic_batch(2, hw[0], hw[1], seed=11 + hw[0])
    ref = O.denoiser_module_call(spec, params, state, noisy)
    got = bf.DenoiserModule(m)(noisy)
    _check_u8(got, ref)
    f = bf.DenoiserModule(m, cast_to_uint8=False)(noisy)
    _check_f32(f, O.denoiser_module_call(spec, params, state, noisy, cast_to_uint8=False))
    m.set_option("base_rows", 0)
    try:
        other = bf.DenoiserModule(m)(noisy)
        f0 = bf.DenoiserModule(m, cast_to_uint8=False)(noisy)
    finally:
        m.set_option("base_rows", 1)
    assert np.abs(other.astype(int) - got.astype(int)).max() <= 1
    assert np.abs(np.asarray(f0, np.float64) - np.asarray(f, np.float64)).mean() / 255.0 <= 2e-6


def test_library_reports_the_block_kernel_it_launched():
    """bf_get_block_kernel: bench.py keys its roofline and the committed counter traffic on this name, so it must follow the
    variant selection (batch size, options), not be derived from options by the caller."""
    cfg, spec, params, state, m = _model(4, seed=3)
    _, small = O.synthetic_batch(2, 32, 32, seed=1)
    _, big = O.synthetic_batch(16, 256, 64, seed=2)             # 4 096 rows of one strip: the streaming kernels
    mod = bf.DenoiserModule(m)
    mod(small)
    assert m.block_kernel() == ("fused_block_h3r_kernel", 4)
    mod(big)
    assert m.block_kernel() == ("fused_block2_h3w_kernel", 2)
    m.set_option("h3_pair", 0)
    mod(big)
    assert m.block_kernel() == ("fused_block_h3v_kernel", 4)
    m.set_option("h3_pair", 1)
    m.set_option("arith", 0)
    mod(big)
    assert m.block_kernel() == ("fused_block_v4_kernel", 4)
    m.set_option("arith", 1)
    cfg3, spec3, params3, state3, m3 = _model(3, seed=3)       # odd count: the single block first, then a pair
    bf.DenoiserModule(m3)(big)
    assert m3.block_kernel() == ("fused_block2_h3w_kernel", 2)
