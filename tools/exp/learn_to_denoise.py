"""Does the engine learn?  Trains the canonical resnet 1x6 from scratch on noisy crops of tests/golden/lena.jpg through the public API
(dataset_builder from an image directory -> train_loop) and measures PSNR of a held-out crop before / after denoising."""
import pathlib, sys, tempfile, time
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import blind_image_denoising_amd as bf
from oracle import bfcnn_oracle as O
from PIL import Image


def psnr(a, b):
    return 10 * np.log10(255.0 ** 2 / np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))


def main(layers=6, epochs=30, batch=16, crop=64):
    lena = Image.open(pathlib.Path(__file__).resolve().parents[2] / "tests" / "golden" / "lena.jpg")
    tmp = pathlib.Path(tempfile.mkdtemp())
    (tmp / "img").mkdir()
    lena.crop((0, 0, 512, 384)).save(tmp / "img" / "train.png")                    # the top three quarters train, the rest is held out
    held = np.asarray(lena.convert("RGB"))[384:512, 0:512][None]
    cfg = O.canonical_config(no_layers=layers)
    cfg["train"].update({"epochs": epochs, "gpu_batches_per_step": 1})
    cfg["train"]["optimizer"]["schedule"]["config"]["learning_rate"] = 2e-3
    cfg["loss"] = {"hinge": 0.0, "cutoff": 255.0, "mae_multiplier": 1.0, "ssim_multiplier": 0.0, "regularization": 0.01}
    cfg["dataset"] = {"batch_size": batch, "color_mode": "rgb", "no_crops_per_image": 64 * batch, "value_range": [0, 255], "clip_value": True,
                      "round_values": True, "random_up_down": True, "random_left_right": True, "input_shape": [crop, crop, 3],
                      "additional_noise": [20, 20.0001], "inputs": [{"directory": str(tmp / "img")}]}
    t0 = time.time()
    model, hist = bf.train_loop(cfg, str(tmp / "run"))
    dt = time.time() - t0
    rng = np.random.default_rng(0)
    noisy = np.clip(np.round(held + rng.normal(0, 20.0, held.shape)), 0, 255).astype(np.uint8)
    den = bf.load_model(str(tmp / "run" / "final"))(noisy)
    print(f"steps {len(hist)}  {dt:.1f} s  loss {hist[0]:.2f} -> {np.mean(hist[-20:]):.2f}   PSNR noisy {psnr(held, noisy):.2f} dB -> denoised {psnr(held, den):.2f} dB")
    return psnr(held, noisy), psnr(held, den)


if __name__ == "__main__":
    main()
