"""Generates tests/golden/unet_v56.npz from DATA files of the reference (run in the build container, where
/root/reference exists; the GPU box only sees the .npz):
  * the trained tensors of bfcnn/pretrained/unet_laplacian_v5.6/model_hydra.keras, flattened in the inventory order of
    oracle.unet_oracle.UnetLaplacianSpec, plus the model config derived from the archive's config.json;
  * two 256x256 crops of the KITTI frames the reference's own test_pretrained.py denoises
    (images/test/kitti/files, tests/bfcnn/constants.py:11-17);
  * known-answer constants read out of bfcnn/pretrained/unet_laplacian_v5.6/denoiser_model.tflite, the same network
    exported by the reference's TFLite converter: the GaussianFilter taps of both Laplacian levels and the per-channel
    int8 quantisation scales of the conv_3 kernels (= max|conv_3 * multiplier| / 127, which pins the function the
    ChannelLearnableMultiplier applies) -- tests/test_unet_pretrained.py checks the oracle against them.
usage: python tests/golden/make_unet_v56_fixture.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def tflite_tensors(path):
    """{tensor name: (float constant or None, per-channel scales or None)} through the minimal flatbuffer reader."""
    sys.path.insert(0, os.path.join(ROOT, "tools", "exp"))
    from tflite_graph import FB
    fb = FB(open(path, "rb").read())
    model = fb.root()
    buffers = fb.tables(model, 4)
    sg = fb.tables(model, 2)[0]
    out = {}
    for t in fb.tables(sg, 0):
        name, ttype, shape = fb.string(t, 3), fb.scalar(t, 1, "b"), fb.ints(t, 0)
        n, base = fb.vector(buffers[fb.scalar(t, 2, "I")], 0)
        const = None
        if ttype == 0 and n:                                   # FLOAT32 with data
            const = np.frombuffer(fb.b, np.float32, n // 4, base).reshape(shape).copy()
        scales = None
        q = fb.field(t, 4)
        if q is not None:
            ns, sb = fb.vector(fb.indirect(q), 2)
            if ns:
                scales = np.frombuffer(fb.b, np.float32, ns, sb).copy()
        out[name] = (const, scales)
    return out


def main():
    from PIL import Image
    from blind_image_denoising_amd import keras_import
    from oracle import unet_oracle as U
    arch = os.path.join(REF, "bfcnn/pretrained/unet_laplacian_v5.6")
    config, h5 = keras_import.read_archive(os.path.join(arch, "model_hydra.keras"))
    spec = U.UnetLaplacianSpec.from_config(config)
    params = keras_import.params_from_archive(config, h5, spec.tensors())
    kitti = os.path.join(REF, "images/test/kitti/files")
    crops = []
    for name, (y, x) in (("kitti_0000000000.png", (100, 400)), ("kitti_0000000017.png", (110, 700))):
        im = np.asarray(Image.open(os.path.join(kitti, name)).convert("RGB"))
        crops.append(im[y:y + 256, x:x + 256])
    tfl = tflite_tensors(os.path.join(arch, "denoiser_model.tflite"))
    pick = lambda frag: next(v for k, v in tfl.items() if k.split(";")[0].endswith(frag))
    kat = {"gauss0": pick("unet_laplacian/gaussian_filter/depthwise")[0],
           "gauss1": pick("unet_laplacian/gaussian_filter_1/depthwise")[0]}
    for blk, ours in (("encoder_0_0", "enc0_0"), ("encoder_1_1", "enc1_1"), ("decoder_0_2", "dec0_2"), ("decoder_1_0", "dec1_0")):
        kat[f"conv3_scales/{ours}"] = pick(f"unet_laplacian/{blk}/conv2d_1/Conv2D")[1]
    for n, frag in (("key", "conv2d"), ("query", "conv2d_1"), ("value", "conv2d_2")):
        kat[f"attn_scales/{n}"] = pick(f"unet_laplacian/convolutional_self_attention/{frag}/Conv2D")[1]
    assert all(v is not None for v in kat.values()), [k for k, v in kat.items() if v is None]
    out = os.path.join(ROOT, "tests", "golden", "unet_v56.npz")
    np.savez_compressed(out, params=params, config=np.frombuffer(json.dumps(config).encode(), np.uint8),
                        kitti=np.stack(crops).astype(np.uint8), **{"kat/" + k: v for k, v in kat.items()})
    print(out, os.path.getsize(out), "bytes;", params.size, "parameters")


if __name__ == "__main__":
    main()
