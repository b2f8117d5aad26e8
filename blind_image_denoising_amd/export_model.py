"""`export_model` (bfcnn/export_model.py:19-190): build the model of a pipeline configuration, restore the latest checkpoint of a
training run, write `pipeline.json` and the model into `output_directory`, optionally run it once.

What it writes is this engine's model directory (`pipeline.json` + `weights.npz`, `save_model`), the directory `load_model` /
`load_denoiser_model` read back as the uint8 -> uint8 denoiser module; TensorFlow's SavedModel and TFLite serialisations of the
reference (`:106-190`) are out of scope -- `to_tflite` is accepted and logged."""
import os
from pathlib import Path
from typing import Dict, Union

import numpy as np

from .constants import *
from .custom_logger import logger
from .utilities import load_config


def export_model(pipeline_config_path: Union[str, Dict, Path], checkpoint_directory: Union[str, Path],
                 output_directory: Union[str, Path], to_tflite: bool = True, test_model: bool = True, device=None):
    from .checkpoint import Checkpoint, CheckpointManager
    from .model import model_builder, save_model
    from .module_denoiser import DenoiserModule
    from .optimizer import optimizer_builder
    if checkpoint_directory is None or not os.path.isdir(str(checkpoint_directory)):
        raise ValueError("Checkpoint directory [{0}] is not valid".format(checkpoint_directory))        # export_model.py:35-38
    if not os.path.isdir(output_directory):
        Path(output_directory).mkdir(parents=True, exist_ok=True)
        if not os.path.isdir(output_directory):
            raise ValueError("Output directory [{0}] is not valid".format(output_directory))
    config = load_config(pipeline_config_path)
    output_directory = str(output_directory)
    logger.info("building model")
    hydra = model_builder(config[MODEL_STR], device=device).hydra
    optimizer = optimizer_builder(config["train"]["optimizer"])[0] if "train" in config and "optimizer" in config["train"] else None
    ckpt = Checkpoint(model=hydra, optimizer=optimizer)
    manager = CheckpointManager(checkpoint=ckpt, directory=str(checkpoint_directory))
    logger.info(f"restoring checkpoint weights from [{checkpoint_directory}]")
    if not manager.restore_latest():
        raise ValueError("Checkpoint directory [{0}] holds no checkpoint".format(checkpoint_directory))
    logger.info(f"restored checkpoint at epoch [{int(ckpt.epoch)}] and step [{int(ckpt.step)}]")
    logger.info("saving configuration pipeline and model")
    save_model(hydra, output_directory, pipeline_config=config)
    if to_tflite:
        logger.info("to_tflite: TensorFlow Lite serialisation is outside this engine; the model directory is what load_model reads")
    module = DenoiserModule(hydra, cast_to_uint8=True)
    if test_model:                                                     # export_model.py:150-190: one call on a random uint8 image
        channels = int(config[MODEL_STR][BACKBONE_STR][INPUT_SHAPE_STR][-1])
        x = np.random.default_rng(0).integers(0, 256, (1, 256, 256, channels), dtype=np.uint8)
        y = module(x)
        if y.shape != x.shape or y.dtype != np.uint8:
            raise ValueError(f"exported module returned {y.dtype} {y.shape} for uint8 {x.shape}")
        logger.info("model test: uint8 {0} -> uint8 {1}".format(x.shape, y.shape))
    return module
