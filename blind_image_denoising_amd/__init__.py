"""
blind_image_denoising_amd -- MI355X-native engine for the bfcnn resnet-denoiser hot path.

Mirrors the public surface of the reference package (bfcnn/__init__.py:127-141) for that path:
`load_model`, `load_denoiser_model`, `models`, `configs`, `model_builder`, `DenoiserModule`,
`loss_function_builder`, `optimizer_builder`, `schedule_builder`, `train_loop`,
`build_pyramid_model`, `build_inverse_pyramid_model`.  All arithmetic runs in hand-written
gfx950 HIP kernels behind the C ABI of include/bfcnn_hip.h (lib/libbfcnn_hip.so); torch is
used only for device memory, streams and torch.distributed.  There is no CPU fallback.
"""
import os
import pathlib

__version__ = "0.1.0"

from .constants import *
from .custom_logger import logger
from .utilities import load_config, save_config, input_shape_fixer
from .model import (BuilderResults, HydraModel, model_builder, describe_resnet, save_model, load_hydra,
                    build_normalize_model, build_denormalize_model)
from .module_denoiser import DenoiserModule, GraphedDenoiserModule
from .loss import loss_function_builder
from .optimizer import optimizer_builder, schedule_builder, deep_supervision_schedule_builder
from .train_loop import (train_loop, build_train_functions, DataParallelTrainer, NativeCommunicator, shard_batch,
                         allreduce_gradients)
from .checkpoint import Checkpoint, CheckpointManager
from .resnet_generic import squeeze_and_excite_block, selector_block
from .pyramid import (build_pyramid_model, build_inverse_pyramid_model, multiscales_generator_fn)
from .dataset import dataset_builder, PrepareData, noise_augment
from .export_model import export_model
from .file_operations import load_image
from . import regularizers
from .custom_layers import RandomOnOff, Multiplier, ChannelwiseMultiplier

current_dir = pathlib.Path(__file__).parent.resolve()

# ---- configs (bfcnn/__init__.py:29-44): every *.json in configs/ ------------------------------
configs_dir = current_dir / "configs"
configs = []
if configs_dir.is_dir():
    for _f in sorted(configs_dir.glob("*.json")):
        try:
            configs.append((os.path.basename(str(_f)), load_config(str(_f))))
        except Exception as _e:     # a broken config must not break import
            logger.error(f"failed to load config [{_f}]: {_e}")

CONFIGS_DICT = {os.path.splitext(name)[0]: cfg for name, cfg in configs}          # bfcnn/__init__.py:41-44

# ---- pretrained registry (bfcnn/__init__.py:48-75) --------------------------------------------
pretrained_dir = current_dir / "pretrained"
models = {}


def _make_loader(directory):
    # bound per directory (the reference closure late-binds its loop variable, __init__.py:58-64)
    def load_denoiser_module():
        return DenoiserModule(load_hydra(str(directory)))
    return load_denoiser_module


if pretrained_dir.is_dir():
    for _d in [d for d in sorted(pretrained_dir.iterdir()) if d.is_dir()]:
        models[str(_d.name)] = {
            "directory": _d,
            DENOISER_STR: _make_loader(_d),
            "configuration": str(_d / PIPELINE_FILE_STR),
            "saved_model_path": str(_d),
        }


def load_model(model_path: str, device=None) -> DenoiserModule:
    """bfcnn/__init__.py:81-97: a registry name, a model directory (pipeline.json + weights.npz written by
    `save_model`), or a `.keras` archive of a trained unet_laplacian hydra / the directory holding `model_hydra.keras`
    (the layout of the reference's bfcnn/pretrained/<name>/).  Returns a callable uint8 [B,H,W,C] -> uint8 [B,H,W,C]."""
    # --- argument checking
    if model_path is None or len(model_path) <= 0:
        raise ValueError("model_path cannot be empty")
    # --- load from pretrained
    if model_path in models:
        return DenoiserModule(load_hydra(models[model_path]["saved_model_path"], device=device))
    # --- load from any directory
    if not os.path.exists(model_path):
        raise ValueError("model_path [{0}] does not exist".format(model_path))
    # --- a trained hydra as keras wrote it (bfcnn/export_model.py:106-110), or the directory holding one
    archive = os.path.join(model_path, "model_hydra.keras") if os.path.isdir(model_path) else str(model_path)
    if archive.endswith(".keras") and os.path.isfile(archive):
        from .unet_laplacian import UnetLaplacianHydra
        return DenoiserModule(UnetLaplacianHydra.from_keras_archive(archive, device=device))
    return DenoiserModule(load_hydra(str(model_path), device=device))


def load_denoiser_model(model_path: str):
    """bfcnn/__init__.py:103-112."""
    if model_path is None or len(model_path) <= 0:
        raise ValueError("model_path cannot be empty")
    if model_path in models:
        return models[model_path][DENOISER_STR]()
    raise ValueError("model_path [{0}] does not exist".format(model_path))


# offer a decent pretrained model (bfcnn/__init__.py:118-122)
load_default_denoiser = list(models.values())[0][DENOISER_STR] if len(models) > 0 else None
