"""Host-side helpers mirroring bfcnn/utilities.py (configuration handling only; every tensor
op of that file that is on the hot path lives in csrc/ as a HIP kernel)."""
import json
import os
from pathlib import Path
from typing import Dict, Iterable, Union

from .custom_logger import logger


def load_config(config: Union[str, Dict, Path]) -> Dict:
    """bfcnn/utilities.py:59-83: dict passthrough or JSON file; ValueError on anything else."""
    try:
        if config is None:
            raise ValueError("config should not be empty")
        if isinstance(config, dict):
            return config
        if isinstance(config, (str, Path)):
            if not os.path.isfile(str(config)):
                raise ValueError("configuration path [{0}] is not valid".format(str(config)))
            with open(str(config), "r") as f:
                return json.load(f)
        raise ValueError("don't know how to handle config [{0}]".format(config))
    except Exception as e:
        logger.error(e)
        raise ValueError(f"failed to load [{config}]")


def save_config(config: Union[str, Dict, Path], filename: Union[str, Path]) -> None:
    """bfcnn/utilities.py:708-731."""
    config = load_config(config)
    with open(str(filename), "w") as f:
        json.dump(obj=config, fp=f, indent=4)


def input_shape_fixer(input_shape: Iterable):
    """bfcnn/utilities.py:89-96: "?", "" and "-1" mean None."""
    input_shape = list(input_shape)
    for i, shape in enumerate(input_shape):
        if shape == "?" or shape == "" or shape == "-1":
            input_shape[i] = None
    return input_shape


def next_power_of_2(n: int) -> int:
    """target size of pad_to_power_of_2 (bfcnn/utilities.py:736-751)."""
    p = 1
    while p < n:
        p <<= 1
    return p
