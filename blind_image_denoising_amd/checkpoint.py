"""
Checkpoint / resume of a training run: the state `tf.train.Checkpoint(step, epoch, optimizer, model)` +
`tf.train.CheckpointManager(max_to_keep)` carry in the reference (bfcnn/train_loop.py:121-181, utilities.py:691-706) --
weights, BatchNorm moving statistics, the Adam slots m / v, the optimizer's iteration count, step and epoch -- in ONE
blob per checkpoint (`ckpt-<step>.npz`, this package's own format; TF's tensor-bundle format is out of scope).
"""
import glob
import json
import os
import re
from typing import Optional

import numpy as np
import torch

from .custom_logger import logger

_CKPT_RE = re.compile(r"ckpt-(\d+)\.npz$")


class Checkpoint:
    """ckpt.step / ckpt.epoch / ckpt.optimizer / ckpt.model as in create_checkpoint (utilities.py:691-706)."""

    def __init__(self, model, optimizer, step: int = 0, epoch: int = 0):
        self.model, self.optimizer, self.step, self.epoch = model, optimizer, int(step), int(epoch)

    # ---- one blob ------------------------------------------------------------------------------
    def state_dict(self):
        w = self.model.get_weights()
        params, state = w if isinstance(w, tuple) else (w, np.zeros(0, np.float32))
        opt = self.optimizer
        d = {"params": np.asarray(params, np.float32), "state": np.asarray(state, np.float32),
             "step": np.int64(self.step), "epoch": np.int64(self.epoch),
             "iterations": np.int64(getattr(opt, "iterations", 0))}
        if getattr(opt, "m", None) is not None:
            d["adam_m"] = opt.m.detach().cpu().numpy()
            d["adam_v"] = opt.v.detach().cpu().numpy()
        return d

    def write(self, path: str) -> str:
        tmp = path + ".tmp.npz"
        np.savez(tmp, **self.state_dict())
        os.replace(tmp, path)                       # a crash mid-write never leaves a half checkpoint under the final name
        return path

    def restore(self, path: str) -> "Checkpoint":
        z = np.load(path)
        n = int(np.asarray(self.model.get_weights()[0] if isinstance(self.model.get_weights(), tuple) else self.model.get_weights()).size)
        if int(z["params"].size) != n:
            raise ValueError(f"checkpoint [{path}] holds {int(z['params'].size)} parameters, the model has {n}")
        if z["state"].size:
            self.model.set_weights(z["params"], z["state"])
        else:
            self.model.set_weights(z["params"])
        self.step, self.epoch = int(z["step"]), int(z["epoch"])
        opt = self.optimizer
        if opt is not None:
            opt.iterations = int(z["iterations"])   # "restore learning rate" (train_loop.py:178-179)
            if "adam_m" in z.files:
                opt._slots(self.model)
                opt.m.copy_(torch.from_numpy(z["adam_m"]).to(opt.m.device))
                opt.v.copy_(torch.from_numpy(z["adam_v"]).to(opt.v.device))
        return self


class CheckpointManager:
    """tf.train.CheckpointManager(checkpoint, directory, max_to_keep) (train_loop.py:158-163): numbered blobs, the oldest
    are deleted, `latest_checkpoint` is what a restarted run resumes from."""

    def __init__(self, checkpoint: Checkpoint, directory: str, max_to_keep: int = 3, checkpoint_name: str = "ckpt"):
        if checkpoint_name != "ckpt":
            raise ValueError("checkpoint_name is fixed to 'ckpt'")
        self.checkpoint, self.directory, self.max_to_keep = checkpoint, directory, int(max_to_keep)
        os.makedirs(directory, exist_ok=True)

    def _all(self):
        found = []
        for p in glob.glob(os.path.join(self.directory, "ckpt-*.npz")):
            m = _CKPT_RE.search(p)
            if m:
                found.append((int(m.group(1)), p))
        return [p for _, p in sorted(found)]

    @property
    def latest_checkpoint(self) -> Optional[str]:
        a = self._all()
        return a[-1] if a else None

    def save(self) -> str:
        path = self.checkpoint.write(os.path.join(self.directory, f"ckpt-{self.checkpoint.step}.npz"))
        with open(os.path.join(self.directory, "checkpoint.json"), "w") as f:
            json.dump({"latest": os.path.basename(path), "step": self.checkpoint.step, "epoch": self.checkpoint.epoch}, f)
        if self.max_to_keep > 0:
            for old in self._all()[:-self.max_to_keep]:
                os.remove(old)
        logger.info(f"saved checkpoint to [{path}]")
        return path

    def restore_latest(self) -> bool:
        latest = self.latest_checkpoint
        if not latest:
            logger.info("!!! Did NOT find checkpoint to restore !!!")
            return False
        logger.info("!!! Found checkpoint to restore !!!")
        self.checkpoint.restore(latest)
        logger.info(f"restored checkpoint at epoch [{self.checkpoint.epoch}] and step [{self.checkpoint.step}]")
        return True
