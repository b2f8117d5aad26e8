"""World-size-2 `gloo` tests (CPU) of the N>1 path: batch sharding, the ONE gradient all-reduce and
the replica-consistency logic of train_loop.DataParallelTrainer.  The per-rank compute is stood in
by the fp64 oracle (the HIP engine needs a GPU); the communication code under test is the product's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import bfcnn_oracle as O
from blind_image_denoising_amd.train_loop import allreduce_gradients, shard_batch


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = O.canonical_config(no_layers=1)
        spec, ls = O.ResnetSpec.from_config(cfg["model"]), O.LossSpec.from_config(cfg["loss"])
        params, state = O.init_params(spec, seed=42)
        clean, noisy = O.synthetic_batch(4, 16, 16, seed=5)           # the GLOBAL batch, same on every rank
        gt = shard_batch(torch.from_numpy(clean.astype(np.float64)), rank, world).numpy()
        x = shard_batch(torch.from_numpy(noisy.astype(np.float64)), rank, world).numpy()
        assert gt.shape[0] == 4 // world
        _, _, _, _, grads, _ = O.train_step_single_gpu(spec, ls, params, state, gt, x)
        g = torch.from_numpy(grads.copy())
        work = allreduce_gradients(g, async_op=True)                  # the one collective of a step
        assert work is not None
        work.wait()
        # identical update on every rank: Adam on the averaged gradient (grad_scale = 1/world)
        p1, _, _ = O.adam_step(params.astype(np.float64), g.numpy() / world, np.zeros(g.numel()), np.zeros(g.numel()),
                               0, 1e-3, global_clipnorm=1.0)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), local=grads, reduced=g.numpy(), params=p1)
        # broadcast_parameters semantics: rank 0's buffers win
        t = torch.full((8,), float(rank))
        dist.broadcast(t, src=0)
        assert torch.all(t == 0)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_allreduce_and_identical_replicas(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    assert not np.allclose(r[0]["local"], r[1]["local"])                      # shards really differ
    assert np.allclose(r[0]["reduced"], r[0]["local"] + r[1]["local"], rtol=0, atol=1e-15)
    assert np.array_equal(r[0]["reduced"], r[1]["reduced"])                   # same sum everywhere
    assert np.array_equal(r[0]["params"], r[1]["params"])                     # replicas stay identical


def test_shard_batch_partitions_the_batch():
    b = torch.arange(24).reshape(6, 4)
    parts = [shard_batch(b, r, 3) for r in range(3)]
    assert torch.equal(torch.cat(parts), b) and all(p.shape[0] == 2 for p in parts)
    with pytest.raises(ValueError, match="not divisible"):
        shard_batch(b, 0, 4)


def test_allreduce_is_a_noop_without_a_process_group():
    g = torch.ones(5)
    assert allreduce_gradients(g) is None and torch.all(g == 1)


# ---- DataParallelTrainer itself with world > 1 ---------------------------------------------------------------------
class _OracleModel:
    """the surface DataParallelTrainer touches (params / state / mark_dirty / device), weights on the CPU"""
    def __init__(self, params, state):
        self.params, self.state = torch.from_numpy(params.copy()), torch.from_numpy(state.copy())
        self.device, self.n_params = torch.device("cpu"), params.size
        self.dirty = 0

    def mark_dirty(self):
        self.dirty += 1


def _trainer_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from blind_image_denoising_amd.train_loop import DataParallelTrainer, TrainFunctions
        cfg = O.canonical_config(no_layers=1)
        spec, ls = O.ResnetSpec.from_config(cfg["model"]), O.LossSpec.from_config(cfg["loss"])
        params, state = O.init_params(spec, seed=40 + rank)                  # replicas start DIFFERENT: broadcast must fix it
        model = _OracleModel(params.astype(np.float64), state.astype(np.float64))
        trainer = DataParallelTrainer.__new__(DataParallelTrainer)
        trainer.model, trainer.optimizer, trainer.group, trainer.world_size, trainer.comm = model, None, None, world, None
        log = []

        def train_step_single_gpu(gt, x, dw, pct, tv):                       # per-rank compute: the fp64 oracle
            total, ml, dl, pred, grads, new_state = O.train_step_single_gpu(spec, ls, model.params.numpy(), model.state.numpy(),
                                                                            gt.numpy(), x.numpy())
            model.state = torch.from_numpy(new_state)
            log.append("compute")
            return torch.tensor(total), ml, [dl], torch.from_numpy(pred), torch.from_numpy(grads.copy())

        slots = {"m": np.zeros(params.size), "v": np.zeros(params.size), "it": 0}

        def apply_grads(opt, grads, tv=None, grad_scale=1.0):
            log.append("apply")
            p1, slots["m"], slots["v"] = O.adam_step(model.params.numpy(), grads.numpy() * grad_scale, slots["m"], slots["v"],
                                                     slots["it"], 1e-3, global_clipnorm=1.0)
            slots["it"] += 1
            model.params = torch.from_numpy(p1)

        trainer.fns = TrainFunctions(None, None, train_step_single_gpu, apply_grads)
        trainer.broadcast_parameters()
        assert model.dirty == 1
        clean, noisy = O.synthetic_batch(4, 16, 16, seed=5)
        gt = shard_batch(torch.from_numpy(clean.astype(np.float64)), rank, world)
        x = shard_batch(torch.from_numpy(noisy.astype(np.float64)), rank, world)
        for _ in range(2):
            out = trainer.step(gt, x, overlap=lambda: log.append("overlap") or "next batch")
            assert out[4] == "next batch"                                     # the hook's result comes back
        assert log == ["compute", "overlap", "apply"] * 2                     # hook between the collective and the update
        np.save(os.path.join(out_dir, f"trainer_rank{rank}.npy"), model.params.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_data_parallel_trainer_world_two(tmp_path):
    """DataParallelTrainer.broadcast_parameters / step with two ranks: rank 0's weights win, the all-reduce sits between
    the local step and the update, the overlap hook runs in between, grad_scale is 1 / world, replicas stay identical and
    equal the one-process emulation of the same two shards."""
    world = 2
    mp.spawn(_trainer_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    p = [np.load(tmp_path / f"trainer_rank{i}.npy") for i in range(world)]
    assert np.array_equal(p[0], p[1])
    cfg = O.canonical_config(no_layers=1)
    spec, ls = O.ResnetSpec.from_config(cfg["model"]), O.LossSpec.from_config(cfg["loss"])
    params, state = O.init_params(spec, seed=40)
    params, states = params.astype(np.float64), [state.astype(np.float64)] * 2
    clean, noisy = O.synthetic_batch(4, 16, 16, seed=5)
    m, v = np.zeros(params.size), np.zeros(params.size)
    for it in range(2):
        g = 0
        for r in range(2):
            sl = slice(2 * r, 2 * r + 2)
            _, _, _, _, gr, states[r] = O.train_step_single_gpu(spec, ls, params, states[r], clean[sl].astype(np.float64),
                                                                noisy[sl].astype(np.float64))
            g = g + gr
        params, m, v = O.adam_step(params, g * 0.5, m, v, it, 1e-3, global_clipnorm=1.0)
    assert np.abs(p[0] - params).max() <= 1e-12


# ---- a multi-output model without BatchNorm state (unet_laplacian's surface) through the same trainer ----------------------
class _MultiOutputModel:
    """what DataParallelTrainer sees of a unet_laplacian graph: several output scales, parameters, NO `state` attribute"""
    multi_output, depth = True, 3

    def __init__(self, params):
        self.params = torch.from_numpy(params.copy())
        self.device, self.n_params = torch.device("cpu"), params.size
        self.dirty = 0

    def mark_dirty(self):
        self.dirty += 1


def _multi_output_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from blind_image_denoising_amd.train_loop import DataParallelTrainer, TrainFunctions
        model = _MultiOutputModel(np.full(8, float(rank + 1)))               # replicas start different
        trainer = DataParallelTrainer.__new__(DataParallelTrainer)
        trainer.model, trainer.optimizer, trainer.group, trainer.world_size, trainer.comm = model, None, None, world, None
        seen = []

        def train_step_single_gpu(gt, x, dw, pct, tv):
            seen.append(tuple(dw))
            if len(dw) != model.depth:                                        # what _build_multi_output_train_functions raises
                raise ValueError(f"{model.depth} output scales need {model.depth} depth weights, got {len(dw)}")
            grads = model.params * 0 + float(rank + 1) * sum(dw)
            return torch.tensor(0.0), {}, [{}] * model.depth, [x] * model.depth, grads

        def apply_grads(opt, grads, tv=None, grad_scale=1.0):
            model.params = model.params - grads * grad_scale

        trainer.fns = TrainFunctions(None, None, train_step_single_gpu, apply_grads)
        trainer.broadcast_parameters()                                        # no `state`: must not raise
        assert torch.all(model.params == 1.0)
        x = torch.zeros(2, 4, 4, 3)
        out = trainer.step(x, x)                                              # scalar depth weight -> one per output scale
        assert len(out) == 5 and out[4] is None
        out = trainer.step(x, x, depth_weight=(0.5, 0.25, 0.25), overlap=lambda: "side")
        assert out[4] == "side"
        assert seen == [(1.0, 1.0, 1.0), (0.5, 0.25, 0.25)]
        np.save(os.path.join(out_dir, f"mo_rank{rank}.npy"), model.params.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_data_parallel_trainer_multi_output_model_world_two(tmp_path):
    """a model with several output scales and no BatchNorm state (unet_laplacian): broadcast_parameters skips the missing state,
    step() hands one depth weight per scale to the step (a scalar is repeated), the summed gradient is scaled by 1 / world."""
    world = 2
    mp.spawn(_multi_output_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    p = [np.load(tmp_path / f"mo_rank{i}.npy") for i in range(world)]
    assert np.array_equal(p[0], p[1])
    # step 1: grads (1 + 2) * 3 / 2 = 4.5 ; step 2: (1 + 2) * 1 / 2 = 1.5
    assert np.allclose(p[0], 1.0 - 4.5 - 1.5)
