"""CPU tests: the C-ABI library loads and exports every symbol include/bfcnn_hip.h declares,
non-compute entry points behave (layout, errors), and the host mirrors the reference's
argument / error behaviour.  No kernel is launched here."""
import ctypes as C
import json
import math
import pathlib
import re

import numpy as np
import pytest
import torch

import blind_image_denoising_amd as bf
from blind_image_denoising_amd import _native as N
from oracle import bfcnn_oracle as O

ROOT = pathlib.Path(__file__).resolve().parent.parent


HEADERS = ("bfcnn_hip.h", "bfcnn_hip_debug.h")      # the drop-in ABI and the single-kernel diagnostic entries


def _header_text():
    return "\n".join((ROOT / "include" / h).read_text() for h in HEADERS)


def _declared_symbols():
    text = _header_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(str(N.LIB_PATH))
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/bfcnn_hip.h but not exported"
    # and the ctypes table binds exactly the declared functions
    assert sorted(N.SIGNATURES) == declared
    assert N.lib().bf_abi_version() == 1


def test_ctypes_signatures_match_the_header_prototypes():
    """every prototype of include/bfcnn_hip.h against _native.SIGNATURES: same number of parameters, pointers bound as pointers,
    int / int64_t / float / uint64_t as the ctypes type of that width (a binding that drifts from the header corrupts calls
    silently: ctypes does not check)."""
    text = _header_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = re.findall(r"\b(?:const\s+char\s*\*|int64_t|int|void|bf_handle)\s*(bf_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S)
    names = [n for n, _ in protos]
    assert sorted(set(names)) == _declared_symbols(), sorted(set(_declared_symbols()) - set(names))
    scalar = {"int": C.c_int, "int32_t": C.c_int, "int64_t": C.c_int64, "float": C.c_float, "uint64_t": C.c_uint64, "unsigned": C.c_uint,
              "double": C.c_double}
    for name, args in protos:
        args = " ".join(args.split())
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        _, argtypes = N.SIGNATURES[name]
        assert len(params) == len(argtypes), (name, len(params), len(argtypes))
        for i, (p, t) in enumerate(zip(params, argtypes)):
            words = p.replace("const", " ").replace("*", " * ").split()
            is_pointer = "*" in words or "[" in p or words[0] in ("bf_handle",)
            if is_pointer:
                assert t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents") or hasattr(t, "_type_"), (name, i, p, t)
            else:
                assert words[0] in scalar, (name, i, p)
                assert C.sizeof(t) == C.sizeof(scalar[words[0]]) and (t is C.c_float) == (words[0] == "float"), (name, i, p, t)


def test_struct_layouts_match_header():
    assert C.sizeof(N.ResnetDesc) == 17 * 4 + 5 * 4
    assert C.sizeof(N.LossDesc) == 8 * 4
    assert C.sizeof(N.TensorInfo) == 104          # 100 rounded up to the int64 alignment


@pytest.mark.parametrize("n,count", [(6, 28784), (18, 84272)])
def test_parameter_inventory_matches_oracle(n, count):
    cfg = O.canonical_config(no_layers=n)["model"]
    m = bf.model_builder(cfg, device="cpu", seed=0).hydra
    spec = O.ResnetSpec.from_config(cfg)
    assert m.n_params == count == spec.param_count()
    assert m.n_state == spec.state_count() == 32 * n
    off = spec.offsets()
    names = {"block%d/conv0/kernel", "block%d/conv1/kernel"}
    for v in m.trainable_variables:
        o, s = off[v.name.replace("bn1", "bn1")]
        assert (v.offset, v.shape) == (o, tuple(s)), v.name
    for v, (name, shape) in zip(m.non_trainable_variables, spec.state_tensors()):
        assert v.name == name and v.shape == tuple(shape)


@pytest.mark.parametrize("nb", [1, 3])
def test_parameter_inventory_of_block_variants(nb):
    """block_kernels of length 1 / 3: variable order conv0, (conv j, bn j gamma) for j >= 1; one BN state pair per j >= 1."""
    cfg = O.canonical_config(no_layers=3)["model"]
    cfg["backbone"].update(block_kernels=[3] * nb, block_filters=[16] * nb)
    m = bf.model_builder(cfg, device="cpu", seed=0).hydra
    spec = O.ResnetSpec.from_config(cfg)
    assert m.n_params == spec.param_count() and m.n_state == spec.state_count() == 3 * (nb - 1) * 32
    off = spec.offsets()
    assert [v.name for v in m.trainable_variables] == [t[0] for t in spec.tensors()]
    for v in m.trainable_variables:
        assert (v.offset, v.shape) == (off[v.name][0], tuple(off[v.name][1])), v.name
    assert [(v.name, v.shape) for v in m.non_trainable_variables] == [(n, tuple(s)) for n, s in spec.state_tensors()]


def test_initial_values_follow_keras_defaults():
    m = bf.model_builder(O.canonical_config(no_layers=2)["model"], device="cpu", seed=3).hydra
    for v in m.trainable_variables:
        a = v.numpy()
        if v.kind == 1:
            assert np.all(a == 1.0)                           # gamma
        else:
            kh, kw, ci, co = a.shape
            std = math.sqrt(2.0 / (kh * kw * (ci + co))) / 0.87962566103423978
            assert np.abs(a).max() <= 2.0 * std + 1e-7        # truncated at 2 sigma
            assert 0.5 * std < a.std() < 1.2 * std
    st = {v.name: v.numpy() for v in m.non_trainable_variables}
    assert np.all(st["block0/bn1/moving_mean"] == 0) and np.all(st["block0/bn1/moving_variance"] == 1)


def test_create_rejects_what_the_reference_rejects():
    cfg = O.canonical_config(no_layers=2)["model"]
    bad = json.loads(json.dumps(cfg)); bad["backbone"]["block_kernels"] = [3, 3, 3, 3]; bad["backbone"]["block_filters"] = [16] * 4
    with pytest.raises(ValueError, match="<= 3"):
        bf.model_builder(bad, device="cpu")
    bad = json.loads(json.dumps(cfg)); bad["backbone"]["block_filters"] = [16]
    with pytest.raises(ValueError):
        bf.model_builder(bad, device="cpu")
    bad = json.loads(json.dumps(cfg)); bad["backbone"]["type"] = "nonsense"
    with pytest.raises(ValueError, match="don't know how to build model"):
        bf.model_builder(bad, device="cpu")
    bad = json.loads(json.dumps(cfg)); bad["backbone"]["type"] = "efficientnet"
    with pytest.raises(NotImplementedError):
        bf.model_builder(bad, device="cpu")
    bad = json.loads(json.dumps(cfg)); bad["backbone"]["filters"] = 48; bad["backbone"]["block_filters"] = [48, 48]
    with pytest.raises(NotImplementedError, match="16"):
        bf.model_builder(bad, device="cpu")
    # 32 / 64 / 128 filters go to the generic resnet model (operator library, inference only)
    ok = json.loads(json.dumps(cfg)); ok["backbone"]["filters"] = 32; ok["backbone"]["block_filters"] = [32, 32]
    assert type(bf.model_builder(ok, device="cpu").hydra).__name__ == "GenericResnetHydra"
    d = N.ResnetDesc()
    h = C.c_void_p()
    assert N.lib().bf_create(C.byref(d), C.byref(h)) == N.BF_EINVAL and not h.value
    assert "struct_size" in N.last_error(None)


def test_workspace_and_packed_queries():
    m = bf.model_builder(O.canonical_config(no_layers=6)["model"], device="cpu").hydra
    L = N.lib()
    assert L.bf_workspace_bytes(m._h, N.BF_MODE_INFERENCE, 64, 256, 256) == 3 * 64 * 256 * 256 * 16 * 4 + N.BF_STATUS_BYTES
    assert L.bf_workspace_bytes(m._h, N.BF_MODE_INFERENCE, 1, 200, 300) == 3 * 256 * 512 * 16 * 4 + N.BF_STATUS_BYTES   # pow2 padded
    assert L.bf_workspace_bytes(m._h, N.BF_MODE_TRAIN, 2, 32, 32) > (3 * 6 + 2) * 2 * 32 * 32 * 16 * 4
    assert L.bf_workspace_bytes(m._h, 0, 0, 8, 8) == -1
    assert L.bf_packed_bytes(m._h) % 256 == 0


def test_load_model_errors_mirror_reference():
    with pytest.raises(ValueError, match="cannot be empty"):
        bf.load_model("")
    with pytest.raises(ValueError, match="cannot be empty"):
        bf.load_model(None)
    with pytest.raises(ValueError, match="does not exist"):
        bf.load_model("/definitely/not/here")
    with pytest.raises(ValueError, match="does not exist"):
        bf.load_denoiser_model("nope")
    with pytest.raises(ValueError, match="should not be None"):
        bf.DenoiserModule(None)


def test_denoiser_module_rejects_non_uint8_rank4(tmp_path):
    m = bf.model_builder(O.canonical_config(no_layers=1)["model"], device="cpu").hydra
    mod = bf.DenoiserModule(m)
    with pytest.raises(ValueError):
        mod(np.zeros((1, 8, 8, 3), np.float32))
    with pytest.raises(ValueError):
        mod(np.zeros((8, 8, 3), np.uint8))
    with pytest.raises(ValueError):
        mod(np.zeros((1, 8, 8, 1), np.uint8))
    assert mod(np.zeros((0, 8, 8, 3), np.uint8)).shape == (0, 8, 8, 3)       # empty batch
    with pytest.raises(RuntimeError, match="no CPU execution path"):
        mod(np.zeros((1, 8, 8, 3), np.uint8))


def test_save_and_load_model_directory(tmp_path):
    cfg = O.canonical_config(no_layers=2)
    m = bf.model_builder(cfg["model"], device="cpu", seed=5).hydra
    bf.save_model(m, str(tmp_path / "m"), cfg)
    mod = bf.load_model(str(tmp_path / "m"), device="cpu")
    p0, s0 = m.get_weights()
    p1, s1 = mod.model_hydra.get_weights()
    assert np.array_equal(p0, p1) and np.array_equal(s0, s1)
    assert bf.load_config(str(tmp_path / "m" / "pipeline.json"))["loss"]["hinge"] == 0.5


def test_load_config_and_shape_fixer():
    assert bf.input_shape_fixer(["?", "", "-1", 3]) == [None, None, None, 3]
    with pytest.raises(ValueError):
        bf.load_config(None)
    with pytest.raises(ValueError):
        bf.load_config("/no/such/file.json")
    assert bf.load_config({"a": 1}) == {"a": 1}
    assert len(bf.configs) == 3 and all("model" in c for _, c in bf.configs)


def test_schedules_match_keras_formulas():
    s = bf.schedule_builder({"type": "exponential_decay", "config": {"decay_rate": 0.9, "decay_steps": 40000, "learning_rate": 1e-3}})
    assert s(0) == 1e-3 and abs(s(20000) - 1e-3 * 0.9 ** 0.5) < 1e-15
    assert abs(s(20000) - O.exponential_decay(1e-3, 40000, 0.9, 20000)) < 1e-18
    c = bf.schedule_builder({"type": "cosine_decay", "config": {"decay_steps": 100, "learning_rate": 1.0, "alpha": 0.1}})
    assert abs(c(0) - 1.0) < 1e-12 and abs(c(100) - 0.1) < 1e-12 and abs(c(1000) - 0.1) < 1e-12
    r = bf.schedule_builder({"type": "cosine_decay_restarts", "config": {"decay_steps": 10, "learning_rate": 1.0}})
    assert abs(r(0) - 1.0) < 1e-12 and r(10) == pytest.approx(0.9 * (1 - 0.001) + 0.001)
    with pytest.raises(ValueError):
        bf.schedule_builder({"type": "nope"})
    with pytest.raises(ValueError):
        bf.schedule_builder({})
    d = bf.deep_supervision_schedule_builder({"type": "linear_low_to_high"}, 3)
    assert np.allclose(d(0.0), [1 / 6, 2 / 6, 3 / 6]) and np.allclose(d(1.0), [3 / 6, 2 / 6, 1 / 6])


def test_optimizer_builder_contract():
    cfg = O.canonical_config()["train"]["optimizer"]
    opt, sched = bf.optimizer_builder(cfg)
    assert opt.global_clipnorm == 1.0 and opt.lr() == 1e-3 and (opt.beta_1, opt.beta_2, opt.epsilon) == (0.9, 0.999, 1e-7)
    with pytest.raises(ValueError):
        bf.optimizer_builder("x")
    with pytest.raises(ValueError, match="don't know how to handle optimizer_type"):
        bf.optimizer_builder({"type": "sgd", "schedule": cfg["schedule"]})
    with pytest.raises(NotImplementedError):
        bf.optimizer_builder({"schedule": cfg["schedule"]})          # default RMSprop: outside the hot path
    # the optimizer section of the reference's shipped unet configs: per-tensor clipnorm + cosine restarts
    opt, sched = bf.optimizer_builder({"type": "ADAM", "gradient_clipping_by_norm_local": 1.0, "schedule": {
        "type": "cosine_decay_restarts", "config": {"t_mul": 1.1, "epsilon": 1e-5, "decay_rate": 0.9, "decay_steps": 40000,
                                                    "learning_rate": 0.001}}})
    assert (opt.clipnorm, opt.global_clipnorm, opt.clipvalue) == (1.0, None, None) and abs(opt.lr() - 1e-3) < 1e-9


def test_loss_builder_contract_and_monitor_values():
    fns = bf.loss_function_builder(O.canonical_config()["loss"])
    assert set(fns) == {"model", "denoiser"}
    d = fns["denoiser"].desc(0.5)
    assert (d.hinge, d.cutoff, d.mae_multiplier, d.regularization, d.depth_weight) == (0.5, 255.0, 1.0, pytest.approx(0.01), 0.5)
    gt = torch.zeros((2, 8, 8, 3))
    with pytest.raises(RuntimeError, match="GPU"):                        # monitoring values come from the HIP kernels only
        fns["denoiser"](gt, gt)


def test_pyramid_type_parsing():
    from blind_image_denoising_amd.pyramid import PyramidType
    assert PyramidType.from_string(" laplacian ") == PyramidType.LAPLACIAN
    for bad in (None, 3, "  "):
        with pytest.raises(ValueError):
            PyramidType.from_string(bad)
    with pytest.raises(KeyError):
        PyramidType.from_string("pyramid")


def test_graphed_module_argument_checks():
    import blind_image_denoising_amd as bf
    with pytest.raises(ValueError):
        bf.GraphedDenoiserModule(object())
    cfg = O.canonical_config(no_layers=1)
    m = bf.DenoiserModule(bf.model_builder(cfg["model"], device="cpu").hydra)
    with pytest.raises(ValueError):
        bf.GraphedDenoiserModule(m, max_shapes=0)
    g = bf.GraphedDenoiserModule(m)
    assert g.captured_shapes() == [] and g.name == m.name
    with pytest.raises(ValueError):
        g(np.zeros((1, 8, 8, 3), np.float32))
