"""Importer for trained `unet_laplacian` hydras saved by Keras 2.x as a `.keras` archive (`model.save`,
bfcnn/export_model.py:106-110): `config.json` gives the graph, `model.weights.h5` (read
with `h5lite`) the tensors.  Returns the model-config dict `UnetLaplacianHydra` / `oracle.unet_oracle` understand plus
the flat float32 parameter vector in their inventory order.

The one archive the reference ships (`pretrained/unet_laplacian_v5.6/model_hydra.keras`) was written by older code
than the snapshot builder; what differs is detected from the archive itself and expressed through config keys of this
package (not reference keys): `convnext_activation`, `encoder_level_activation`, `output_normalization_at_heads`,
`attention_full_resolution`, `attention_activation`, `upsample_linear` (DESIGN.md section 7 says what each stands for and
how the exported TFLite graph next to the archive pins them)."""
import json
import re
import zipfile
from typing import Dict, List, Tuple

import numpy as np

from .h5lite import H5File


def _walk_layers(model: Dict):
    for layer in model["config"]["layers"]:
        yield layer
        if layer["class_name"] == "Functional":
            yield from _walk_layers(layer)


def _suffix(name: str) -> int:
    m = re.search(r"_(\d+)$", name)
    return int(m.group(1)) if m else 0


def _by_creation(names: List[str]) -> List[str]:
    return sorted(set(names), key=_suffix)


def _leaky_name(alpha: float) -> str:
    for name, a in (("leaky_relu_01", 0.1), ("leaky_relu", 0.3), ("leaky_relu_001", 0.01)):
        if abs(alpha - a) < 1e-6:
            return name
    raise NotImplementedError(f"LeakyReLU(alpha={alpha})")


def config_from_archive_graph(graph: Dict) -> Dict:
    """model-config dict (`backbone` / `denoiser`) of the hydra described by the archive's config.json."""
    layers = list(_walk_layers(graph))
    by_class: Dict[str, List[Dict]] = {}
    for layer in layers:
        by_class.setdefault(layer["class_name"], []).append(layer)
    blocks = {layer["config"]["name"]: layer["config"] for layer in by_class.get("ConvNextBlock", [])}
    enc = sorted(n for n in blocks if n.startswith("encoder_"))
    if not enc:
        raise NotImplementedError("archive holds no unet_laplacian ConvNext encoder")
    attn = by_class.get("ConvolutionalSelfAttention", [])
    depth = 1 + max(int(n.split("_")[1]) for n in enc) + (1 if attn else 0)
    width = 1 + max(int(n.split("_")[2]) for n in enc)
    e0 = blocks["encoder_0_0"]
    dec = sorted(n for n in blocks if n.startswith("decoder_"))
    k_of = lambda c: int(c["conv_params_1"]["kernel_size"][0] if isinstance(c["conv_params_1"]["kernel_size"], (list, tuple))
                         else c["conv_params_1"]["kernel_size"])
    base = next(l for l in by_class["Conv2D"] if tuple(l["config"]["kernel_size"]) == (5, 5))["config"]
    leaky = by_class.get("LeakyReLU", [])
    activation = _leaky_name(float(leaky[0]["config"]["alpha"])) if leaky else "linear"
    # encoder level output: snapshot applies LayerNorm + activation before the Laplacian split; the archive's graph feeds
    # the last Add of the level straight into the smoothing filter
    smooth = by_class.get("GaussianFilter", []) or by_class.get("AveragePooling2D", [])
    level_raw = bool(smooth) and all(str(l["inbound_nodes"][0][0][0]).startswith("add") for l in smooth)
    n_out_ln = len(by_class.get("LayerNormalization", []))
    heads = [l for l in graph["config"]["layers"] if l["config"]["name"].startswith("denoiser_head_")]
    head_convs = [l["config"] for l in heads[0]["config"]["layers"] if l["class_name"] == "Conv2D"]
    head_leaky = [l["config"] for l in heads[0]["config"]["layers"] if l["class_name"] == "LeakyReLU"]
    if any(l["class_name"] in ("Concatenate", "BatchNormalization") for l in layers):
        raise NotImplementedError("archive graph uses concat skips / batch norm")
    if not any(l["class_name"] == "UpSampling2D" and l["config"].get("interpolation") == "bilinear" for l in layers):
        raise NotImplementedError("archive graph does not use bilinear upsampling")
    backbone = {
        "type": "unet_laplacian", "input_shape": ["?", "?", int(base.get("batch_input_shape", [None, None, None, 3])[-1] or 3)],
        "depth": depth, "width": width, "filters": int(base["filters"]),
        "use_bn": False, "use_ln": e0.get("ln_params") is not None, "use_bias": bool(base.get("use_bias", False)),
        "use_concat": False, "use_gamma": bool(e0.get("use_gamma", True)), "use_complex_base": False, "use_mix_project": False,
        "use_self_attention": bool(attn), "use_attention_gates": False, "use_output_normalization": n_out_ln > 0,
        "use_laplacian": True, "use_laplacian_averaging": not by_class.get("GaussianFilter"),
        "encoder_kernel_size": k_of(e0), "decoder_kernel_size": k_of(blocks[dec[0]]) if dec else 1,
        "multiple_scale_outputs": True, "activation": activation, "upsample_type": "upsample_laplacian_conv2d",
        "downsample_type": "strides", "value_range": [0, 255],
        "convnext_activation": e0["conv_params_2"].get("activation", activation),
        "encoder_level_activation": not level_raw, "output_normalization_at_heads": level_raw and n_out_ln == depth,
        # UpSampling2D fed by a Conv2D and feeding the Add directly: 1x1 -> bilinear x2 with no activation in between
        "upsample_linear": all(str(l["inbound_nodes"][0][0][0]).startswith("conv2d") for l in by_class["UpSampling2D"]),
    }
    if attn:
        ac = attn[0]["config"]
        backbone["attention_activation"] = ac.get("attention_activation", "leaky_relu")
        # no attention_resolution in the saved layer config and no resize operator in the exported graph: every pixel
        # of the deepest level is a token; such layers carry a second LayerNorm (`ln_1`)
        backbone["attention_full_resolution"] = "attention_resolution" not in ac
    denoiser = {"filters": int(head_convs[0]["filters"]), "use_bn": False, "use_ln": False, "use_bias": False,
                "activation": _leaky_name(float(head_leaky[0]["alpha"])) if head_leaky else head_convs[0].get("activation", "linear"),
                "output_channels": int(head_convs[-1]["filters"])}
    return {"backbone": backbone, "denoiser": denoiser}


def _weights_by_layer(h5: Dict[str, np.ndarray]) -> Tuple[Dict[str, Dict[str, np.ndarray]], List[Dict[str, np.ndarray]]]:
    """({backbone layer name: {sub-layer or "": array}}, [per-head {layer name: array}]) from the dataset paths."""
    dep = "_layer_checkpoint_dependencies/"
    backbone: Dict[str, Dict[str, np.ndarray]] = {}
    heads: Dict[str, Dict[str, np.ndarray]] = {}
    for path, arr in h5.items():
        parts = [p for p in path.replace(dep, "").split("/") if p]
        if parts[-2:] != ["vars", "0"]:
            raise NotImplementedError(f"layer with more than one variable: {path}")
        parts = parts[:-2]
        if len(parts) >= 3 and parts[1] == "functional":                      # hydra / backbone wrapper / unet_laplacian
            backbone.setdefault(parts[2], {})["/".join(parts[3:])] = arr
        elif len(parts) == 2:
            heads.setdefault(parts[0], {})[parts[1]] = arr
        else:
            raise NotImplementedError(f"unexpected dataset path {path}")
    return backbone, [heads[k] for k in _by_creation(list(heads))]


def params_from_archive(model_config: Dict, h5: Dict[str, np.ndarray], inventory: List[Tuple[str, Tuple[int, ...], str]]) -> np.ndarray:
    """flat float32 vector in `inventory` order.  Keras numbers same-class layers in creation order, which is the
    builder's graph order: encoder levels top-down, then decoder levels bottom-up, then the output LayerNorms full
    resolution first."""
    bb = model_config["backbone"]
    depth, width = bb["depth"], bb["width"]
    layers, heads = _weights_by_layer(h5)
    blocks = _by_creation([n for n in layers if n.startswith("conv_next_block")])
    attns = _by_creation([n for n in layers if n.startswith("convolutional_self_attention")])
    convs = _by_creation([n for n in layers if re.fullmatch(r"conv2d(_\d+)?", n)])
    lns = _by_creation([n for n in layers if re.fullmatch(r"layer_normalization(_\d+)?", n)])
    n_conv_levels = depth - (1 if attns else 0)
    named: Dict[str, np.ndarray] = {"base/kernel": layers[convs[0]][""]}
    order = [f"enc{d}_{w}" for d in range(n_conv_levels) for w in range(width)] + \
            [f"dec{d}_{w}" for d in reversed(range(depth - 1)) for w in range(width)]
    if len(order) != len(blocks) or len(attns) not in (0, width) or len(convs) != 1 + 2 * (depth - 1):
        raise NotImplementedError("archive layer counts do not match a unet_laplacian of this depth / width")
    for prefix, name in zip(order, blocks):
        w = layers[name]
        named[f"{prefix}/dw/kernel"], named[f"{prefix}/pw1/kernel"], named[f"{prefix}/pw2/kernel"] = w["conv_1"], w["conv_2"], w["conv_3"]
        if "ln" in w:
            named[f"{prefix}/ln/gamma"] = w["ln"]
        if "gamma" in w:
            named[f"{prefix}/gamma/w"] = w["gamma"]
    for i, name in enumerate(attns):
        w, prefix = layers[name], f"enc{depth - 1}_{i}"
        named[f"{prefix}/ln/gamma"] = w["ln_0"]
        for n in ("key", "query", "value"):
            named[f"{prefix}/{n}/kernel"] = w[f"{n}_conv"]
        if "ln_1" in w:
            named[f"{prefix}/ln1/gamma"] = w["ln_1"]
        named[f"{prefix}/out/kernel"], named[f"{prefix}/gamma/w"] = w["output_fn"], w["gamma"]
    for d in range(depth - 1):                                       # down projections, then up projections deepest first
        named[f"down{d}/kernel"] = layers[convs[1 + d]][""]
    for j, d in enumerate(reversed(range(depth - 1))):
        named[f"up{d}/kernel"] = layers[convs[depth + j]][""]
    if bb.get("output_normalization_at_heads"):
        for d, name in enumerate(lns):
            named[f"enc{d}/out_ln/gamma" if d == depth - 1 else f"dec{d}/out_ln/gamma"] = layers[name][""]
    elif lns:
        raise NotImplementedError("archive with in-line output LayerNorms: creation order not mapped")
    for i, h in enumerate(heads):
        c = _by_creation(list(h))
        named[f"head{i}/conv0/kernel"], named[f"head{i}/conv1/kernel"] = h[c[0]], h[c[1]]
    parts = []
    for name, shape, _ in inventory:
        if name not in named:
            raise ValueError(f"archive holds no tensor for {name}")
        a = np.asarray(named.pop(name), np.float32)
        if a.size != int(np.prod(shape)):
            raise ValueError(f"{name}: archive shape {a.shape} vs inventory {shape}")
        parts.append(a.ravel())
    if named:
        raise ValueError(f"archive tensors without a place in the graph: {sorted(named)}")
    return np.concatenate(parts)


def read_archive(path: str) -> Tuple[Dict, Dict[str, np.ndarray]]:
    with zipfile.ZipFile(path) as z:
        graph = json.loads(z.read("config.json"))
        h5 = dict(H5File(z.read("model.weights.h5")).datasets())
    return config_from_archive_graph(graph), h5
