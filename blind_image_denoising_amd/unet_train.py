"""
Training of the `unet_laplacian` hydra (bfcnn/train_loop.py:259-321 with a multi-output model): the
training-mode forward, one denoiser loss per output scale times its depth weight, the model's regularisers, and the gradient
of the total with respect to every trainable tensor -- what `tf.GradientTape` does in the reference -- as an explicit walk of
the graph of bfcnn/backbone_unet_laplacian.py:281-606 over the operator library's C-ABI entry points (`bf_op_*`: forward
operators of unet_ops.hip, backward primitives of train_prims.hip).  PyTorch holds the tensors; it computes nothing.

Scope: the graph family of configs/unet_laplacian_v5.json and v6.json -- ConvNext blocks (depthwise k x k, LayerNorm, 1x1 C->4C + activation,
1x1 4C->C, ChannelLearnableMultiplier, StochasticDepth, Add), self-attention blocks on the deepest level (resize to 16x16,
LayerNorm, query / key / value, dot-product attention with dropout, resize back, output convolution, multiplier), level
LayerNorm + activation, the averaging / Gaussian Laplacian split, strided down-sampling + 1x1 or 2x2 stride-2 convolution, `upsample_laplacian_conv2d` / nearest or bilinear + 3x3 convolution,
per-scale denoiser heads, AdditiveAttentionGate in front of the decoder Add (v3 / v4) -- with any depth / width / filters.
The graph revision of the reference's trained archive (tests/golden/unet_v56.npz: GELU in the MLP and on query / key / value,
row-wise full-resolution attention with a second LayerNorm, no level activation, 1x1-then-resize up-sampling, the output
LayerNorms in front of the heads) trains as well, so the shipped network can be fine-tuned.  So do mix projections, maxpool
down-sampling and plain bilinear / nearest up-sampling; conv2d_transpose up-sampling raises NotImplementedError (as in inference).

Exact fp32 throughout (the split-f16 inference operators are not used here): gradients are compared with the torch-autograd
oracle (oracle/unet_torch.py) in tests/test_gpu_unet_train.py.
"""
import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _native as N
from . import unet_laplacian as UL
from .pyramid import avg_pool2_valid, upsample_2x

SOFTORTHONORMAL = (0.01, 0.0, 1e-4)        # bfcnn/constants.py:19-21: lambda, l1, l2
MULTIPLIER_L1 = 1e-6                        # ChannelLearnableMultiplier's regulariser (custom_layers.py:267)
KERNEL_L2 = 0.01                            # keras "l2" string regulariser
GATE_L2 = 1e-4                              # AdditiveAttentionGate's default kernel regulariser (custom_layers.py:726)


def _call(fn_name: str, *args):
    N.check(getattr(N.lib(), fn_name)(*args), None, fn_name)


class _Ops:
    """the backward primitives as tensor-in / tensor-out calls sharing one scratch buffer"""

    def __init__(self, device, scratch_floats: int):
        self.device = device
        self.scratch = torch.empty(int(scratch_floats), dtype=torch.float32, device=device)

    def _s(self):
        return N.ptr(self.scratch), self.scratch.numel()

    def act_bwd(self, out, dy, act, pre=None):
        """dy * act'(.): from the activation's OUTPUT for the sign-preserving ones, from its input `pre` for GELU"""
        code, a = UL._act(act)
        if code == 0:
            return dy
        dx = torch.empty_like(dy)
        if code == 3:
            if pre is None:
                raise ValueError("the GELU derivative needs the pre-activation")
            _call("bf_op_act_bwd", N.ptr(pre), N.ptr(dy), N.ptr(dx), dy.numel(), code, a, 0, N.stream_ptr(dy))
        else:
            _call("bf_op_act_bwd", N.ptr(out), N.ptr(dy), N.ptr(dx), dy.numel(), code, a, 1, N.stream_ptr(dy))
        return dx

    def act_bwd_alpha(self, out, dy, alpha):
        dx = torch.empty_like(dy)
        _call("bf_op_act_bwd", N.ptr(out), N.ptr(dy), N.ptr(dx), dy.numel(), 2, float(alpha), 1, N.stream_ptr(dy))
        return dx

    def matmul_wgrad(self, x, dy, dw):
        cin, cout = x.shape[-1], dy.shape[-1]
        sp, sn = self._s()
        _call("bf_op_matmul_wgrad", N.ptr(x), N.ptr(dy), N.ptr(dw), x.numel() // cin, cin, cout, sp, sn, N.stream_ptr(x))

    def dwconv_wgrad(self, x, dy, dw, k):
        B, H, W, Cc = x.shape
        sp, sn = self._s()
        _call("bf_op_dwconv_wgrad", N.ptr(x), N.ptr(dy), N.ptr(dw), B, H, W, Cc, k, sp, sn, N.stream_ptr(x))

    def layernorm_bwd(self, x, gamma, dy, dgamma):
        Cc = x.shape[-1]
        dx = torch.empty_like(x)
        sp, sn = self._s()
        _call("bf_op_layernorm_bwd", N.ptr(x), N.ptr(gamma), N.ptr(dy), N.ptr(dx), N.ptr(dgamma), x.numel() // Cc, Cc, UL.LN_EPSILON,
              sp, sn, N.stream_ptr(x))
        return dx

    def scale_add(self, res, t, m, s):
        B = t.shape[0]
        Cc = t.shape[-1]
        out = torch.empty_like(t)
        _call("bf_op_scale_add", N.ptr(res), N.ptr(t), N.ptr(m), N.ptr(s), N.ptr(out), B, t.numel() // (B * Cc), Cc, N.stream_ptr(t))
        return out

    def scale_add_bwd(self, t, m, s, dy, dm):
        B, Cc = t.shape[0], t.shape[-1]
        dt = torch.empty_like(t)
        sp, sn = self._s()
        _call("bf_op_scale_add_bwd", N.ptr(t), N.ptr(m), N.ptr(s), N.ptr(dy), N.ptr(dt), N.ptr(dm), B, t.numel() // (B * Cc), Cc,
              sp, sn, N.stream_ptr(t))
        return dt

    def add(self, a, b):
        """a + b (new tensor)"""
        return self.scale_add(a, b, None, None)

    def transpose(self, w2d):
        a, b = w2d.shape
        out = torch.empty((b, a), dtype=torch.float32, device=w2d.device)
        _call("bf_op_transpose2d", N.ptr(w2d), N.ptr(out), a, b, N.stream_ptr(w2d))
        return out


class UnetTrainGraph:
    """train_step_single_gpu for a UnetLaplacianHydra: `step(gt, noisy, depth_weights, ...)` returns the per-scale predictions
    and fills `grads` (flat, laid out like model.params) and the loss slots."""

    def __init__(self, model: "UL.UnetLaplacianHydra", loss_config: Dict, soft_orthonormal: Optional[bool] = None):
        self.m = model
        bad = []
        if model.downsample_type not in ("strides", "conv2d", "maxpool"): bad.append(f"downsample_type {model.downsample_type}")
        if model.upsample_type not in ("upsample_laplacian_conv2d", "upsample_nearest_conv2d", "upsample_bilinear_conv2d", "bilinear",
                                       "nn", "nearest"):
            bad.append(f"upsample_type {model.upsample_type}")
        if model.activation == "gelu": bad.append("gelu outside the convnext MLP / the attention projections")
        if model.activation == "linear": bad.append("linear activation")
        if getattr(model, "use_concat", False) and not model.use_mix_project and model.dec_k not in (1, 3, 5):
            bad.append(f"use_concat without use_mix_project and decoder_kernel_size {model.dec_k}")
        if any(model.level_filters(d) == 256 and not model._is_attention(d) for d in range(model.depth)):
            bad.append("a 256-channel ConvNext level (runs at inference only)")
        if bad:
            raise NotImplementedError("unet_laplacian training is built for the configs/unet_laplacian_v5.json graph family: " + ", ".join(bad))
        self.loss_config = dict(loss_config)
        bb = model.config["backbone"]
        self.soft_orthonormal = bool(bb.get("use_soft_orthonormal_regularization", False)) if soft_orthonormal is None else soft_orthonormal
        self.off = {name: (off, shape, kind) for name, shape, kind, off in model.trainable_variables}
        self.ops = None
        self.totals = None

    # ---- parameters ------------------------------------------------------------------------------------------------------
    def W(self, name) -> torch.Tensor:
        off, shape, _ = self.off[name]
        n = int(np.prod(shape))
        t = self.m.params[off:off + n]
        if off % 4:
            t = t.clone()
        return t.view(shape)

    def G(self, name, grads) -> torch.Tensor:
        """the slice of the flat gradient a tensor's gradient is written to (16-byte aligned staging when the slice is not)"""
        off, shape, _ = self.off[name]
        n = int(np.prod(shape))
        if off % 4:
            buf = torch.empty(n, dtype=torch.float32, device=grads.device)
            self._unaligned.append((buf, off, n))
            return buf
        return grads[off:off + n]

    # ---- one training step ----------------------------------------------------------------------------------------------------
    def step(self, gt: torch.Tensor, noisy: torch.Tensor, depth_weights: Sequence[float], grads: torch.Tensor,
             depth_scale: Optional[Dict[str, torch.Tensor]] = None, attn_scale: Optional[Dict[str, torch.Tensor]] = None):
        m = self.m
        dev = m.device
        gt = gt.to(device=dev, dtype=torch.float32).contiguous()
        noisy = noisy.to(device=dev).contiguous()
        if noisy.dtype != torch.uint8:
            noisy = noisy.to(torch.float32)
        B, H, Wd, _ = noisy.shape
        if H % (1 << (m.depth - 1)) or Wd % (1 << (m.depth - 1)):
            raise ValueError(f"training needs sizes divisible by {1 << (m.depth - 1)} (got {H}x{Wd})")
        depth_scale, attn_scale = depth_scale or {}, attn_scale or {}
        npix0 = B * H * Wd
        need = max(8 * 1024 * 1024, int(N.lib().bf_op_denoiser_loss_scratch_floats(B, H, Wd, m.out_channels)) + 1024,
                   B * 256 * 256 + 1024, npix0 * 4)
        if self.ops is None or self.ops.scratch.numel() < need:
            self.ops = _Ops(dev, need)
        ops = self.ops
        self._unaligned = []
        a = m.activation
        back = []                          # closures, run in reverse

        def pack(w2d):
            return UL.pack_pointwise(w2d.contiguous())

        def pointwise_act(x_, wp, cout, act, alpha=None):
            """(act(x W), what its derivative is taken from): the fused epilogue for the sign-preserving activations, the
            product kept and the activation as its own pass for GELU"""
            if UL._act(act)[0] == 3:
                pre_ = UL.pointwise(x_, wp, cout)
                y_ = UL.dwconv_ln(pre_.view(1, 1, -1, 32), None, None, act).view(pre_.shape)
                return y_, pre_
            y_ = UL.pointwise(x_, wp, cout, act) if alpha is None else UL.pointwise(x_, wp, cout, act, alpha=alpha)
            return y_, None

        # -- blocks -------------------------------------------------------------------------------------------------------
        def convnext(prefix, x, k):
            """ConvNextBlock + the residual Add (+ depth scale).  The first decoder block behind a Concatenate without mix projection
            maps 2 C -> C channels: no Add, no StochasticDepth there (backbone_unet_laplacian.py:557-560)"""
            Cin = x.shape[-1]
            wdw = self.W(f"{prefix}/dw/kernel")                                    # [k,k,Cin,1]
            Hh = int(self.off[f"{prefix}/pw1/kernel"][1][-1])
            Cc = int(self.off[f"{prefix}/pw2/kernel"][1][-1])
            skip = Cin == Cc
            w1, w2 = self.W(f"{prefix}/pw1/kernel").view(Cin, Hh), self.W(f"{prefix}/pw2/kernel").view(Hh, Cc)
            t1 = UL.dwconv_mult(x, wdw, None)
            gamma = self.W(f"{prefix}/ln/gamma") if m.use_ln else None
            t2 = UL.dwconv_ln(t1, None, gamma) if m.use_ln else t1
            # the hidden layer in chunks of at most 256 units (the widest 1x1 convolution of the operator library): one chunk up
            # to 64 channels, two for the 128-channel levels of the 4-level models
            Hc = min(Hh, 256)
            chunks = range(Hh // Hc)
            w1c = [w1 if Hc == Hh else w1[:, j * Hc:(j + 1) * Hc].contiguous() for j in chunks]
            w2c = [w2[j * Hc:(j + 1) * Hc] for j in chunks]
            am = m.mlp_activation
            t3, t3pre = zip(*[pointwise_act(t2, pack(w1c[j]), Hc, am) for j in chunks])
            t4 = None
            for j in chunks:
                t4 = UL.pointwise(t3[j], pack(w2c[j]), Cc, res=t4)
            wm = self.W(f"{prefix}/gamma/w") if m.use_gamma else None
            mult = UL.channel_multiplier(wm) if m.use_gamma else None
            s = depth_scale.get(prefix) if skip else None
            out = ops.scale_add(x if skip else None, t4, mult, s)

            def bwd(dout):
                dm = torch.empty(Cc, dtype=torch.float32, device=dev) if m.use_gamma else None
                dt4 = ops.scale_add_bwd(t4, mult, s, dout, dm)
                if m.use_gamma:
                    _call("bf_op_multiplier_bwd", N.ptr(wm), N.ptr(dm), N.ptr(self.G(f"{prefix}/gamma/w", grads)), Cc, N.stream_ptr(dm))
                gw2, gw1 = self.G(f"{prefix}/pw2/kernel", grads).view(Hh, Cc), self.G(f"{prefix}/pw1/kernel", grads).view(Cin, Hh)
                dt2 = None
                for j in chunks:
                    ops.matmul_wgrad(t3[j], dt4, gw2[j * Hc:(j + 1) * Hc])
                    dt3 = ops.act_bwd(t3[j], UL.pointwise(dt4, pack(ops.transpose(w2c[j].contiguous())), Hc), am, t3pre[j])
                    if Hc == Hh:
                        ops.matmul_wgrad(t2, dt3, gw1)
                    else:                                          # a column block of the kernel gradient: staged, then copied in
                        blk = torch.empty((Cin, Hc), dtype=torch.float32, device=dev)
                        ops.matmul_wgrad(t2, dt3, blk)
                        gw1[:, j * Hc:(j + 1) * Hc].copy_(blk)
                    dt2 = UL.pointwise(dt3, pack(ops.transpose(w1c[j])), Cin, res=dt2)
                dt1 = ops.layernorm_bwd(t1, gamma, dt2, self.G(f"{prefix}/ln/gamma", grads)) if m.use_ln else dt2
                ops.dwconv_wgrad(x, dt1, self.G(f"{prefix}/dw/kernel", grads), k)
                wf = torch.empty_like(wdw)
                _call("bf_op_flip_hw", N.ptr(wdw), N.ptr(wf), k, Cin, N.stream_ptr(wdw))
                dx = UL.dwconv_mult(dt1, wf, None)
                return ops.add(dout, dx) if skip else dx
            return out, bwd

        def attention(prefix, x):
            Bc, Hc, Wc, Cc = x.shape
            A = m.filters
            rows = m.attention_rows                  # the trained archive's graph: no resize, one sequence per image row
            rh, rw = (Hc, Wc) if rows else m.attention_resolution
            NB, T = (Bc * rh, rw) if rows else (Bc, rh * rw)
            r = x if rows else UL.resize_bilinear(x, rh, rw)
            gamma = self.W(f"{prefix}/ln/gamma") if m.use_ln else None
            n_ = UL.dwconv_ln(r, None, gamma) if m.use_ln else r
            qact, alpha = (m.attention_activation, None) if m.attention_activation else ("leaky_relu", m.attention_alpha)
            # keras reads [query, value, key]; the archive's graph hands the key convolution over as the value and the value
            # convolution as the key (unet_laplacian._attention)
            order = ("query", "key", "value") if rows else ("query", "value", "key")
            ws = {n: self.W(f"{prefix}/{n}/kernel").view(Cc, A) for n in order}
            (q, qpre), (v, vpre), (k_, kpre) = (tuple(None if z is None else z.view(NB, T, A) for z in pointwise_act(n_, pack(ws[n]), A, qact, alpha))
                                                for n in order)
            ps = attn_scale.get(prefix)
            o = torch.empty((NB, T, A), dtype=torch.float32, device=dev)
            P = torch.empty((NB, T, T), dtype=torch.float32, device=dev)
            _call("bf_op_attention_train", N.ptr(q), N.ptr(v), N.ptr(k_), N.ptr(ps), N.ptr(o), N.ptr(P), NB, T, A, N.stream_ptr(q))
            o4 = o.view(Bc, rh, rw, A)
            gamma1 = self.W(f"{prefix}/ln1/gamma") if (rows and m.use_ln) else None
            if rows:
                u = UL.dwconv_ln(o4, None, gamma1) if m.use_ln else o4
            else:
                u = UL.resize_bilinear(o4, Hc, Wc)
            wo = self.W(f"{prefix}/out/kernel").view(A, Cc)
            t = UL.pointwise(u, pack(wo), Cc)
            wm = self.W(f"{prefix}/gamma/w")
            mult = UL.channel_multiplier(wm)
            s = depth_scale.get(prefix)
            out = ops.scale_add(x, t, mult, s)

            def bwd(dout):
                dm = torch.empty(Cc, dtype=torch.float32, device=dev)
                dt = ops.scale_add_bwd(t, mult, s, dout, dm)
                _call("bf_op_multiplier_bwd", N.ptr(wm), N.ptr(dm), N.ptr(self.G(f"{prefix}/gamma/w", grads)), Cc, N.stream_ptr(dm))
                ops.matmul_wgrad(u, dt, self.G(f"{prefix}/out/kernel", grads))
                du = UL.pointwise(dt, pack(ops.transpose(wo)), A)
                if rows:
                    do = ops.layernorm_bwd(o4, gamma1, du, self.G(f"{prefix}/ln1/gamma", grads)) if m.use_ln else du
                else:
                    do = torch.empty((Bc, rh, rw, A), dtype=torch.float32, device=dev)
                    _call("bf_op_resize_bilinear_bwd", N.ptr(du), N.ptr(do), Bc, rh, rw, A, Hc, Wc, N.ptr(ops.scratch), N.stream_ptr(du))
                dq, dv, dk = (torch.empty((NB, T, A), dtype=torch.float32, device=dev) for _ in range(3))
                dS = torch.empty((NB, T, T), dtype=torch.float32, device=dev)
                _call("bf_op_attention_bwd", N.ptr(q), N.ptr(v), N.ptr(k_), N.ptr(ps), N.ptr(P), N.ptr(do), N.ptr(dq), N.ptr(dv),
                      N.ptr(dk), N.ptr(dS), NB, T, A, N.stream_ptr(q))
                dn = None
                for name, y, ypre, dy in zip(order, (q, v, k_), (qpre, vpre, kpre), (dq, dv, dk)):
                    dp = (ops.act_bwd_alpha(y, dy, alpha) if alpha is not None else ops.act_bwd(y, dy, qact, ypre)).view(Bc, rh, rw, A)
                    ops.matmul_wgrad(n_, dp, self.G(f"{prefix}/{name}/kernel", grads))
                    part = UL.pointwise(dp, pack(ops.transpose(ws[name])), Cc)
                    dn = part if dn is None else ops.add(dn, part)
                dr = ops.layernorm_bwd(r, gamma, dn, self.G(f"{prefix}/ln/gamma", grads)) if m.use_ln else dn
                if rows:
                    return ops.add(dout, dr)
                dxb = torch.empty_like(x)
                _call("bf_op_resize_bilinear_bwd", N.ptr(dr), N.ptr(dxb), Bc, Hc, Wc, Cc, rh, rw, N.ptr(ops.scratch), N.stream_ptr(dr))
                return ops.add(dout, dxb)
            return out, bwd

        def norm_act(name, x, act):
            """y = act(LayerNorm(x) * gamma)"""
            gamma = self.W(name)
            y = UL.dwconv_ln(x, None, gamma, act)

            def bwd(dy):
                return ops.layernorm_bwd(x, gamma, ops.act_bwd(y, dy, act), self.G(name, grads))
            return y, bwd

        def conv1x1_act(name, x, cout, act):
            cin = x.shape[-1]
            w = self.W(name).view(cin, cout)
            y = UL.pointwise(x, pack(w), cout, act)

            def bwd(dy):
                dp = ops.act_bwd(y, dy, act)
                ops.matmul_wgrad(x, dp, self.G(name, grads))
                return UL.pointwise(dp, pack(ops.transpose(w)), cin)
            return y, bwd

        def conv3x3_act(name, x, cout, act):
            """Conv2D 3 x 3, same, + activation (the convolution behind an UpSampling2D: upsampling.py:52-76)"""
            Bc, Hc, Wc, cin = x.shape
            w = self.W(name)                                                      # [3,3,cin,cout]
            y = UL.conv2d(x, UL.pack_conv(w.contiguous()), cout, 3, 1, act)

            def bwd(dy):
                dp = ops.act_bwd(y, dy, act)
                sp, sn = ops._s()
                _call("bf_op_conv2d_wgrad", N.ptr(x), 0, N.ptr(dp), N.ptr(self.G(name, grads)), Bc, Hc, Wc, cin, cout, 3, 0, 0.0, 0.0, sp, sn,
                      N.stream_ptr(dp))
                wf = torch.empty_like(w)                                          # data gradient: taps flipped, every tap transposed
                _call("bf_op_flip_hw", N.ptr(w), N.ptr(wf), 3, cin * cout, N.stream_ptr(w))
                wt = torch.empty((3, 3, cout, cin), dtype=torch.float32, device=dev)
                for t_ in range(9):
                    _call("bf_op_transpose2d", N.ptr(wf.view(9, cin, cout)[t_]), N.ptr(wt.view(9, cout, cin)[t_]), cin, cout, N.stream_ptr(wf))
                return UL.conv2d(dp, UL.pack_conv(wt), cin, 3, 1, "linear")
            return y, bwd

        def conv2x2_s2_act(name, x, cout, act):
            """Conv2D 2 x 2, strides 2, same (downsample_type "conv2d", downsampling.py:45-55); even sizes: no padding, every output
            sees its own 2 x 2 block, so the weight gradient is four 1 x 1 weight gradients on the strided slices of x and the data
            gradient the transposed convolution with the same kernel array"""
            Bc, Hc, Wc, cin = x.shape
            if Hc % 2 or Wc % 2:
                raise ValueError(f"conv2d down-sampling in training needs even sizes (got {Hc}x{Wc})")
            w = self.W(name)                                                      # [2,2,cin,cout]
            y = UL.conv2d(x, UL.pack_conv(w.contiguous()), cout, 2, 2, act)

            def bwd(dy):
                dp = ops.act_bwd(y, dy, act)
                gw = self.G(name, grads).view(4, cin * cout)
                xs = torch.empty((Bc, Hc // 2, Wc // 2, cin), dtype=torch.float32, device=dev)
                flat = x.reshape(-1)
                for ky in range(2):
                    for kx in range(2):
                        src = flat[(ky * Wc + kx) * cin:]                         # x[:, ky::2, kx::2, :] as a strided slice from a shifted base
                        _call("bf_strided_slice2", N.ptr(src), N.ptr(xs), Bc, Hc, Wc, cin, N.stream_ptr(x))
                        ops.matmul_wgrad(xs, dp, gw[ky * 2 + kx])
                return UL.conv2d_transpose(dp, w.contiguous(), 2, "linear")     # keras Conv2D kernel [k,k,cin,cout] = Conv2DTranspose's [k,k,out,in]
            return y, bwd

        def attention_gate(d, enc, up):
            """AdditiveAttentionGate (custom_layers.py:805-832) and the Add behind it (backbone_unet_laplacian.py:497-519):
            x = enc * sigmoid(4 * scale(conv_o(leaky_relu_0.1(conv_x(LN up) + conv_y(LN enc))))) + up.
            Returns (x, backward: dx -> (d enc, d up))."""
            Cc = enc.shape[-1]
            pre = f"gate{d}"
            gx = self.W(f"{pre}/x_ln/gamma") if m.use_ln else None
            gy = self.W(f"{pre}/y_ln/gamma") if m.use_ln else None
            wx, wy, wo = (self.W(f"{pre}/{n_}/kernel").view(Cc, Cc) for n_ in ("x", "y", "o"))
            lx = UL.dwconv_ln(up, None, gx) if m.use_ln else up
            ly = UL.dwconv_ln(enc, None, gy) if m.use_ln else enc
            z = ops.add(UL.pointwise(lx, pack(wx), Cc), UL.pointwise(ly, pack(wy), Cc))
            sact = UL.dwconv_ln(z, None, None, "leaky_relu_01")
            o_raw = UL.pointwise(sact, pack(wo), Cc)
            wm = self.W(f"{pre}/scale/w")
            mult = UL.channel_multiplier(wm)
            o = ops.scale_add(None, o_raw, mult, None)
            out = torch.empty_like(enc)
            _call("bf_op_sigmoid_gate", N.ptr(enc), N.ptr(o), N.ptr(up), N.ptr(out), enc.numel(), N.stream_ptr(enc))

            def bwd(dout):
                denc, do = torch.empty_like(enc), torch.empty_like(enc)
                _call("bf_op_sigmoid_gate_bwd", N.ptr(enc), N.ptr(o), N.ptr(dout), N.ptr(denc), N.ptr(do), enc.numel(), N.stream_ptr(enc))
                dm = torch.empty(Cc, dtype=torch.float32, device=dev)
                do_raw = ops.scale_add_bwd(o_raw, mult, None, do, dm)
                _call("bf_op_multiplier_bwd", N.ptr(wm), N.ptr(dm), N.ptr(self.G(f"{pre}/scale/w", grads)), Cc, N.stream_ptr(dm))
                ops.matmul_wgrad(sact, do_raw, self.G(f"{pre}/o/kernel", grads))
                dz = ops.act_bwd(sact, UL.pointwise(do_raw, pack(ops.transpose(wo)), Cc), "leaky_relu_01")
                ops.matmul_wgrad(lx, dz, self.G(f"{pre}/x/kernel", grads))
                ops.matmul_wgrad(ly, dz, self.G(f"{pre}/y/kernel", grads))
                dlx = UL.pointwise(dz, pack(ops.transpose(wx)), Cc)
                dly = UL.pointwise(dz, pack(ops.transpose(wy)), Cc)
                dup = ops.layernorm_bwd(up, gx, dlx, self.G(f"{pre}/x_ln/gamma", grads)) if m.use_ln else dlx
                dency = ops.layernorm_bwd(enc, gy, dly, self.G(f"{pre}/y_ln/gamma", grads)) if m.use_ln else dly
                return ops.add(denc, dency), ops.add(dout, dup)
            return out, bwd

        # -- forward ----------------------------------------------------------------------------------------------------------
        wb = self.W("base/kernel")
        x = UL.first_conv(noisy, wb, H, Wd, a, True, m.v_min, m.v_max, arith=0)
        x0 = x

        def base_bwd(dx):
            dpre = ops.act_bwd(x0, dx, a)
            sp, sn = ops._s()
            _call("bf_op_conv2d_wgrad", N.ptr(noisy), int(noisy.dtype == torch.uint8), N.ptr(dpre), N.ptr(self.G("base/kernel", grads)),
                  B, H, Wd, m.in_channels, m.filters, 5, 1, m.v_min, m.v_max, sp, sn, N.stream_ptr(dpre))
            return None
        # gradient bookkeeping: `flow` is a list of (forward value, backward closure chain); the graph below is a chain per
        # level with two fan-outs (the Laplacian split; the level outputs feeding a head and the next decoder level)
        enc_chain: List = []                # closures of the encoder path, applied in reverse to the gradient of its running value
        lap: Dict[int, torch.Tensor] = {}
        lap_bwd_at: Dict[int, int] = {}
        k_g = m.gauss_k
        gauss = None if m.use_laplacian_averaging else torch.from_numpy(UL.gaussian_kernel((k_g, k_g))).to(dev)
        split_info = {}
        for d in range(m.depth):
            for w_ in range(m.width):
                pre = f"enc{d}_{w_}"
                x, b_ = attention(pre, x) if m._is_attention(d) else convnext(pre, x, m.enc_k)
                enc_chain.append(("op", b_))
            la = a if m.level_activation else "linear"
            if m.use_output_normalization and m.use_ln and not m.output_norm_at_heads:
                x, b_ = norm_act(f"enc{d}/out_ln/gamma", x, la)
                enc_chain.append(("op", b_))
            elif la != "linear":
                y = x
                x = UL.dwconv_ln(y, None, None, la)
                enc_chain.append(("op", (lambda yy: (lambda dy: ops.act_bwd(yy, dy, la)))(x)))
            if d != m.depth - 1:
                ds = 2 if m.downsample_type == "strides" else 1              # conv2d / maxpool take the full-resolution smooth map
                lp, down = UL.smooth_split(x, k_g, gauss, ds)
                lap[d] = lp
                Bx, Hx, Wx, Cx = x.shape
                enc_chain.append(("split", d, (Bx, Hx, Wx, Cx), ds))
                if ds == 2:
                    x, b_ = conv1x1_act(f"down{d}/kernel", down, m.level_filters(d + 1), a)
                elif m.downsample_type == "maxpool":                        # MaxPooling2D(2, 2, same) + 1x1 (downsampling.py:56-68)
                    x, b_c = conv1x1_act(f"down{d}/kernel", UL.maxpool2(down), m.level_filters(d + 1), a)

                    def b_(dy, down=down, b_c=b_c):
                        dmp = b_c(dy)
                        dd = torch.empty_like(down)
                        _call("bf_op_maxpool2_bwd", N.ptr(down), N.ptr(dmp), N.ptr(dd), down.shape[0], down.shape[1], down.shape[2],
                              down.shape[3], N.stream_ptr(dmp))
                        return dd
                else:
                    x, b_ = conv2x2_s2_act(f"down{d}/kernel", down, m.level_filters(d + 1), a)
                enc_chain.append(("op", b_))
        deep = x                                                      # nodes[depth - 1]

        # decoder
        outs = {m.depth - 1: deep}
        dec_chain: Dict[int, List] = {}
        for d in reversed(range(m.depth - 1)):
            low = outs[d + 1]
            Cc = m.level_filters(d)
            bil = m.upsample_type != "upsample_nearest_conv2d"
            conv_first = m.upsample_type == "upsample_laplacian_conv2d" and m.upsample_linear   # upsampling.py:80-90: 1x1, then resize
            plain = m.upsample_type in ("bilinear", "nn", "nearest")          # UpSampling2D alone (upsampling.py:103-116)
            if plain:
                if low.shape[-1] != Cc:
                    raise ValueError(f"Add of [{Cc}] and [{low.shape[-1]}] channels: upsample_type [{m.upsample_type}] needs equal "
                                     f"filters on both levels")
                bil = m.upsample_type == "bilinear"
                up, b_up = upsample_2x(low, bilinear=bil), (lambda g_: g_)
            elif conv_first:
                c_, b_up = conv1x1_act(f"up{d}/kernel", low, Cc, "linear")
                up = upsample_2x(c_, bilinear=True)
            elif m.upsample_type == "upsample_laplacian_conv2d":
                up, b_up = conv1x1_act(f"up{d}/kernel", upsample_2x(low, bilinear=bil), Cc, a)
            else:
                up, b_up = conv3x3_act(f"up{d}/kernel", upsample_2x(low, bilinear=bil), Cc, a)
            b_gate = None
            if m.use_attention_gates:
                x, b_gate = attention_gate(d, lap[d], up)
            elif m.use_concat:                                                # Concatenate([encoder feature, upsampled]) (:516-517)
                x = torch.empty(up.shape[:-1] + (2 * Cc,), dtype=torch.float32, device=dev)
                _call("bf_op_concat_channels", N.ptr(lap[d]), N.ptr(up), None, N.ptr(x), up.numel() // Cc, Cc, Cc, 0, N.stream_ptr(up))
            else:
                x = ops.add(lap[d], up)
            chain = [("up", b_up, low.shape, bil, b_gate, conv_first)]
            if m.use_mix_project:                                             # backbone_unet_laplacian.py:521-527
                x, b_ = conv1x1_act(f"mix{d}/kernel", x, Cc, a)
                chain.append(("op", b_))
            for w_ in range(m.width):
                x, b_ = convnext(f"dec{d}_{w_}", x, m.dec_k)
                chain.append(("op", b_))
            if m.use_output_normalization and m.use_ln and not m.output_norm_at_heads:
                x, b_ = norm_act(f"dec{d}/out_ln/gamma", x, "linear")
                chain.append(("op", b_))
            outs[d] = x
            dec_chain[d] = chain

        # heads + losses
        ld = N.LossDesc()
        ld.struct_size = C.sizeof(N.LossDesc)
        lc = self.loss_config
        ld.hinge, ld.cutoff = float(lc.get("hinge", 0.0)), float(lc.get("cutoff", 255.0))
        ld.mae_multiplier, ld.mse_multiplier = float(lc.get("mae_multiplier", 1.0)), float(lc.get("mse_multiplier", 0.0))
        ld.ssim_multiplier, ld.regularization = float(lc.get("ssim_multiplier", 1.0)), float(lc.get("regularization", 1.0))
        gts = [gt]
        for _ in range(m.depth - 1):
            gts.append(avg_pool2_valid(gts[-1], clip_values=True, round_values=True))       # multiscales_generator_fn
        preds, scale_losses, dfeat = [], [], {}
        total = torch.zeros(3, dtype=torch.float32, device=dev)       # [0] total loss, [1] regularisation value, [2] [1] * regularization
        for i in range(m.depth):
            f, b_head_ln = outs[i], None
            if m.use_output_normalization and m.use_ln and m.output_norm_at_heads:     # the archive's graph: only the heads read the
                f, b_head_ln = norm_act(m._out_ln_name(i), f, "linear")                # normalised maps, the decoder the raw ones
            Bc, Hc, Wc, Cc = f.shape
            w0 = self.W(f"head{i}/conv0/kernel").view(Cc, m.head_filters)
            w1 = self.W(f"head{i}/conv1/kernel").view(m.head_filters, m.out_channels).contiguous()
            h0 = UL.pointwise(f, pack(w0), m.head_filters, m.head_activation)
            pred = UL.head_out(h0, w1, Hc, Wc, False, True, m.v_min, m.v_max)
            preds.append(pred)
            losses = torch.zeros(N.BF_LOSS_COUNT, dtype=torch.float32, device=dev)
            dpred = torch.empty_like(pred)
            ld.depth_weight = float(depth_weights[i])
            sp, sn = ops._s()
            _call("bf_op_denoiser_loss", N.ptr(pred), N.ptr(gts[i]), Bc, Hc, Wc, m.out_channels, C.byref(ld), N.ptr(dpred), N.ptr(losses),
                  sp, sn, N.stream_ptr(pred))
            scale_losses.append(losses)
            _call("bf_op_axpy", N.ptr(total), N.ptr(losses[N.BF_LOSS_TOTAL:N.BF_LOSS_TOTAL + 1]), 1.0, 0, 1, N.stream_ptr(total))
            dh0 = torch.empty_like(h0)
            sp, sn = ops._s()
            _call("bf_op_head_out_bwd", N.ptr(h0), N.ptr(w1), N.ptr(dpred), N.ptr(dh0), N.ptr(self.G(f"head{i}/conv1/kernel", grads)),
                  Bc * Hc * Wc, m.head_filters, m.out_channels, 1, m.v_min, m.v_max, sp, sn, N.stream_ptr(h0))
            dh0p = ops.act_bwd(h0, dh0, m.head_activation)
            ops.matmul_wgrad(f, dh0p, self.G(f"head{i}/conv0/kernel", grads))
            dfeat[i] = UL.pointwise(dh0p, pack(ops.transpose(w0)), Cc)
            if b_head_ln is not None:
                dfeat[i] = b_head_ln(dfeat[i])

        # -- backward ---------------------------------------------------------------------------------------------------------
        # decoder levels top (d = 0) to bottom: each yields the gradient of the Laplacian skip and of the level below
        dlap: Dict[int, torch.Tensor] = {}
        dlow = {}
        for d in range(m.depth - 1):
            g = dfeat[d] if d not in dlow else ops.add(dfeat[d], dlow[d])
            chain = dec_chain[d]
            for item in reversed(chain[1:]):
                g = item[1](g)
            _, b_up, low_shape, bil, b_gate, conv_first = chain[0]
            if b_gate is not None:
                dlap[d], g = b_gate(g)                                   # x = gate(lap[d], up) + up
            elif m.use_concat:                                          # x = [lap[d] | up]: the two halves of the gradient
                Ch = g.shape[-1] // 2
                halves = [torch.empty(g.shape[:-1] + (Ch,), dtype=torch.float32, device=dev) for _ in range(2)]
                for h_, o_ in zip(halves, (0, Ch)):
                    _call("bf_op_slice_channels", N.ptr(g), N.ptr(h_), g.numel() // (2 * Ch), 2 * Ch, o_, Ch, N.stream_ptr(g))
                dlap[d], g = halves
            else:
                dlap[d] = g                                             # x = lap[d] + up
            if conv_first:                                              # up = resize(conv(low))
                dc = torch.empty(tuple(low_shape[:3]) + (g.shape[-1],), dtype=torch.float32, device=dev)
                _call("bf_op_upsample2x_bwd", N.ptr(g), N.ptr(dc), low_shape[0], low_shape[1], low_shape[2], g.shape[-1], 1, N.stream_ptr(g))
                dlow[d + 1] = b_up(dc)
                continue
            du2 = b_up(g)
            dl = torch.empty(low_shape, dtype=torch.float32, device=dev)
            _call("bf_op_upsample2x_bwd", N.ptr(du2), N.ptr(dl), low_shape[0], low_shape[1], low_shape[2], low_shape[3], int(bil),
                  N.stream_ptr(du2))
            dlow[d + 1] = dl
        last = m.depth - 1
        g = dfeat[last] if last not in dlow else ops.add(dfeat[last], dlow[last])
        for item in reversed(enc_chain):
            if item[0] == "op":
                g = item[1](g)
            else:                                                      # the split of level d: g = d(down), dlap[d] = d(lap)
                _, d, (Bx, Hx, Wx, Cx), ds = item
                dx = torch.empty((Bx, Hx, Wx, Cx), dtype=torch.float32, device=dev)
                _call("bf_op_smooth_split_bwd_ex", N.ptr(dlap[d]), N.ptr(g), N.ptr(gauss), N.ptr(dx), Bx, Hx, Wx, Cx, k_g, ds,
                      N.stream_ptr(g))
                g = dx
        base_bwd(g)

        # -- regularisers: value into total[1], gradients added times `regularization` -------------------------------------------
        reg = float(ld.regularization)
        sob = torch.empty(2 * 512 * 512, dtype=torch.float32, device=dev)
        for name, shape, kind, off in m.trainable_variables:
            if kind == "ln_gamma":
                continue
            n = int(np.prod(shape))
            gslice = self._grad_view(name, grads)
            w = self.W(name)
            leaf = name.split("/")[1]
            is_gate = name.startswith("gate")            # AdditiveAttentionGate convolutions: soft-orthonormal with the flag, else l2(1e-4)
            soft = leaf in ("key", "query", "value", "out") or (leaf in ("pw1", "pw2") and self.soft_orthonormal) or \
                (is_gate and kind == "conv" and self.soft_orthonormal)
            if kind == "multiplier":
                _call("bf_op_reg_elementwise", N.ptr(w), N.ptr(gslice), n, N.BF_REG_L1, MULTIPLIER_L1, reg, N.ptr(total[1:2]), N.stream_ptr(w))
            elif soft:
                _call("bf_op_reg_soft_orthonormal", N.ptr(w), N.ptr(gslice), shape[2], shape[3], SOFTORTHONORMAL[0], SOFTORTHONORMAL[1],
                      SOFTORTHONORMAL[2], reg, N.ptr(total[1:2]), N.ptr(sob), N.stream_ptr(w))
            else:
                _call("bf_op_reg_elementwise", N.ptr(w), N.ptr(gslice), n, N.BF_REG_L2, GATE_L2 if is_gate else KERNEL_L2, reg,
                      N.ptr(total[1:2]), N.stream_ptr(w))
        for buf, off, n in self._unaligned:                            # staged gradients of tensors at unaligned offsets (a copy)
            grads[off:off + n].copy_(buf)
        _call("bf_op_axpy", N.ptr(total[2:3]), N.ptr(total[1:2]), reg, 0, 1, N.stream_ptr(total))
        _call("bf_op_axpy", N.ptr(total), N.ptr(total[2:3]), 1.0, 0, 1, N.stream_ptr(total))
        self.totals = total
        return preds, scale_losses, total

    def _grad_view(self, name, grads):
        off, shape, _ = self.off[name]
        n = int(np.prod(shape))
        for buf, o, nn in self._unaligned:
            if o == off:
                return buf
        return grads[off:off + n]
