"""
Training of the resnet backbones outside the 16-filter 3x3 engine (`GenericResnetHydra`: the shipped bottleneck / depthwise
config `resnet_color_1x6_bn_32x128x32_1x3x1_..._depthwise`, per-position kernels / filters / groups, `add_gates`):
bfcnn/train_loop.py:259-312 -- training-mode forward (BatchNormalization on batch statistics, moving statistics updated),
denoiser loss, the builder's regularisers, and the gradient of the total for every trainable tensor -- as an explicit
forward / backward walk of the graph of bfcnn/backbone_resnet.py:36-298 + backbone_blocks.py:163-246 over the operator
library's C-ABI entry points (forward operators of unet_ops.hip, backward primitives of train_prims.hip, BatchNorm / gate /
layout operators of train_generic.hip).  PyTorch holds the tensors; it computes nothing.

Exact fp32.  Gradients are compared with the torch-autograd oracle (oracle/resnet_generic_torch.py) in
tests/test_gpu_resnet_generic_train.py.
"""
import ctypes as C
from typing import Dict

import numpy as np
import torch

from . import _native as N
from . import unet_laplacian as UL
from .resnet_generic import BN_EPSILON, SELECTOR_GLOBAL_LEAKY, GenericResnetHydra
from .unet_train import _Ops, _call

BN_MOMENTUM = 0.995                # DEFAULT_BN_MOMENTUM (bfcnn/constants.py:10)
REG_COEF = 0.01                    # keras "l1" / "l2" string regularisers
CHANNELWISE_L1 = 0.1               # DEFAULT_CHANNELWISE_MULTIPLIER_L1 (bfcnn/constants.py:13)
MULTIPLIER_L1 = 1.0                # DEFAULT_MULTIPLIER_L1 (:12)


class GenericResnetTrainGraph:
    """train_step_single_gpu for a GenericResnetHydra: `step(gt, noisy, grads)` returns (prediction, loss slots, totals[3]) and
    fills `grads` (flat, laid out like model.params); model.state (moving statistics) is updated in place."""

    def __init__(self, model: GenericResnetHydra, loss_config: Dict):
        self.m = model
        self.loss_config = dict(loss_config)
        self.off = {name: (off, shape, kind) for name, shape, kind, off in model.trainable_variables}
        self.soff = {name: (off, shape) for name, shape, off in model.non_trainable_variables}
        self.ops = None
        self.totals = None

    # ---- parameters / state ------------------------------------------------------------------------------------------------
    def W(self, name) -> torch.Tensor:
        off, shape, _ = self.off[name]
        n = int(np.prod(shape))
        t = self.m.params[off:off + n]
        if off % 4:
            t = t.clone()
        return t.view(shape)

    def G(self, name, grads) -> torch.Tensor:
        off, shape, _ = self.off[name]
        n = int(np.prod(shape))
        if off % 4:
            buf = torch.empty(n, dtype=torch.float32, device=grads.device)
            self._unaligned.append((buf, off, n))
            return buf
        return grads[off:off + n]

    def _grad_view(self, name, grads):
        off, shape, _ = self.off[name]
        for buf, o, nn in self._unaligned:
            if o == off:
                return buf
        return grads[off:off + int(np.prod(shape))]

    def S(self, name) -> torch.Tensor:
        off, shape = self.soff[name]
        return self.m.state[off:off + int(np.prod(shape))]

    def regularizer(self, name: str, kind: str):
        """backbone_resnet.py:128-176, backbone_blocks.py:146-160, model.py:297-342"""
        bb, dn = self.m.config["backbone"], self.m.config["denoiser"]
        if kind == "bn_gamma":
            return None
        if kind in ("channelwise", "multiplier"):                  # handled where the regularisers are summed (their own coefficients)
            return kind
        if name.startswith("base/"):
            return bb.get("kernel_regularizer", "l1")
        if name.startswith("head/"):
            return dn.get("kernel_regularizer", "l2")
        if "/gate/" in name:
            return "l2"
        if "/selector/" in name:                                   # custom_layers_selector.py:88
            return (bb.get("selector_params") or {}).get("kernel_regularizer", "l1")
        j = int(name.split("/")[1][4:])
        br = bb.get("block_regularizer") or [bb.get("kernel_regularizer", "l1")] * len(self.m.block_kernels)
        return br[j]

    # ---- one training step -------------------------------------------------------------------------------------------------
    def step(self, gt: torch.Tensor, noisy: torch.Tensor, grads: torch.Tensor, depth_weight: float = 1.0, drop_scale=None):
        """drop_scale: {block index: per-sample factor [B] on the device} = RandomOnOff's draw (0 or 1 / (1 - rate))"""
        m = self.m
        drop_scale = drop_scale or {}
        dev = m.device
        gt = gt.to(device=dev, dtype=torch.float32).contiguous()
        noisy = noisy.to(device=dev).contiguous()
        if noisy.dtype != torch.uint8:
            noisy = noisy.to(torch.float32)
        B, H, Wd, _ = noisy.shape
        npix = B * H * Wd
        L = N.lib()
        cmax = max([m.filters] + [c for c in self._channels()])
        need = max(8 * 1024 * 1024, int(L.bf_op_denoiser_loss_scratch_floats(B, H, Wd, m.out_channels)) + 1024, npix * 4,
                   int(L.bf_op_gate_scratch_floats(B, cmax)) + 64, int(L.bf_op_bn_train_scratch_floats(cmax)) + 64,
                   npix * m.filters if m.selector else 0)        # the selector's resize / dense adjoints (at most one row set per pixel)
        if self.ops is None or self.ops.scratch.numel() < need:
            self.ops = _Ops(dev, need)
        ops = self.ops
        self._unaligned = []
        nb = len(m.block_kernels)
        f32 = dict(dtype=torch.float32, device=dev)

        def pack(w2d):
            return UL.pack_pointwise(w2d.contiguous())

        def conv_op(name, x, j):
            """convolution j of a block (no normalisation / activation): returns (y, backward closure dy -> dx)"""
            kk, cf, dm, g = m.block_kernels[j], m.block_filters[j], m.block_depthwise[j], m.block_groups[j]
            cin = x.shape[-1]
            w = self.W(name)
            if dm != -1:
                # DepthwiseConv2D(depth_multiplier = dm) = plain depthwise convolution of the channel-repeated tensor
                y = UL.dwconv_mult(x, w, None)

                def bwd(dy):
                    Bc, Hc, Wc, _ = x.shape
                    xr = x
                    if dm != 1:
                        xr = torch.empty((Bc, Hc, Wc, cin * dm), **f32)
                        _call("bf_op_channel_repeat", N.ptr(x), N.ptr(xr), Bc * Hc * Wc, cin, dm, N.stream_ptr(x))
                    ops.dwconv_wgrad(xr, dy, self.G(name, grads), kk)                  # [k,k,C*dm] == [k,k,C,dm] in memory
                    wf = torch.empty_like(w)
                    _call("bf_op_flip_hw", N.ptr(w), N.ptr(wf), kk, cin * dm, N.stream_ptr(w))
                    dxr = UL.dwconv_mult(dy, wf.view(kk, kk, cin * dm, 1), None)
                    if dm == 1:
                        return dxr
                    dx = torch.empty_like(x)
                    _call("bf_op_channel_group_sum", N.ptr(dxr), N.ptr(dx), Bc * Hc * Wc, cin, dm, N.stream_ptr(dxr))
                    return dx
                return y, bwd
            if kk == 1:
                dense = w.view(cin // g, cf)
                if g != 1:
                    dense = torch.empty((cin, cf), **f32)
                    _call("bf_op_group_kernel", N.ptr(w), N.ptr(dense), cin, cf, g, 0, N.stream_ptr(w))
                y = UL.pointwise(x, pack(dense), cf)

                def bwd(dy):
                    gw = self.G(name, grads)
                    if g == 1:
                        ops.matmul_wgrad(x, dy, gw)
                    else:
                        dd = torch.empty((cin, cf), **f32)
                        ops.matmul_wgrad(x, dy, dd)
                        _call("bf_op_group_kernel", N.ptr(gw), N.ptr(dd), cin, cf, g, 1, N.stream_ptr(dd))
                    return UL.pointwise(dy, pack(ops.transpose(dense)), cin)
                return y, bwd
            # k x k convolution; Conv2D(groups = g) as the block-diagonal dense one, tap by tap (keras kernel [k][k][cin / g][cf]): the
            # kernel gradient is taken dense and its diagonal blocks are written back (backbone_blocks.py:196-205, conv2d groups)
            wd = w
            if g != 1:
                wd = torch.empty((kk, kk, cin, cf), **f32)
                for t_ in range(kk * kk):
                    _call("bf_op_group_kernel", N.ptr(w.view(kk * kk, cin // g, cf)[t_]), N.ptr(wd.view(kk * kk, cin, cf)[t_]), cin, cf, g, 0,
                          N.stream_ptr(w))
            y = UL.conv2d(x, UL.pack_conv(wd.contiguous()), cf, kk, 1, "linear")

            def bwd(dy, w=wd):
                Bc, Hc, Wc, _ = x.shape
                sp, sn = ops._s()
                gw = self.G(name, grads)
                gd = gw if g == 1 else torch.empty((kk, kk, cin, cf), **f32)
                _call("bf_op_conv2d_wgrad", N.ptr(x), 0, N.ptr(dy), N.ptr(gd), Bc, Hc, Wc, cin, cf, kk, 0, 0.0, 0.0,
                      sp, sn, N.stream_ptr(dy))
                if g != 1:
                    for t_ in range(kk * kk):
                        _call("bf_op_group_kernel", N.ptr(gw.view(kk * kk, cin // g, cf)[t_]), N.ptr(gd.view(kk * kk, cin, cf)[t_]), cin, cf, g, 1,
                              N.stream_ptr(gd))
                # data gradient = convolution with the taps flipped and every tap transposed
                wf = torch.empty_like(w)
                _call("bf_op_flip_hw", N.ptr(w), N.ptr(wf), kk, cin * cf, N.stream_ptr(w))
                wt = torch.empty((kk, kk, cf, cin), **f32)
                for t_ in range(kk * kk):
                    _call("bf_op_transpose2d", N.ptr(wf.view(kk * kk, cin, cf)[t_]), N.ptr(wt.view(kk * kk, cf, cin)[t_]), cin, cf,
                          N.stream_ptr(wf))
                return UL.conv2d(dy, UL.pack_conv(wt), cin, kk, 1, "linear")
            return y, bwd

        def bn_step(base, c, a):
            """BatchNormalization on batch statistics (+ activation a), moving statistics updated: (y, backward dy -> dc)"""
            code, alpha = UL._act(a)
            Cc = c.shape[-1]
            gamma = self.W(base + "/gamma")
            save = torch.empty(2 * Cc, **f32)
            y = torch.empty_like(c)
            sp, sn = ops._s()
            _call("bf_op_bn_train_fwd", N.ptr(c), N.ptr(gamma), N.ptr(y), N.ptr(save), N.ptr(self.S(base + "/moving_mean")),
                  N.ptr(self.S(base + "/moving_variance")), c.numel() // Cc, Cc, BN_EPSILON, BN_MOMENTUM, code, alpha,
                  sp, sn, N.stream_ptr(c))

            def bwd(dy):
                dpre = ops.act_bwd(y, dy, a)
                dx = torch.empty_like(c)
                sp, sn = ops._s()
                _call("bf_op_bn_train_bwd", N.ptr(c), N.ptr(gamma), N.ptr(save), N.ptr(dpre), N.ptr(dx), N.ptr(self.G(base + "/gamma", grads)),
                      c.numel() // Cc, Cc, sp, sn, N.stream_ptr(c))
                return dx
            return y, bwd

        def mult_step(name, t, s_=None):
            """ChannelwiseMultiplier / Multiplier: t * relu(w0 + 1) [* the per-sample RandomOnOff factor]; name None: the factor alone"""
            Cc = t.shape[-1]
            mvec = None
            if name is not None:
                w0 = self.W(name)
                nw = w0.numel()
                mvec = torch.empty(Cc, **f32)
                _call("bf_op_relu_shift", N.ptr(w0), nw, 1.0, N.ptr(mvec), Cc, N.stream_ptr(mvec))
            y = ops.scale_add(None, t, mvec, s_)

            def bwd(dy):
                dm = torch.empty(Cc, **f32) if name is not None else None
                dt = ops.scale_add_bwd(t, mvec, s_, dy, dm)
                if name is not None:
                    _call("bf_op_relu_shift_bwd", N.ptr(w0), nw, 1.0, N.ptr(dm), N.ptr(self.G(name, grads)), Cc, N.stream_ptr(dm))
                return dt
            return y, bwd

        def prefilter_step(i, x, pre, pool, Ct, chain_):
            """the selector's optional pre-filters (custom_layers_selector.py:160-185) forward, their adjoints appended to chain_"""
            from .resnet_generic import SELECTOR_EPSILON
            if pre.get("conv1x1"):
                name = f"block{i}/selector/pre/kernel"
                xin, w = x, self.W(name).view(x.shape[-1], Ct)
                x = UL.pointwise(xin, pack(w), Ct)

                def b_conv(dy, xin=xin, w=w, name=name):
                    ops.matmul_wgrad(xin, dy, self.G(name, grads))
                    return UL.pointwise(dy, pack(ops.transpose(w)), int(xin.shape[-1]))
                chain_.append(b_conv)
            Bx, Hx, Wx, Cx = x.shape
            if pre.get("gn"):                                    # per sample: BatchNorm's batch-statistics forward / backward with gamma 1
                xin = x
                ones, mm, mv = (torch.ones(Cx, **f32) for _ in range(3))
                saves = torch.empty((Bx, 2 * Cx), **f32)
                x = torch.empty_like(xin)
                for b in range(Bx):
                    xb, ob, sb = xin[b], x[b], saves[b]
                    sp_, sn_ = ops._s()
                    _call("bf_op_bn_train_fwd", N.ptr(xb), N.ptr(ones), N.ptr(ob), N.ptr(sb), N.ptr(mm), N.ptr(mv), Hx * Wx, Cx, SELECTOR_EPSILON,
                          0.0, 0, 0.0, sp_, sn_, N.stream_ptr(xin))

                def b_gn(dy, xin=xin, ones=ones, saves=saves):
                    dx = torch.empty_like(xin)
                    dgamma = torch.empty(Cx, **f32)
                    for b in range(Bx):
                        xb, db, ob, sb = xin[b], dy[b], dx[b], saves[b]
                        sp_, sn_ = ops._s()
                        _call("bf_op_bn_train_bwd", N.ptr(xb), N.ptr(ones), N.ptr(sb), N.ptr(db), N.ptr(ob), N.ptr(dgamma), Hx * Wx, Cx, sp_, sn_,
                              N.stream_ptr(xin))
                    return dx
                chain_.append(b_gn)
            if pre.get("ln"):
                xin = x

                def pooled(t):
                    o = torch.empty_like(t)
                    _call("bf_op_avgpool_same", N.ptr(t), N.ptr(o), Bx, Hx, Wx, Cx, pool[0], pool[1], 1, 1, N.stream_ptr(t))
                    return o

                def pooled_t(t):                                 # the pooling's adjoint
                    o = torch.empty_like(t)
                    _call("bf_op_avgpool_same_bwd", N.ptr(t), N.ptr(o), Bx, Hx, Wx, Cx, pool[0], pool[1], 1, 1, 0, N.stream_ptr(t))
                    return o
                mean = pooled(xin)
                sq = torch.empty_like(xin)
                _call("bf_op_center_scale", N.ptr(xin), N.ptr(mean), None, N.ptr(sq), xin.numel(), SELECTOR_EPSILON, N.stream_ptr(xin))
                var = pooled(sq)
                x = torch.empty_like(xin)
                _call("bf_op_center_scale", N.ptr(xin), N.ptr(mean), N.ptr(var), N.ptr(x), xin.numel(), SELECTOR_EPSILON, N.stream_ptr(xin))

                def b_ln(dy, xin=xin, mean=mean, var=var):
                    dd, dv = torch.empty_like(xin), torch.empty_like(xin)
                    _call("bf_op_center_scale_bwd", N.ptr(xin), N.ptr(mean), N.ptr(var), N.ptr(dy), N.ptr(dd), N.ptr(dv), xin.numel(),
                          SELECTOR_EPSILON, N.stream_ptr(dy))
                    t = pooled_t(dv)
                    tot = torch.empty_like(xin)
                    _call("bf_op_center_sq_bwd", N.ptr(xin), N.ptr(mean), N.ptr(t), N.ptr(dd), N.ptr(tot), xin.numel(), N.stream_ptr(dd))
                    back = pooled_t(tot)
                    _call("bf_op_axpy", N.ptr(tot), N.ptr(back), -1.0, 0, tot.numel(), N.stream_ptr(tot))     # d (x - pool x)
                    return tot
                chain_.append(b_ln)
            for key, high in (("lp", 0), ("hp", 1)):
                if pre.get(key):
                    xin = x
                    x = torch.empty_like(xin)
                    _call("bf_op_pass_filter", N.ptr(xin), N.ptr(x), xin.numel(), 4.0, 4, high, N.stream_ptr(xin))

                    def b_pf(dy, xin=xin, high=high):
                        dx = torch.empty_like(xin)
                        _call("bf_op_pass_filter_bwd", N.ptr(xin), N.ptr(dy), N.ptr(dx), xin.numel(), 4.0, 4, high, N.stream_ptr(dy))
                        return dx
                    chain_.append(b_pf)
            return x

        def selector_step(i, x1, x2, sel):
            """selector_block (custom_layers_selector.py:81-330) in place of the skip Add: out = x1 s + x2 (1 - s), s = F(2.5 - u),
            u = up(relu(leaky(pool(sel) W0) W1)).  scale_type GLOBAL is the same chain with one window over the whole image
            (Dense layers, slope SELECTOR_GLOBAL_LEAKY).  Returns (out, backward: d out -> (d x1, d x2, d sel))."""
            sp = m.selector
            st, soft = sp["scale_type"], int(sp["activation_type"] == "soft")
            Ct = x1.shape[-1]
            pre_chain = []                                       # adjoints of the optional pre-filters, in forward order
            if sp.get("pre"):
                sel = prefilter_step(i, sel, sp["pre"], sp["pool"], Ct, pre_chain)
            Bs, Hs, Ws, Cs = sel.shape
            if st == "global":
                stride, pools, alpha0, kind = (Hs, Ws), [(Hs, Ws)], SELECTOR_GLOBAL_LEAKY, "dense"
            else:
                stride, pool, alpha0, kind = sp["stride"], sp["pool"], 0.3, "conv"
                if Hs % stride[0] or Ws % stride[1]:
                    raise ValueError(f"selector_block: the image ({Hs}x{Ws}) must be a multiple of the strides {stride}")
                pools = [(pool[0] // 2, pool[1] // 2), pool, (pool[0] * 2, pool[1] * 2)] if st == "multiscale" else [pool]
            OH, OW = Hs // stride[0], Ws // stride[1]
            rows = Bs * OH * OW
            parts = []
            for ph, pw in pools:
                pm = torch.empty((Bs, OH, OW, Cs), **f32)
                _call("bf_op_avgpool_same", N.ptr(sel), N.ptr(pm), Bs, Hs, Ws, Cs, ph, pw, stride[0], stride[1], N.stream_ptr(sel))
                parts.append(pm)
            if st == "mixed":                                    # local means next to the image's global mean on the same grid
                gm = torch.empty((Bs, OH, OW, Cs), **f32)
                sp_, sn_ = ops._s()
                _call("bf_op_channel_mean_broadcast", N.ptr(sel), N.ptr(gm), Bs, Hs * Ws, Cs, OH * OW, sp_, sn_, N.stream_ptr(sel))
                parts.append(gm)
            if len(parts) > 1:
                cat = torch.empty((Bs, OH, OW, Cs * len(parts)), **f32)
                _call("bf_op_concat_channels", N.ptr(parts[0]), N.ptr(parts[1]), N.ptr(parts[2]) if len(parts) > 2 else None, N.ptr(cat),
                      rows, Cs, Cs, Cs if len(parts) > 2 else 0, N.stream_ptr(sel))
            else:
                cat = parts[0]
            Cc = cat.shape[-1]
            n0, n1 = f"block{i}/selector/{kind}0/kernel", f"block{i}/selector/{kind}1/kernel"
            w0, w1 = self.W(n0), self.W(n1)
            C8 = int(w0.shape[-1])
            u = torch.empty((Bs, OH, OW, Ct), **f32)
            _call("bf_op_dense2", N.ptr(cat), N.ptr(w0), None, N.ptr(w1), None, N.ptr(u), rows, Cc, Ct, C8, 2, alpha0, 4, N.stream_ptr(cat))
            up = UL.resize_bilinear(u, Hs, Ws)
            out = torch.empty_like(x1)
            _call("bf_op_selector_mix", N.ptr(x1), N.ptr(x2), N.ptr(up), N.ptr(out), x1.numel(), soft, N.stream_ptr(x1))

            def bwd(dout):
                dx1, dx2, dup = torch.empty_like(x1), torch.empty_like(x1), torch.empty_like(x1)
                _call("bf_op_selector_mix_bwd", N.ptr(x1), N.ptr(x2), N.ptr(up), N.ptr(dout), N.ptr(dx1), N.ptr(dx2), N.ptr(dup), x1.numel(),
                      soft, N.stream_ptr(dout))
                du = torch.empty_like(u)
                _call("bf_op_resize_bilinear_bwd", N.ptr(dup), N.ptr(du), Bs, OH, OW, Ct, Hs, Ws, N.ptr(ops.scratch), N.stream_ptr(dup))
                dcat = torch.empty_like(cat)
                sp_, sn_ = ops._s()
                _call("bf_op_dense2_bwd", N.ptr(cat), N.ptr(w0), N.ptr(w1), N.ptr(du), N.ptr(dcat), N.ptr(self.G(n0, grads)),
                      N.ptr(self.G(n1, grads)), rows, Cc, Ct, C8, alpha0, sp_, sn_, N.stream_ptr(du))
                dsel = torch.empty_like(sel)
                for k_, (ph, pw) in enumerate(pools):
                    dpart = dcat
                    if len(parts) > 1:
                        dpart = torch.empty((Bs, OH, OW, Cs), **f32)
                        _call("bf_op_slice_channels", N.ptr(dcat), N.ptr(dpart), rows, Cc, k_ * Cs, Cs, N.stream_ptr(dcat))
                    _call("bf_op_avgpool_same_bwd", N.ptr(dpart), N.ptr(dsel), Bs, Hs, Ws, Cs, ph, pw, stride[0], stride[1], int(k_ > 0),
                          N.stream_ptr(dpart))
                if st == "mixed":                                # d mean: the column sums of its gradient, spread over the image
                    dgm = torch.empty((Bs, OH, OW, Cs), **f32)
                    _call("bf_op_slice_channels", N.ptr(dcat), N.ptr(dgm), rows, Cc, Cs, Cs, N.stream_ptr(dcat))
                    spread = torch.empty_like(sel)
                    sp_, sn_ = ops._s()
                    _call("bf_op_channel_mean_broadcast", N.ptr(dgm), N.ptr(spread), Bs, OH * OW, Cs, Hs * Ws, sp_, sn_, N.stream_ptr(dgm))
                    _call("bf_op_axpy", N.ptr(dsel), N.ptr(spread), float(OH * OW) / float(Hs * Ws), 0, dsel.numel(), N.stream_ptr(dsel))
                for b_pre in reversed(pre_chain):
                    dsel = b_pre(dsel)
                return dx1, dx2, dsel
            return out, bwd

        # -- forward -------------------------------------------------------------------------------------------------------------
        wb = self.W("base/kernel")
        # GELU is not sign-preserving: its derivative needs the pre-activation, so the convolution output is kept and the activation
        # runs as its own pass (the other activations are fused and differentiated from their output)
        gelu = lambda a_: UL._act(a_)[0] == 3
        act_only = lambda t_, a_: UL.dwconv_ln(t_.view(1, 1, -1, 32), None, None, a_).view(t_.shape)
        if gelu(m.base_activation):
            f0pre = UL.first_conv(noisy, wb, H, Wd, "linear", True, m.v_min, m.v_max, arith=0)
            f = act_only(f0pre, m.base_activation)
        else:
            f0pre = None
            f = UL.first_conv(noisy, wb, H, Wd, m.base_activation, True, m.v_min, m.v_max, arith=0)
        f0 = f
        chain = []                                   # per block: closure d(block output) -> d(block input)
        if m.add_initial_bn:
            f, b_ = bn_step("initial_bn", f, "linear")
            chain.append(b_)
        for i in range(m.no_layers):
            t = f
            steps = []
            for j in range(nb):
                c, b_conv = conv_op(f"block{i}/conv{j}/kernel", t, j)
                a = m.block_activation[j]
                code, alpha = UL._act(a)
                Cc = c.shape[-1]
                if j >= 1 and m.use_bn:
                    gamma = self.W(f"block{i}/bn{j}/gamma")
                    save = torch.empty(2 * Cc, **f32)
                    y = torch.empty_like(c)
                    sp, sn = ops._s()
                    _call("bf_op_bn_train_fwd", N.ptr(c), N.ptr(gamma), N.ptr(y), N.ptr(save), N.ptr(self.S(f"block{i}/bn{j}/moving_mean")),
                          N.ptr(self.S(f"block{i}/bn{j}/moving_variance")), c.numel() // Cc, Cc, BN_EPSILON, BN_MOMENTUM,
                          0 if gelu(a) else code, alpha, sp, sn, N.stream_ptr(c))
                    ypre = None
                    if gelu(a):
                        ypre, y = y, act_only(y, a)

                    def b_norm(dy, c=c, y=y, ypre=ypre, gamma=gamma, save=save, a=a, Cc=Cc, name=f"block{i}/bn{j}/gamma"):
                        dpre = ops.act_bwd(y, dy, a, ypre)
                        dx = torch.empty_like(c)
                        sp, sn = ops._s()
                        _call("bf_op_bn_train_bwd", N.ptr(c), N.ptr(gamma), N.ptr(save), N.ptr(dpre), N.ptr(dx), N.ptr(self.G(name, grads)),
                              c.numel() // Cc, Cc, sp, sn, N.stream_ptr(c))
                        return dx
                else:
                    y = c if code == 0 else act_only(c, a)

                    def b_norm(dy, y=y, c=c, a=a):
                        return ops.act_bwd(y, dy, a, c)
                steps.append((b_conv, b_norm))
                t = y
                if j == 0:
                    first, n_first = t, len(steps)                    # x_1st_conv: the selector layer; steps[:n_first] produce it
                if j == 1 and m.add_gates:
                    w0, w1 = self.W(f"block{i}/gate/dense0/kernel"), self.W(f"block{i}/gate/dense1/kernel")
                    C8 = w0.shape[1]
                    gsave = torch.empty(int(L.bf_op_gate_save_floats(B, Cc, C8)), **f32)
                    gout = torch.empty_like(t)
                    sp, sn = ops._s()
                    _call("bf_op_gate_fwd", N.ptr(t), N.ptr(w0), N.ptr(w1), None, N.ptr(gout), N.ptr(gsave), B, H * Wd, Cc, C8, sp, sn,
                          N.stream_ptr(t))

                    def b_gate(dy, t=t, w0=w0, w1=w1, gsave=gsave, Cc=Cc, C8=C8, i=i):
                        dx = torch.empty_like(t)
                        sp, sn = ops._s()
                        _call("bf_op_gate_bwd", N.ptr(t), N.ptr(w0), N.ptr(w1), N.ptr(gsave), N.ptr(dy), N.ptr(dx),
                              N.ptr(self.G(f"block{i}/gate/dense0/kernel", grads)), N.ptr(self.G(f"block{i}/gate/dense1/kernel", grads)),
                              B, H * Wd, Cc, C8, sp, sn, N.stream_ptr(t))
                        return dx
                    steps.append((None, b_gate))
                    t = gout
            # backbone_blocks.py:215-225: ChannelwiseMultiplier, Multiplier, RandomOnOff in front of the Add
            tails = ([f"block{i}/channelwise/w0"] if m.add_channelwise else []) + ([f"block{i}/multiplier/w0"] if m.add_multiplier else [])
            ds_ = drop_scale.get(i)
            if ds_ is not None and not tails:
                tails = [None]
            for q_, name_ in enumerate(tails):
                t, b_ = mult_step(name_, t, ds_ if q_ == len(tails) - 1 else None)
                steps.append((None, b_))
            b_sel = None
            if m.selector:                                            # selector_block instead of the Add (backbone_blocks.py:227-239)
                f, b_sel = selector_step(i, f, t, first)
            else:
                f = ops.add(f, t)                                     # Add()([x, previous_layer]) (backbone_blocks.py:242)

            def b_block(dout, steps=steps, b_sel=b_sel, n_first=n_first):
                if b_sel is not None:
                    dskip, g, dsel = b_sel(dout)
                else:
                    dskip, g, dsel = dout, dout, None
                for idx in range(len(steps) - 1, -1, -1):
                    if dsel is not None and idx == n_first - 1:       # g is the gradient at the first convolution's output here
                        g = ops.add(g, dsel)
                    b_conv, b_norm = steps[idx]
                    g = b_norm(g)
                    if b_conv is not None:
                        g = b_conv(g)
                return ops.add(dskip, g)
            chain.append(b_block)

        if m.add_final_bn:                                            # backbone_resnet.py:274-287
            f, b_ = bn_step("final_bn", f, "linear")
            chain.append(b_)
        Cf = m.filters
        if m.add_concat_input:
            # Concatenate([features, the backbone's normalised input]) (backbone_resnet.py:277-279) as the inference path runs it: padded
            # with zero channels to the width the head's matrix kernel takes; the closing multipliers and the head's first kernel are
            # padded likewise (factor 1 / zero rows) and their gradients sliced back.  The image carries no gradient.
            cin = m.in_channels
            cf, Cp = Cf + cin, next(c for c in (32, 64, 128, 256) if c >= Cf + cin)
            cat = torch.empty((B, H, Wd, Cp), **f32)
            _call("bf_op_concat_input", N.ptr(f), N.ptr(noisy), int(noisy.dtype == torch.uint8), N.ptr(cat), B, H, Wd, H, Wd, Cf, cin, Cp,
                  m.v_min, m.v_max, N.stream_ptr(f))

            def b_cat(dcat):
                df = torch.empty((B, H, Wd, Cf), **f32)
                _call("bf_op_slice_channels", N.ptr(dcat), N.ptr(df), npix, Cp, 0, Cf, N.stream_ptr(dcat))
                return df
            chain.append(b_cat)
            f = cat
            ones_pad = torch.ones(Cp - cf, **f32) if Cp > cf else None

            def padded_mult(name_, t):
                """mult_step on the padded tensor: the factor's cf entries (or its one scalar) followed by ones"""
                w0_ = self.W(name_)
                nw = w0_.numel()
                mv = torch.empty(cf, **f32)
                _call("bf_op_relu_shift", N.ptr(w0_), nw, 1.0, N.ptr(mv), cf, N.stream_ptr(mv))
                mp = mv
                if ones_pad is not None:
                    mp = torch.empty(Cp, **f32)
                    _call("bf_op_concat_channels", N.ptr(mv), N.ptr(ones_pad), None, N.ptr(mp), 1, cf, Cp - cf, 0, N.stream_ptr(mv))
                y = ops.scale_add(None, t, mp, None)

                def bwd(dy):
                    dmp = torch.empty(Cp, **f32)
                    dt = ops.scale_add_bwd(t, mp, None, dy, dmp)
                    _call("bf_op_relu_shift_bwd", N.ptr(w0_), nw, 1.0, N.ptr(dmp), N.ptr(self.G(name_, grads)), cf, N.stream_ptr(dmp))
                    return dt
                return y, bwd
        for name_ in (["channelwise/w0"] if m.add_channelwise else []) + (["multiplier/w0"] if m.add_multiplier else []):
            f, b_ = padded_mult(name_, f) if m.add_concat_input else mult_step(name_, f)
            chain.append(b_)

        # -- head + loss ---------------------------------------------------------------------------------------------------------
        ld = N.LossDesc()
        ld.struct_size = C.sizeof(N.LossDesc)
        lc = self.loss_config
        ld.hinge, ld.cutoff = float(lc.get("hinge", 0.0)), float(lc.get("cutoff", 255.0))
        ld.mae_multiplier, ld.mse_multiplier = float(lc.get("mae_multiplier", 1.0)), float(lc.get("mse_multiplier", 0.0))
        ld.ssim_multiplier, ld.regularization = float(lc.get("ssim_multiplier", 0.0)), float(lc.get("regularization", 1.0))
        ld.depth_weight = float(depth_weight)
        w0 = self.W("head/conv0/kernel").view(-1, m.head_filters)
        Ch = int(f.shape[-1])                                         # Cf, or the padded width behind add_concat_input
        if Ch != w0.shape[0]:                                         # zero rows for the padding channels
            w0p = torch.empty((Ch, m.head_filters), **f32)
            zrows = torch.zeros((Ch - w0.shape[0]) * m.head_filters, **f32)
            _call("bf_op_concat_channels", N.ptr(w0), N.ptr(zrows), None, N.ptr(w0p), 1, w0.numel(), zrows.numel(), 0, N.stream_ptr(w0))
            w0 = w0p
        w1 = self.W("head/conv1/kernel").view(m.head_filters, m.out_channels).contiguous()
        h0 = UL.pointwise(f, pack(w0), m.head_filters, m.head_activation)
        pred = UL.head_out(h0, w1, H, Wd, False, True, m.v_min, m.v_max)
        losses = torch.zeros(N.BF_LOSS_COUNT, **f32)
        dpred = torch.empty_like(pred)
        total = torch.zeros(3, **f32)                                # [0] total loss, [1] regularisation value, [2] [1] * regularization
        sp, sn = ops._s()
        _call("bf_op_denoiser_loss", N.ptr(pred), N.ptr(gt), B, H, Wd, m.out_channels, C.byref(ld), N.ptr(dpred), N.ptr(losses), sp, sn,
              N.stream_ptr(pred))
        _call("bf_op_axpy", N.ptr(total), N.ptr(losses[N.BF_LOSS_TOTAL:N.BF_LOSS_TOTAL + 1]), 1.0, 0, 1, N.stream_ptr(total))
        dh0 = torch.empty_like(h0)
        sp, sn = ops._s()
        _call("bf_op_head_out_bwd", N.ptr(h0), N.ptr(w1), N.ptr(dpred), N.ptr(dh0), N.ptr(self.G("head/conv1/kernel", grads)), npix,
              m.head_filters, m.out_channels, 1, m.v_min, m.v_max, sp, sn, N.stream_ptr(h0))
        dh0p = ops.act_bwd(h0, dh0, m.head_activation)
        g0 = self.G("head/conv0/kernel", grads)
        if Ch * m.head_filters != g0.numel():                         # the padded rows' gradient is dropped (those channels are zero)
            gp = torch.empty(Ch * m.head_filters, **f32)
            ops.matmul_wgrad(f, dh0p, gp)
            _call("bf_op_slice_channels", N.ptr(gp), N.ptr(g0), 1, gp.numel(), 0, g0.numel(), N.stream_ptr(gp))
        else:
            ops.matmul_wgrad(f, dh0p, g0)
        g = UL.pointwise(dh0p, pack(ops.transpose(w0)), Ch)

        # -- backward ------------------------------------------------------------------------------------------------------------
        for b_block in reversed(chain):
            g = b_block(g)
        dpre = ops.act_bwd(f0, g, m.base_activation, f0pre)
        sp, sn = ops._s()
        _call("bf_op_conv2d_wgrad", N.ptr(noisy), int(noisy.dtype == torch.uint8), N.ptr(dpre), N.ptr(self.G("base/kernel", grads)),
              B, H, Wd, m.in_channels, m.filters, m.kernel_size, 1, m.v_min, m.v_max, sp, sn, N.stream_ptr(dpre))

        # -- regularisers: value into total[1], gradients added times `regularization` ------------------------------------------
        reg = float(ld.regularization)
        n_mult = 0
        for name, shape, kind, off in m.trainable_variables:
            rk = self.regularizer(name, kind)
            if rk in (None, "none"):
                continue
            if rk in ("channelwise", "multiplier"):
                w = self.W(name)
                _call("bf_op_reg_elementwise", N.ptr(w), N.ptr(self._grad_view(name, grads)), int(np.prod(shape)), N.BF_REG_L1,
                      CHANNELWISE_L1 if rk == "channelwise" else MULTIPLIER_L1, reg, N.ptr(total[1:2]), N.stream_ptr(w))
                n_mult += rk == "multiplier"
                continue
            if rk not in ("l1", "l2"):
                raise NotImplementedError(f"regularizer {rk}")
            w = self.W(name)
            _call("bf_op_reg_elementwise", N.ptr(w), N.ptr(self._grad_view(name, grads)), int(np.prod(shape)),
                  N.BF_REG_L1 if rk == "l1" else N.BF_REG_L2, REG_COEF, reg, N.ptr(total[1:2]), N.stream_ptr(w))
        if n_mult:
            # Multiplier hands its regulariser to the non-trainable w1 (= 1.0) as well (custom_layers.py:1067-1074): L1(1.0) of a
            # constant 1.0 per layer in model.losses, no gradient
            ones = torch.ones(n_mult, **f32)
            _call("bf_op_reg_elementwise", N.ptr(ones), None, n_mult, N.BF_REG_L1, MULTIPLIER_L1, reg, N.ptr(total[1:2]), N.stream_ptr(ones))
        for buf, off, n in self._unaligned:
            grads[off:off + n].copy_(buf)
        _call("bf_op_axpy", N.ptr(total[2:3]), N.ptr(total[1:2]), reg, 0, 1, N.stream_ptr(total))
        _call("bf_op_axpy", N.ptr(total), N.ptr(total[2:3]), 1.0, 0, 1, N.stream_ptr(total))
        m.mark_dirty()                                               # the folded inference weights no longer match the state
        self.totals = total
        return pred, losses, total

    def _channels(self):
        m = self.m
        cin = m.filters
        for cf, dm in zip(m.block_filters, m.block_depthwise):
            cin = cin * dm if dm != -1 else cf
            yield cin
