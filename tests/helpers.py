"""Shared test helpers: device transfer and the C-ABI diagnostic wrappers."""
import ctypes as C

import numpy as np
import torch

from blind_image_denoising_amd import _native as N


def dev(a, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy()


def assert_close(got, ref, rel=2e-5, what=""):
    """|got - ref| <= rel * max(1, max|ref|): fp32 accumulation against the fp64 oracle."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    err = np.abs(got - ref).max() if got.size else 0.0
    scale = max(1.0, np.abs(ref).max() if ref.size else 0.0)
    assert err <= rel * scale, f"{what}: max err {err:.3e} > {rel:.1e} * {scale:.3e}"


def conv3x3_gpu(x, w, epi=0, scale=None, shift=None, res=None, mask=None, transpose_flip=0, want_stats=False):
    """single 3x3 C16 convolution through bf_debug_conv3x3."""
    L = N.lib()
    B, H, W, _ = x.shape
    xd, wd = dev(x), dev(w)
    out = torch.full((B, H, W, 16), float("nan"), dtype=torch.float32, device="cuda")
    sd = dev(scale) if scale is not None else None
    hd = dev(shift) if shift is not None else None
    rd = dev(res) if res is not None else None
    md = dev(mask) if mask is not None else None
    grid = L.bf_debug_conv3x3_grid(B, H, W)
    stats = torch.zeros(grid * 32, dtype=torch.float32, device="cuda") if want_stats else None
    scratch = torch.zeros(2 * 2304 + 64, dtype=torch.float32, device="cuda")
    rc = L.bf_debug_conv3x3(N.ptr(xd), N.ptr(wd), N.ptr(out), N.ptr(sd), N.ptr(hd), N.ptr(rd), N.ptr(md), N.ptr(stats),
                            N.ptr(scratch), B, H, W, epi, transpose_flip, N.stream_ptr(xd))
    assert rc == 0, rc
    if want_stats:
        return host(out), host(stats).reshape(grid, 32)
    return host(out)


def conv3x3_h3_gpu(x, w, epi=0, res=None, mask=None, transpose_flip=0, want_stats=False):
    """single split-f16 3x3 C16 convolution on fp32 NHWC (the training convolution) through bf_debug_conv3x3_h3."""
    L = N.lib()
    B, H, W, _ = x.shape
    xd, wd = dev(x), dev(w)
    out = torch.full((B, H, W, 16), float("nan"), dtype=torch.float32, device="cuda")
    rd = dev(res) if res is not None else None
    md = dev(mask) if mask is not None else None
    grid = L.bf_debug_conv3x3_grid(B, H, W)
    stats = torch.zeros(grid * 32, dtype=torch.float32, device="cuda") if want_stats else None
    scratch = torch.full((int(L.bf_debug_conv3x3_h3_scratch_floats()),), float("nan"), dtype=torch.float32, device="cuda")
    rc = L.bf_debug_conv3x3_h3(N.ptr(xd), N.ptr(wd), N.ptr(out), N.ptr(rd), N.ptr(md), N.ptr(stats), N.ptr(scratch),
                               B, H, W, epi, transpose_flip, N.stream_ptr(xd))
    assert rc == 0, rc
    if want_stats:
        return host(out), host(stats).reshape(grid, 32)
    return host(out)


def fused_block_gpu(x, w1, w2, scale, shift, act1_relu=1):
    L = N.lib()
    B, H, W, _ = x.shape
    xd = dev(x)
    out = torch.full((B, H, W, 16), float("nan"), dtype=torch.float32, device="cuda")
    scratch = torch.zeros(2 * 2304 + 64, dtype=torch.float32, device="cuda")
    w1d, w2d, sd, hd = dev(w1), dev(w2), dev(scale), dev(shift)
    rc = L.bf_debug_fused_block(N.ptr(xd), N.ptr(w1d), N.ptr(w2d), N.ptr(sd), N.ptr(hd), N.ptr(out), N.ptr(scratch),
                                B, H, W, act1_relu, N.stream_ptr(xd))
    assert rc == 0, rc
    return host(out)


def fused_block_h3_gpu(x, w1, w2, scale, shift, act1_relu=1):
    """the split-f16 fused block (fused_h3.hip) on fp32 NHWC tensors through bf_debug_fused_block_h3."""
    L = N.lib()
    B, H, W, _ = x.shape
    xd = dev(x)
    out = torch.full((B, H, W, 16), float("nan"), dtype=torch.float32, device="cuda")
    # garbage-filled scratch: the kernel must not depend on pre-zeroed activations / dump lines
    scratch = torch.full((int(L.bf_debug_fused_block_h3_scratch_floats(B, H, W)),), float("nan"), dtype=torch.float32,
                         device="cuda")
    w1d, w2d, sd, hd = dev(w1), dev(w2), dev(scale), dev(shift)
    rc = L.bf_debug_fused_block_h3(N.ptr(xd), N.ptr(w1d), N.ptr(w2d), N.ptr(sd), N.ptr(hd), N.ptr(out), N.ptr(scratch),
                                   B, H, W, act1_relu, N.stream_ptr(xd))
    assert rc == 0, rc
    return host(out)


def fused_block2_h3_gpu(x, w4, scale2, shift2, act1_relu=1, reverse=0):
    """TWO consecutive split-f16 fused blocks in one launch (fused_h3w.hip) on fp32 NHWC tensors through bf_debug_fused_block2_h3:
    w4 = [4,3,3,16,16] (conv1, conv2 of block a, then of block b), scale2 / shift2 = [2,16]."""
    L = N.lib()
    B, H, W, _ = x.shape
    xd = dev(x)
    out = torch.full((B, H, W, 16), float("nan"), dtype=torch.float32, device="cuda")
    scratch = torch.full((int(L.bf_debug_fused_block2_h3_scratch_floats(B, H, W)),), float("nan"), dtype=torch.float32,
                         device="cuda")
    wd, sd, hd = dev(np.ascontiguousarray(w4)), dev(np.ascontiguousarray(scale2)), dev(np.ascontiguousarray(shift2))
    rc = L.bf_debug_fused_block2_h3(N.ptr(xd), N.ptr(wd), N.ptr(sd), N.ptr(hd), N.ptr(out), N.ptr(scratch), B, H, W, act1_relu,
                                    reverse, N.stream_ptr(xd))
    assert rc == 0, rc
    return host(out)


def wgrad_gpu(x, dy, h3=False):
    L = N.lib()
    B, H, W, _ = x.shape
    xd, dd = dev(x), dev(dy)
    partial = torch.zeros(int(L.bf_debug_wgrad_partial_floats(B, H, W)), dtype=torch.float32, device="cuda")
    dw = torch.full((3, 3, 16, 16), float("nan"), dtype=torch.float32, device="cuda")
    fn = L.bf_debug_wgrad3x3_h3 if h3 else L.bf_debug_wgrad3x3
    rc = fn(N.ptr(xd), N.ptr(dd), N.ptr(partial), N.ptr(dw), B, H, W, N.stream_ptr(xd))
    assert rc == 0, rc
    return host(dw)


def conv3x3_h3_pre_gpu(x, c, scale, shift, w, relu=1, reverse=0):
    """conv3x3_h3 with the affine + add formed on load: returns (y = x + scale * c + shift, [relu] conv(y))."""
    L = N.lib()
    B, H, W, _ = x.shape
    xd, cd, sd, hd, wd = dev(x), dev(c), dev(scale), dev(shift), dev(w)
    y = torch.full((B, H, W, 16), float("nan"), dtype=torch.float32, device="cuda")
    out = torch.full((B, H, W, 16), float("nan"), dtype=torch.float32, device="cuda")
    scratch = torch.full((int(L.bf_debug_conv3x3_h3_scratch_floats()),), float("nan"), dtype=torch.float32, device="cuda")
    rc = L.bf_debug_conv3x3_h3_pre(N.ptr(xd), N.ptr(cd), N.ptr(sd), N.ptr(hd), N.ptr(y), N.ptr(wd), N.ptr(out), N.ptr(scratch),
                                   B, H, W, relu, reverse, N.stream_ptr(xd))
    assert rc == 0, rc
    return host(y), host(out)


def fwd_block_h3t_gpu(x, w0, w1, c=None, scale=None, shift=None, relu=1, reverse=0, want_t=True):
    """the training-mode forward of one [3,3] block in one kernel (train_fwd_h3t.hip): returns (a, t, c_out, stats[32]) with
    a = x + scale * c + shift (None without c), t = [relu] conv_0(a) (None unless want_t), c_out = conv_1(t), stats = sum | sum of squares."""
    L = N.lib()
    B, H, W, _ = x.shape
    xd, w0d, w1d = dev(x), dev(w0), dev(w1)
    cd = dev(c) if c is not None else None
    sd = dev(scale) if c is not None else None
    hd = dev(shift) if c is not None else None
    nan = lambda: torch.full((B, H, W, 16), float("nan"), dtype=torch.float32, device="cuda")
    a_out = nan() if c is not None else None
    t_out = nan() if want_t else None
    c_out = nan()
    stats = torch.full((32,), float("nan"), dtype=torch.float32, device="cuda")
    scratch = torch.full((int(L.bf_debug_fwd_block_h3t_scratch_floats(B, H, W)),), float("nan"), dtype=torch.float32, device="cuda")
    rc = L.bf_debug_fwd_block_h3t(N.ptr(xd), N.ptr(cd), N.ptr(sd), N.ptr(hd), N.ptr(w0d), N.ptr(w1d), N.ptr(a_out), N.ptr(t_out),
                                  N.ptr(c_out), N.ptr(stats), N.ptr(scratch), B, H, W, relu, reverse, N.stream_ptr(xd))
    assert rc == 0, rc
    return (host(a_out) if a_out is not None else None, host(t_out) if t_out is not None else None, host(c_out), host(stats))


def bwd_block_h3t_gpu(a, g, c, coef, w0, w1, bnc=None, relu=1, reverse=0):
    """the backward of one [3,3] block in one kernel with T recomputed (train_bwd_h3t.hip): returns (out, dw1, dw0[, stats[32]])."""
    L = N.lib()
    B, H, W, _ = a.shape
    ad, gd, cd, kd, w0d, w1d = dev(a), dev(g), dev(c), dev(coef), dev(w0), dev(w1)
    bd = dev(bnc) if bnc is not None else None
    out = torch.full((B, H, W, 16), float("nan"), dtype=torch.float32, device="cuda")
    dw1 = torch.full((3, 3, 16, 16), float("nan"), dtype=torch.float32, device="cuda")
    dw0 = torch.full((3, 3, 16, 16), float("nan"), dtype=torch.float32, device="cuda")
    stats = torch.full((32,), float("nan"), dtype=torch.float32, device="cuda") if bnc is not None else None
    scratch = torch.full((int(L.bf_debug_bwd_block_h3t_scratch_floats(B, H, W)),), float("nan"), dtype=torch.float32, device="cuda")
    rc = L.bf_debug_bwd_block_h3t(N.ptr(ad), N.ptr(gd), N.ptr(cd), N.ptr(kd), N.ptr(w0d), N.ptr(w1d), N.ptr(bd), N.ptr(out), N.ptr(dw1),
                                  N.ptr(dw0), N.ptr(stats), N.ptr(scratch), B, H, W, relu, reverse, N.stream_ptr(ad))
    assert rc == 0, rc
    if stats is not None:
        return host(out), host(dw1), host(dw0), host(stats)
    return host(out), host(dw1), host(dw0)


def bwd3x3_h3_gpu(x, g, w, epi, c=None, coef=None, res=None, bnc=None, reverse=0, dbuf=0):
    """the fused backward kernel of one convolution: returns (dx, dw[, stats [grid, 32]])."""
    L = N.lib()
    B, H, W, _ = x.shape
    xd, gd, wd = dev(x), dev(g), dev(w)
    cd = dev(c) if c is not None else None
    kd = dev(coef) if coef is not None else None
    rd = dev(res) if res is not None else None
    bd = dev(bnc) if bnc is not None else None
    out = torch.full((B, H, W, 16), float("nan"), dtype=torch.float32, device="cuda")
    dw = torch.full((3, 3, 16, 16), float("nan"), dtype=torch.float32, device="cuda")
    grid = L.bf_debug_bwd3x3_h3_grid_ex(B, H, W, dbuf)
    stats = torch.full((grid * 32,), float("nan"), dtype=torch.float32, device="cuda") if epi & N.EPI_BNBWD else None
    scratch = torch.full((int(L.bf_debug_bwd3x3_h3_scratch_floats(B, H, W)),), float("nan"), dtype=torch.float32, device="cuda")
    rc = L.bf_debug_bwd3x3_h3(N.ptr(xd), N.ptr(gd), N.ptr(cd), N.ptr(kd), N.ptr(wd), N.ptr(out), N.ptr(rd), N.ptr(bd), N.ptr(dw),
                              N.ptr(stats), N.ptr(scratch), B, H, W, epi, reverse | (dbuf << 1), 1, N.stream_ptr(xd))
    assert rc == 0, rc
    if stats is not None:
        return host(out), host(dw), host(stats).reshape(grid, 32)
    return host(out), host(dw)


def compare_or_explain_by_ties(compare, oracle_step, find_ties, max_ties=6):
    """A training step of the GPU engine against the fp64 oracle, with the ONE legitimate source of disagreement checked instead
    of assumed: `compare(ref)` raises AssertionError on a mismatch with ref = oracle_step(flips).  On a mismatch the oracle lists
    the elements that sit within fp32 rounding of a kink of the graph (ReLU gates, hinge / cutoff thresholds, the denormaliser's
    clip: oracle.training_step_ties).  No such element -> the mismatch is a fault and is raised.  Otherwise the GPU result must
    equal the oracle evaluated with some subset of exactly those gates (the `max_ties` closest to their kink, if there are more) taken
    on the other side (what fp32 may do there); if no subset explains it, the mismatch is raised.  Returns the number of flipped gates (0: plain agreement)."""
    import itertools
    try:
        compare(oracle_step(None))
        return 0
    except AssertionError as first:
        ties = find_ties()
        if not ties:
            raise AssertionError(f"mismatch and no element within rounding of a kink: {first}") from first
        # more candidates than can be enumerated: only the `max_ties` CLOSEST to their kink may be flipped (2^max_ties oracle steps at
        # most); the others keep the oracle's side, so a mismatch they would explain is raised -- stricter, never more lenient
        cand = sorted(ties, key=lambda t: abs(t[2]))[:max_ties]
        for n in range(1, len(cand) + 1):
            for combo in itertools.combinations(cand, n):
                try:
                    compare(oracle_step(list(combo)))
                    return n
                except AssertionError:
                    pass
        raise AssertionError(f"mismatch that no side of the {len(cand)} closest of {len(ties)} near-kink elements explains: {first}") from first


def oracle_is_on_a_kink(oracle_grads, noisy, tol_rel, segments=None, eps=2e-4, trials=3, seed=0):
    """The tie hypothesis of the autograd-oracle sweeps, CHECKED on the oracle alone: does the fp64 oracle's own gradient jump when
    its input moves by `eps` (on the 0..255 scale: a relative change of 1e-6, far below what moves a smooth gradient past the bar)?
    oracle_grads(noisy) -> flat fp64 gradient; segments = [(offset, size)] of the gradient tensors (each is judged against its own
    largest entry, as the comparison does).  True: some ReLU / hinge / clip input sits within rounding of its kink, fp32 may take
    the other side there and a mismatch on this input proves nothing.  False: the gradient is smooth around this input and a
    mismatch is a fault."""
    rng = np.random.default_rng(seed)
    g0 = oracle_grads(noisy)
    segments = segments or [(0, g0.size)]
    floor = 1e-3 * float(np.abs(g0).max())
    for _ in range(trials):
        g1 = oracle_grads(noisy + eps * rng.choice([-1.0, 1.0], size=noisy.shape))
        for o, n in segments:
            scale = max(float(np.abs(g0[o:o + n]).max()), floor, 1e-30)
            if np.abs(g1[o:o + n] - g0[o:o + n]).max() > 0.25 * tol_rel * scale:
                return True
    return False
