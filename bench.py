#!/usr/bin/env python3
"""
bench.py -- denoised images/sec (256x256x3) of the resnet hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one DenoiserModule.__call__ pass (uint8 in -> uint8 out, one bf_forward_u8 C-ABI call)
over one synthetic batch that is already resident in HBM.  Workload = BASELINE.json's metric
config: resnet_color_1x18_bn_16x3x3, batch 128 per GPU, 256x256x3, random-init glorot weights with
non-trivial BN statistics, smooth-field + gaussian-noise uint8 images.  The batch is
sharded by replication across ranks (weak scaling: independent images, no data-path collective).

The JSON line also carries
  roofline     : dominant kernel = the fused residual block (one launch per block), its average launch
                 duration measured with HIP events on the launch stream over the timed region.
                 Default arithmetic (split-f16 on the f16 matrix cores, fused_block_h3r_kernel): the
                 kernel's nearer roof is HBM -- bound "hbm", achieved = algorithmic bytes per launch
                 (B*H*W * 128 B/px: read x, write y, fp32-equivalent storage) / launch duration, peak
                 8 TB/s; the matrix-pipe view of the same launch is reported beside it ("mfma").
                 --arith 0 (exact fp32 on the f32 matrix cores, fused_block_v4_kernel): bound "mfma",
                 achieved = B*H*W * 9216 FLOP/px / duration against the 157.3 TFLOP/s fp32 MFMA peak.
  cpu_baseline : the oracle's C port (oracle/bfcnn_port.c, fp32, OpenMP, all host cores) timed on
                 a bounded sample of the same workload, rank 0 / N=1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
MFMA_F16_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md, BF16/F16 dense
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md, HBM3E
BYTES_PER_PX_BLOCK = 128            # read x + write y, 16 channels x 4 B (fp32 NHWC or split-f16 hi/lo planes)
H3_MFMA_FLOP_PER_PX = 2 * 15 * 16384 / 16 + 16384 / 16   # issued f16 MFMA FLOPs: 15 (+1 residual) MFMAs of 16x16x32 per 16 px and conv
FLOP_PER_PX_BLOCK = 2 * 4608        # two 3x3 16->16 convolutions


def gflop_per_image(no_layers, h, w, k=3, cin=3, hf=32, cout=3):
    per_px = 2 * k * k * cin * 16 + no_layers * FLOP_PER_PX_BLOCK + 2 * (16 * hf + hf * cout)
    return per_px * h * w / 1e9


def host_cores():
    """the host cores this process may actually use: a GPU box hands out a SHARE of its 256 hardware threads (a cgroup CPU quota),
    which os.sched_getaffinity does not show -- 256 torch threads on a 16-core share ran a training step in 84 s.  The quota
    (cgroup v2 cpu.max / v1 cfs_quota_us) bounds the affinity count; OMP_NUM_THREADS, when the launcher set it, bounds both."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    try:
        n = min(n, max(1, int(os.environ["OMP_NUM_THREADS"])))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline_port(spec, params, state, noisy_u8, budget_s=10.0):
    """the oracle's C port on the host cores: bounded sample, ~budget_s of CPU work."""
    from oracle import port
    h = port.lib(rebuild=True)                       # -march=native on THIS box
    port.forward_u8(spec, params, state, noisy_u8[:1], h)          # warm-up / page-in
    t0 = time.perf_counter()
    port.forward_u8(spec, params, state, noisy_u8[:1], h)
    one = time.perf_counter() - t0
    # ~budget_s of CPU work (the contract asks for 10-30 s): the bench batch, repeated as often as that takes
    n = int(max(2, min(noisy_u8.shape[0], budget_s / max(one, 1e-3))))
    t0 = time.perf_counter()
    port.forward_u8(spec, params, state, noisy_u8[:n], h)
    first = time.perf_counter() - t0
    reps = 1 + int(max(0, min(63, round((budget_s - first) / max(first, 1e-3)))))
    for _ in range(reps - 1):
        port.forward_u8(spec, params, state, noisy_u8[:n], h)
    dt = time.perf_counter() - t0
    return {"value": n * reps / dt, "unit": "images/s", "cores": int(h.bfcnn_port_max_threads()), "kind": "port",
            "sample": f"{reps} x {n} images of the same 1x18 256x256x3 uint8 workload ({dt:.1f} s), oracle/bfcnn_port.c fp32 OpenMP "
                      f"(CPU restatement, not TensorFlow)"}


def cpu_baseline_torch(spec, params, state, noisy_u8, budget_s=10.0, nthreads=None):
    """the same graph on PyTorch-CPU (oneDNN convolutions, channels_last, all host cores; SURVEY 8d leg ii): the
    restatement's tensors loaded into torch.nn.functional calls -- a second CPU implementation, not TensorFlow."""
    import torch
    import torch.nn.functional as F
    if nthreads is None:            # the cores this process may run on (a GPU box hands out a share of its 256)
        nthreads = host_cores()
    torch.set_num_threads(int(nthreads))
    off = spec.offsets()
    P = lambda name: torch.from_numpy(np.asarray(params[off[name][0]:off[name][0] + int(np.prod(off[name][1]))], np.float32).reshape(off[name][1]))
    conv_w = lambda name: P(name).permute(3, 2, 0, 1).contiguous(memory_format=torch.channels_last)       # HWIO -> OIHW
    base = conv_w("base/kernel")
    blocks = []
    for i in range(spec.no_layers):
        g = P(f"block{i}/bn1/gamma")
        mean = torch.from_numpy(np.asarray(state[i * 32:i * 32 + 16], np.float32))
        var = torch.from_numpy(np.asarray(state[i * 32 + 16:i * 32 + 32], np.float32))
        sc = g / torch.sqrt(var + spec.bn_eps)
        blocks.append((conv_w(f"block{i}/conv0/kernel"), conv_w(f"block{i}/conv1/kernel"), sc.view(1, -1, 1, 1), (-sc * mean).view(1, -1, 1, 1)))
    h0, h1 = conv_w("head/conv0/kernel"), conv_w("head/conv1/kernel")

    def forward(u8):
        x = torch.from_numpy(u8).permute(0, 3, 1, 2).float().contiguous(memory_format=torch.channels_last)
        x = torch.clamp(x, spec.v_min, spec.v_max) / (spec.v_max - spec.v_min) - 0.5
        x = F.conv2d(x, base, padding=base.shape[-1] // 2)
        for w0, w1, sc, sh in blocks:
            t = F.relu(F.conv2d(x, w0, padding=1))
            x = x + F.conv2d(t, w1, padding=1) * sc + sh
        y = torch.tanh(2.0 * F.conv2d(F.conv2d(x, h0), h1)) * 0.51
        y = (torch.clamp(y, -0.5, 0.5) + 0.5) * (spec.v_max - spec.v_min) + spec.v_min
        return torch.round(y).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().numpy()

    with torch.no_grad():
        forward(noisy_u8[:1])
        t0 = time.perf_counter()
        out1 = forward(noisy_u8[:1])
        one = time.perf_counter() - t0
        n = int(max(2, min(noisy_u8.shape[0], 16, budget_s / max(one, 1e-3))))
        t0 = time.perf_counter()
        forward(noisy_u8[:n])
        first = time.perf_counter() - t0
        reps = 1 + int(max(0, min(15, round((budget_s - first) / max(first, 1e-3)))))
        for _ in range(reps - 1):
            forward(noisy_u8[:n])
        dt = time.perf_counter() - t0
    return {"value": n * reps / dt, "unit": "images/s", "cores": nthreads, "kind": "port",
            "sample": f"{reps} x {n} images of the same workload ({dt:.1f} s), torch-CPU {torch.__version__} conv2d (oneDNN, channels_last, "
                      f"{nthreads} threads) with the restatement's tensors"}, out1


def all_agree(torch, dist, ok):
    """True when EVERY rank says ok (MIN over the ranks).  A guarded stage that holds collectives must be followed by this before
    the next collective: a rank that failed alone (RCCL dlopen, a duplicate device in a rehearsal) would otherwise leave its
    peers inside a collective it never enters, and the bench would hang instead of recording a note."""
    if dist is None:
        return bool(ok)
    dev = "cpu" if dist.get_backend() == "gloo" else "cuda"
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def time_gradient_exchange(torch, bf, model, prep, clean_dev, world, dist):
    """what the ONE collective of a training step costs on this box, next to the work it is supposed to hide behind
    (DataParallelTrainer.step(overlap=...): the next batch's on-device corruption): HIP-event time per call of
    torch.distributed.all_reduce and of the C ABI's bf_allreduce_grads on the flat gradient buffer, in the process group of this
    run (world > 1) or in a one-rank RCCL group made for the measurement (world = 1: launch + kernel cost without the wire)."""
    import socket
    out = {"payload_floats": int(model.n_params), "world": int(world)}
    own_group = False
    try:
        import torch.distributed as d
        if dist is None:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            d.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                 device_id=torch.device("cuda", torch.cuda.current_device()))
            own_group = True
        g = torch.zeros(model.n_params, dtype=torch.float32, device="cuda")

        def timed(fn, n=50):
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / n

        out["torch_distributed_all_reduce_us"] = timed(lambda: d.all_reduce(g))
        # the C ABI's collective: every stage that holds a collective is agreed on by all ranks before the next one starts
        comm, note = None, None
        if os.environ.get("BF_BENCH_REHEARSE") == "1" and world > 1:
            note = "rehearsal (ranks share a GPU over gloo): RCCL cannot form a communicator with duplicate devices, skipped"
        else:
            try:
                comm = bf.NativeCommunicator(torch.device("cuda", torch.cuda.current_device()))
            except Exception as e:
                note = str(e)[:200]
        if not all_agree(torch, d if (dist is not None or own_group) else None, comm is not None):
            if comm is not None:
                comm.close()
            comm = None
            note = note or "another rank could not create its communicator"
        if comm is not None:
            out["c_abi_bf_allreduce_grads_us"] = timed(lambda: comm.allreduce(g))
            # the same calls seen from the communicator's OWN stream (events recorded there: the all-reduce kernels alone)
            cs = comm._stream
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record(cs)
            for _ in range(50):
                comm.launch(g)
            e1.record(cs)
            comm.wait(g)
            torch.cuda.synchronize()
            out["c_abi_bf_allreduce_grads_on_its_stream_us"] = e0.elapsed_time(e1) * 1e3 / 50
            comm.close()
        else:
            out["c_abi_bf_allreduce_grads_us"] = None
            out["c_abi_note"] = note
        out["overlap_work_us"] = timed(lambda: prep(clean_dev), n=20)
        out["overlap_work"] = "PrepareData on the next batch (flips + truncated-normal noise on the device, bf_noise_augment)"
    except Exception as e:
        out["note"] = f"not measured: {str(e)[:200]}"
    finally:
        if own_group:
            try:
                import torch.distributed as d
                d.destroy_process_group()
            except Exception:
                pass
    return out


def cpu_train_baseline_torch(spec, ls, params, state, S, budget_s=12.0, nthreads=None):
    """the training step of configs[3] on PyTorch-CPU, fp32, all host cores ("the same graph on all host cores"):
    normalise -> base conv -> N x [conv + ReLU -> conv -> BatchNorm (training statistics, scale only) -> + skip] -> head ->
    tanh(2x) * 0.51 -> denormalise; L1 with hinge / cutoff (keras relu threshold, loss.py:40-65) + L1 / L2 kernel regularisers;
    autograd backward; Adam with global-norm clipping.  Independent restatement on torch.nn.functional (oneDNN), not TensorFlow."""
    import torch
    import torch.nn.functional as F
    if nthreads is None:
        nthreads = host_cores()
    torch.set_num_threads(int(nthreads))
    off = spec.offsets()
    flat = torch.tensor(np.asarray(params, np.float32), requires_grad=True)
    V = lambda name: flat[off[name][0]:off[name][0] + int(np.prod(off[name][1]))].view(*off[name][1])
    W = lambda name: V(name).permute(3, 2, 0, 1)                                  # HWIO -> OIHW
    opt = torch.optim.Adam([flat], lr=1e-3, betas=(0.9, 0.999), eps=1e-7)

    def step(gt, noisy):
        x = (torch.clamp(noisy, spec.v_min, spec.v_max) / (spec.v_max - spec.v_min) - 0.5).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
        f = F.conv2d(x, W("base/kernel"), padding=spec.kernel_size // 2)
        reg = V("base/kernel").abs().sum() * 0.01
        for i in range(spec.no_layers):
            t = F.relu(F.conv2d(f, W(f"block{i}/conv0/kernel"), padding=1))
            c = F.conv2d(t, W(f"block{i}/conv1/kernel"), padding=1)
            c = F.batch_norm(c, None, None, weight=V(f"block{i}/bn1/gamma"), bias=None, training=True, eps=spec.bn_eps)
            f = f + c
            reg = reg + (V(f"block{i}/conv0/kernel").abs().sum() + V(f"block{i}/conv1/kernel").abs().sum()) * 0.01
        h = F.conv2d(F.conv2d(f, W("head/conv0/kernel")), W("head/conv1/kernel"))
        reg = reg + ((V("head/conv0/kernel") ** 2).sum() + (V("head/conv1/kernel") ** 2).sum()) * 0.01
        p = torch.tanh(2.0 * h) * 0.51
        pred = ((torch.clamp(p, -0.5, 0.5) + 0.5) * (spec.v_max - spec.v_min) + spec.v_min).permute(0, 2, 3, 1)
        a = (gt - pred).abs()
        d = torch.where((a > ls.hinge) & (a < ls.cutoff), a, torch.zeros_like(a)) + torch.where(a >= ls.cutoff, torch.full_like(a, ls.cutoff), torch.zeros_like(a))
        total = d.mean() * ls.mae_multiplier + reg * ls.regularization
        opt.zero_grad(set_to_none=True)
        total.backward()
        torch.nn.utils.clip_grad_norm_([flat], 1.0)
        opt.step()
        return float(total.detach())

    nb = 2
    from oracle import bfcnn_oracle as O
    c1, n1 = O.synthetic_batch(nb, S, S, sigma=20.0, seed=99)
    gt, noisy = torch.from_numpy(c1.astype(np.float32)), torch.from_numpy(n1.astype(np.float32))
    first_loss = step(gt, noisy)                                                  # warm-up (oneDNN primitive creation)
    t0 = time.perf_counter()
    nrep = 0
    while nrep < 1 or (time.perf_counter() - t0 < budget_s and nrep < 64):
        step(gt, noisy)
        nrep += 1
    dt = (time.perf_counter() - t0) / nrep
    return {"value": nb / dt, "unit": "images/s", "cores": int(nthreads), "kind": "port", "first_loss": first_loss,
            "sample": f"{nrep} training steps (forward + L1 loss + autograd backward + clipped Adam) of {nb} x {S}x{S} images, "
                      f"torch-CPU fp32 (oneDNN, channels_last, {nthreads} threads): CPU restatement of the same graph, not TensorFlow"}


def cpu_baseline(spec, params, state, noisy_u8):
    """both CPU legs of SURVEY 8d; the faster one is the reported baseline, the other rides along."""
    a = cpu_baseline_port(spec, params, state, noisy_u8)
    try:
        b, _ = cpu_baseline_torch(spec, params, state, noisy_u8, nthreads=a["cores"])      # same core count as the C port's OpenMP team
    except Exception as e:                                # torch-CPU leg is a bonus: never fail the bench on it
        b = {"value": 0.0, "unit": "images/s", "cores": 0, "kind": "port", "sample": f"torch-CPU leg failed: {e}"}
    best, other = (a, b) if a["value"] >= b["value"] else (b, a)
    best = dict(best)
    best["other_leg"] = other
    return best


def pmc_traffic(layers, batch, size, fused, kernel=None):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc pass of THIS
    workload (profiles/pmc_traffic.json: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide
    coalesced reads on gfx950, plus WRITE_SIZE); None when no matching profile is committed."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            for e in json.load(f):
                if (e["layers"], e["batch"], e["size"], e["fused"]) == (layers, batch, size, fused) and \
                        (kernel is None or e.get("kernel") == kernel):
                    return e["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def pyramid_bench(args, torch, bf, O, rank, local_rank, world, dist):
    """configs[4]'s resampling part (pyramid.py + upsampling.py): 3-level Laplacian pyramid + its inverse on a
    [32,512,512,3] float32 batch; HBM-bound kernels, so the figure of merit is algorithmic GB/s against the 8 TB/s roof.
    Algorithmic bytes (4 B elements, n = B*H*W*C): split level l (n_l elements): x read once, down (n_l / 4) and x - up(down)
    (n_l) written -> 2.25 n_l (one kernel, bf_laplacian_split); merge level l: reads n_l/4 + n_l, writes n_l -> 2.25 n_l."""
    B, S, C, levels = (32 if args.batch == 128 else args.batch), (512 if args.size == 256 else args.size), 3, 3
    x = (torch.rand((B, S, S, C), device=f"cuda:{local_rank}") - 0.5).contiguous()
    cfg = {"type": "laplacian", "levels": levels, "kernel_size": (5, 5)}
    pyr, inv = bf.build_pyramid_model((S, S, C), cfg), bf.build_inverse_pyramid_model((S, S, C), cfg)

    def step():
        return inv(pyr(x))
    for _ in range(max(args.warmup, 1)):
        y = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    n = B * S * S * C
    # "Laplacian split fused: read 1, write 1 + 1/4" per level; the merge reads 1/4 + 1 and writes 1: 2.25 n each
    nbytes = 4 * sum((2.25 + 2.25) * n / 4 ** l for l in range(levels - 1))
    err = float((y - x).abs().mean().item())
    if rank == 0:
        gbs = nbytes * args.steps / elapsed / 1e9
        cpu = None
        if not args.no_cpu_baseline:
            # the oracle's NumPy restatement of the same split + merge (oracle/bfcnn_oracle.py laplacian_pyramid / inverse) on a bounded
            # sample: single images of the same size until about 8 s have passed
            xs = x[:1].cpu().numpy().astype(np.float32)
            t1, reps = time.perf_counter(), 0
            while reps < 1 or time.perf_counter() - t1 < 8.0:
                back = O.inverse_laplacian_pyramid(O.laplacian_pyramid(xs, levels, (5, 5)))
                reps += 1
            dt = time.perf_counter() - t1
            cpu = {"value": reps / dt, "unit": "images/s", "cores": 1, "kind": "port",
                   "sample": f"{reps} x one {S}x{S}x{C} float32 image through oracle/bfcnn_oracle.py laplacian_pyramid + inverse_laplacian_pyramid "
                             f"(NumPy restatement, not TensorFlow; {dt:.1f} s)",
                   "round_trip_mean_abs_error": float(np.abs(back - xs).mean())}
        print(json.dumps({
            "metric": "laplacian pyramid split + merge images/sec (512x512x3, 3 levels)", "value": B * args.steps / elapsed,
            "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"laplacian pyramid (avg-pool 5x5 s2 SAME, bilinear x2) + inverse, batch={B} {S}x{S}x{C} float32"},
            "round_trip_mean_abs_error": err,      # reference test bar: < 1e-7 (tests/bfcnn/test_pyramid.py)
            "roofline": {"bound": "hbm", "kernel": "lap_split_kernel + upsample2x_band_kernel (one launch per level and direction)", "achieved": gbs,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_step": nbytes,
                         # rounds 1-2 counted the split as two kernels (3.5 n per level: x read twice, down read back); same time
                         # against that figure, for comparison with their lines only
                         "two_kernel_accounting": {"bytes_per_step": 4 * sum((3.5 + 2.25) * n / 4 ** l for l in range(levels - 1)),
                                                   "gbs": 4 * sum((3.5 + 2.25) * n / 4 ** l for l in range(levels - 1)) * args.steps / elapsed / 1e9}},
            **({"cpu_baseline": cpu} if cpu is not None else {})}), flush=True)


def unet_flop_per_px(m):
    """algorithmic FLOP (2 x MAC) per level-0 pixel of the built unet_laplacian graph: 1x1 / k x k convolutions only
    (LayerNorm, resize, adds and the fixed-size attention are not counted)."""
    total = 2.0 * 25 * m.in_channels * m.filters
    for d in range(m.depth):
        C, frac = m.level_filters(d), 0.25 ** d
        if m._is_attention(d):
            per = 0.0          # fixed 16x16 token grid: q, k, v, out 1x1s and the attention do not scale with the image
        else:
            per = m.width * (2.0 * m.enc_k ** 2 * C + 2 * 2.0 * C * 4 * C)
        if d < m.depth - 1:
            per += m.width * (2.0 * m.dec_k ** 2 * C + 2 * 2.0 * C * 4 * C)          # decoder blocks of this level
            per += 0.25 * 2.0 * C * m.level_filters(d + 1) * 2                         # down 1x1 + up 1x1 (both at 1/4 res)
        per += 2.0 * C * m.head_filters + 2.0 * m.head_filters * m.out_channels if d == 0 else 0.0
        total += per * frac
    return total


def unet_bench(args, torch, bf, O, rank, local_rank, world, dist):
    rec = unet_run(args, torch, bf, O, rank, local_rank, world, dist, args.steps, args.warmup, cpu=not args.no_cpu_baseline)
    if rec is not None:
        print(json.dumps(rec), flush=True)


def unet_run(args, torch, bf, O, rank, local_rank, world, dist, steps, warmup, cpu=True, B=None, S=None):
    """configs[4]: unet_laplacian (v5 graph: depth 3, width 3, 32/64/128 filters) inference, batch 32 512x512x3,
    uint8 -> uint8 through DenoiserModule.__call__ (only the full-resolution head is evaluated, as the module keeps
    output 0).  Images are independent: N ranks = N replicas, no collective.  Returns the record on rank 0 (`--mode unet` prints
    it; the default mode carries a short run of it as the `unet` sub-record)."""
    from oracle import unet_oracle as U
    if B is None:
        B = 32 if args.batch == 128 else args.batch
    if S is None:
        S = 512 if args.size == 256 else args.size
    trained = args.unet_graph == "v5.6"
    if trained:
        # the reference's trained network (graph revision and tensors of pretrained/unet_laplacian_v5.6, committed as data
        # in tests/golden/unet_v56.npz; same size as the v5 graph: 334 976 parameters)
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "unet_v56.npz"))
        cfg = {"model": json.loads(bytes(z["config"]).decode())}
        params = np.asarray(z["params"])
    else:
        cfg = U.canonical_config()
    spec = U.UnetLaplacianSpec.from_config(cfg["model"])
    if not trained:
        params = U.init_params(spec, seed=42)
    model = bf.model_builder(cfg["model"], device=f"cuda:{local_rank}").hydra
    model.set_weights(params)
    for kv in args.opt:                                   # A/B switches of the unet host (fuse_chain, fuse_up_block, arith)
        k, v = kv.split("=")
        try:
            model.set_option(k, int(v))
        except ValueError:
            pass                                          # an option of another mode's model
    module = bf.DenoiserModule(model)
    clean, base = O.synthetic_batch(4, S, S, sigma=20.0, seed=1234 + rank)
    noisy = torch.from_numpy(np.concatenate([base] * ((B + 3) // 4), axis=0)[:B]).cuda()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(max(warmup, 1)):
        out = module(noisy)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = module(noisy)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return None
    # parity of one small crop against the oracle (the full 512^2 image takes the fp64 NumPy oracle minutes)
    crop = base[:1, :64, :64]
    ref = U.denoiser_module_call(spec, params, crop)
    got = module(torch.from_numpy(np.ascontiguousarray(crop)).cuda()).cpu().numpy()
    diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    flop = unet_flop_per_px(model) * B * S * S
    tf = flop * steps / elapsed / 1e12
    # dominant kernel, timed live with events on the launch stream: the level-0 encoder ConvNext block
    # (uh_enc32_kernel: x read once, out written once = 2 * C * 4 B per pixel algorithmic)
    from blind_image_denoising_amd import unet_laplacian as UL
    P = model._pack()
    x0 = torch.randn((B, S, S, 32), device=f"cuda:{local_rank}")
    blk = lambda: UL.convnext_block_h3(x0, P["enc0_0/dw/kernel"], P["enc0_0/ln/gamma"], P["enc0_0/mlp_h3"], P["enc0_0/gamma/w"],
                                       model.mlp_activation)
    for _ in range(3):
        blk()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nl = 20
    e0.record()
    for _ in range(nl):
        blk()
    e1.record()
    torch.cuda.synchronize()
    launch_us = e0.elapsed_time(e1) * 1e3 / nl
    blk_bytes = B * S * S * 32 * 4 * 2
    gbs = blk_bytes / launch_us / 1e3
    blk_flop = B * S * S * (2.0 * 25 * 32 + 2 * 2.0 * 32 * 128)
    rec = {
        "metric": "denoised images/sec (512x512x3), unet_laplacian 3-scale", "value": world * B * steps / elapsed,
        "unit": "images/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (ConvNext MLPs: f16x2 split hi+lo, fp32 accumulate)", "data": "synthetic",
        "config": {"workload": f"unet_laplacian {'v5.6 trained archive graph' if trained else 'v5 graph'} (depth 3, width 3, "
                               f"filters 32/64/128, attention on the deepest "
                               f"level) inference, batch={B}/GPU {S}x{S}x3 uint8->uint8 (DenoiserModule.__call__)",
                   "batch_per_gpu": B, "parallelism": f"replicas x{world}, no collective"},
        "parity": {"max_abs_lsb": int(diff.max()), "mean_abs_lsb": float(diff.mean()), "checked": "one 64x64 crop vs oracle"},
        "end_to_end_tflops": tf,
        "roofline": {"bound": "hbm", "kernel": "uh_enc32u_kernel (level-0 encoder ConvNext block: 2.8 of the 12 ms graph "
                                               "are this kernel)", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": gbs / HBM_PEAK_GBS, "traffic": pmc_traffic(0, B, S, True, kernel="uh_enc32u_kernel"),
                     "algorithmic_bytes_per_launch": blk_bytes,
                     "launch_us": launch_us,
                     "mfma": {"dtype": "f16 (split hi/lo, 3 products)", "algorithmic_tflops": blk_flop / launch_us / 1e6,
                              "peak_tflops": MFMA_F16_PEAK_TFLOPS}}}
    if trained:
        mae = lambda a, b: float(np.abs(a.astype(np.float64) - b.astype(np.float64)).mean())
        den = out[:4].cpu().numpy()
        rec["denoising"] = {"mae_noisy": mae(clean, base), "mae_denoised": mae(clean, den),
                            "note": "synthetic smooth fields + truncated normal noise, std 20 (SURVEY 8d); trained weights"}
    if cpu:
        # leg (i), the reported baseline: the same graph on torch-CPU fp32 with all the host cores this process may use
        # (oracle/unet_torch.py: the restatement's forward on torch.nn.functional tensor ops; whole SxS images, no scaling);
        # leg (ii): the fp64 NumPy oracle on one 256x256 crop, scaled by the pixel ratio, kept beside it as `other_leg`
        small = base[:1, :256, :256]
        t0 = time.perf_counter()
        nrep = 0
        while nrep < 1 or (time.perf_counter() - t0 < 5.0 and nrep < 16):
            U.denoiser_module_call(spec, params, small)
            nrep += 1
        dt = (time.perf_counter() - t0) / nrep
        oracle_leg = {"value": 1.0 / (dt * (S * S) / (256.0 * 256.0)), "unit": "images/s", "cores": 1, "kind": "port",
                      "sample": f"{nrep} x one 256x256 crop through oracle/unet_oracle.py (fp64 NumPy restatement, not TensorFlow; "
                                f"BLAS threads as NumPy picks them), scaled by the pixel ratio to {S}x{S}"}
        try:
            rec["cpu_baseline"] = cpu_unet_baseline_torch(spec, params, base[:1], ref_crop=(crop, ref))
            rec["cpu_baseline"]["other_leg"] = oracle_leg
        except Exception as e:                              # the torch-CPU leg must never fail the bench
            rec["cpu_baseline"] = dict(oracle_leg, note=f"torch-CPU leg failed: {str(e)[:200]}")
    return rec


def cpu_unet_baseline_torch(spec, params, image_u8, budget_s=10.0, nthreads=None, ref_crop=None):
    """unet_laplacian inference on PyTorch-CPU, fp32, all host cores this process may use: oracle/unet_torch.py's forward (the
    NumPy restatement on torch tensor ops: conv2d / layer norm / interpolate; first output only, round + uint8 as the module does)
    on whole images of the bench's size.  A CPU restatement of the same graph, not TensorFlow."""
    import torch
    from oracle import unet_torch as UT
    if nthreads is None:
        nthreads = host_cores()
    torch.set_num_threads(int(nthreads))
    P = UT.views(spec, torch.from_numpy(np.asarray(params, np.float32)))

    def forward(u8):
        with torch.no_grad():
            y = UT.hydra(spec, P, torch.from_numpy(u8.astype(np.float32)))[0]
            return torch.round(y).clamp(0, 255).to(torch.uint8).numpy()
    note = {}
    if ref_crop is not None:                               # the leg computes what the oracle computes (+-1 LSB: fp32 against fp64)
        c, r = ref_crop
        note["max_abs_lsb_vs_oracle_on_the_parity_crop"] = int(np.abs(forward(c).astype(np.int32) - r.astype(np.int32)).max())
    forward(image_u8[:1])                                  # warm-up (oneDNN primitive creation)
    t0 = time.perf_counter()
    nrep = 0
    while nrep < 1 or (time.perf_counter() - t0 < budget_s and nrep < 64):
        forward(image_u8[:1])
        nrep += 1
    dt = (time.perf_counter() - t0) / nrep
    S = image_u8.shape[1]
    return dict({"value": 1.0 / dt, "unit": "images/s", "cores": int(nthreads), "kind": "port",
                 "sample": f"{nrep} x one {S}x{S}x3 uint8 image through oracle/unet_torch.py's forward in fp32 (torch-CPU {torch.__version__}, "
                           f"{nthreads} threads; CPU restatement of the same graph, not TensorFlow; {dt * nrep:.1f} s)"}, **note)


def generic_bench(args, torch, bf, O, rank, local_rank, world, dist):
    """The one resnet configuration the reference ships (bfcnn/configs/resnet_color_1x6_bn_32x128x32_1x3x1_128x128_depthwise_l1_relu.json:
    7x7 base 3->32, six blocks of 1x1 32->32 + ReLU, depthwise 3x3 x4 + BN + ReLU, grouped 1x1 128->32 + BN, Add; backbone_resnet.py:149-176)
    through the generic operator path (resnet_generic.py; exact fp32 on the fp32 matrix cores): inference, batch 64, 256 x 256 uint8 ->
    uint8 through DenoiserModule.__call__, random-init weights.  Images are independent: N ranks = N replicas, no collective."""
    from oracle import resnet_generic_oracle as G
    from blind_image_denoising_amd import _native as N
    B, S = (64 if args.batch == 128 else args.batch), args.size
    cfg = G.shipped_config()
    spec = G.GenericResnetSpec.from_config(cfg)
    params, state = G.init_params(spec, seed=42)
    model = bf.model_builder(cfg, device=f"cuda:{local_rank}").hydra
    model.set_weights(params, state)
    module = bf.DenoiserModule(model)
    _, base = O.synthetic_batch(4, S, S, sigma=20.0, seed=1234 + rank)
    noisy = torch.from_numpy(np.concatenate([base] * ((B + 3) // 4), axis=0)[:B]).cuda()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(max(args.warmup, 1)):
        out = module(noisy)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = module(noisy)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return
    crop = base[:1, :64, :64]
    ref = G.denoiser_module_call(spec, params, state, crop)
    got = module(torch.from_numpy(np.ascontiguousarray(crop)).cuda()).cpu().numpy()
    diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    # dominant kernel, timed live with events on the launch stream through its C-ABI operator: the whole bottleneck block (1x1 32 -> 32,
    # depthwise 3x3 x4 + folded BN + ReLU, 1x1 128 -> 32 + folded BN + Add) in one kernel, ug_bneck_kernel (csrc/generic_h3.hip; split-f16
    # GEMMs, the 128-channel tensor never leaves the registers); six launches per forward.  Algorithmic bytes: the 32-channel map read
    # once (it is also the skip) and written once = 2 * 128 B per pixel.  Matrix work counted as issued: 3 f16 products per fp32 one.
    L = N.lib()
    px = B * S * S
    x = torch.randn((B, S, S, 32), device="cuda")
    y = torch.empty_like(x)
    from blind_image_denoising_amd import unet_laplacian as UL
    pk = UL.pack_bneck_h3(torch.randn((32, 32), device="cuda") * 0.2, torch.randn((3, 3, 32, 4), device="cuda") * 0.1,
                          torch.randn((128, 32), device="cuda") * 0.1)
    b1, b2 = torch.randn(128, device="cuda") * 0.1, torch.randn(32, device="cuda") * 0.1
    blk = lambda: N.check(L.bf_op_bneck_block_h3(N.ptr(x), N.ptr(y), N.ptr(pk), None, 1, 0.0, N.ptr(b1), 1, 0.0, N.ptr(b2), 1, 0.0, 1, B, S, S,
                                                  N.stream_ptr(x)), None, "bf_op_bneck_block_h3")
    for _ in range(12):                                   # clocks settle over the first few launches (491 -> 411 us)
        blk()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nl = 30
    e0.record()
    for _ in range(nl):
        blk()
    e1.record()
    torch.cuda.synchronize()
    launch_us = e0.elapsed_time(e1) * 1e3 / nl
    nbytes = px * 32 * 4 * 2
    flop = px * (2.0 * 32 * 32 + 2.0 * 9 * 128 + 2.0 * 128 * 32)
    issued = px * 3 * (2.0 * 32 * 32 * (340.0 / 256.0) + 2.0 * 128 * 32)          # split-f16 products, the leading 1x1 on the haloed tile
    per_px = 2.0 * 49 * 3 * 32 + 6 * (2.0 * 32 * 32 + 2.0 * 9 * 128 + 2.0 * 128 * 32) + 2.0 * (32 * 32 + 32 * 3)
    value = world * B * args.steps / elapsed
    rec = {
        "metric": "denoised images/sec (256x256x3), shipped resnet bottleneck config", "value": value, "unit": "images/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 (block 1x1 convolutions: f16x2 split hi+lo, fp32 accumulate)", "data": "synthetic",
        "config": {"workload": f"resnet_color_1x6_bn_32x128x32_1x3x1 (7x7 base, 1x1 -> depthwise 3x3 x4 -> grouped 1x1, BN, ReLU) inference, "
                               f"batch={B}/GPU {S}x{S}x3 uint8->uint8 (DenoiserModule.__call__)", "batch_per_gpu": B,
                   "parallelism": f"replicas x{world}, no collective"},
        "parity": {"max_abs_lsb": int(diff.max()), "mean_abs_lsb": float(diff.mean()), "checked": "one 64x64 crop vs oracle/resnet_generic_oracle.py"},
        "end_to_end_tflops": value / world * per_px * S * S / 1e12,
        "roofline": {"bound": "hbm", "kernel": "ug_bneck_kernel (1x1 32 -> 32 + depthwise 3x3 x4 + BN + ReLU + 1x1 128 -> 32 + BN + Add; 6 launches per forward)",
                     "achieved": nbytes / launch_us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / launch_us / 1e3 / HBM_PEAK_GBS,
                     "dtype": "f16x2 split operands, fp32 accumulate; depthwise in fp32", "traffic": pmc_traffic(6, B, S, True, "ug_bneck_kernel"),
                     "launch_us": launch_us,
                     "algorithmic_bytes_per_launch": nbytes,
                     "mfma": {"issued_tflops": issued / launch_us / 1e6, "peak": MFMA_F16_PEAK_TFLOPS, "frac": issued / launch_us / 1e6 / MFMA_F16_PEAK_TFLOPS,
                              "useful_gflop_per_launch": flop / 1e9},
                     "valu": {"depthwise_fma_per_launch": px * 9.0 * 128, "note": "fp32 FMAs of the depthwise on the vector ALU: 36 v_pk_fma_f32 "
                              "per pixel group and chunk; the kernel's longest pipe (DESIGN.md 4.6)"}}}
    if not args.no_cpu_baseline:
        # the same graph on torch-CPU fp32 with the host cores this process may use (oracle/resnet_generic_torch.py's forward: the
        # restatement on torch.nn.functional ops), whole images; the fp64 NumPy oracle on the parity crop beside it
        try:
            rec["cpu_baseline"] = cpu_generic_baseline_torch(spec, params, state, base[:2], ref_crop=(crop, ref))
        except Exception as e:
            rec["cpu_baseline"] = {"note": f"torch-CPU leg failed: {str(e)[:200]}"}
    print(json.dumps(rec), flush=True)


def cpu_generic_baseline_torch(spec, params, state, images_u8, budget_s=10.0, nthreads=None, ref_crop=None):
    import torch
    from oracle import resnet_generic_torch as GT
    if nthreads is None:
        nthreads = host_cores()
    torch.set_num_threads(int(nthreads))
    P = GT.views(spec, torch.from_numpy(np.asarray(params, np.float32)))
    St = GT.state_views(spec, torch.from_numpy(np.asarray(state, np.float32)))

    def forward(u8):
        with torch.no_grad():
            y = GT.hydra(spec, P, St, torch.from_numpy(u8.astype(np.float32)), False)[0]
            return torch.round(y).clamp(0, 255).to(torch.uint8).numpy()
    note = {}
    if ref_crop is not None:
        c, r = ref_crop
        note["max_abs_lsb_vs_oracle_on_the_parity_crop"] = int(np.abs(forward(c).astype(np.int32) - r.astype(np.int32)).max())
    forward(images_u8)
    t0 = time.perf_counter()
    nrep = 0
    while nrep < 1 or (time.perf_counter() - t0 < budget_s and nrep < 64):
        forward(images_u8)
        nrep += 1
    dt = (time.perf_counter() - t0) / nrep
    S = images_u8.shape[1]
    return dict({"value": images_u8.shape[0] / dt, "unit": "images/s", "cores": int(nthreads), "kind": "port",
                 "sample": f"{nrep} x {images_u8.shape[0]} {S}x{S}x3 uint8 images through oracle/resnet_generic_torch.py's forward in fp32 (torch-CPU "
                           f"{torch.__version__}, {nthreads} threads; CPU restatement of the same graph, not TensorFlow; {dt * nrep:.1f} s)"}, **note)


def latency_bench(args, torch, bf, O, rank, local_rank, world, dist):
    """configs[0]/[1] territory: single-image latency of DenoiserModule.__call__ (uint8 256x256x3 -> uint8), launched
    op by op on the stream and as ONE captured HIP graph (the C ABI neither allocates nor synchronises, so the whole call
    is capturable); resnet 1x6 / 1x18 and unet_laplacian v5 (512x512)."""
    from oracle import unet_oracle as U
    dev = f"cuda:{local_rank}"
    rows = {}

    def measure(name, module, img):
        x = torch.from_numpy(img).to(dev)
        for _ in range(5):
            module(x)
        torch.cuda.synchronize()
        n = max(args.steps, 50)
        t0 = time.perf_counter()
        for _ in range(n):
            module(x)
        torch.cuda.synchronize()
        t_stream = (time.perf_counter() - t0) / n
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = module(x)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t0) / n
        # the packaged form of the same thing: GraphedDenoiserModule (input copy into the graph's static tensor + replay + output copy)
        gm = bf.GraphedDenoiserModule(module)
        for _ in range(5):
            gm(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            gm(x)
        torch.cuda.synchronize()
        rows[name] = {"stream_us": t_stream * 1e6, "graph_us": t_graph * 1e6, "graphed_module_us": (time.perf_counter() - t0) / n * 1e6}
        return out

    for layers in (6, 18):
        cfg = O.canonical_config(no_layers=layers)
        spec = O.ResnetSpec.from_config(cfg["model"])
        params, state = O.init_params(spec, seed=42, nontrivial_bn=True)
        m = bf.model_builder(cfg["model"], device=dev).hydra
        m.set_weights(params, state)
        for kv in args.opt:
            k, v = kv.split("=")
            m.set_option(k, int(v))
        _, img = O.synthetic_batch(1, 256, 256, sigma=20.0, seed=1234)
        measure(f"resnet_1x{layers} 1x256x256x3", bf.DenoiserModule(m), img)
    ucfg = U.canonical_config()
    um = bf.model_builder(ucfg["model"], device=dev).hydra
    um.set_weights(U.init_params(U.UnetLaplacianSpec.from_config(ucfg["model"]), seed=42))
    _, img = O.synthetic_batch(1, 512, 512, sigma=20.0, seed=1234)
    measure("unet_laplacian_v5 1x512x512x3", bf.DenoiserModule(um), img)
    if rank == 0:
        best = rows["resnet_1x18 1x256x256x3"]
        print(json.dumps({
            "metric": "single-image latency, DenoiserModule.__call__ (us)", "value": best["graph_us"], "unit": "us",
            "n_gpus": 1, "steps": max(args.steps, 50), "warmup": 5, "ms_per_step": best["graph_us"] / 1e3,
            "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f16x2 split (hi+lo, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "batch 1 uint8->uint8, launched op by op on the stream vs replayed as one captured HIP graph"},
            "latency": rows}), flush=True)


def train_bench(args, torch, bf, O, rank, local_rank, world, dist):
    rec = train_run(args, torch, bf, O, rank, local_rank, world, dist, args.steps, args.warmup, cpu=not args.no_cpu_baseline)
    if rec is not None:
        print(json.dumps(rec), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def train_parity_crop(torch, bf, O, cfg, layers, device):
    """the training step of the SAME network on a crop the fp64 oracle finishes in a second (2 x 48 x 48), head kernels scaled by
    0.1 as in tests/test_gpu_training.py::test_config4_network_on_a_reduced_crop_matches_oracle (a freshly initialised 18-block
    network otherwise sits on the denormaliser's clip, whose derivative is discontinuous): loss and gradient error against the oracle."""
    spec = O.ResnetSpec.from_config(cfg["model"])
    ls = O.LossSpec.from_config(cfg["loss"])
    params, state = O.init_params(spec, seed=42, nontrivial_bn=True)
    for name, (o, sh) in spec.offsets().items():
        if name.startswith("head"):
            params[o:o + int(np.prod(sh))] *= 0.1
    m = bf.model_builder(cfg["model"], device=device).hydra
    m.set_weights(params, state)
    fns = bf.build_train_functions(m, bf.loss_function_builder(cfg["loss"]))
    clean, noisy = O.synthetic_batch(2, 48, 48, seed=21)
    gt, x = clean.astype(np.float32), noisy.astype(np.float32)
    total, _, _, _, grads = fns.train_step_single_gpu(torch.from_numpy(gt), torch.from_numpy(x), (1.0,), 0.0, None)
    r_total, _, _, _, r_grads, _ = O.train_step_single_gpu(spec, ls, params, state, gt.astype(np.float64), x.astype(np.float64))
    g = grads.cpu().numpy().astype(np.float64)
    worst = 0.0
    for name, (o, sh) in spec.offsets().items():
        n = int(np.prod(sh))
        worst = max(worst, float(np.abs(g[o:o + n] - r_grads[o:o + n]).max() / max(np.abs(r_grads[o:o + n]).max(), 1e-6)))
    return {"loss_rel_err": float(abs(total.item() - r_total) / abs(r_total)), "grad_max_err_of_tensor_max": worst,
            "bars": {"loss_rel_err": 1e-5, "grad_max_err_of_tensor_max": 6e-4},
            "checked": f"2 x 48 x 48 crop, all {layers} blocks, vs oracle/bfcnn_oracle.py train_step_single_gpu (fp64)"}


def train_kernels(N, model):
    """the block kernels the model's LAST bf_train_step launched, as the LIBRARY reports them (bf_get_train_kernels)"""
    name = N.lib().bf_get_train_kernels(model._h)
    return name.decode() if name else ""


def train_run(args, torch, bf, O, rank, local_rank, world, dist, steps, warmup, cpu=True, exchange=True, parity=True, roofline=True,
              native_collective=False, check_replicas=False, B=None, S=None):
    """configs[3]: resnet_color_1x18 training step, L1 loss (hinge 0.5), additive-gaussian synthetic batch, global batch =
    --batch x world sharded over the ranks, one sum-all-reduce of the flat fp32 gradient buffer, fused clip + Adam.
    EVERY rank calls this (it holds collectives); the record comes back on rank 0 (`--mode train` prints it; the default mode carries
    short runs of it as the `train` / `train_dp` sub-records)."""
    if B is None:
        B = 32 if args.batch == 128 else args.batch        # per-GPU shard (256 global on 8 GPUs)
    if S is None:
        S = args.size
    layers = args.layers
    cfg = O.canonical_config(no_layers=layers)
    if args.loss == "shipped":        # the loss section of the reference's shipped configs (configs/unet_laplacian_v5.json)
        cfg["loss"].update({"hinge": 3.5, "cutoff": 255.0, "mae_multiplier": 1.0, "mse_multiplier": 0.5, "ssim_multiplier": 1.0})
    spec = O.ResnetSpec.from_config(cfg["model"])
    params, state = O.init_params(spec, seed=42 + (rank if check_replicas else 0), nontrivial_bn=False)   # (replica check: ranks START different)
    model = bf.model_builder(cfg["model"], device=f"cuda:{local_rank}").hydra
    model.set_weights(params, state)
    for kv in args.opt:
        k, v = kv.split("=")
        model.set_option(k, int(v))
    opt, _ = bf.optimizer_builder(cfg["train"]["optimizer"])
    own_group = False
    if getattr(args, "force_collective", False) and dist is None:
        import socket
        import torch.distributed as d1
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        d1.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        own_group, native_collective = True, "force"
    trainer = bf.DataParallelTrainer(model, bf.loss_function_builder(cfg["loss"]), opt, native_collective=native_collective)
    trainer.broadcast_parameters()
    clean, noisy = O.synthetic_batch(min(B, 8), S, S, sigma=20.0, seed=1234 + rank)
    reps = (B + clean.shape[0] - 1) // clean.shape[0]
    clean_dev = torch.from_numpy(np.concatenate([clean] * reps)[:B].astype(np.float32)).cuda()
    # fresh corruption every step, on the device (bfcnn/dataset.py:126-239 -> bf_noise_augment): additive truncated-normal
    # noise sigma ~ U[5, 40] and whole-batch flips, each with probability 1/2 -- inside the timed region
    prep = bf.PrepareData({"random_left_right": True, "random_up_down": True, "additional_noise": [5, 40]}, seed=77 + rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # the corruption of batch k+1 is issued between the launch of step k's gradient all-reduce and the point where the
    # compute stream waits for it (DataParallelTrainer.step(overlap=...)): the one piece of a step that does not depend on
    # the reduced gradients
    total = None
    batch = prep(clean_dev)
    for _ in range(max(warmup, 1)):
        total, _, _, _, batch = trainer.step(*batch, overlap=lambda: prep(clean_dev))
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        total, _, _, _, batch = trainer.step(*batch, overlap=lambda: prep(clean_dev))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    replicas = None
    if check_replicas and dist is not None:
        # data-parallel replicas must hold IDENTICAL parameters after the same all-reduced steps (local BatchNorm statistics differ by
        # design): rank 0's vector is broadcast and every rank reports its largest difference
        ref = model.params.clone()
        dist.broadcast(ref, src=0)
        dmax = (model.params - ref).abs().max().reshape(1).to(torch.float64)
        dist.all_reduce(dmax, op=dist.ReduceOp.MAX)
        replicas = float(dmax.item())
    # the dominant kernel INSIDE real steps: option "timing" brackets every bwd_block_h3t_kernel launch with its own HIP-event pair
    # on the launch stream (a debug-entry loop on the same tensors runs 20 % slower: there the 671 MB of operands never sit in the
    # 256 MB Infinity Cache, in a step the gradient was written by the launch before)
    live = None
    if roofline:                                           # (every rank: the steps hold the collective)
        import ctypes as C
        from blind_image_denoising_amd import _native as N0
        model.set_option("timing", 1)
        for _ in range(max(min(steps, 10), 5)):
            total, _, _, _, batch = trainer.step(*batch, overlap=lambda: prep(clean_dev))
        torch.cuda.synchronize()
        ms, ln = C.c_float(), C.c_int()
        if N0.lib().bf_get_timing(model._h, C.byref(ms), C.byref(ln)) == 0 and int(ln.value) > 0:
            live = (float(ms.value) * 1e3 / int(ln.value), int(ln.value))
        model.set_option("timing", 0)
    if own_group:
        if trainer.comm is not None:
            trainer.comm.close()
            trainer.comm = None
        import torch.distributed as d1
        d1.destroy_process_group()
    exch = time_gradient_exchange(torch, bf, model, prep, clean_dev, world, dist) if exchange else None      # every rank takes part
    if trainer.comm is not None:
        trainer.comm.close()
    if rank != 0:
        return None
    from blind_image_denoising_amd import _native as N
    # algorithmic FLOPs per image: fwd + dgrad + wgrad of every 3x3 16->16 conv, fwd + wgrad of the base conv, head fwd + bwd
    per_px = 3 * layers * FLOP_PER_PX_BLOCK + 2 * 2 * 9 * 3 * 16 + 3 * 2 * (16 * 32 + 32 * 3)
    value = B * world * steps / elapsed
    rec = {
        "metric": "training images/sec (256x256x3), resnet_1x18 data-parallel step", "value": value, "unit": "images/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 (convolutions: f16x2 split hi+lo, fp32 accumulate)",
        "data": "synthetic",
        "config": {"workload": f"resnet_color_1x{layers}_bn_16x3x3 training step ("
                               f"{'L1 hinge 3.5 + 0.5 RMSE + SSIM' if args.loss == 'shipped' else 'L1 hinge 0.5'}, Adam, global clipnorm 1), "
                               f"batch={B}/GPU {S}x{S}x3 float32 corrupted on the device every step, one all-reduce of {model.n_params} fp32 gradients",
                   "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                   "collective": ("bf_allreduce_grads (C ABI, RCCL, own stream)" if native_collective else "torch.distributed.all_reduce")
                                 if world > 1 else "none (one rank)"},
        "last_total_loss": float(total.item()),
        "end_to_end_tflops": value / world * per_px * S * S / 1e12,
        "block_kernels": train_kernels(N, model)}
    if replicas is not None:
        rec["replicas_max_abs_param_diff"] = replicas
        rec["replicas_identical"] = replicas == 0.0
    if roofline:
        rec["roofline"] = train_roofline(torch, N, model, layers, B, S, rec["block_kernels"], live)
    if parity:
        try:
            rec["parity"] = train_parity_crop(torch, bf, O, cfg, layers, f"cuda:{local_rank}")
        except Exception as e:
            rec["parity"] = {"note": f"not checked: {str(e)[:200]}"}
    if exch is not None:
        rec["gradient_exchange"] = exch
    if os.environ.get("BF_BENCH_REHEARSE") == "1":
        rec["rehearsal"] = True              # ranks shared a GPU over gloo: exercises the launch path, not a measurement
    if dist is not None:
        rec["world_size_seen"] = int(dist.get_world_size())
    rec["visible_gpus"] = int(torch.cuda.device_count())
    if world == 1 and cpu:
        # leg (i): the same graph on torch-CPU fp32 autograd with all host cores (the reported baseline); leg (ii): the oracle's
        # training step (fp64 NumPy restatement) on ONE image of the same shape, kept beside it
        ls = O.LossSpec.from_config(cfg["loss"])
        c1, n1 = O.synthetic_batch(1, S, S, sigma=20.0, seed=99)
        t0 = time.perf_counter()
        nrep = 0
        while nrep < 1 or (time.perf_counter() - t0 < 6.0 and nrep < 16):           # ~6 s of CPU work
            O.train_step_single_gpu(spec, ls, params, state, c1.astype(np.float64), n1.astype(np.float64))
            nrep += 1
        dt = (time.perf_counter() - t0) / nrep
        oracle_leg = {"value": 1.0 / dt, "unit": "images/s", "cores": 1, "kind": "port",
                      "sample": f"{nrep} x one {S}x{S} image through oracle/bfcnn_oracle.py train_step_single_gpu (fp64 NumPy "
                                f"restatement of forward + loss + backward, BLAS threads as NumPy picks them; not TensorFlow)"}
        try:
            try:                                # the C port's OpenMP team size is what the inference legs use on this box
                from oracle import port
                nthreads = min(host_cores(), int(port.lib(rebuild=False).bfcnn_port_max_threads()))
            except Exception:
                nthreads = host_cores()
            rec["cpu_baseline"] = cpu_train_baseline_torch(spec, ls, params, state, S, nthreads=nthreads)
            rec["cpu_baseline"]["other_leg"] = oracle_leg
        except Exception as e:                      # the torch-CPU leg must never fail the bench
            rec["cpu_baseline"] = dict(oracle_leg, note=f"torch-CPU leg failed: {e}")
    return rec


def train_roofline(torch, N, model, layers, B, S, kernels="", live=None):
    """roofline of the training step's dominant kernel, timed live with HIP events on the launch stream through the C ABI's
    single-kernel entry.  With the one-kernel block backward (bwd_block_h3t_kernel: BatchNorm backward on load, T recomputed, both
    weight gradients, both data gradients, the skip's gradient and the next BatchNorm's sums; one launch per block, ~55 % of the step):
    algorithmic bytes = A_i, g, C_i, C_{i-1} read once and dA' written once (5 * 64 B per pixel).  Otherwise (option / shape): the fused
    backward of a block's second convolution, bwd3x3_h3_kernel<true, 8> (dy, conv_out and x read, dx written: 4 * 64 B per pixel)."""
    L = N.lib()
    px = B * S * S
    if "bwd_block_h3t_kernel" in kernels:
        r = lambda *sh: torch.randn(*sh, device="cuda")
        a, g, c, bnc = r(B, S, S, 16), r(B, S, S, 16) * 0.1, r(B, S, S, 16), r(B, S, S, 16)
        coef = torch.cat([torch.ones(16), torch.full((16,), 0.1), torch.full((16,), 0.01)]).cuda()
        w0, w1 = r(3, 3, 16, 16) * 0.1, r(3, 3, 16, 16) * 0.1
        out, dw1, dw0, st = torch.empty_like(a), torch.empty(2304, device="cuda"), torch.empty(2304, device="cuda"), torch.empty(32, device="cuda")
        scr = torch.empty(int(L.bf_debug_bwd_block_h3t_scratch_floats(B, S, S)), device="cuda")
        calls = [0]

        def run(flags):
            N.check(L.bf_debug_bwd_block_h3t(N.ptr(a), N.ptr(g), N.ptr(c), N.ptr(coef), N.ptr(w0), N.ptr(w1), N.ptr(bnc), N.ptr(out), N.ptr(dw1),
                                             N.ptr(dw0), N.ptr(st), N.ptr(scr), B, S, S, 1, flags | (calls[0] & 1), N.stream_ptr(a)), None, "bwd_block_h3t")
            calls[0] += 1
        run(0)                                             # packs the weights into the scratch
        for _ in range(3):
            run(2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        nl = 20
        e0.record()
        for _ in range(nl):
            run(2)                                         # the kernel alone, alternating walking directions as the step does
        e1.record()
        torch.cuda.synchronize()
        alone_us = e0.elapsed_time(e1) * 1e3 / nl
        launch_us, timed_launches = live if live is not None else (alone_us, nl)
        nbytes = px * 64 * 5
        gbs = nbytes / launch_us / 1e3
        # matrix work: data + weight gradient of both convolutions = 4 * 4 608 FLOP per pixel algorithmic; issued on the f16 matrix pipe per
        # row of a 128-column strip: 27 groups x 15 MFMAs (recomputed conv_0 and the two data gradients on the 144-column grid, 3 split
        # products) + 216 MFMAs of the two weight gradients, 16 384 FLOP each
        alg = px * 4 * 4608 / launch_us / 1e6
        issued = px * (27 * 15 + 216) * 16384 / 128 / launch_us / 1e6
        return {"bound": "hbm", "kernel": "bwd_block_h3t_kernel", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                "traffic": pmc_traffic(layers, B, S, True, "bwd_block_h3t_kernel"), "algorithmic_bytes_per_launch": nbytes,
                "launch_us": launch_us, "launches_of_this_kernel_per_step": layers, "launches_timed": timed_launches,
                "timed": "one HIP-event pair per launch inside real training steps (option timing)" if live is not None else "single-kernel entry, back to back",
                "launch_us_alone_cold_operands": alone_us,
                "note": "the launch is bound by matrix / LDS issue at three waves per SIMD (stamps: profiles/r04_bwd_block_stamps.txt), not by HBM",
                "mfma": {"dtype": "f16 (split hi/lo, 3 products)", "algorithmic_tflops": alg, "issued_tflops": issued,
                         "peak_tflops": MFMA_F16_PEAK_TFLOPS, "issued_frac": issued / MFMA_F16_PEAK_TFLOPS}}
    xw = torch.relu(torch.randn((B, S, S, 16), device="cuda"))
    gw = torch.randn((B, S, S, 16), device="cuda") * 0.1
    cw = torch.randn((B, S, S, 16), device="cuda")
    coef = torch.cat([torch.ones(16), torch.full((16,), 0.1), torch.full((16,), 0.01)]).cuda()
    wk = torch.randn((3, 3, 16, 16), device="cuda") * 0.1
    dxw = torch.empty_like(xw)
    dw = torch.empty(2304, device="cuda")
    scr = torch.empty(int(L.bf_debug_bwd3x3_h3_scratch_floats(B, S, S)), device="cuda")
    calls = [0]

    def wg():
        N.check(L.bf_debug_bwd3x3_h3(N.ptr(xw), N.ptr(gw), N.ptr(cw), N.ptr(coef), N.ptr(wk), N.ptr(dxw), None, None, N.ptr(dw),
                                     None, N.ptr(scr), B, S, S, N.EPI_MASK, calls[0] & 1, 1 if calls[0] == 0 else 0,
                                     N.stream_ptr(xw)), None, "bwd3x3_h3")
        calls[0] += 1
    for _ in range(3):
        wg()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nl = 20
    e0.record()
    for _ in range(nl):
        wg()
    e1.record()
    torch.cuda.synchronize()
    launch_us = e0.elapsed_time(e1) * 1e3 / nl
    wbytes = px * 16 * 4 * 4
    gbs = wbytes / launch_us / 1e3
    return {"bound": "hbm", "kernel": "bwd3x3_h3_kernel<true, 8> (+ reduce_partials_kernel)", "achieved": gbs, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": pmc_traffic(layers, B, S, True, "bwd3x3_h3_kernel<true, 8>"), "algorithmic_bytes_per_launch": wbytes,
            "launch_us": launch_us, "launches_of_this_kernel_per_step": layers}


def train_dp_record(args, torch, bf, O, rank, local_rank, world, dist, steps=10, warmup=3):
    """N > 1 in the default mode: the data-parallel training step of configs[3] over this run's process group, once with
    torch.distributed's all-reduce and once with the C ABI's own (bf_allreduce_grads on the communicator's stream), replicas that
    START different checked identical afterwards, and what the exchange costs.  Every rank calls this."""
    rec = {"world_size_seen": int(dist.get_world_size()), "backend": dist.get_backend()}
    rehearse = os.environ.get("BF_BENCH_REHEARSE") == "1"
    for key, native in (("torch_all_reduce", False), ("native_collective", True)):
        if native and rehearse:
            if rank == 0:
                rec[key] = {"note": "rehearsal (ranks share a GPU over gloo): RCCL cannot form a communicator with duplicate devices, skipped"}
            continue
        ok, r, err = True, None, ""
        try:
            r = train_run(args, torch, bf, O, rank, local_rank, world, dist, steps, warmup, cpu=False, exchange=not native, parity=False,
                          roofline=False, native_collective=native, check_replicas=True)
        except Exception as e:                      # (NativeCommunicator raises on ALL ranks together, see train_loop.py)
            ok, err = False, str(e)[:200]
        if not all_agree(torch, dist, ok):
            if rank == 0:
                rec[key] = {"note": f"failed on a rank: {err}"}
            continue
        if rank == 0:
            keep = ("value", "unit", "steps", "warmup", "ms_per_step", "last_total_loss", "replicas_max_abs_param_diff", "replicas_identical",
                    "block_kernels", "gradient_exchange", "rehearsal")
            rec[key] = {k: r[k] for k in keep if k in r}
            rec[key]["config"] = r["config"]
    return rec if rank == 0 else None


def self_launch(n):
    import socket
    import subprocess
    with socket.socket() as sk:                       # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes needs it on this driver)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env)
    raise SystemExit(proc.returncode)


def block_kernel_info(N, model):
    """name of the kernel that ran most of the residual-block launches of the model's last forward and the number of block
    launches of one forward, as the LIBRARY reports them (bf_get_block_kernel): never derived from option values here."""
    lpf = C.c_int()
    name = N.lib().bf_get_block_kernel(model._h, C.byref(lpf))
    return (name.decode() if name else ""), int(lpf.value)


def parity_sample(B):
    """indices of the images checked against the oracle: spread over the batch (band scheduling gives different CUs different images)"""
    return sorted(set([0, B // 3, (2 * B) // 3, B - 1]))


def block_roofline(kernel, launches_per_forward, layers, B, S, launch_s, traffic):
    """roofline object of the dominant residual-block kernel.  Algorithmic work per launch: one activation read and
    one written = B*S*S*128 bytes; 9 216 FLOP per pixel and BLOCK, times the blocks a launch runs (2 for fused_block2_h3w_kernel,
    1 for the one-block kernels, 1/2 for one convolution per launch)."""
    px = B * S * S
    blocks_per_launch = layers / max(launches_per_forward, 1)
    flop = px * FLOP_PER_PX_BLOCK * blocks_per_launch
    nbytes = px * BYTES_PER_PX_BLOCK
    tf, gbs = flop / launch_s / 1e12, nbytes / launch_s / 1e9
    common = {"kernel": kernel, "launch_us": launch_s * 1e6, "launches_per_forward": launches_per_forward,
              "blocks_per_launch": blocks_per_launch, "algorithmic_bytes_per_launch": nbytes,
              "algorithmic_gflop_per_launch": flop / 1e9, "traffic": traffic}
    if kernel == "fused_block2_h3w_kernel":
        # two blocks per launch: the activation between them never leaves the CU, the launch is bound by the f16 matrix pipe under
        # the 1 400 W package cap (233 us without its MFMAs, 349 us without its memory instructions, 437-455 us with
        # both).  issued = 3 split products + residual, on the 144-column grid of a 128-column strip (x 9/8)
        issued = px * H3_MFMA_FLOP_PER_PX * blocks_per_launch * 9 / 8 / launch_s / 1e12
        return dict(common, bound="mfma", achieved=tf, peak=MFMA_F16_PEAK_TFLOPS, unit="TFLOP/s", frac=tf / MFMA_F16_PEAK_TFLOPS,
                    dtype="f16 (split hi/lo, 3 products), fp32 accumulate", issued_tflops=issued,
                    issued_frac=issued / MFMA_F16_PEAK_TFLOPS,
                    hbm={"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS},
                    bound_note="neither roof is reached: the launch is limited by the issue rate of its MFMA + ds_read + epilogue mix at three "
                               "waves per SIMD (~21 cycles per MFMA slot) under the 1 400 W package cap (clock 1.9 GHz); of the two roofs the "
                               "matrix pipe is the nearer (issued_frac against hbm.frac) -- DESIGN 4.1c, profiles/r03_mfma_mix.txt, "
                               "r03_power_probe.txt, r04_mfma_fp8_mix.txt")
    if kernel.startswith("fused_block_h3"):
        issued = px * H3_MFMA_FLOP_PER_PX * blocks_per_launch / launch_s / 1e12
        return dict(common, bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS,
                    mfma={"dtype": "f16 (split hi/lo, 3 products)", "algorithmic_tflops": tf, "issued_tflops": issued,
                          "peak_tflops": MFMA_F16_PEAK_TFLOPS, "issued_frac": issued / MFMA_F16_PEAK_TFLOPS})
    return dict(common, bound="mfma", achieved=tf, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s", frac=tf / MFMA_F32_PEAK_TFLOPS,
                dtype="f32")


def sub_record(model, module, noisy, noisy_host, spec, params, state, O, N, torch, arith, steps, warmup, S, compact=0, check=None):
    """one more timed loop of the default workload on `model` with another arithmetic / batch / activation layout / image size (rank 0,
    N = 1).  check: indices of the images compared with the oracle (default: parity_sample)."""
    B = int(noisy.shape[0])
    model.set_option("arith", arith)
    model.set_option("h3_compact", compact)
    model.set_option("timing", 1)
    for _ in range(warmup):
        out = module(noisy)
    torch.cuda.synchronize()
    model.set_option("timing", 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = module(noisy)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms, ln = C.c_float(), C.c_int()
    N.check(N.lib().bf_get_timing(model._h, C.byref(ms), C.byref(ln)), model._h)
    launch_s = float(ms.value) / 1e3 / max(int(ln.value), 1)
    kernel, lpf = block_kernel_info(N, model)
    idx = parity_sample(B) if check is None else list(check)
    ref = O.denoiser_module_call(spec, params, state, noisy_host[idx])
    diff = np.abs(out[idx].cpu().numpy().astype(np.int32) - ref.astype(np.int32))
    px = B * S * S
    rec = {"value": B * steps / elapsed, "unit": "images/s", "batch_per_gpu": B, "steps": steps, "warmup": warmup,
           "ms_per_step": elapsed / steps * 1e3,
           "parity": {"mae_vs_oracle_lsb": float(diff.mean()), "max_abs_lsb": int(diff.max()), "checked_images": len(idx)},
           "dtype": "f32" if arith == 0 else "f16x2 split (hi+lo, fp32 accumulate)",
           "arithmetic": "exact fp32 MFMA" if arith == 0 else "split-f16 MFMA (f16x3), fp32 accumulate",
           "roofline": block_roofline(kernel, lpf, spec.no_layers, B, S, launch_s, None)}
    if compact:
        # 48 instead of 64 bytes per pixel between the launches (fp8 lo planes): `achieved` stays on the ALGORITHMIC 128 B per pixel and
        # block of SURVEY 8(d), `stored_gbs` is what the layout actually moves
        rec["activation_layout"] = "compact split-planar: f16 hi planes + fp8 (e4m3, x 2^12) lo planes, 48 B per pixel"
        rec["roofline"]["stored_bytes_per_px_block"] = 96
        rec["roofline"]["stored_gbs"] = px * 96 / launch_s / 1e9
    model.set_option("arith", 1)
    model.set_option("h3_compact", 0)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # >= 100 timed iterations after >= 20 warm-up ones
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=128, help="images per GPU")
    ap.add_argument("--layers", type=int, default=18)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sub-records", action="store_true", help="default mode: skip the fp32_exact / batch64 sub-records")
    ap.add_argument("--unfused", action="store_true", help="one kernel per convolution (A/B only)")
    ap.add_argument("--fused-tile", type=int, default=None, help="exact-fp32 fused-block tile geometry variant (A/B only)")
    ap.add_argument("--arith", type=int, default=1, help="1 = split-f16 fused blocks (default), 0 = exact-fp32 fused blocks")
    ap.add_argument("--h3-variant", type=int, default=None, help="split-f16 kernel variant (A/B only)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT", help="bf_set_option on the model (A/B only)")
    ap.add_argument("--loss", choices=["l1", "shipped"], default="l1",
                    help="--mode train: l1 = BASELINE configs[3] (L1 only); shipped = L1 + RMSE + SSIM as the reference's configs")
    ap.add_argument("--force-collective", action="store_true",
                    help="--mode train with one GPU: run the step's gradient all-reduce anyway, through the C ABI's communicator in a one-rank "
                         "RCCL group (the two-stream trace of a data-parallel step, tools/dp_trace.py)")
    ap.add_argument("--unet-graph", choices=["v5", "v5.6"], default="v5",
                    help="--mode unet: v5 = snapshot builder graph, random weights; v5.6 = the reference's trained network")
    ap.add_argument("--mode", choices=["inference", "train", "pyramid", "unet", "latency", "generic"], default="inference",
                    help="train: BASELINE.json configs[3] -- one data-parallel training step per step (L1 loss, "
                         "batch sharded over the ranks, ONE gradient all-reduce, clip + Adam); not the headline metric")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` by itself: THIS process never touches a GPU (no torch import, no HIP call); it starts
        # one rank per GPU through torch.distributed.run as a child, relays the child's output (rank 0 prints the JSON
        # line) and exits with its code.
        return self_launch(args.gpus)

    import torch
    import blind_image_denoising_amd as bf
    from blind_image_denoising_amd import _native as N
    from oracle import bfcnn_oracle as O          # checker + synthetic workload generator only

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU (or run bench.py by itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU execution path")
    # BF_BENCH_REHEARSE=1: the N ranks share the visible GPU(s) and rendezvous over gloo -- exercises the launch / barrier /
    # max-over-ranks path on a one-GPU box; the line it prints says "rehearsal" and is not a measurement
    rehearse = os.environ.get("BF_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        # a scaling run must prove that the collective backend saw N ranks on N devices
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
    if not rehearse and torch.cuda.device_count() < min(args.gpus, world):
        raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} device(s) are visible to rank {rank}")

    if args.mode == "train":
        return train_bench(args, torch, bf, O, rank, local_rank, world, dist)
    if args.mode == "pyramid":
        return pyramid_bench(args, torch, bf, O, rank, local_rank, world, dist)
    if args.mode == "unet":
        return unet_bench(args, torch, bf, O, rank, local_rank, world, dist)
    if args.mode == "latency":
        return latency_bench(args, torch, bf, O, rank, local_rank, world, dist)
    if args.mode == "generic":
        return generic_bench(args, torch, bf, O, rank, local_rank, world, dist)
    cfg = O.canonical_config(no_layers=args.layers)
    spec = O.ResnetSpec.from_config(cfg["model"])
    params, state = O.init_params(spec, seed=42, nontrivial_bn=True)
    model = bf.model_builder(cfg["model"], device=f"cuda:{local_rank}").hydra
    model.set_weights(params, state)
    if args.unfused:
        model.set_option("fused_blocks", 0)
    if args.fused_tile is not None:
        model.set_option("fused_tile", args.fused_tile)
    if args.h3_variant is not None:
        model.set_option("h3_variant", args.h3_variant)
    model.set_option("arith", args.arith)
    for kv in args.opt:
        k, v = kv.split("=")
        model.set_option(k, int(v))
    h3 = bool(args.arith) and not args.unfused
    model.set_option("timing", 1)
    module = bf.DenoiserModule(model)

    B, S = args.batch, args.size
    # 16 distinct synthetic images tiled to the batch (generation cost only; every image is processed)
    _, base = O.synthetic_batch(min(B, 16), S, S, sigma=20.0, seed=1234 + rank)
    reps = (B + base.shape[0] - 1) // base.shape[0]
    noisy_host = np.concatenate([base] * reps, axis=0)[:B]
    noisy = torch.from_numpy(noisy_host).cuda()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    for _ in range(max(args.warmup, 1)):
        out = module(noisy)
    barrier()
    model.set_option("timing", 1)                  # restart the event window: the timed region only
    block_ms, launches = [], 0
    ms, ln = C.c_float(), C.c_int()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = module(noisy)
    barrier()
    elapsed = time.perf_counter() - t0
    # HIP events on the launch stream bracket the residual-block launches of EVERY timed step (ring of 256 steps)
    N.check(N.lib().bf_get_timing(model._h, C.byref(ms), C.byref(ln)), model._h)
    block_ms, launches = float(ms.value), int(ln.value)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # a >= 2 s loop of the same step behind the timed region: the package sits at its power cap, so the steady-state rate is a few
    # percent below what a 0.1-0.5 s window shows (profiles/r03_power_probe.txt); every rank runs it, rank 0 reports its own
    sustained = None
    if not args.no_sub_records:
        n_sus = max(int(2.2 / max(elapsed / args.steps, 1e-4)), args.steps)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_sus):
            out = module(noisy)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        sustained = {"value": B * world * n_sus / dt, "unit": "images/s", "steps": n_sus, "seconds": dt, "ms_per_step": dt / n_sus * 1e3,
                     "note": "same step, same inputs, run back to back behind the timed region (rank 0's clock, no barrier inside)"}
    # N > 1: the ONE collective of the path -- the data-parallel training step of configs[3] over this run's process group
    train_dp = None
    if world > 1 and not args.no_sub_records:
        train_dp = train_dp_record(args, torch, bf, O, rank, local_rank, world, dist)

    if rank == 0:
        # parity of the timed configuration: four images spread over the batch against the fp64 oracle (+-1 LSB bar)
        idx = parity_sample(B)
        ref = O.denoiser_module_call(spec, params, state, noisy_host[idx])
        got = out[idx].cpu().numpy()
        diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
        images = B * world * args.steps
        value = images / elapsed
        avg_launch_s = block_ms / 1e3 / max(launches, 1)
        kernel, lpf = block_kernel_info(N, model)
        roofline = block_roofline(kernel, lpf, args.layers, B, S, avg_launch_s, pmc_traffic(args.layers, B, S, not args.unfused, kernel))
        roofline["launches_in_timed_region"] = launches
        result = {
            "metric": "denoised images/sec (256x256x3) + MAE vs ref, resnet_1x18",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16x2 split (hi+lo, fp32 accumulate)" if h3 else "f32", "data": "synthetic",
            "config": {"workload": f"resnet_color_1x{args.layers}_bn_16x3x3 inference, batch={B}/GPU {S}x{S}x3 uint8->uint8 "
                                   f"(DenoiserModule.__call__ via bf_forward_u8)",
                       "batch_per_gpu": B, "global_batch": B * world, "height": S, "width": S,
                       "blocks": args.layers, "fused_blocks": not args.unfused,
                       "arithmetic": "split-f16 MFMA (f16x3), fp32 accumulate" if h3 else "exact fp32 MFMA",
                       "parallelism": f"replicas x{world}, no collective"},
            "parity": {"mae_vs_oracle_lsb": float(diff.mean()), "max_abs_lsb": int(diff.max()), "checked_images": len(idx)},
            "end_to_end_tflops": value / world * gflop_per_image(args.layers, S, S) / 1e3,
            "roofline": roofline,
        }
        if rehearse:
            result["rehearsal"] = True              # ranks shared a GPU over gloo: exercises the launch path, not a measurement
        if dist is not None:
            result["world_size_seen"] = int(dist.get_world_size())
        result["visible_gpus"] = int(torch.cuda.device_count())
        if sustained is not None:
            result["sustained"] = sustained
        if train_dp is not None:
            result["train_dp"] = train_dp
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(spec, params, state, noisy_host)
        if world == 1 and not args.no_sub_records and h3 and args.h3_variant is None:
            # the same workload on the other arithmetic / at the north star's batch, timed by this very run (fewer steps):
            #   fp32_exact : --arith 0, exact fp32 on the f32 matrix cores (fused_block_v4_kernel), vs the 157.3 TF fp32 MFMA peak
            #   batch64    : default arithmetic at batch 64 (BASELINE.json north_star: "1x18 resnet 3x3 conv stack at batch 64")
            #   fp32_exact_batch64 : the exact-fp32 kernel at that batch (the north star's ">= 60 % MFMA roofline" reading)
            #   compact24  : default arithmetic, activations between the launches as f16 hi + fp8 lo planes (set_option h3_compact 1)
            result["fp32_exact"] = sub_record(model, module, noisy, noisy_host, spec, params, state, O, N, torch, arith=0,
                                              steps=max(args.steps // 4, 10), warmup=max(args.warmup // 4, 3), S=S)
            result["fp32_exact_batch64"] = sub_record(model, module, noisy[:64].contiguous(), noisy_host[:64], spec, params, state, O, N,
                                                      torch, arith=0, steps=max(args.steps // 4, 10), warmup=max(args.warmup // 4, 3), S=S)
            result["batch64"] = sub_record(model, module, noisy[:64].contiguous(), noisy_host[:64], spec, params, state, O, N, torch,
                                           arith=1, steps=max(args.steps // 2, 10), warmup=max(args.warmup // 2, 3), S=S)
            result["compact24"] = sub_record(model, module, noisy, noisy_host, spec, params, state, O, N, torch, arith=1,
                                             steps=max(args.steps // 2, 10), warmup=max(args.warmup // 2, 3), S=S, compact=1)
            #   wide_512   : the same network and pixel count on 512 x 512 images (batch / 4): wider than the 256 columns of the one-block
            #                streaming kernel, two blocks per launch on 128-column strips (round 3; the tile kernel before)
            if S == 256 and B % 4 == 0:
                _, wide4 = O.synthetic_batch(4, 512, 512, sigma=20.0, seed=4321)
                wide_host = np.concatenate([wide4] * (B // 16 + 1), axis=0)[:B // 4]
                result["wide_512"] = sub_record(model, module, torch.from_numpy(wide_host).cuda(), wide_host, spec, params, state, O, N, torch,
                                                arith=1, steps=max(args.steps // 4, 10), warmup=max(args.warmup // 4, 3), S=512,
                                                check=[0, B // 4 - 1])
                result["wide_512"]["workload"] = f"resnet_color_1x{spec.no_layers}_bn_16x3x3 inference, batch={B // 4} 512x512x3 uint8"
            model.set_option("arith", 1)
            model.set_option("h3_compact", 0)
        if world == 1 and not args.no_sub_records:
            # BASELINE configs[3] and configs[4] in the driver-run line: short runs of `--mode train` / `--mode unet` (their own
            # roofline, parity against the oracle on a crop, the kernels the library says it launched; CPU legs only in their modes)
            try:
                result["train"] = train_run(args, torch, bf, O, rank, local_rank, world, None, steps=max(args.steps // 5, 10),
                                            warmup=max(args.warmup // 4, 3), cpu=False, exchange=False, B=32, S=256)
            except Exception as e:
                result["train"] = {"note": f"failed: {str(e)[:300]}"}
            try:
                result["unet"] = unet_run(args, torch, bf, O, rank, local_rank, world, None, steps=max(args.steps // 10, 10),
                                          warmup=max(args.warmup // 4, 3), cpu=False, B=32, S=512)
            except Exception as e:
                result["unet"] = {"note": f"failed: {str(e)[:300]}"}
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if train_dp is not None and any(isinstance(v, dict) and v.get("replicas_identical") is False for v in train_dp.values()):
        raise SystemExit("train_dp: data-parallel replicas differ after the all-reduced steps")


if __name__ == "__main__":
    main()
