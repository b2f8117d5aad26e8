"""DenoiserModule: the inference drop-in boundary (bfcnn/module_denoiser.py:15-75)."""
import numpy as np
import torch

from .constants import DENOISER_STR


class _DeferredStatus:
    """The f16-range status words of calls that handed back DEVICE tensors: each is copied to its own pinned host slot behind its
    call (stream-ordered, nothing synchronises) and looked at when a later call starts or when the caller asks.  The engine clears
    the device word at the start of every forward, so one shared slot would only ever show the LAST call of a pipelined loop
    (the host runs ahead of the GPU: the earlier events are not complete when the next call polls); here every call keeps its
    slot until it has been read, and a slot about to be reused is waited for first.  Host-array calls do not need any of this:
    they synchronise anyway and check at once."""
    SLOTS = 16

    def __init__(self):
        self.host, self.pending, self.free, self.carried = None, [], [], 0

    @staticmethod
    def _capturing() -> bool:
        # a call that is being captured into a HIP graph may neither allocate pinned memory nor query events, and an event
        # recorded inside a capture cannot be queried afterwards: captured calls are not tracked here (after a replay, ask
        # HydraModel.check_status(), which reads the device word)
        return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()

    def post(self, status_dev: torch.Tensor):
        if self._capturing():
            return
        if self.host is None:
            self.host = torch.zeros(self.SLOTS, dtype=torch.int32).pin_memory()
            self.free = list(range(self.SLOTS))
        if not self.free:                                   # every slot is waiting to be read: read the oldest now
            slot, event = self.pending.pop(0)
            event.synchronize()
            self.carried |= int(self.host[slot])
            self.free.append(slot)
        slot = self.free.pop(0)
        self.host[slot:slot + 1].copy_(status_dev.reshape(-1)[:1], non_blocking=True)
        event = torch.cuda.Event()
        event.record(torch.cuda.current_stream(status_dev.device))
        self.pending.append((slot, event))

    def poll(self, wait: bool = False) -> int:
        """OR of the status words of every posted call that is known by now (all of them with wait=True); each is reported once."""
        if self._capturing():
            return 0
        status, self.carried = self.carried, 0
        while self.pending:
            slot, event = self.pending[0]
            if wait:
                event.synchronize()
            elif not event.query():
                break                                       # completion is in stream order: the later ones are not done either
            self.pending.pop(0)
            status |= int(self.host[slot])
            self.free.append(slot)
        return status


class DenoiserModule:
    """denoising inference module: uint8 [B,H,W,C] -> uint8 [B,H,W,C].

    Reference: cast -> pad_to_power_of_2 -> hydra -> [take output 0] -> remove_padding ->
    tf.round -> uint8 (module_denoiser.py:46-75).  Here the whole chain is ONE C-ABI call
    (bf_forward_u8): the cast/normalise is fused into the base-convolution kernel, the padding
    is virtual, and denormalise + crop + round-half-even + cast are fused into the head kernel."""

    def __init__(self, model_hydra, cast_to_uint8: bool = True):
        # --- argument checking (module_denoiser.py:31-33)
        if model_hydra is None:
            raise ValueError("model_denoise should not be None")
        self.name = DENOISER_STR
        self._cast_to_uint8 = cast_to_uint8
        self._model_hydra = model_hydra
        self._deferred = _DeferredStatus()

    @property
    def model_hydra(self):
        return self._model_hydra

    def check_status(self, wait: bool = True) -> bool:
        """f16-range status of the last call that returned device tensors (waits for it by default).  On a hit: with
        `auto_exact_fallback` the model is switched to the exact-fp32 kernels for every later call and False comes back
        (the tensor that call returned is NOT trustworthy); without it FloatingPointError is raised."""
        from . import _native as N
        hydra = self._model_hydra
        if not (self._deferred.poll(wait) & N.BF_STATUS_F16_RANGE):
            return True
        if not getattr(hydra, "auto_exact_fallback", False):
            raise FloatingPointError("an activation left the f16 range inside the split-f16 kernels during the previous "
                                     "call: its output is not trustworthy; use set_option('arith', 0)")
        from .custom_logger import logger
        logger.warning("an activation left the f16 range inside the split-f16 kernels during the previous call (its output "
                       "is not trustworthy): switching this model to the exact-fp32 kernels")
        hydra.set_option("arith", 0)
        return False

    def __call__(self, image):
        """image: uint8 tensor of rank 4 (the input_signature of module_denoiser.py:43-45)."""
        was_numpy = isinstance(image, np.ndarray)
        if was_numpy:
            image = torch.from_numpy(np.ascontiguousarray(image))
        if not isinstance(image, torch.Tensor):
            raise ValueError("image must be a torch.Tensor or numpy array")
        if image.dtype != torch.uint8 or image.dim() != 4:
            raise ValueError(f"input must be a uint8 tensor of shape [B,H,W,C], "
                             f"got {image.dtype} {tuple(image.shape)}")
        hydra = self._model_hydra
        if image.shape[-1] != hydra.desc.in_channels:
            raise ValueError(f"expected {hydra.desc.in_channels} channels, got {image.shape[-1]}")
        if image.shape[0] == 0:
            out = torch.empty((0,) + tuple(image.shape[1:3]) + (hydra.desc.out_channels,), dtype=torch.uint8)
            return out.numpy() if was_numpy else out
        hydra._require_gpu()
        self.check_status(wait=False)          # the previous device-tensor call, if its status has arrived by now
        image = image.to(hydra.device).contiguous()
        # one C-ABI call: cast, virtual power-of-two padding, hydra, crop and (cast_to_uint8) round + cast all run in the
        # engine; a multi-output hydra keeps its first, full-resolution output (module_denoiser.py:62-65)
        out = hydra.infer_u8(image, self._cast_to_uint8)
        if was_numpy and not hydra.check_status(raise_on_overflow=not hydra.auto_exact_fallback):
            # host arrays are handed back (the stream is synchronised anyway) and an activation left the f16 range:
            # switch this model to the exact-fp32 kernels for good and repeat the call
            hydra.set_option("arith", 0)
            return self(image.cpu().numpy())
        if was_numpy:
            return out.cpu().numpy()
        st = hydra.status_tensor() if hasattr(hydra, "status_tensor") else None
        if st is not None:
            self._deferred.post(st)            # no synchronisation: looked at by the next call / check_status()
        return out


class GraphedDenoiserModule:
    """A DenoiserModule whose call is replayed as ONE captured HIP graph per input shape -- the serving form of the module.

    The C ABI behind `DenoiserModule.__call__` neither allocates nor synchronises (include/bfcnn_hip.h, "graph-capturable"), so a
    whole call -- eleven to twenty launches for resnet 1x18, about sixty operator calls from Python for unet_laplacian -- can be
    captured once per shape and then costs the host a single launch.  What that buys is HOST time (one launch instead of a walk over
    the graph in Python), not latency: a batch-1 call is bound by its kernels on the GPU either way (`bench.py --mode latency`:
    resnet 1x18 133 us replayed, 142 us launched on the stream; unet_laplacian v5 at 512 x 512: 750-780 us either way).  The first call with a
    new [B,H,W,C] runs the module twice directly (packing, workspace, kernel attributes), captures it on a static input tensor and
    keeps the graph; later calls copy the image into that tensor and replay.  At most `max_shapes` graphs are kept (least recently
    used goes first).  The result is a fresh tensor unless `copy_output=False` (then it is the graph's own output buffer, valid until
    the next call with that shape).  Same argument checks, return types and f16-range status handling as DenoiserModule
    (module_denoiser.py:43-75 is what both compute).  A graph holds the kernels, options and buffer addresses of the moment it was
    captured, and the captured kernels read the PACKED weights, which only a direct call refreshes: every graph therefore remembers
    the model's `version` (bumped by `set_weights`, `mark_dirty` -- a training or optimizer step -- and `set_option`, the automatic
    switch to the exact-fp32 kernels included) and the address of its workspace, and a graph whose model has moved on since is
    re-captured by itself on its next use (two direct calls + one capture); `invalidate()` drops all of them by hand."""

    def __init__(self, module: DenoiserModule, max_shapes: int = 8, copy_output: bool = True):
        if not isinstance(module, DenoiserModule):
            raise ValueError("module must be a DenoiserModule")
        if max_shapes < 1:
            raise ValueError("max_shapes must be at least 1")
        self._module, self._max, self._copy = module, int(max_shapes), bool(copy_output)
        self._graphs = {}                      # shape -> (graph, static_in, static_out); insertion order = recency
        self.name = module.name

    @property
    def model_hydra(self):
        return self._module.model_hydra

    def check_status(self, wait: bool = True) -> bool:
        return self._module.check_status(wait)

    def captured_shapes(self):
        return list(self._graphs.keys())

    def invalidate(self):
        """drops every captured graph (after set_option: a graph replays the kernels that were selected when it was captured)"""
        self._graphs.clear()

    def _workspace_ptr(self):
        ws = getattr(self._module.model_hydra, "_workspace", None)
        return ws.data_ptr() if isinstance(ws, torch.Tensor) else 0

    def _stamp(self):
        """what a captured graph depends on besides the input shape: the model's weights / options and its workspace"""
        return (getattr(self._module.model_hydra, "version", 0), self._workspace_ptr())

    def _capture(self, image: torch.Tensor):
        static_in = image.clone()
        side = torch.cuda.Stream(device=image.device)
        side.wait_stream(torch.cuda.current_stream(image.device))
        with torch.cuda.stream(side):          # warm-up off the capture: one-time packing / workspace / attribute calls
            self._module(static_in)
            self._module(static_in)
        torch.cuda.current_stream(image.device).wait_stream(side)
        torch.cuda.synchronize(image.device)
        self._module.check_status(wait=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = self._module(static_in)
        return graph, static_in, static_out, self._stamp()

    def __call__(self, image):
        was_numpy = isinstance(image, np.ndarray)
        if was_numpy:
            image = torch.from_numpy(np.ascontiguousarray(image))
        if not isinstance(image, torch.Tensor):
            raise ValueError("image must be a torch.Tensor or numpy array")
        if image.dtype != torch.uint8 or image.dim() != 4:
            raise ValueError(f"input must be a uint8 tensor of shape [B,H,W,C], "
                             f"got {image.dtype} {tuple(image.shape)}")
        hydra = self._module.model_hydra
        if image.shape[-1] != hydra.desc.in_channels:
            raise ValueError(f"expected {hydra.desc.in_channels} channels, got {image.shape[-1]}")
        if image.shape[0] == 0:
            return self._module(image.numpy() if was_numpy else image)
        hydra._require_gpu()
        self._module.check_status(wait=False)  # (may switch the model to the exact-fp32 kernels: that bumps its version)
        image = image.to(hydra.device).contiguous()
        key = tuple(image.shape)
        entry = self._graphs.pop(key, None)
        if entry is not None and entry[3] != self._stamp():
            entry = None                       # weights, options or the engine's workspace changed since this graph was captured
        if entry is None:
            if len(self._graphs) >= self._max:
                self._graphs.pop(next(iter(self._graphs)))
            entry = self._capture(image)
        self._graphs[key] = entry              # most recently used last
        graph, static_in, static_out, _ = entry
        static_in.copy_(image)
        graph.replay()
        if was_numpy:
            out = static_out.cpu().numpy()     # synchronises: the status word of this replay can be read at once
            if not hydra.check_status(raise_on_overflow=not hydra.auto_exact_fallback):
                hydra.set_option("arith", 0)   # the captured graphs hold the split-f16 kernels: drop them
                self._graphs.clear()
                return self(image.cpu().numpy())
            return out
        st = hydra.status_tensor() if hasattr(hydra, "status_tensor") else None
        if st is not None:
            self._module._deferred.post(st)
        return static_out.clone() if self._copy else static_out
