"""Host -> device traffic of the hot path without stream synchronisation.

`tensor.new_tensor(numpy_array)` / `torch.tensor(list, device='cuda')` copy from pageable memory: the host
blocks until the stream has drained (torch.cuda.set_sync_debug_mode flags each one).  The step makes ~170
such copies of tiny per-sample matrices and shape constants; here they go through pinned staging buffers
(`to_device_async`) or are built once and cached (`const_tensor`)."""
from __future__ import annotations

import numpy as np
import torch

_CONST: dict = {}


class _PinnedArena:
    """One pinned staging buffer per process, bump-allocated and reused round-robin.  `tensor.pin_memory()`
    per copy is not an option: a pinned block can only be recycled once its copy has executed, the host runs
    far ahead of the GPU, so every call ends in a fresh hipHostMalloc (a slow, serialising driver call).
    A slice is overwritten only after a full lap (8 MB = a few hundred training steps of these tiny copies);
    at the wrap-around the stream is drained once, so no copy still in flight can see its source overwritten."""

    def __init__(self, nbytes=8 << 20):
        self.buf = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        self.pos = 0

    def stage(self, src):
        """copy the CPU tensor `src` into the arena -> pinned view with src's dtype / shape"""
        n = src.numel() * src.element_size()
        if n > self.buf.numel() // 4:
            return src.pin_memory()
        start = (self.pos + 15) & ~15
        if start + n > self.buf.numel():
            torch.cuda.synchronize()             # once per lap: every copy staged so far has executed
            start = 0
        self.pos = start + n
        view = self.buf[start:start + n].view(src.dtype).view(src.shape)
        view.copy_(src)
        return view


_ARENA = None


def to_device_async(array, device, dtype=torch.float32):
    """numpy / nested list -> device tensor via the pinned arena, non-blocking."""
    global _ARENA
    device = torch.device(device)
    t = torch.as_tensor(np.ascontiguousarray(np.asarray(array)), dtype=dtype)
    if device.type != "cuda":
        return t.to(device)
    if _ARENA is None:
        _ARENA = _PinnedArena()
    return _ARENA.stage(t).to(device, non_blocking=True)


def const_tensor(values, device, dtype=torch.float32):
    """small constant (shape tables, normalisers ...) built once per (value, device, dtype) -- never written
    to by callers."""
    device = torch.device(device)
    arr = np.asarray(values)
    key = (arr.shape, arr.tobytes(), str(arr.dtype), str(device), dtype)
    t = _CONST.get(key)
    if t is None:
        t = to_device_async(arr, device, dtype)
        _CONST[key] = t
    return t
