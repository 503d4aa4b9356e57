"""Host -> device traffic of the hot path without stream synchronisation.

`tensor.new_tensor(numpy_array)` / `torch.tensor(list, device='cuda')` copy from pageable memory: the host
blocks until the stream has drained (torch.cuda.set_sync_debug_mode flags each one).  The step makes ~170
such copies of tiny per-sample matrices and shape constants; here they go through pinned staging buffers
(`to_device_async`) or are built once and cached (`const_tensor`)."""
from __future__ import annotations

import numpy as np
import torch

_CONST: dict = {}


def to_device_async(array, device, dtype=torch.float32):
    """numpy / nested list -> device tensor via pinned memory, non-blocking."""
    device = torch.device(device)
    t = torch.as_tensor(np.ascontiguousarray(np.asarray(array)), dtype=dtype)
    if device.type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


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
