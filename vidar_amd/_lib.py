"""ctypes loader for libvidar_hip.so.  There is NO fallback: if the library is missing or a call
fails the product raises, it never routes to a CPU/eager path."""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libvidar_hip.so"
_lib = None
BAD_ARG = -22


class VidarHipError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        # torch bundles its own libamdhip64.so.7; it must be the copy already loaded when our
        # library is dlopen'ed, otherwise two HIP runtimes fight over the device (hipErrorNoDevice).
        import torch  # noqa: F401
        if not LIB_PATH.exists():
            raise VidarHipError(
                f"{LIB_PATH} not found: build it with `python -m vidar_amd.build` "
                f"(or __graft_entry__.build()); vidar_amd has no CPU fallback")
        _lib = ctypes.CDLL(str(LIB_PATH))
        dcn = os.environ.get("VIDAR_DCN_VARIANT")                # A/B of the DCNv2 col2im gather (LDS window / global loads)
        if dcn is not None:
            _lib.vidar_dcn_set_variant(int(dcn))
        order = os.environ.get("VIDAR_MSDA_ITEM_ORDER")          # A/B of the MSDA gather kernels' item order (tools, bench)
        if order is not None:
            _lib.vidar_msda_set_item_order(int(order))
    return _lib


def check(rc: int, what: str):
    if rc == 0:
        return
    if rc == BAD_ARG:
        raise ValueError(f"{what}: invalid argument")
    raise VidarHipError(f"{what}: HIP error {rc}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def workspace(nbytes_fn, *dims, like):
    """Caller-owned device scratch for an op: `nbytes_fn(*dims)` bytes from torch's caching allocator on `like`'s
    device (so it lives on the op's device and stream and is counted by torch's memory statistics).
    -> (tensor or None, pointer, size_t): keep the tensor alive until the call has been enqueued."""
    import torch
    nbytes_fn.restype = ctypes.c_size_t
    n = int(nbytes_fn(*dims))
    if n == 0:
        return None, None, ctypes.c_size_t(0)
    ws = torch.empty(n, dtype=torch.uint8, device=like.device)
    return ws, ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(n)


def stream_of(t):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


# --------------------------------------------------------------------------------------------------
# optional per-op timing with HIP events on torch's current stream (the stream every op launches on).
# Disabled by default (zero overhead beyond one attribute test); bench.py switches it on to
# measure the dominant kernel live inside the timed region.
# --------------------------------------------------------------------------------------------------
class OpTimer:
    def __init__(self):
        self.enabled = False
        self.records = {}          # name -> list[(start_event, end_event, algorithmic_bytes)]

    def reset(self):
        self.records = {}

    class _Span:
        __slots__ = ("t", "name", "nbytes", "e0")

        def __init__(self, t, name, nbytes):
            self.t, self.name, self.nbytes = t, name, nbytes

        def __enter__(self):
            import torch
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

        def __exit__(self, *exc):
            import torch
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.t.records.setdefault(self.name, []).append((self.e0, e1, self.nbytes))

    class _Null:
        def __enter__(self): return None
        def __exit__(self, *exc): return False

    _null = _Null()

    def span(self, name, nbytes=0):
        return OpTimer._Span(self, name, nbytes) if self.enabled else OpTimer._null

    def summary(self):
        """name -> dict(calls, total_ms, avg_ms, bytes_per_call) ; synchronises."""
        import torch
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = [a.elapsed_time(b) for a, b, _ in recs]
            out[name] = dict(calls=len(ms), total_ms=sum(ms), avg_ms=sum(ms) / len(ms),
                             bytes_per_call=sum(r[2] for r in recs) / len(recs))
        return out


TIMER = OpTimer()
