"""ctypes loader for libvidar_hip.so.  There is NO fallback: if the library is missing or a call
fails the product raises, it never routes to a CPU/eager path."""
from __future__ import annotations

import ctypes
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


def stream_of(t):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
