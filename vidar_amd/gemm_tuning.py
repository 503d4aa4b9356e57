"""Library-GEMM selection for the step: PyTorch's TunableOp picks, per GEMM shape, the fastest fp32 solution among
the rocBLAS and hipBLASLt kernels (same arithmetic, different tilings / split-K).  The Linear / value-projection /
1x1-convolution GEMMs are ~48 % of the hot path and ~50 % of the full step, and the libraries' default heuristics
leave a lot on the table for these tall-skinny fp32 shapes: hot path 157.7 -> 130.4 ms/step with tuned solutions
(MI355X, profiles/r02_*).

`enable()` switches TunableOp on, preloads the solutions tuned on MI355X that ship with the package
(`tunableop_gfx950.csv`; rejected by torch if the ROCm / hipBLASLt / rocBLAS / PyTorch versions differ) and, by default,
tunes shapes it has not seen during the first steps (warm-up).  SpatialCrossAttention's GEMM height depends on the number of
visible queries; `BEVFormerEncoder.plan_frames` rounds it up to a multiple of 256 so that the set of shapes stays small.
Not used by the parity tests."""
from __future__ import annotations

import os
from pathlib import Path

SHIPPED = Path(__file__).with_name("tunableop_gfx950.csv")


def enable(tune_missing: bool = True, results_file=None, rank: int = 0, max_tuning_ms: int = 30):
    """-> dict describing what was enabled (for logs / the bench JSON)"""
    import torch
    if not torch.cuda.is_available():
        return dict(enabled=False, reason="no GPU")
    import torch.cuda.tunable as tn
    tn.enable(True)
    tn.tuning_enable(bool(tune_missing))
    tn.set_max_tuning_duration(int(max_tuning_ms))
    out = results_file or os.environ.get("VIDAR_TUNABLEOP_FILE") or f"/tmp/vidar_tunableop_rank{rank}.csv"
    tn.set_filename(str(out), insert_device_ordinal=False)
    loaded = False
    if SHIPPED.exists():
        try:
            loaded = bool(tn.read_file(str(SHIPPED)))
        except RuntimeError:
            loaded = False
    return dict(enabled=True, tune_missing=bool(tune_missing), shipped_solutions=str(SHIPPED.name) if loaded else None,
                results_file=str(out))


def freeze():
    """stop TUNING new shapes (tuned / shipped solutions stay in use).  Call after warm-up: under DDP a rank that
    meets a new GEMM shape mid-training (e.g. a new padded SpatialCrossAttention length) would otherwise spend
    seconds tuning while the other ranks wait at the gradient all-reduce.  No-op without a GPU."""
    import torch
    if torch.cuda.is_available():
        import torch.cuda.tunable as tn
        if tn.is_enabled():
            tn.tuning_enable(False)


def thaw():
    """allow tuning again (warm-up of another model / config); no-op unless TunableOp was enabled"""
    import torch
    if torch.cuda.is_available():
        import torch.cuda.tunable as tn
        if tn.is_enabled():
            tn.tuning_enable(True)


def count_results():
    import torch.cuda.tunable as tn
    try:
        return len(tn.get_results())
    except RuntimeError:
        return 0
