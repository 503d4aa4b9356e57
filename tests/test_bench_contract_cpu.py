"""CPU: bench.py's contract pieces that do not need a GPU -- it refuses to run without one (no CPU fallback
for the measured path), the driver's flags parse, `roofline.traffic` comes from a committed PMC summary that
exists, and the bounded CPU-baseline leg (the only part allowed to call the oracle) returns the documented
record."""
import json
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]


def test_refuses_to_run_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout)


def test_driver_flags_parse_and_defaults_are_single_gpu():
    sys.path.insert(0, str(ROOT))
    import bench
    old = sys.argv
    try:
        sys.argv = ["bench.py"]
        a = bench.parse()
        assert a.gpus == 1 and a.steps > 0 and a.warmup >= 0 and a.config == "vidar_1_8_nusc_1future"
        assert not a.cpu_baseline_full                      # the full-size CPU step is opt-in
        sys.argv = ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"]
        a = bench.parse()
        assert (a.gpus, a.steps, a.warmup) == (8, 20, 5)
    finally:
        sys.argv = old


def test_pmc_traffic_is_backed_by_committed_profiles():
    sys.path.insert(0, str(ROOT))
    import bench
    nbytes, src = bench.pmc_traffic("msda_bwd[L=4,P=8]")
    assert nbytes and nbytes > 8.7e8                        # at least the algorithmic bytes
    files = [p for p in (ROOT / "profiles").glob("r02_pmc_*_SIZE_kbench_msda.csv")]
    assert len(files) == 2 and "r02_pmc_" in src
    assert bench.pmc_traffic("no such kernel") == (None, None)


def test_cpu_baseline_leg_reports_a_bounded_sample():
    """the subprocess the bench launches on rank 0 (small BEV here to keep the suite fast)"""
    sys.path.insert(0, str(ROOT))
    import bench
    rec = bench.cpu_baseline("vidar_1_8_nusc_1future", threads=4, with_backbone=False, reduced=True)
    assert rec["kind"] == "port" and rec["unit"] == "samples/s" and rec["cores"] == 4
    assert rec["value"] > 0 and "bounded sample" in rec["sample"] and "x that" in rec["sample"]
    json.dumps(rec)


def test_gemm_tuning_is_a_no_op_without_a_gpu_and_ships_validated_solutions():
    from vidar_amd import gemm_tuning
    if not torch.cuda.is_available():
        assert gemm_tuning.enable() == dict(enabled=False, reason="no GPU")
    lines = gemm_tuning.SHIPPED.read_text().splitlines()
    validators = [l for l in lines if l.startswith("Validator,")]
    entries = [l.split(",") for l in lines if l and not l.startswith("Validator,")]
    assert {v.split(",")[1] for v in validators} >= {"PT_VERSION", "GCN_ARCH_NAME", "ROCBLAS_VERSION", "HIPBLASLT_VERSION"}
    assert any("gfx950" in v for v in validators)
    assert len(entries) > 100 and all(len(e) == 4 and e[0].startswith("Gemm") and float(e[3]) > 0 for e in entries)
    assert len({(e[0], e[1]) for e in entries}) == len(entries)          # one solution per (op, shape)
