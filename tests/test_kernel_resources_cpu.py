"""The built library's code objects, read without a GPU (tools/kernel_resources.py): every kernel is a gfx950
wave64 kernel with no scratch memory and no VGPR spills, and the LDS-heavy ones fit the CU as DESIGN.md says."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
SO = ROOT / "vidar_amd" / "libvidar_hip.so"


@pytest.fixture(scope="module")
def rows():
    if not SO.exists():
        pytest.skip("libvidar_hip.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    import hashlib
    import kernel_resources
    before = hashlib.sha256(SO.read_bytes()).hexdigest()
    out = kernel_resources.table(SO)
    # reading the metadata must never touch the product library (llvm-objcopy rewrites its input when no output is named)
    assert hashlib.sha256(SO.read_bytes()).hexdigest() == before
    return out


def test_every_kernel_is_wave64_without_scratch_or_vgpr_spills(rows):
    assert len(rows) >= 50
    for r in rows:
        assert r["wave"] == 64, r
        assert r["scratch"] == 0 and r["vgpr_spill"] == 0 and not r["dyn_stack"], r
        assert r["agpr"] == 0, r                 # the unified file is all arch VGPRs (the MFMA GEMM keeps its accumulators there too)
        # the MFMA GEMM holds 64 accumulators + two staged operands (+ hi/lo fragments): 3 waves per SIMD by design; the
        # few-output 3x3 convolution fits 3 workgroups of 4 waves per CU in LDS (its slab), so 3 waves per SIMD is all it can use
        by_design_3 = r["kernel"].startswith(("gemm_mfma_kernel<", "conv3x3_few_kernel<"))
        assert r["waves_per_simd"] >= (3 if by_design_3 else 4), r


def test_hot_kernels_are_present_with_the_documented_footprints(rows):
    by = {r["kernel"]: r for r in rows}
    for name in ("msda_fwd_kernel", "msda_bwd_tile_kernel", "msda_bwd_locw_kernel", "msda_bin_scan_kernel",
                 "lr_gather_fwd_kernel", "lr_prob_bwd_kernel", "ray_ce_bwd_kernel", "ray_gumbel_fwd_kernel",
                 "knn1_d3_scan_kernel", "dcn_im2col_pair_kernel", "dcn_col2im_gather_kernel", "colsum_kernel",
                 "dvxlr_march_kernel<256>", "dvr_render_kernel<256>", "sca_project_kernel", "drop_add_ln_fwd_kernel"):
        assert name in by, name
    tile = by["msda_bwd_tile_kernel"]            # 4 waves x 10 KB accumulation windows + the staged records
    assert 40 * 1024 <= tile["lds"] <= 48 * 1024 and tile["wgs_per_cu_lds"] == 3
    assert by["msda_fwd_kernel"]["waves_per_simd"] == 8        # the gather needs every wave slot to hide L2 latency
    assert by["msda_fwd_kernel"]["vgpr"] <= 64
