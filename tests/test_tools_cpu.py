"""CPU: the small host-side helpers under tools/ that produce committed evidence -- kernel categories of
tools/prof_summary.py and the merge step of tools/pretune_gemms.py."""
import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _load(name):
    spec = importlib.util.spec_from_file_location(name, ROOT / "tools" / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    sys.path.insert(0, str(ROOT))
    spec.loader.exec_module(mod)
    return mod


def test_prof_summary_sorts_kernels_into_the_five_categories():
    ps = _load("prof_summary")
    cat = ps.category
    assert cat("Cijk_Ailk_Bljk_SB_MT64x64x32_MI16x16x4x1_SN_1LDSB1") == "library GEMM (hipBLASLt / rocBLAS)"
    assert cat("miopenSp3AsmConv_v30_3_1_gfx9_fp32_f2x3_stride1") == "MIOpen convolution"
    assert cat("igemm_wrw_gtcx35_nhwc_fp32_bx0_ex0_bt128x128x16") == "MIOpen convolution"
    assert cat("void at::native::vectorized_elementwise_kernel<4, at::native::CUDAFunctor_add<float>>") == "torch eager"
    assert cat("void (anonymous namespace)::msda_bwd_tile_kernel(long const*, long const*)") == "vidar_amd HIP kernels"
    assert cat("(anonymous namespace)::dcn_im2col_pair_kernel(float const*)") == "vidar_amd HIP kernels"
    assert cat("(anonymous namespace)::colsum_kernel(float const*, float*, long, int)") == "vidar_amd HIP kernels"
    assert cat("__amd_rocclr_fillBufferAligned") == "other (fills, copies, collectives)"


def test_pretune_merge_adds_only_new_shapes_and_refuses_other_library_versions(tmp_path, monkeypatch):
    pt = _load("pretune_gemms")
    from vidar_amd import gemm_tuning
    shipped = tmp_path / "shipped.csv"
    head = "Validator,PT_VERSION,2.10.0\nValidator,GCN_ARCH_NAME,gfx950\n"
    shipped.write_text(head + "GemmTunableOp_float_NN,nn_1_2_3,Gemm_A,0.1\n")
    monkeypatch.setattr(gemm_tuning, "SHIPPED", shipped)
    extra = tmp_path / "extra.csv"
    extra.write_text(head + "GemmTunableOp_float_NN,nn_1_2_3,Gemm_B,0.05\nGemmTunableOp_float_NT,nt_4_5_6,Gemm_C,0.2\n")
    pt.merge(str(extra))
    lines = shipped.read_text().splitlines()
    assert lines.count("GemmTunableOp_float_NN,nn_1_2_3,Gemm_A,0.1") == 1          # the shipped solution wins
    assert "GemmTunableOp_float_NT,nt_4_5_6,Gemm_C,0.2" in lines and len(lines) == 4
    other = tmp_path / "other.csv"
    other.write_text("Validator,PT_VERSION,2.11.0\nValidator,GCN_ARCH_NAME,gfx950\nGemmTunableOp_float_NT,nt_7_8_9,Gemm_D,0.2\n")
    import pytest
    with pytest.raises(SystemExit):
        pt.merge(str(other))


def test_devcode_kernel_parser_and_resource_name_shortening():
    """tools/devcode.py splits an llvm-objdump listing into per-kernel instruction lists without the address / encoding
    comments (so that two builds compare equal when only addresses moved); tools/kernel_resources.py shortens names"""
    sys.path.insert(0, str(ROOT / "tools"))          # devcode imports kernel_resources from its own directory
    kernel_resources = _load("kernel_resources")
    devcode = _load("devcode")
    text = ["", "0000000000001000 <_ZN1a3fooEv>:",
            "\ts_load_dwordx2 s[0:1], s[4:5], 0x0                         // 000000001000: C0060002 00000000",
            "\ts_endpgm                                                   // 000000001008: BF810000",
            "0000000000001100 <_ZN1a3barEv>:",
            "\tv_mov_b32_e32 v0, 0                                        // 000000001100: 7E000280"]
    k = devcode.kernels(text)
    assert list(k) == ["_ZN1a3fooEv", "_ZN1a3barEv"]
    assert k["_ZN1a3fooEv"] == ["s_load_dwordx2 s[0:1], s[4:5], 0x0", "s_endpgm"]
    moved = [ln.replace("000000001", "000000009") for ln in text]
    assert devcode.kernels(moved)["_ZN1a3fooEv"] == k["_ZN1a3fooEv"]
    assert kernel_resources.short("void (anonymous namespace)::dvr_render_kernel<256>(float const*, int)") == "dvr_render_kernel<256>"
    assert kernel_resources.short("(anonymous namespace)::msda_fwd_kernel(float const*, long const*)") == "msda_fwd_kernel"
