"""Build oracle/_ref: the REFERENCE's own native sources compiled for the host CPU.

TEST INFRASTRUCTURE.  Runs only where /root/reference exists (the build container); the GPU box
uses the prebuilt .so files that travel with the repo snapshot (oracle/_ref is git-ignored, not
gpurun-ignored).  Reference sources are read where they lie and are never copied into the repo:

  * third_lib/chamfer_dist/chamferdist/chamferdist/{ext.cpp,knn_cpu.cpp} compile unmodified
    -> oracle/_ref/ref_chamferdist_C*.so   (pybind module: knn_points_idx / knn_points_backward)
  * third_lib/dvr/dvr.cu, third_lib/dvxlr/dvxlr.cu, third_lib/dvxlr/dvxlr_v2.cu are CUDA; they are
    compiled as host C++ by (a) force-including oracle/ref_shim/ref_prelude.h, which defines
    __global__/blockIdx/... and re-states AT_DISPATCH_FLOATING_TYPES for torch 2.10, and (b) piping
    the source through one regex that turns `kernel<scalar_t><<<blocks, threads>>>(` into
    `vidar_ref_launch(kernel<scalar_t>, blocks, threads, ` on its way into g++'s stdin.  The kernel
    bodies (all arithmetic) are compiled exactly as written, one host call per CUDA thread.
    -> oracle/_ref/ref_dvr*.so, ref_dvxlr*.so, ref_dvxlr_v2*.so
  * projects/mmdet3d_plugin/bevformer/backbones/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh -- the
    reference's in-tree copy of the deformable-attention bilinear sampling and its gradients
    (`dcnv3_im2col_bilinear`, `dcnv3_col2im_bilinear_gm` and the two kernels around them): the same
    shim; the filter drops the two CUDA-only includes, the __shared__-memory backward variants and
    the `<<<>>>` host launchers, and oracle/ref_dcnv3_bind.cpp is appended to the same translation unit
    -> oracle/_ref/ref_dcnv3.so   (pins oracle/msda.py, tests/test_oracle_msda.py)

gcc on x86-64 without -mfma cannot fuse multiply-adds, so these builds evaluate the reference's
expressions in strict IEEE order; -ffp-contract=off is passed anyway.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import sysconfig
from pathlib import Path

HERE = Path(__file__).resolve().parent
OUT = HERE / "_ref"
REF = Path(os.environ.get("VIDAR_REFERENCE", "/root/reference"))
LAUNCH = re.compile(r"(\w+<scalar_t>)\s*<<<\s*(\w+)\s*,\s*(\w+)\s*>>>\s*\(")

CU_MODULES = {
    "ref_dvr": ("third_lib/dvr/dvr.cu", "REF_DVR"),
    "ref_dvxlr": ("third_lib/dvxlr/dvxlr.cu", "REF_DVXLR"),
    "ref_dvxlr_v2": ("third_lib/dvxlr/dvxlr_v2.cu", "REF_DVXLR_V2"),
}


DCNV3_CUH = "projects/mmdet3d_plugin/bevformer/backbones/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh"


def _dcnv3_host_text(text: str) -> str:
    """Keep the device functions, `dcnv3_im2col_gpu_kernel` and `dcnv3_col2im_gpu_kernel_gm`."""
    text = re.sub(r"#include <(ATen/cuda/CUDAContext\.h|THC/THCAtomics\.cuh)>\n", "", text)
    shm = text.index("template <typename scalar_t, unsigned int blockSize>")
    gm = text.index("template <typename scalar_t>\n__global__ void dcnv3_col2im_gpu_kernel_gm(")
    host = text.index("template <typename scalar_t>\nvoid dcnv3_im2col_cuda(")
    return text[:shm] + text[gm:host]


def _torch_flags():
    import torch
    from torch.utils.cpp_extension import include_paths, library_paths

    inc = [f"-I{p}" for p in include_paths()] + [f"-I{sysconfig.get_paths()['include']}"]
    libdirs = library_paths()
    ld = [f"-L{p}" for p in libdirs] + [f"-Wl,-rpath,{p}" for p in libdirs]
    ld += ["-lc10", "-ltorch_cpu", "-ltorch", "-ltorch_python"]
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cxx = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-w",
           f"-D_GLIBCXX_USE_CXX11_ABI={abi}"]
    return inc, ld, cxx


def available() -> bool:
    return REF.is_dir()


def so_path(name: str) -> Path:
    return OUT / f"{name}.so"


def _needs(out: Path, deps: list[Path]) -> bool:
    return not out.exists() or any(d.stat().st_mtime > out.stat().st_mtime for d in deps)


def build(verbose: bool = True) -> list[Path]:
    if not available():
        raise RuntimeError(f"reference tree not found at {REF}")
    OUT.mkdir(exist_ok=True)
    inc, ld, cxx = _torch_flags()
    built = []
    me = [Path(__file__), HERE / "ref_bind.cpp", *sorted((HERE / "ref_shim").glob("*.h"))]

    # --- CUDA sources through the host shim -----------------------------------------------------
    for name, (rel, macro) in CU_MODULES.items():
        src = REF / rel
        out = so_path(name)
        if _needs(out, [src, *me]):
            text = LAUNCH.sub(r"vidar_ref_launch(\1, \2, \3, ", src.read_text())
            obj_k = OUT / f"{name}_kernels.o"
            obj_b = OUT / f"{name}_bind.o"
            cmd = ["g++", *cxx, *inc, f"-I{HERE / 'ref_shim'}", "-include",
                   str(HERE / "ref_shim" / "ref_prelude.h"), "-x", "c++", "-c", "-", "-o", str(obj_k)]
            if verbose:
                print(f"[build_ref] {rel} -> {obj_k.name}", flush=True)
            subprocess.run(cmd, input=text.encode(), check=True)
            subprocess.run(["g++", *cxx, *inc, f"-D{macro}", f"-DTORCH_EXTENSION_NAME={name}",
                            "-c", str(HERE / "ref_bind.cpp"), "-o", str(obj_b)], check=True)
            subprocess.run(["g++", "-shared", str(obj_k), str(obj_b), *ld, "-o", str(out)], check=True)
            obj_k.unlink()
            obj_b.unlink()
        built.append(out)

    # --- ops_dcnv3 bilinear sampling kernels (the in-tree MSDA arithmetic) -------------------------
    src = REF / DCNV3_CUH
    out = so_path("ref_dcnv3")
    bind = HERE / "ref_dcnv3_bind.cpp"
    if _needs(out, [src, bind, *me]):
        if verbose:
            print(f"[build_ref] {DCNV3_CUH} -> {out.name}", flush=True)
        text = _dcnv3_host_text(src.read_text()) + "\n" + bind.read_text()
        subprocess.run(["g++", *cxx, *inc, f"-I{HERE / 'ref_shim'}", "-include",
                        str(HERE / "ref_shim" / "ref_prelude.h"), "-DTORCH_EXTENSION_NAME=ref_dcnv3",
                        "-x", "c++", "-shared", "-", *ld, "-o", str(out)], input=text.encode(), check=True)
    built.append(out)

    # --- chamferdist CPU KNN, unmodified ----------------------------------------------------------
    cd = REF / "third_lib/chamfer_dist/chamferdist/chamferdist"
    out = so_path("ref_chamferdist_C")
    srcs = [cd / "ext.cpp", cd / "knn_cpu.cpp"]
    if _needs(out, [*srcs, Path(__file__)]):
        if verbose:
            print("[build_ref] chamferdist ext.cpp + knn_cpu.cpp", flush=True)
        subprocess.run(["g++", *cxx, *inc, f"-I{cd}", "-DTORCH_EXTENSION_NAME=ref_chamferdist_C",
                        "-shared", *map(str, srcs), *ld, "-o", str(out)], check=True)
    built.append(out)
    return built


def load(name: str):
    """Import a prebuilt oracle/_ref module (pybind11 torch extension) by file path."""
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)

    path = so_path(name)
    if not path.exists():
        raise FileNotFoundError(f"{path} missing: run `python oracle/build_ref.py` where "
                                f"/root/reference is mounted")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    for p in build():
        print(p)
