"""Per-kernel resource table of the built library, read from the code objects' own metadata (no GPU needed):

    python tools/kernel_resources.py [vidar_amd/libvidar_hip.so] > profiles/rNN_kernel_resources.txt

For every gfx950 kernel: VGPRs (arch + acc), SGPRs, static LDS bytes, scratch bytes, spill counts, the
workgroup-size bound, and the occupancy those allow (waves per SIMD: 512 unified VGPRs per lane, 8 wave slots;
workgroups per CU from the 160 KB LDS).  tests/test_kernel_resources_cpu.py asserts the invariants the kernels
are written to (wave64, no scratch, no VGPR spills) on the same data.  s-spill = scalar registers parked in VGPR
lanes (v_writelane / v_readlane, no memory traffic): non-zero only in the DCN kernels, whose per-tap addressing
holds more uniform values than the 106 SGPRs."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

import yaml

ROOT = Path(__file__).resolve().parents[1]
LLVM = Path("/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
LDS_PER_CU = 160 * 1024
VGPRS_PER_SIMD_LANE = 512          # unified arch + acc file, allocation granule 8
WAVE_SLOTS_PER_SIMD = 8


def demangle(names):
    import shutil
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not filt:
        return list(names)
    out = subprocess.run([filt], input="\n".join(names), text=True, capture_output=True)
    return out.stdout.splitlines() if out.returncode == 0 else list(names)


def short(name):
    """'(anonymous namespace)::msda_fwd_kernel(float const*, ...)' -> 'msda_fwd_kernel'; template args are kept."""
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    depth = 0
    for i, ch in enumerate(name):          # cut the parameter list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def code_objects(so_path, work):
    """The gfx950 code objects of a HIP shared library: one clang offload bundle per translation unit in .hip_fatbin."""
    fat = work / "fat.bin"
    # llvm-objcopy rewrites its INPUT in place when no output is named: always name one (never touch the library)
    subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(so_path),
                    str(work / "objcopy_out.discard")], check=True)
    blob = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    objs = []
    for i, p in enumerate(starts):
        end = starts[i + 1] if i + 1 < len(starts) else len(blob)
        bundle, co = work / f"bundle{i}.bin", work / f"co{i}.o"
        bundle.write_bytes(blob[p:end])
        subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={bundle}",
                        f"--targets={TARGET}", f"--output={co}"], check=True)
        objs.append(co)
    return objs


def kernels_of(co):
    notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], text=True, capture_output=True, check=True).stdout
    doc = notes[notes.index("---"):notes.rindex("...")]
    meta = yaml.safe_load(doc)
    assert meta["amdhsa.target"].endswith("gfx950"), meta["amdhsa.target"]
    return meta.get("amdhsa.kernels", [])


def occupancy(k):
    regs = k[".vgpr_count"] + k.get(".agpr_count", 0)
    alloc = max(8, -(-regs // 8) * 8)
    waves_simd = min(WAVE_SLOTS_PER_SIMD, VGPRS_PER_SIMD_LANE // alloc)
    lds = k[".group_segment_fixed_size"]
    wgs_lds = LDS_PER_CU // lds if lds else None
    return waves_simd, wgs_lds


def table(so_path):
    rows = []
    with tempfile.TemporaryDirectory() as d:
        for co in code_objects(Path(so_path), Path(d)):
            for k in kernels_of(co):
                rows.append(k)
    names = demangle([k[".name"] for k in rows])
    out = []
    for k, n in zip(rows, names):
        waves, wgs = occupancy(k)
        out.append(dict(kernel=short(n), vgpr=k[".vgpr_count"], agpr=k.get(".agpr_count", 0), sgpr=k[".sgpr_count"],
                        lds=k[".group_segment_fixed_size"], scratch=k[".private_segment_fixed_size"],
                        vgpr_spill=k.get(".vgpr_spill_count", 0), sgpr_spill=k.get(".sgpr_spill_count", 0),
                        dyn_stack=bool(k.get(".uses_dynamic_stack", False)), max_wg=k[".max_flat_workgroup_size"],
                        wave=k[".wavefront_size"], waves_per_simd=waves, wgs_per_cu_lds=wgs))
    return sorted(out, key=lambda r: r["kernel"])


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else ROOT / "vidar_amd" / "libvidar_hip.so"
    rows = table(so)
    print(f"# {Path(so).name}: {len(rows)} gfx950 kernels (llvm-readelf --notes of the bundled code objects)")
    print(f"{'kernel':58s} {'vgpr':>4s} {'agpr':>4s} {'sgpr':>4s} {'lds B':>6s} {'scratch':>7s} {'v-spill':>7s} {'s-spill':>7s} "
          f"{'max wg':>6s} {'waves/SIMD':>10s} {'WG/CU (LDS)':>11s}")
    for r in rows:
        print(f"{r['kernel'][:58]:58s} {r['vgpr']:4d} {r['agpr']:4d} {r['sgpr']:4d} {r['lds']:6d} {r['scratch']:7d} "
              f"{r['vgpr_spill']:7d} {r['sgpr_spill']:7d} {r['max_wg']:6d} {r['waves_per_simd']:10d} "
              f"{'-' if r['wgs_per_cu_lds'] is None else r['wgs_per_cu_lds']:>11}")


if __name__ == "__main__":
    main()
