"""Device-code tools that need no GPU.

    python tools/devcode.py diff A.hip B.hip [-- extra hipcc flags]    # is the gfx950 code of two sources identical?
    python tools/devcode.py head-diff vidar_amd/csrc/dcn.hip [-- flags] # working tree vs the committed file (HEAD)
    python tools/devcode.py asm vidar_amd/csrc/dcn.hip KERNEL [-- flags] # disassembly of the kernels whose name contains KERNEL

Used to prove that a refactor, or a compile-time variant that is off by default, leaves the shipped device code
untouched (the round's rule: nothing that changes device code goes in without a GPU run), and to read what the
compiler made of a loop (loads in flight, s_waitcnt placement).  Per-file flags are those of vidar_amd/build.py."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
from kernel_resources import LLVM, code_objects  # noqa: E402
from vidar_amd.build import COMMON, HIPCC, PER_FILE  # noqa: E402


def disasm(src, extra, work, name=None):
    src = Path(src)
    so = work / f"{src.stem}_{abs(hash((str(src), tuple(extra))))}.so"
    flags = COMMON + PER_FILE.get(name or src.name, []) + list(extra)
    subprocess.run([HIPCC, *flags, "-shared", str(src), "-o", str(so)], check=True)
    sub = work / (so.stem + "_co")
    sub.mkdir()
    text = []
    for co in code_objects(so, sub):
        out = subprocess.run([str(LLVM / "llvm-objdump"), "-d", str(co)], text=True, capture_output=True, check=True).stdout
        text += [ln for ln in out.splitlines() if "file format" not in ln]
    return text


def kernels(text):
    cur, body = None, {}
    for ln in text:
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            cur = m.group(1)
            body[cur] = []
        elif cur and ln.strip():
            body[cur].append(re.sub(r"\s*//.*", "", ln).strip())
    return body


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        i = argv.index("--")
        argv, extra = argv[:i], argv[i + 1:]
    cmd = argv[0]
    with tempfile.TemporaryDirectory() as d:
        work = Path(d)
        if cmd in ("diff", "head-diff"):
            if cmd == "diff":
                a, b, name = Path(argv[1]), Path(argv[2]), Path(argv[2]).name
            else:
                b = Path(argv[1]).resolve()
                name = b.name
                a = work / ("HEAD_" + name)
                a.write_bytes(subprocess.run(["git", "-C", str(ROOT), "show", f"HEAD:{b.relative_to(ROOT)}"],
                                             capture_output=True, check=True).stdout)
            ka, kb = kernels(disasm(a, extra, work, name)), kernels(disasm(b, extra, work, name))
            changed = [k for k in sorted(set(ka) | set(kb)) if ka.get(k) != kb.get(k)]
            if not changed:
                print(f"identical device code: {len(kb)} kernels")
                return 0
            for k in changed:
                print(f"differs: {k[:100]}  ({len(ka.get(k, []))} -> {len(kb.get(k, []))} instructions)")
            return 1
        if cmd == "asm":
            for k, body in kernels(disasm(argv[1], extra, work)).items():
                if argv[2] in k:
                    print(f"== {k}  ({len(body)} instructions)")
                    for n, ln in enumerate(body):
                        print(f"{n:5d}  {ln}")
            return 0
    print(__doc__)
    return 2


if __name__ == "__main__":
    sys.exit(main())
