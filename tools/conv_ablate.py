"""Timing ablations of csrc/conv3x3_mfma.hip: what keeps an fp32-MFMA kernel fed from LDS at half the matrix peak?

    bash:  for k in 0 1 2 4 8 16 6 14; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
               -Iinclude -Ividar_amd/csrc -DVIDAR_CONV_ABL=$k vidar_amd/csrc/conv3x3_mfma.hip -o tools/_abl/conv_abl_$k.so; done
    python tools/conv_ablate.py

Each variant library drops one part of the kernel (results are then WRONG -- only the time is of interest):
  1 no border masks   2 operands not read from LDS   4 no global fetch / LDS staging   8 no per-chunk barrier
  16 no MFMA (one FMA per operand pair instead)
(round 5 also timed three restructurings this way -- float4 weights, operands read a pair / a chunk ahead behind a
sched_barrier: profiles/r05_conv3x3_ablation.log -- none moved the kernel by more than 6 %; they were not kept)"""
import ctypes
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from tools.kbench import timeit  # noqa: E402

VARIANTS = (0, 1, 2, 4, 8, 16, 6, 14)
NAMES = {0: "full kernel", 1: "no border masks", 2: "no LDS operand reads", 4: "no fetch / staging", 8: "no barrier",
         16: "no MFMA", 6: "no LDS reads, no fetch / staging",
         14: "MFMA only (no LDS reads, fetch, staging, barrier)"}


def main():
    torch.zeros(1, device="cuda")
    for (N, C, H, W) in [(24, 256, 58, 100), (6, 256, 58, 100)]:
        x = torch.randn(N, C, H, W, device="cuda")
        w = torch.randn(27, C, 3, 3, device="cuda") * 0.02
        b = torch.randn(27, device="cuda")
        out = torch.empty(N, 27, H, W, device="cuda")
        ws = torch.empty(C * 9 * 32 * 4, dtype=torch.uint8, device="cuda")
        flops = 2 * 32 * C * 9 * N * H * W
        for k in VARIANTS:
            so = ROOT / "tools" / "_abl" / f"conv_abl_{k}.so"
            if not so.exists():
                continue
            L = ctypes.CDLL(str(so))
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            p = lambda t: ctypes.c_void_p(t.data_ptr())
            for chunk in (8,):
                fn = lambda: L.vidar_conv3x3_few_f32(p(x), p(w), p(b), p(out), N, C, H, W, 27, p(ws), ctypes.c_size_t(ws.numel()), st)
                assert fn() == 0
                ref = torch.nn.functional.conv2d(x, w, b, padding=1)
                err = float((out - ref).abs().max() / ref.abs().max())
                ms = timeit(fn, warm=3, it=20)
                print(json.dumps({"case": [N, C, H, W], "variant": k, "what": NAMES[k], "chunk": chunk, "ms": round(ms, 4), "rel_err": float(f"{err:.2e}"),
                                  "frac_fp32_mfma_padded": round(flops / ms / 1e9 / 157.3, 3)}), flush=True)


if __name__ == "__main__":
    main()
