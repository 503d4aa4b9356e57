"""Offline GEMM tuning for the data-dependent shapes of the step (run on an MI355X; writes a TunableOp csv).

SpatialCrossAttention's GEMM height is (cameras x padded visible-query count); `BEVFormerEncoder.plan_frames` pads
the count to a multiple of 256, so real data meets a handful of lengths around Q/4.  bench.py / train.fit FREEZE
tuning after warm-up (a rank tuning mid-training would stall the others at the all-reduce); a length that was never
tuned then runs on the library's default heuristic.  This script tunes the six GEMMs of `MSDeformableAttention3D`'s
`sampling_offsets` / `attention_weights` Linears (forward, grad_input, grad_weight) for every padded length in a
range and merges the winners into the shipped file:

    python tools/pretune_gemms.py --lo 4096 --hi 16384 --cams 6 --out gpurun_out/tunableop_sca_lengths.csv
    python tools/pretune_gemms.py --merge gpurun_out/tunableop_sca_lengths.csv      # -> vidar_amd/tunableop_gfx950.csv
"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def merge(extra):
    from vidar_amd.gemm_tuning import SHIPPED
    have = SHIPPED.read_text().splitlines()
    keys = {tuple(l.split(",")[:2]) for l in have if l and not l.startswith("Validator,")}
    validators = [l for l in have if l.startswith("Validator,")]
    new_lines = Path(extra).read_text().splitlines()
    if [l for l in new_lines if l.startswith("Validator,")] != validators:
        raise SystemExit("library versions of the two files differ: re-tune instead of merging")
    add = [l for l in new_lines if l and not l.startswith("Validator,") and tuple(l.split(",")[:2]) not in keys]
    SHIPPED.write_text("\n".join(have + add) + "\n")
    print(f"merged {len(add)} new solutions into {SHIPPED} ({len(keys) + len(add)} total)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lo", type=int, default=4096)
    ap.add_argument("--hi", type=int, default=16384)
    ap.add_argument("--cams", type=int, nargs="*", default=[6])
    ap.add_argument("--out", default="/tmp/tunableop_sca_lengths.csv")
    ap.add_argument("--merge")
    a = ap.parse_args()
    if a.merge:
        return merge(a.merge)
    import torch
    import torch.cuda.tunable as tn
    from vidar_amd import gemm_tuning
    gemm_tuning.enable(tune_missing=True, results_file=a.out)
    dev = torch.device("cuda")
    lins = [torch.nn.Linear(256, n).to(dev) for n in (512, 256)]       # sampling_offsets (8 heads x 4 levels x 8 points x 2), attention_weights
    for cams in a.cams:
        for length in range(a.lo, a.hi + 1, 256):
            x = torch.randn(cams, length, 256, device=dev, requires_grad=True)
            for lin in lins:
                y = lin(x)
                y.backward(torch.ones_like(y))
            torch.cuda.synchronize()
    tn.write_file(a.out)
    print(f"{gemm_tuning.count_results()} solutions in {a.out}")


if __name__ == "__main__":
    main()
