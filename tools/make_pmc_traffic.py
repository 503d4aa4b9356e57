"""profiles/pmc_traffic.json (what bench.py reports as `roofline.traffic`) from the committed PMC summaries.

    python tools/make_pmc_traffic.py profiles/r03_pmc_msda_sca profiles/r03_pmc_FETCH_SIZE_calibration.csv \
        profiles/r03_pmc_WRITE_SIZE_calibration.csv > profiles/pmc_traffic.json

FETCH_SIZE / WRITE_SIZE (KB per dispatch, `tools/pmc_pass.sh` over `tools/kbench.py msda_sca`: the
SpatialCrossAttention shape B=6, Nq=10^4, L=4, P=8 on random reference points) are CALIBRATED on this library's own
access pattern first: `tools/micro/fetch_calib.hip` gathers every 128-byte line of a 1 GiB buffer exactly once,
8 lanes x float4 per line in random order -- a known byte count.  On gfx950 the counter reports half of it (like the
wide streaming read of MI355X_MICROARCH.md), so fetched bytes = FETCH_SIZE x 1024 x (known / reported); WRITE_SIZE is
exact for 128-byte line writes."""
import csv
import json
import sys
from pathlib import Path


def table(path):
    return {(r["kernel"], r["counter"]): float(r["mean_value"]) for r in csv.DictReader(open(path))}


def main(pmc_dir, fetch_cal, write_cal, pattern="random reference points", target="msda_sca", nq="10000"):
    fc, wc = table(fetch_cal), table(write_cal)
    nlines = (1 << 30) // 128
    fetch_factor = ((1 << 30) + 4 * nlines) / (fc[("calib_gather128", "FETCH_SIZE")] * 1024)
    write_factor = (1 << 30) / (wc[("calib_write128", "WRITE_SIZE")] * 1024)
    d = Path(pmc_dir)
    f, w = table(d / "pmc_FETCH_SIZE.csv"), table(d / "pmc_WRITE_SIZE.csv")
    kb = lambda t, k, c: t.get((k, c), 0.0) * 1024
    B, Nv, Nq, H, C, L, P = 6, 30825, int(nq), 8, 32, 4, 8
    fwd_alg = 4 * (B * Nv * H * C + B * Nq * H * L * P * 3 + B * Nq * H * C)
    bwd_alg = fwd_alg + 4 * (B * Nq * H * C + B * Nv * H * C + B * Nq * H * L * P * 3)
    src = (f"{d}/pmc_{{FETCH,WRITE}}_SIZE.csv (tools/pmc_pass.sh: rocprofv3 --pmc in separate passes over `tools/kbench.py "
           f"{target}`, SCA shape B=6 Nq={Nq} L=4 P=8, {pattern}); FETCH_SIZE x {fetch_factor:.3f} and WRITE_SIZE x "
           f"{write_factor:.3f} from the calibration on a known byte count ({fetch_cal}, {write_cal})")
    out = {"queries_per_camera": Nq, "calibration": {"fetch_factor": round(fetch_factor, 4), "write_factor": round(write_factor, 4),
                           "pattern": "random 128-byte lines, 8 lanes x float4, 1 GiB buffer, every line once",
                           "source": f"{fetch_cal}, {write_cal}, tools/micro/fetch_calib.hip"}}
    bwd_kernels = ["msda_bin_kernel<false>", "msda_bin_scan_kernel", "msda_bin_kernel<true>", "msda_bwd_tile_kernel",
                   "msda_bwd_locw_kernel"]
    per = {k: dict(fetch_bytes=kb(f, k, "FETCH_SIZE") * fetch_factor, write_bytes=kb(w, k, "WRITE_SIZE") * write_factor)
           for k in bwd_kernels}
    memset = 4.0 * B * Nv * H * C                   # grad_value is zeroed by the call (hipMemsetAsync, not in the kernel list)
    fb = sum(v["fetch_bytes"] for v in per.values()); wb = sum(v["write_bytes"] for v in per.values()) + memset
    out["msda_bwd[L=4,P=8]"] = dict(bytes=fb + wb, fetch_bytes=fb, write_bytes=wb, algorithmic_bytes=bwd_alg,
                                    ratio=round((fb + wb) / bwd_alg, 2), per_kernel=per, source=src)
    k = "msda_fwd_kernel"
    fb, wb = kb(f, k, "FETCH_SIZE") * fetch_factor, kb(w, k, "WRITE_SIZE") * write_factor
    out["msda_fwd[L=4,P=8]"] = dict(bytes=fb + wb, fetch_bytes=fb, write_bytes=wb, algorithmic_bytes=fwd_alg,
                                    ratio=round((fb + wb) / fwd_alg, 2), source=src)
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:7])
