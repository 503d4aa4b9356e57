"""Summarise a rocprofv3 results .db (kernel trace): per kernel name -> calls, avg us, total ms.
    python tools/prof_summary.py RESULTS.db [--steps K] [--csv] [substring ...]
With --steps K only the dispatches between the first and the last `vidar_marker_kernel` launch (the timed
region of bench.py) are counted and the columns are per step."""
import sqlite3
import sys


OWN = ("msda_", "dcn_", "affine_act", "affine_grad", "lr_", "ray_", "knn", "sca_", "drop_add_ln", "dvr", "dvxlr", "vidar_",
       "rows_", "bev_", "latent_", "colsum")


def category(name):
    """library GEMMs (Tensile kernels of hipBLASLt / rocBLAS) / MIOpen convolutions / this library's HIP kernels /
    torch eager kernels / the rest (fills, copies, RCCL)"""
    short = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if "(anonymous namespace)::" in name and "at::" not in name and "rocprim" not in name:
        return "vidar_amd HIP kernels"          # every kernel of libvidar_hip.so lives in an anonymous namespace
    if short.startswith("Cijk_") or "Cijk_" in short[:40]:
        return "library GEMM (hipBLASLt / rocBLAS)"
    if short.startswith(("miopen", "igemm_", "batched_transpose", "naive_conv", "gridwise_", "SubTensorOp", "MIOpen")) \
            or "ck::" in short[:60] or "Conv" in short[:40]:
        return "MIOpen convolution"
    if short.startswith("at::") or "at::native" in short[:80] or short.startswith(("c10::", "torch::")):
        return "torch eager"
    if any(short.startswith(p) for p in OWN):
        return "vidar_amd HIP kernels"
    return "other (fills, copies, collectives)"


def main(argv):
    path = argv[0]
    steps, csv_out, subs = 0, False, []
    it = iter(argv[1:])
    for a in it:
        if a == "--steps":
            steps = int(next(it))
        elif a == "--csv":
            csv_out = True
        else:
            subs.append(a)
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    sym = [t for t in tabs if "info_kernel_symbol" in t][0]
    where = ""
    div = 1.0
    if steps:
        marks = [r[0] for r in c.execute(f"select d.start from {kd} d join {sym} s on d.kernel_id=s.id "
                                         f"where s.display_name like '%vidar_marker_kernel%' order by d.start")]
        assert len(marks) >= 2, "markers not found"
        where = f"where d.start > {marks[0]} and d.start < {marks[-1]} and s.display_name not like '%vidar_marker_kernel%'"
        div = float(steps)
    rows = list(c.execute(f"select s.display_name, count(*), avg(d.end-d.start), sum(d.end-d.start), min(d.end-d.start) "
                          f"from {kd} d join {sym} s on d.kernel_id=s.id {where} group by s.display_name order by 4 desc"))
    total = sum(r[3] for r in rows)
    ncalls = sum(r[1] for r in rows)
    cats = {}
    for name, n, avg, tot, mn in rows:
        k = category(name)
        c0 = cats.setdefault(k, [0, 0.0])
        c0[0] += n; c0[1] += tot
    if csv_out:
        print("Name,Calls,TotalMs,AverageUs,MinUs,Percentage")
    else:
        print(f"{'kernel':70s} {'calls':>8s} {'avg_us':>10s} {'min_us':>10s} {'total_ms':>10s} {'%':>6s}")
    for name, n, avg, tot, mn in rows:
        if subs and not any(s in name for s in subs):
            continue
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        if csv_out:
            print(f"\"{short}\",{n / div:.2f},{tot / 1e6 / div:.4f},{avg / 1e3:.2f},{mn / 1e3:.2f},{100.0 * tot / total:.3f}")
        else:
            print(f"{short:70s} {n / div:8.1f} {avg / 1e3:10.1f} {mn / 1e3:10.1f} {tot / 1e6 / div:10.2f} {100.0 * tot / total:6.2f}")
    unit = " per step" if steps else ""
    tail = [f"# {ncalls / div:.0f} launches{unit}, {total / 1e6 / div:.2f} ms kernel time{unit}"]
    for k, (n, tot) in sorted(cats.items(), key=lambda kv: -kv[1][1]):
        tail.append(f"#   {k:34s} {n / div:8.1f} launches {tot / 1e6 / div:9.2f} ms{unit}")
    print("\n".join(tail))
    print("\n".join(tail), file=sys.stderr)


if __name__ == "__main__":
    main(sys.argv[1:])
