"""Summarise a rocprofv3 results .db (kernel trace): per kernel name -> calls, avg us, total ms.
    python tools/prof_summary.py gpurun_out/x/prof/run_results.db [substring ...]"""
import sqlite3
import sys


def main(path, subs):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    sym = [t for t in tabs if "info_kernel_symbol" in t][0]
    rows = c.execute(f"select s.display_name, count(*), avg(d.end-d.start), sum(d.end-d.start), min(d.end-d.start), d.grid_size_x "
                     f"from {kd} d join {sym} s on d.kernel_id=s.id group by s.display_name, d.grid_size_x order by 4 desc")
    print(f"{'kernel':60s} {'grid':>10s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'total_ms':>10s}")
    for name, n, avg, tot, mn, grid in rows:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:]
        if subs and not any(s in name for s in subs):
            continue
        print(f"{short:60s} {grid:10d} {n:6d} {avg / 1e3:10.1f} {mn / 1e3:10.1f} {tot / 1e6:10.2f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
