"""Per-kernel statistics of the TIMED region of bench.py from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python bench.py ...
    python tools/trace_stats.py DIR/run_kernel_trace.csv STEPS > profiles/rNN_step_kernel_stats.csv

Only dispatches between the two `vidar_marker_kernel` launches are counted, so MIOpen's solver search
and other warm-up work do not pollute the numbers (plain `--stats` cannot separate them)."""
import csv
import sys
from collections import defaultdict

path, steps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
name_k = "Kernel_Name" if "Kernel_Name" in rows[0] else "Kernel Name"
marks = sorted(int(r["Start_Timestamp"]) for r in rows if "vidar_marker_kernel" in r[name_k])
assert len(marks) >= 2, "markers not found"
lo, hi = marks[0], marks[-1]
agg = defaultdict(list)
for r in rows:
    t = int(r["Start_Timestamp"])
    if lo < t < hi and "vidar_marker_kernel" not in r[name_k]:
        agg[r[name_k]].append(int(r["End_Timestamp"]) - t)
total = sum(sum(v) for v in agg.values())
w = csv.writer(sys.stdout)
w.writerow(["Name", "CallsPerStep", "TotalMsPerStep", "AverageUs", "Percentage"])
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([k[:160], round(len(v) / steps, 2), round(sum(v) / steps / 1e6, 4), round(sum(v) / len(v) / 1e3, 2),
                round(100.0 * sum(v) / total, 3)])
print(f"# timed region: {steps} steps, wall {(hi - lo) / steps / 1e6:.2f} ms/step, "
      f"kernel time {total / steps / 1e6:.2f} ms/step", file=sys.stderr)
