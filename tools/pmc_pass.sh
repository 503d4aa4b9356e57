#!/bin/bash
# One rocprofv3 PMC pass per counter (separate passes, kernel-trace only -- never combined with sys/hip traces):
#   tools/pmc_pass.sh <outdir> "<counter> <counter,counter,...> ..." <command...>
# (a comma-separated group is collected in ONE pass)
# Writes <outdir>/pmc_<first counter of the group>.csv = "kernel,counter,dispatches,mean,min,max" per kernel of this library
# (and of the calibration micro-benchmark).
set -u
out=$(mkdir -p "$1" && cd "$1" && pwd); shift
counters=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for grp in $counters; do
  c=${grp%%,*}
  rm -rf /tmp/pmc_$c
  ( cd "$root" && timeout 300 rocprofv3 --pmc ${grp//,/ } --output-format csv -d /tmp/pmc_$c -o run -- "$@" ) > /tmp/pmc_$c.log 2>&1 < /dev/null
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" "$out/pmc_$c.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
    if "at::native" in name:
        continue
    if "anonymous namespace" in name:
        name = name.split("::")[1].split("(")[0]
    elif name.startswith("calib_"):
        name = name.split("(")[0]
    else:
        continue
    agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as f:
    f.write("kernel,counter,dispatches,mean_value,min,max\n")
    for (k, c), v in sorted(agg.items()):
        f.write(f"{k},{c},{len(v)},{sum(v)/len(v):.1f},{min(v):.1f},{max(v):.1f}\n")
PY
  else
    echo "no counter csv for $c"; tail -5 /tmp/pmc_$c.log
  fi
done
cat "$out"/pmc_*.csv
