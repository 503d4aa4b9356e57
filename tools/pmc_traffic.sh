#!/bin/bash
# HBM traffic of the HIP kernels from PMC counters (separate passes, as the microarch guide asks):
#   tools/pmc_traffic.sh <outdir> <kbench group...>
# Writes <outdir>/pmc_{FETCH_SIZE,WRITE_SIZE}.csv reduced to "kernel,counter,value" rows.
set -u
out=$(mkdir -p "$1" && cd "$1" && pwd); shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o run -- python "$GRAFT_REPO_ROOT/tools/kbench.py" "$@" > /tmp/pmc_$c.log 2>&1 < /dev/null
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" "$out/pmc_$c.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
    if "anonymous namespace" in name and "at::native" not in name:
        agg[(name.split("::")[1].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
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
