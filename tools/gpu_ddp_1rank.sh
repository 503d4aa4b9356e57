#!/bin/bash
# the N > 1 code path of bench.py on ONE rank of a real RCCL group (process group, exchange modes, ddp / ddp_ab blocks)
#     gpurun --timeout 900 -- 'bash tools/gpu_ddp_1rank.sh [tag]'
set -u
cd "$(dirname "$0")/.."
tag=${1:-ddp_1rank}; out=gpurun_out/$tag; mkdir -p $out
VIDAR_FORCE_DDP=1 timeout 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 1 --steps 10 --warmup 3 --extra-configs "" --no-cpu-baseline --no-kernel-rooflines > $out/bench_ddp_1rank.json 2> $out/bench_ddp_1rank.err
echo "rc $?"; grep "^\[bench\]" $out/bench_ddp_1rank.err
python - $out/bench_ddp_1rank.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d["ms_per_step"], 1), "ms/step;", "ddp:", {k: v for k, v in d.get("ddp", {}).items() if k != "mode"})
for r in d.get("ddp_ab", []):
    print("  ", r.get("mode"), r.get("ms_per_step"), r.get("error"), (r.get("ddp") or {}).get("bucket_bytes"), (r.get("ddp") or {}).get("early_bucket_overlapped"))
PY
