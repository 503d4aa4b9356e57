"""CPU: bench.py's contract pieces that do not need a GPU -- it refuses to run without one (no CPU fallback
for the measured path), the driver's flags parse, `roofline.traffic` comes from a committed PMC summary that
exists, and the bounded CPU-baseline leg (the only part allowed to call the oracle) returns the documented
record."""
import json
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]


def test_refuses_to_run_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout)


def test_driver_flags_parse_and_defaults_are_single_gpu():
    sys.path.insert(0, str(ROOT))
    import bench
    old = sys.argv
    try:
        sys.argv = ["bench.py"]
        a = bench.parse()
        assert a.gpus == 1 and a.steps > 0 and a.warmup >= 0 and a.config == "vidar_1_8_nusc_1future"
        assert not a.cpu_baseline_full                      # the full-size CPU step is opt-in
        sys.argv = ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"]
        a = bench.parse()
        assert (a.gpus, a.steps, a.warmup) == (8, 20, 5)
    finally:
        sys.argv = old


def test_pmc_traffic_is_backed_by_committed_profiles():
    """`roofline.traffic` = profiles/pmc_traffic.json, which tools/make_pmc_traffic.py derives from the committed PMC
    summaries with the FETCH_SIZE calibration applied (gfx950 reports half of the fetched bytes, also for this
    library's random 128-byte line gathers: tools/micro/fetch_calib.hip)."""
    sys.path.insert(0, str(ROOT))
    import bench
    nbytes, src = bench.pmc_traffic("msda_bwd[L=4,P=8]")
    assert nbytes and nbytes > 7.5e8                        # at least the algorithmic bytes (756 MB at 7 680 queries / camera)
    # counted on the access pattern AND at the size the in-model kernel time belongs to: spatially coherent queries
    # (round 4), 7 680 padded visible queries per camera (round 5; the SURVEY shape has 10^4)
    pmc_dir = "r05_pmc_msda_sca_coherent_nq7680"
    assert pmc_dir in src and "coherent" in src and "calibration" in src and "Nq=7680" in src
    for f in ("pmc_FETCH_SIZE.csv", "pmc_WRITE_SIZE.csv"):
        assert (ROOT / "profiles" / pmc_dir / f).exists()
    assert bench.pmc_traffic("no such kernel") == (None, None)
    # the json is reproducible from the csv files
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "make_pmc_traffic.py"), "profiles/" + pmc_dir,
                        "profiles/r03_pmc_FETCH_SIZE_calibration.csv", "profiles/r03_pmc_WRITE_SIZE_calibration.csv",
                        "spatially coherent queries, head-major item order", "msda_sca_coherent", "7680"],
                       capture_output=True, text=True, cwd=ROOT, timeout=60)
    made = json.loads(r.stdout)
    have = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
    assert made == have
    cal = have["calibration"]
    assert abs(cal["fetch_factor"] - 2.0) < 0.02 and abs(cal["write_factor"] - 1.0) < 0.01
    bwd = have["msda_bwd[L=4,P=8]"]
    assert abs(bwd["bytes"] - (bwd["fetch_bytes"] + bwd["write_bytes"])) < 1 and 1 < bwd["ratio"] <= 3.0


def test_cpu_baseline_leg_reports_a_bounded_sample():
    """the subprocess the bench launches on rank 0 (small BEV here to keep the suite fast)"""
    sys.path.insert(0, str(ROOT))
    import bench
    rec = bench.cpu_baseline("vidar_1_8_nusc_1future", threads=4, with_backbone=False, reduced=True)
    assert rec["kind"] == "port" and rec["unit"] == "samples/s" and rec["cores"] == 4
    assert rec["value"] > 0 and "bounded sample" in rec["sample"] and "x that" in rec["sample"]
    json.dumps(rec)


def test_gemm_tuning_is_a_no_op_without_a_gpu_and_ships_validated_solutions():
    from vidar_amd import gemm_tuning
    if not torch.cuda.is_available():
        assert gemm_tuning.enable() == dict(enabled=False, reason="no GPU")
    lines = gemm_tuning.SHIPPED.read_text().splitlines()
    validators = [l for l in lines if l.startswith("Validator,")]
    entries = [l.split(",") for l in lines if l and not l.startswith("Validator,")]
    assert {v.split(",")[1] for v in validators} >= {"PT_VERSION", "GCN_ARCH_NAME", "ROCBLAS_VERSION", "HIPBLASLT_VERSION"}
    assert any("gfx950" in v for v in validators)
    assert len(entries) > 100 and all(len(e) == 4 and e[0].startswith("Gemm") and float(e[3]) > 0 for e in entries)
    assert len({(e[0], e[1]) for e in entries}) == len(entries)          # one solution per (op, shape)


def test_cpu_baseline_record_weights_op_rows_by_the_gpu_steps_launch_counts(monkeypatch):
    """`cpu_baseline` = op-level full-size CPU rows x the GPU step's own launches per step (a lower bound on the CPU
    step); surface-only ops (dvr family, evaluation KNN) are listed with calls_per_step 0 and stay out of the sum."""
    sys.path.insert(0, str(ROOT))
    import bench
    child = dict(cores=16, ops=[dict(op="msda_fwd[L=4,P=8]", cpu_ms=6000.0, shape="", impl="", threads=16),
                                dict(op="ray_ce_bwd", cpu_ms=800.0, shape="", impl="", threads=16),
                                dict(op="dvxlr.render[M=30000]", cpu_ms=1500.0, shape="", impl="", threads=16),
                                dict(op="not measured on the GPU", cpu_ms=1.0, shape="", impl="", threads=16)])
    monkeypatch.setattr(bench, "cpu_baseline_subprocess", lambda args, mode="ops": dict(child))
    args = type("A", (), dict(cpu_baseline_step=False, cpu_baseline_full=False, cpu_threads=16))()
    gpu_ops = {"msda_fwd[L=4,P=8]": dict(calls=60, avg_ms=0.3, total_ms=18.0, bytes_per_call=1),
               "ray_ce_bwd": dict(calls=10, avg_ms=0.9, total_ms=9.0, bytes_per_call=1)}
    rec = bench.cpu_baseline_record(args, gpu_ops, steps=2, kernel_rows=[dict(kernel="dvxlr.render[M=30000]", avg_ms=0.25)])
    assert rec["kind"] == "port" and rec["cores"] == 16 and rec["unit"] == "samples/s"
    assert rec["step_lower_bound_ms"] == 6000.0 * 30 + 800.0 * 5
    assert abs(rec["value"] - 1e3 / rec["step_lower_bound_ms"]) < 1e-12
    rows = {r["op"]: r for r in rec["ops"]}
    assert rows["msda_fwd[L=4,P=8]"]["speedup"] == 20000.0 and rows["dvxlr.render[M=30000]"]["calls_per_step"] == 0
    assert "speedup" not in rows["not measured on the GPU"]
    json.dumps(rec)


def _bench_group_worker(rank, world, port, out):
    """bench.py's grouped code path on CPU: barrier + synchronize bracket, MAX all-reduce of the elapsed time, the
    `ddp` block (initial and rebuilt buckets), rank-seeded samples -- so the first real multi-GPU run cannot die on
    plumbing (the GPU path differs only in the device the tensors live on)."""
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import time
    import numpy as np
    import torch.distributed as dist
    import bench
    from oracle import cpu_ops
    from test_plugin_cpu import _small_batch
    from vidar_amd import train as T
    torch.set_num_threads(2)
    r, l, w = T.init_distributed()
    torch.manual_seed(1234); np.random.seed(1000 + rank)
    cfg, batch = _small_batch("vidar_1_8_nusc_1future", seed=100 + rank)
    model = T.build_model(cfg).train()
    ddp = T.wrap_ddp(model, l)
    opt = T.build_optimizer(model)
    calls = []

    def step():
        if rank == 1:
            time.sleep(0.05)                       # the slow rank sets the time every rank reports
        with cpu_ops.patched():
            T.train_step(ddp, opt, batch)
        calls.append(1)

    marks = []
    elapsed = bench.timed_steps(step, steps=2, warmup=1, grouped=True, cuda=False,
                                after_warmup=lambda: marks.append("warm"), markers=marks.append)
    info = bench.ddp_info(ddp, w)
    out[rank] = dict(elapsed=elapsed, calls=len(calls), marks=marks, info=info,
                     first=float(batch["gt_points"][0][0, 0]))
    dist.barrier(); dist.destroy_process_group()


def test_grouped_timing_and_ddp_block_two_ranks_gloo():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = mp.Manager().dict()
    mp.spawn(_bench_group_worker, args=(2, port, out), nprocs=2, join=True)
    a, b = out[0], out[1]
    assert a["calls"] == b["calls"] == 3 and a["marks"] == ["warm", 1, 2]
    assert a["elapsed"] == b["elapsed"] and a["elapsed"] >= 0.1          # MAX over ranks (rank 1 sleeps 2 x 50 ms)
    assert a["first"] != b["first"], "ranks must time different samples"
    for i in (a["info"], b["info"]):
        assert i["world_size"] == 2 and i["backend"] == "gloo" and i["allreduce_bytes_per_step"] > 0
        # the default exchange is ONE flat all-reduce per step (train.FlatAllReduce); torch's DDP with its rebuilt
        # buckets is the VIDAR_DDP=torch alternative (tests/test_ddp_cpu.py runs both)
        assert i["buckets"] == 1 and i["bucket_bytes"] == [i["allreduce_bytes_per_step"]] and "flat" in i["mode"]
    assert a["info"]["allreduce_bytes_per_step"] == b["info"]["allreduce_bytes_per_step"]


def test_plain_gpus_n_relaunches_as_n_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher must not silently measure ONE rank: it re-executes itself under
    torch.distributed.run with N ranks on 127.0.0.1 (the driver's own multi-GPU command line); inside a launcher
    (WORLD_SIZE set) nothing is re-launched, and a --gpus that disagrees with the launcher fails loudly."""
    sys.path.insert(0, str(ROOT))
    import bench
    old = sys.argv
    try:
        argv = ["--gpus", "4", "--steps", "20", "--warmup", "5"]
        sys.argv = ["bench.py", *argv]
        a = bench.parse()
        monkeypatch.delenv("WORLD_SIZE", raising=False)
        cmd = bench.relaunch_command(a, argv, port=29512)
        assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
        assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
        assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29512"
        assert cmd[-len(argv) - 1].endswith("bench.py") and cmd[-len(argv):] == argv
        monkeypatch.setenv("WORLD_SIZE", "4")
        assert bench.relaunch_command(a, argv) is None
        monkeypatch.setenv("WORLD_SIZE", "2")
        with pytest.raises(SystemExit):
            bench.relaunch_command(a, argv)
        monkeypatch.delenv("WORLD_SIZE")
        sys.argv = ["bench.py", "--gpus", "1"]
        assert bench.relaunch_command(bench.parse(), ["--gpus", "1"]) is None
    finally:
        sys.argv = old


def test_relaunched_ranks_reach_the_rank_environment():
    """end to end on this CPU box: the re-launch really starts N ranks (each then refuses to run without a GPU --
    the measured path has no CPU form -- which is what the ranks' output must say, N times)."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600)
    out = r.stderr + r.stdout
    assert r.returncode != 0 and "re-running as" in out and out.count("needs a GPU") >= 2


def test_bench_line_records_the_ab_switches():
    """two bench lines are only comparable when they ran with the same switch settings: the line carries them, and
    reading them leaves them as they were"""
    import bench
    first = bench.library_switches()
    assert first == bench.library_switches()
    assert first["dvr_traversal"] == -1 and first["gemm_variant"] == 0 and first["msda_item_order"] == 1
    assert first["gradient_exchange"] in ("flat", "torch")
