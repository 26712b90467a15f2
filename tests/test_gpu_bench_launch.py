"""bench.py's launch forms: the plain `python3 bench.py --gpus N` must start
its own ranks (a child torch.distributed.run, spawned before the parent touches a GPU) and relay
ONE JSON line; `--mode inproc` drives N device shards from one process (the Go binding's form).
Rehearsed on one card: SZG_BENCH_ONE_GPU=1 puts every shard on device 0, SZG_BENCH_BACKEND=gloo
exchanges on the CPU (RCCL wants one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(extra, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines          # exactly ONE line on stdout
    return json.loads(lines[0])


def test_plain_command_starts_its_own_ranks():
    out = run_bench(["--gpus", "2", "--steps", "48", "--warmup", "8", "--rows", "200000", "--settle-seconds", "0.2"],
                    {"SZG_BENCH_ONE_GPU": "1", "SZG_BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["config"]["mode"] == "ranks"
    assert out["steps"] == 48 and out["warmup"] == 8
    assert len(out["ranks"]) == 2 and {r["rank"] for r in out["ranks"]} == {0, 1}
    assert sum(r["rows"] for r in out["ranks"]) == 200000
    assert out["parity"]["ids_identical"] == out["parity"]["queries_checked"] > 0
    assert out["roofline"]["achieved"] > 0 and out["host_us_per_query"] > 0
    assert out["value"] > 0


def test_inproc_mode_and_unchanged_single_gpu_form():
    out = run_bench(["--gpus", "2", "--mode", "inproc", "--steps", "48", "--warmup", "8", "--rows", "200000",
                     "--settle-seconds", "0.2"], {"SZG_BENCH_ONE_GPU": "1"})
    assert out["n_gpus"] == 2 and out["config"]["mode"] == "inproc"
    assert out["parity"]["ids_identical"] == out["parity"]["queries_checked"] > 0
    assert out["host_us_per_query"] > 0
    one = run_bench(["--gpus", "1", "--steps", "32", "--warmup", "8", "--rows", "100000", "--settle-seconds", "0.2",
                     "--cpu-seconds", "1"], {})
    assert one["n_gpus"] == 1 and one["config"]["mode"] == "single"
    assert one["roofline"]["bound"] == "hbm" and one["cpu_baseline"]["kind"] == "port"
    assert one["batched"]["ids_identical_to_single_query_path"]
    assert one["batched"]["roofline"]["bound"] == "hbm" and one["batched"]["roofline"]["achieved"] > 0   # bfloat16 sweep
    assert one["repeats"] == 5 and one["spread"]["min"] <= one["value"] <= one["spread"]["max"]   # the median repeat
    assert one["lone_call"]["ms"] > 0 and 0 < one["lone_call"]["hbm_frac"] < 1
    assert one["parity"]["ids_identical"] == one["parity"]["queries_checked"]


def test_ranks_form_reports_transport_and_fixed_overhead():
    out = run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--rows", "250112", "--settle-seconds", "0.2"],
                    {"SZG_BENCH_ONE_GPU": "1", "SZG_BENCH_BACKEND": "gloo"})
    assert "host transport" in out["exchange_transport"]
    assert out["fixed_overhead_ms"] >= 0
    for r in out["ranks"]:
        assert r["exchanges_per_call"] == 1             # 20 queries: one local call, ONE all-gather
        assert r["fixed_overhead_ms"] >= 0 and r["sweeps_ms"] > 0
    assert out["parity"]["ids_identical"] == out["parity"]["queries_checked"] > 0


def test_rccl_exchange_inside_the_library_single_rank():
    """The N>1 code path with ONE rank (what a 1-GPU box can run of it): torch.distributed only hands the
    communicator id round; the all-gather is the library's own ncclAllGather."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.update({"SZG_BENCH_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "300", "--warmup", "8",
                        "--rows", "100000", "--settle-seconds", "0.2", "--no-extras"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert out["rccl_ranks"] == 1 and out["rccl_ranks_in_library"] == 1
    assert out["exchange_transport"].startswith("rccl")
    assert out["ranks"][0]["exchanges_per_call"] == 2   # 300 queries: two micro-batches of <= 256, pipelined
    assert out["ranks"][0]["zero_copy_staging"] in (0, 1)
    assert out["parity"]["ids_identical"] == out["parity"]["queries_checked"] > 0


def test_replicas_mode_answers_query_slices_without_a_collective():
    """SURVEY.md 8e's zero-collective alternative: every rank holds all rows and answers its slice of the K queries
    (rehearsed with two ranks on one card; gloo only carries the barriers and the timing reduction)."""
    out = run_bench(["--gpus", "2", "--parallelism", "replicas", "--steps", "21", "--warmup", "4", "--rows", "100000",
                     "--settle-seconds", "0.2", "--cpu-seconds", "1", "--repeats", "3"],
                    {"SZG_BENCH_ONE_GPU": "1", "SZG_BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and "replicas" in out["config"]["parallelism"]
    assert out["exchange_transport"] == "none (replicas)"
    assert sorted(r["queries"] for r in out["ranks"]) == [10, 11]        # 21 queries split over the two replicas
    assert all(r["rows"] == 100000 and r["exchanges_per_call"] == 0 for r in out["ranks"])
    assert out["parity"]["ids_identical"] == out["parity"]["queries_checked"] > 0
    assert out["repeats"] == 3 and out["value"] > 0
