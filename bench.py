#!/usr/bin/env python3
"""bench.py -- queries/sec of the exact (brute-force) scan behind Collection.Search.

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--mode ranks|inproc] [--parallelism rows|replicas]

A "step" is one exact query over the whole synthetic corpus (one sweep of the packed
rows through the fused HIP scan; merges, float64 rerank and result assembly included).
Default workload = BASELINE.json's headline: 1M x 768 float32, cosine (angular)
distance, k=10, one query per sweep.

N > 1, two forms of the same row sharding (contiguous 64-row-aligned ranges, every
query goes to every shard, "scaling": "strong"):
  --mode ranks   (default) one process per GPU under torch.distributed.run; the per-shard
                 exact top-(k+1) lists are merged after ONE RCCL all-gather per micro-batch.
                 Called without WORLD_SIZE in the environment, `python bench.py --gpus N`
                 starts the N ranks itself (a child torch.distributed.run, spawned before
                 this process has touched a GPU) and relays the one JSON line.
  --mode inproc  ONE process, one handle with devices=[0..N-1] -- the form the Go binding
                 uses (go/syzgy_gpu.go): shards and merge inside the library.

  --parallelism replicas   (N > 1, --mode ranks) the zero-collective alternative SURVEY.md 8e names for a corpus
                 that fits one card: every rank holds ALL rows and answers its slice of the K queries (concurrent
                 Searches under RLock, collection.go:570, one replica per GPU); no exchange in the data path.  The
                 default stays rows sharded + all-gather (north_star); the line says which mode a number used.

The timed region -- EXACTLY K steps between barrier + synchronize on both sides, results on the host -- is repeated
R = --repeats times (default 5) on FRESH query sets; "value" is the MEDIAN repeat (ms_per_step x steps = that
repeat's call), "spread" carries min / max / repeats: one 9 ms call on one box of a pool that differs by +-4 % is not
a measurement.

Rank 0 prints ONE JSON line: metric/value/unit/..., plus
  "roofline":     HBM roofline of the fused scan kernel from HIP events recorded on the
                  library's scan stream inside the timed region;
  "cpu_baseline": the CPU oracle (a C port of the reference's Go scan; there is no Go
                  toolchain in this image) on this box's host cores, bounded sample (N=1);
  "batched":      the shared multi-query sweep on the matrix cores (its own fixed query set);
  "sketch_prepass": the optional 8-bit sketch pre-pass on the headline workload (same answers);
  "batched_quantized": the shared sweep on 64- / 16- / 8- / 4-bit copies of the headline shape (960 queries per call);
  "lone_call":    sequential ONE-query szg_search_topk calls -- the shape the unchanged Go Search API has;
  "host_us_per_query", "ranks", "rccl_ranks", "other_workloads" (cfg2/cfg3/cfg4/cfg5 per-GPU
                  shards: queries/s and one short roofline object each, peak 8000 GB/s).
The printed line is the compact form; the full objects (kernels, launch times, PMC traffic, host time
breakdown) go to stderr as "bench detail: {...}".
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TF = 157.3

WORKLOADS = {
    # name: (rows, dim, bits, metric, k, radius)   metric 0 = Euclidean, 1 = Cosine
    "headline": (1_000_000, 768, 32, 1, 10, 0.0),      # BASELINE.json metric / north_star target
    "cfg2": (1_000_000, 384, 32, 1, 10, 0.0),
    "cfg3": (1_000_000, 768, 8, 1, 10, 0.0),
    "cfg4": (10_000_000, 768, 32, 0, 100, 0.0),
    "cfg5": (100_000_000, 384, 4, 1, 0, 0.42),          # radius search: K is ignored (collection.go:598-605)
    "plumbing": (10_000, 128, 32, 1, 10, 0.0),
}
# the per-GPU shards of the 8-GPU configurations (what one card sweeps per query there)
SHARD_ROWS = {"cfg4": 1_250_000, "cfg5": 12_500_000}
SEED = 0x53595A4700000000
DTYPE = {4: "u4", 8: "u8", 16: "u16", 32: "f32", 64: "f64"}


# Libraries print to fd 1 (RCCL's version banner on communicator creation, for one): keep the real
# stdout for the JSON line and point fd 1 at stderr for everything else.
_REAL_STDOUT = os.fdopen(os.dup(1), "w")
os.dup2(2, 1)
sys.stdout = sys.stderr


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def _rf(r):
    return {key: r[key] for key in ("bound", "achieved", "unit", "frac", "traffic") if key in r}


def compact(obj):
    """The printed line keeps the contract's fields in full and one short object per extra leg (a driver may
    keep only a tail of stdout); everything else goes to stderr as "bench detail"."""
    o = dict(obj)
    if "batched" in o:
        b = o["batched"]
        o["batched"] = {"value": b["value"], "unit": b["unit"], "queries_per_sweep": b["queries_per_sweep"],
                        "avg_sweep_ms": b["avg_sweep_ms"], "kernel": b.get("kernel", "").split(" ")[0],
                        "roofline": _rf(b.get("roofline", {})),
                        "ids_identical_to_single_query_path": b["ids_identical_to_single_query_path"]}
        if "end_to_end_hbm_frac" in b:
            o["batched"]["end_to_end_hbm_frac"] = b["end_to_end_hbm_frac"]
        if "spread" in b:
            o["batched"]["spread"] = b["spread"]
    if "lone_call" in o:
        o["lone_call"] = {key: o["lone_call"][key] for key in ("ms", "queries_per_s", "hbm_frac")}
    if "spread" in o:
        o["spread"] = {key: o["spread"][key] for key in ("min", "max", "repeats")}
    if "sketch_prepass" in o:
        k = o["sketch_prepass"]
        o["sketch_prepass"] = {"option": "sketch=1 (off by default; DESIGN.md 4.5)", "value": k["value"], "unit": k["unit"],
                               "settled_by_the_sketch": "%d/%d" % (k["settled_by_the_sketch"], k["queries"]),
                               "identical_to_full_precision_path": k["ids_and_distances_identical_to_full_precision_path"],
                               "roofline": _rf(k["roofline"])}
    if "other_workloads" in o:
        o["other_workloads"] = {
            name: ({"error": w["error"]} if "error" in w else
                   dict({"value": w["value"], "roofline": _rf(w["roofline"]),
                         "identical_to_oracle": "%d/%d" % (w["parity"]["identical_to_oracle"], w["parity"]["queries_checked"])},
                        **({"batched_96_queries_per_s": w["batched"]["value"]} if "batched" in w else {})))
            for name, w in o["other_workloads"].items()}
    if "batched_quantized" in o:
        o["batched_quantized"] = {
            name: ({"error": w["error"]} if "error" in w else
                   {"value": w["value"], "avg_pass_ms": w["avg_pass_ms"], "queries_per_pass": w.get("queries_per_pass"),
                    "roofline": _rf(w["roofline"]),
                    "identical_to_single_query_path": w["ids_and_distances_identical_to_single_query_path"]})
            for name, w in o["batched_quantized"].items()}
    if "host_us_breakdown" in o:
        del o["host_us_breakdown"]
    return o


def emit(obj):
    """The ONE JSON line, on the process's real stdout (compact form; the full objects go to stderr)."""
    log("bench detail: " + json.dumps(obj))
    _REAL_STDOUT.write(json.dumps(compact(obj)) + "\n")
    _REAL_STDOUT.flush()


def host_cores():
    """Cores this process may actually use (affinity and cgroup quota), capped at the
    16-core share a one-GPU box gets."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def recorded_traffic(workload, rows, sweeps_per_launch):
    """HBM bytes per scan launch as RECORDED by the PMC passes under profiles/ (FETCH_SIZE with
    the gfx950 x2 correction + WRITE_SIZE; separate rocprofv3 runs, not this run), scaled from
    the sweeps per launch of the PMC run to this run's.  (None, None) when no pass exists for
    this workload at this row count."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            e = json.load(f)[workload]
        if int(e.get("rows", WORKLOADS.get(workload, (0,))[0])) != int(rows):
            return None, None
        return (int(e["hbm_bytes_per_launch"] / float(e.get("sweeps_per_launch", 1)) * sweeps_per_launch),
                "profiles/traffic.json (separate rocprofv3 --pmc passes, not this run)")
    except Exception:
        return None, None


def recorded_pass_traffic(key, rows):
    """HBM bytes per PASS of a shared sweep as recorded by the PMC passes under profiles/ (traffic.json, keys
    mq_<bits>bit), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            e = json.load(f)[key]
        if int(e["rows"]) != int(rows):
            return None
        return int(e["hbm_bytes_per_pass"])
    except Exception:
        return None


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the ranks as a CHILD
    torch.distributed.run (this process has not touched a GPU and never will) and relay
    rank 0's JSON line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    log("bench.py: starting %d ranks: %s" % (n, " ".join(cmd)))
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{"):
            line = ln
        elif ln.strip():
            log(ln)
    if line is not None:
        _REAL_STDOUT.write(line + "\n")
        _REAL_STDOUT.flush()
    if p.returncode != 0 or line is None:
        log("bench.py: the rank launcher exited with %d%s" % (p.returncode, "" if line else " and no JSON line"))
        sys.exit(p.returncode or 1)
    sys.exit(0)


def roofline_of(stats, rows, row_bytes, bits, metric, kind):
    """HBM roofline object of the scan launches timed in `stats` (HIP events on the scan stream)."""
    launches = max(stats["timed_launches"], 1)
    scan_ms = stats["scan_ms"] / launches
    bytes_per_launch = stats["scan_bytes"] / max(stats["scan_launches"], 1)
    achieved = bytes_per_launch / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    return {
        "bound": "hbm",
        "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": None,
        "kernel": "szg::scan_kernel<%d,%d,%s>" % (bits, metric, kind),
        "bytes_per_launch": int(bytes_per_launch),
        "sweeps_per_launch": round(bytes_per_launch / float(max(rows * row_bytes, 1)), 2),
        "avg_launch_ms": round(scan_ms, 5),
        "launches": int(stats["timed_launches"]),
    }


def calibrated_radius(ix, queries, radius, target_hits=500):
    """cfg5's radius (calibrate so a query has 1e2-1e3 hits): the distance of the
    target_hits-th neighbour of the first query, unless the named radius already lands there."""
    r, d = ix.search_radius(queries[0], radius, capacity=None)
    if 50 <= len(r) <= 5000:
        return radius, len(r)
    kk = min(target_hits, ix.rows)
    _, dd, cnt = ix.search_topk(queries[0], kk)
    rad = float(dd[0, int(cnt[0]) - 1])
    r, d = ix.search_radius(queries[0], rad)
    return rad, len(r)


def radius_searches(ix, queries, radius):
    """Radius searches of a query list: szg_search_radius_batch -- the collect sweeps walked query-major, 16 per
    launch, up to three launches in flight, the hits re-ranked from the device-side counters (concurrent
    single-query callers, the reference's Searches under RLock, are coalesced into the same launches)."""
    return ix.search_radius_batch(queries, radius)


def timed_leg(ix, queries, k, radius, n_settle=16):
    """Side workloads (one card, one query per sweep): a short settle, then the queries with HIP
    events on; returns (queries/s, stats, hits per query or None)."""
    ix.set_option("multi_query", 0)
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < 0.4:   # clocks / TLBs settle (untimed)
        if radius > 0:
            radius_searches(ix, queries[:16], radius)
        else:
            ix.search_topk(queries[:n_settle], k)
    ix.set_timing(True)
    ix.reset_stats()
    t0 = time.perf_counter()
    hits = None
    if radius > 0:
        hits = sum(len(r) for r, _ in radius_searches(ix, queries, radius)) / float(len(queries))
    else:
        ix.search_topk(queries, k)
    el = time.perf_counter() - t0
    st = ix.stats()
    ix.set_timing(False)
    return len(queries) / el, st, hits


def side_workload(name, n_queries, devices):
    """One of the other BASELINE configs on ONE card (the 8-GPU ones at their per-GPU shard size)."""
    import oracle as orc
    from syzgydb_amd import ScanIndex
    from syzgydb_amd.synth import synth_vectors
    n_rows, dim, bits, metric, k, radius = WORKLOADS[name]
    rows = SHARD_ROWS.get(name, n_rows)
    seed = SEED + sorted(WORKLOADS).index(name)
    t0 = time.time()
    with ScanIndex(dim, bits, metric, devices=devices) as ix:
        ix.synth(rows, seed)
        q = synth_vectors(seed + 1, 0, max(n_queries, 16), dim)
        hits = None
        if radius > 0:
            radius, _ = calibrated_radius(ix, q, radius)
        qps, st, hits = timed_leg(ix, q[:n_queries], k, radius)
        rf = roofline_of(st, rows, ix.row_bytes, bits, metric, "collect" if radius > 0 else "topk")
        rf["traffic"], src = recorded_traffic(name if radius == 0 else name + "_radius", rows, rf["sweeps_per_launch"])
        if src:
            rf["traffic_source"] = src
        out = {
            "workload": "%s%s: %d x %d, %d-bit, %s, %s, 1 query per sweep" % (
                name, " (per-GPU shard of %d rows / 8)" % n_rows if name in SHARD_ROWS else "", rows, dim, bits,
                "cosine" if metric else "euclidean", "radius %.6g" % radius if radius > 0 else "k=%d" % k),
            "value": round(qps, 1), "unit": "queries/s", "dtype": DTYPE[bits], "roofline": rf,
            "escalations": int(st["escalations"]), "full_replays": int(st["full_replays"]),
        }
        if hits is not None:
            out["hits_per_query"] = round(hits, 1)
        if radius > 0:
            # the same radius search for a batch of 96 queries through the shared sweeps (one pass of the shard per
            # up to 96 queries, hits re-ranked and sorted on the device): what a batching caller gets
            qb = synth_vectors(seed + 7, 0, 96, dim)
            ix.set_option("multi_query", 1)
            ix.search_radius_batch(qb, radius)
            ix.search_radius_batch(qb, radius)
            ix.set_timing(True)
            ix.reset_stats()
            t0b = time.perf_counter()
            hb = ix.search_radius_batch(qb, radius)
            eb = time.perf_counter() - t0b
            bst = ix.stats()
            ix.set_timing(False)
            ix.set_option("multi_query", 0)
            h1 = ix.search_radius_batch(qb[:8], radius)
            pass_ms = bst["scan_ms"] / max(bst["timed_launches"], 1)
            out["batched"] = {
                "queries": 96, "value": round(96 / eb, 1), "unit": "queries/s",
                "queries_per_pass": round(bst["mq_queries"] / max(bst["mq_launches"], 1), 1),
                "avg_pass_ms": round(pass_ms, 5),
                "pass_GBps": round(rows * ix.row_bytes / (pass_ms * 1e-3) / 1e9, 1) if pass_ms else 0.0,
                "hits_per_query": round(sum(len(r) for r, _ in hb) / 96.0, 1),
                "identical_to_one_sweep_per_query": bool(all((a[0] == b[0]).all() and (a[1] == b[1]).all()
                                                             for a, b in zip(hb[:8], h1))),
            }
        # parity on the first rows of the same corpus (a second, small handle): ids and float64 distances
        nchk = min(rows, 50_000)
        with ScanIndex(dim, bits, metric, devices=devices) as small:
            small.synth(nchk, seed)
            small.set_option("multi_query", 0)
            ref_rows = small.read_rows(0, nchk)
            same = 0
            for i in range(2):
                if radius > 0:
                    rr, dd = small.search_radius(q[i], radius + 0.02)  # a few hits on the small corpus too
                    er, ed, _ = orc.search_exact(ref_rows, dim, bits, metric, q[i], radius=radius + 0.02)
                else:
                    rr, dd, cc = small.search_topk(q[i], k)
                    rr, dd = rr[0, :cc[0]], dd[0, :cc[0]]
                    er, ed, _ = orc.search_exact(ref_rows, dim, bits, metric, q[i], k=k)
                same += int(len(rr) == len(er) and (rr == er).all() and (dd == ed).all())
            out["parity"] = {"rows": nchk, "queries_checked": 2, "identical_to_oracle": same}
    log("side workload %s: %.1f s, %.0f GB/s" % (name, time.time() - t0, rf["achieved"]))
    return out


def batched_leg(bits, n_rows, dim, metric, k, devices, seed):
    """The shared sweep on a quantized copy of the headline shape: 960 queries in one call (the exact int8 sweep for
    4-bit rows; the bfloat16 sweep for 64-, 16- and -- their codes being exact in bfloat16 -- tiled 8-bit rows)."""
    from syzgydb_amd import ScanIndex
    from syzgydb_amd.synth import synth_vectors
    with ScanIndex(dim, bits, metric, devices=devices) as ix:
        ix.synth(n_rows, seed)
        qb = synth_vectors(seed + 2, 0, 960, dim)
        ix.search_topk(qb, k)
        ix.search_topk(qb, k)
        times = []
        for _ in range(5):  # the median of five calls (a call is 2-11 ms)
            t0 = time.perf_counter()
            b_rows, b_dist, _ = ix.search_topk(qb, k)
            times.append(time.perf_counter() - t0)
        elapsed = sorted(times)[2]
        ix.set_timing(True)
        ix.reset_stats()
        ix.search_topk(qb, k)
        st = ix.stats()
        ix.set_timing(False)
        ix.set_option("multi_query", 0)
        s_rows, s_dist, _ = ix.search_topk(qb[:32], k)
        pass_ms = st["scan_ms"] / max(st["timed_launches"], 1)
        gbps = n_rows * ix.row_bytes / (pass_ms * 1e-3) / 1e9 if pass_ms else 0.0
        kern = {16: "szg::mq_score_bf16d_kernel<6,cosine,collect> (v_mfma_f32_16x16x32_bf16, 16-bit codes decoded on the fly)",
                64: "szg::mq_score_bf16s_kernel<6,cosine,collect,64> (v_mfma_f32_16x16x32_bf16, float64 narrowed on the fly)"}.get(
                    bits, "szg::mq_score_i8s_kernel (v_mfma_i32_16x16x64_i8, exact integer)")
        if bits == 8 and st["mq_bf16_sweeps"]:
            kern = "szg::mq_score_bf16d8_kernel<6,cosine,collect> (v_mfma_f32_16x16x32_bf16, 8-bit codes exact in bfloat16: 96 queries per pass)"
        return {"workload": "%d x %d, %d-bit, cosine, k=%d, 960 queries in one call" % (n_rows, dim, bits, k),
                "value": round(960 / elapsed, 1), "unit": "queries/s",
                "spread": {"min": round(960 / max(times), 1), "max": round(960 / min(times), 1), "calls": 5},
                "queries_per_pass": round(st["mq_queries"] / max(st["mq_launches"], 1), 2),
                "avg_pass_ms": round(pass_ms, 5), "kernel": kern,
                "roofline": {"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(gbps / HBM_PEAK_GBS, 4),
                             "traffic": recorded_pass_traffic("mq_%dbit" % bits, n_rows)},
                "escalations": int(st["escalations"]),
                "ids_and_distances_identical_to_single_query_path":
                    bool((b_rows[:32] == s_rows).all() and (b_dist[:32] == s_dist).all())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS) + ["all"],
                    help="'all' = the headline plus every other config's roofline (the default at N=1 too)")
    ap.add_argument("--mode", default="ranks", choices=["ranks", "inproc"],
                    help="N>1: one process per GPU + RCCL all-gather, or one process driving N devices")
    ap.add_argument("--parallelism", default="rows", choices=["rows", "replicas"],
                    help="N>1 ranks: rows sharded over the GPUs + all-gather (default), or every rank holds all rows "
                         "and answers its slice of the queries (no collective)")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of K steps on fresh queries; value = the median")
    ap.add_argument("--rows", type=int, default=0, help="override the workload's row count")
    ap.add_argument("--exchange-every", type=int, default=0,
                    help="N>1 ranks: queries per all-gather micro-batch (0 = 256, or steps/4 for short runs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--settle-seconds", type=float, default=2.0,
                    help="untimed sweeps before the warm-up steps (clock / TLB settling)")
    ap.add_argument("--no-cpu", action="store_true", help="skip cpu_baseline and the recall check")
    ap.add_argument("--no-extras", action="store_true", help="skip the batched leg and the other workloads")
    ap.add_argument("--verify", type=int, default=4, help="queries re-checked against the oracle")
    args = ap.parse_args()
    if args.workload == "all":
        args.workload, args.no_extras = "headline", False

    if os.environ.get("SZG_BENCH_FORCE_DIST") == "1" and "RANK" not in os.environ:
        # the N>1 code path with ONE rank (a 1-GPU box): a rendezvous of its own
        os.environ.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                           "MASTER_PORT": str(free_port())})
        args.gpus = 1
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and args.mode == "ranks" and env_world is None:
        spawn_ranks(args.gpus)  # does not return
    args.repeats = max(1, args.repeats)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    if args.mode == "ranks" and world != args.gpus:
        log("bench.py: --gpus %d but WORLD_SIZE=%d; using the launcher's world size" % (args.gpus, world))
        args.gpus = world
    inproc = args.mode == "inproc" and args.gpus > 1
    if inproc and world != 1:
        log("bench.py: --mode inproc runs in ONE process (WORLD_SIZE=%d)" % world)
        sys.exit(2)

    import numpy as np
    from syzgydb_amd import ScanIndex
    from syzgydb_amd.synth import synth_vectors
    from syzgydb_amd.sharded import Comm, ShardedSearcher, shard_range

    n_rows, dim, bits, metric, k, radius = WORKLOADS[args.workload]
    if args.rows:
        n_rows = args.rows
    seed = SEED + sorted(WORKLOADS).index(args.workload)
    one_gpu = os.environ.get("SZG_BENCH_ONE_GPU") == "1"  # rehearsal: every shard on device 0

    dist = None
    torch = None
    # rehearsal knobs (1-GPU box): SZG_BENCH_BACKEND=gloo exchanges on the CPU,
    # SZG_BENCH_ONE_GPU=1 puts every rank's shard on device 0
    backend = os.environ.get("SZG_BENCH_BACKEND", "nccl")
    if one_gpu:
        local_rank = 0
    # SZG_BENCH_FORCE_DIST=1: take the N>1 code path (process group, all-gather, merge)
    # with a single rank -- lets a 1-GPU box exercise the RCCL plumbing
    dist_path = world > 1 or os.environ.get("SZG_BENCH_FORCE_DIST") == "1"
    replicas = args.parallelism == "replicas" and args.mode == "ranks" and world > 1
    rccl_ranks = None
    if dist_path:
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        # ranks the collective library itself reaches: an all-reduce of ones
        t = torch.ones(1, dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t)
        rccl_ranks = int(t.item())
    else:
        try:
            import torch  # only for torch.cuda.synchronize() around the timed region
        except Exception:  # pragma: no cover
            torch = None

    if inproc:
        devices = [0] * args.gpus if one_gpu else list(range(args.gpus))
        lo, hi = 0, n_rows
    elif replicas:
        devices = [local_rank]
        lo, hi = 0, n_rows            # every rank holds the whole corpus
    else:
        devices = [local_rank]
        lo, hi = shard_range(n_rows, rank, world)
    ix = ScanIndex(dim, bits, metric, devices=devices)
    t0 = time.time()
    ix.synth(hi - lo, seed, first_row=lo)  # corpus generated in HBM, reference encoding rules
    ix.set_row_base(lo)
    log("[rank %d] corpus rows [%d, %d) x %d B resident on device(s) %s in %.2f s" % (
        rank, lo, hi, ix.row_bytes, devices, time.time() - t0))
    queries = synth_vectors(seed + 1, 0, args.warmup + args.steps * args.repeats, dim)
    qw = queries[: args.warmup]
    q_rep = [queries[args.warmup + r * args.steps: args.warmup + (r + 1) * args.steps] for r in range(args.repeats)]
    qt = q_rep[0]
    # replicas: this rank's slice of the K queries of a repeat (contiguous, the remainder spread over the first ranks)
    my_lo, my_hi = (rank * args.steps // world, (rank + 1) * args.steps // world) if replicas else (0, args.steps)

    # the headline is one query per sweep (HBM roofline); the shared multi-query
    # sweep is measured separately below and reported under "batched"
    ix.set_option("multi_query", 0)
    hits_per_query = None
    if radius > 0:
        # radius calibrated once on rank 0's shard-independent view: every rank derives it from
        # the same full-corpus rule when it holds everything, else rank 0 broadcasts it
        if world == 1:
            radius, _ = calibrated_radius(ix, qt, radius)
        else:
            rad = torch.tensor([radius], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.broadcast(rad, 0)
            radius = float(rad.item())

    # queries per all-gather: the library answers runs of <= 128 queries with one local call and one exchange and
    # pipelines longer ones in micro-batches of 256 behind a worker thread (szg_search_topk_sharded)
    chunk = args.steps if args.steps <= 128 else 256
    searcher = None
    transport = None
    if dist_path and not replicas:
        # The exchange lives in the library (csrc/scan_comm.cpp): its own RCCL communicator on this rank's card,
        # the 128-byte id handed round through the process group once.  Should RCCL refuse inside the library on
        # ANY rank, every rank falls back to the group's own all-gather as host transport (and says so below).
        comm = None
        if backend == "nccl":
            try:
                comm = Comm.from_process_group(device=local_rank)
                ok = 1
            except Exception as e:  # pragma: no cover
                log("[rank %d] in-library RCCL communicator failed: %s" % (rank, e))
                ok = 0
            t = torch.tensor([ok], dtype=torch.int64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if int(t.item()) == 1:
                transport = "rccl (ncclAllGather inside libsyzgy_scan.so)"
            else:
                if comm is not None:
                    comm.close()
                comm = Comm.from_process_group(via_torch_device=torch.device("cuda", local_rank))
                transport = "torch.distributed nccl all-gather as host transport (in-library RCCL init failed)"
        else:
            comm = Comm.from_process_group()
            transport = "%s all-gather on the CPU as host transport (rehearsal)" % backend
        searcher = ShardedSearcher(index=ix, comm=comm)
        if k > 0:
            comm.reserve(min(max(args.steps, args.warmup, 1), 256), k)  # staging of the largest micro-batch, up front

    def run(q):
        if replicas:
            q = q[my_lo:my_hi]
            if len(q) == 0:
                return np.zeros((0, max(k, 1)), np.uint64), np.zeros((0, max(k, 1)))
        if radius > 0:
            if searcher is not None:
                outs = searcher.search_radius_batch(q, radius)
            else:
                outs = radius_searches(ix, q, radius)
            return [o[0] for o in outs], [o[1] for o in outs]
        if searcher is not None:
            r, d, _, _ = searcher.search_stream(q, k)
            return r, d
        r, d, _ = ix.search_topk(q, k)
        return r, d

    def sync():
        if torch is not None and (backend == "nccl" or not dist_path) and torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # settle the card first (clocks, TLBs, allocator): a fresh process measures 3-5 % low for
    # its first second or two of sweeps.  Untimed, outside the W warm-up steps, same on every rank.
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < args.settle_seconds:
        if radius > 0:
            ix.search_radius(qt[0], radius)
        else:
            ix.search_topk(qt[:64], k)
    if args.warmup:
        run(qw)
    # First use is not what is measured: two UNTIMED calls with exactly the timed call's shape (K queries: the
    # same batch boundaries, micro-batches, staging and all-gather sizes; the warm-up queries repeated, so the
    # timed queries themselves stay unseen), a fixed number on every rank.  Without them a 20-step run at N > 1
    # paid the allocation of its exchange buffers and the first all-gather of that size inside the timed region.
    shape_q = np.resize(qw if args.warmup else qt, (len(qt), dim))
    for _ in range(2):
        run(shape_q)
    ix.set_timing(True)
    ix.reset_stats()
    if searcher is not None:
        searcher.comm.reset_stats()
    # R timed regions of EXACTLY K steps each, every one on queries nobody has seen, each bracketed by barrier +
    # synchronize on both sides and reduced to the MAX over ranks; the median repeat is the value
    rep_elapsed, rep_mine, rep_scan_ms = [], [], []
    res_rows = res_dist = None
    scan_ms_before = 0.0
    for rep in range(args.repeats):
        sync()
        t0 = time.perf_counter()
        rr, rd = run(q_rep[rep])  # returns when every result is on the host
        if torch is not None and (backend == "nccl" or not dist_path) and torch.cuda.is_available():
            torch.cuda.synchronize()
        el = time.perf_counter() - t0   # this rank's K steps, device idle
        rep_mine.append(el)
        scan_ms_now = ix.stats()["scan_ms"]
        rep_scan_ms.append(scan_ms_now - scan_ms_before)   # HIP-event time of this repeat's sweeps
        scan_ms_before = scan_ms_now
        if dist is not None:
            dist.barrier()
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        rep_elapsed.append(el)
        if rep == 0:
            res_rows, res_dist = rr, rd   # (the parity legs check the first repeat's queries)
    order = sorted(range(args.repeats), key=lambda i: rep_elapsed[i])
    med = order[len(order) // 2]
    elapsed = rep_elapsed[med]
    stats = ix.stats()
    ix.set_timing(False)
    if radius > 0:
        hits_per_query = float(np.mean([len(r) for r in res_rows]))
    my_elapsed = rep_mine[med]
    n_timed = max(args.steps * args.repeats, 1)     # queries behind the cumulative statistics
    per_rank = None
    if dist is not None:
        mine = roofline_of(stats, hi - lo, ix.row_bytes, bits, metric, "")
        cst = searcher.comm.stats() if searcher is not None else {"host_us": 0.0, "exchanges": 0, "exchange_us": 0.0,
                                                                  "status_rounds": 0, "chain_rounds": 0, "zero_copy": 0}
        row = {"rank": rank, "device": devices[0], "rows": hi - lo, "queries": my_hi - my_lo,
               "elapsed_s": round(my_elapsed, 6),
               "scan_GBps": mine["achieved"], "avg_launch_ms": mine["avg_launch_ms"],
               "sweeps_ms": round(rep_scan_ms[med], 4),
               # everything in this rank's timed region (the median repeat) that is not a sweep: pipeline fill /
               # drain, host work that did not hide behind sweeps, the exchange
               "fixed_overhead_ms": round(1e3 * my_elapsed - rep_scan_ms[med], 4),
               "host_us_per_query": round((stats["host_prep_us"] + stats["host_finish_us"] + stats["host_enqueue_us"] +
                                           cst["host_us"]) / n_timed, 2),
               "exchanges_per_call": round(cst["exchanges"] / float(args.repeats), 2),
               "exchange_ms_per_batch": round(1e-3 * cst["exchange_us"] / max(cst["exchanges"], 1), 3),
               "status_rounds": int(cst["status_rounds"]), "zero_copy_staging": int(cst["zero_copy"])}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, row)

    out = None
    if rank == 0:
        qps = args.steps / elapsed
        rf = roofline_of(stats, hi - lo if not inproc else (n_rows + args.gpus - 1) // args.gpus, ix.row_bytes,
                         bits, metric, "collect" if radius > 0 else "topk")
        n_dev = args.gpus
        if world == 1 and not inproc:
            rf["traffic"], src = recorded_traffic(args.workload, n_rows, rf["sweeps_per_launch"])
            if src:
                rf["traffic_source"] = src
        host_us = (stats["host_prep_us"] + stats["host_finish_us"] + stats["host_enqueue_us"]) / n_timed
        if searcher is not None:
            host_us += searcher.comm.stats()["host_us"] / n_timed
        out = {
            "metric": "queries/sec, exact scan 1M x 768 cosine k=10" if args.workload == "headline"
            else "queries/sec, exact scan (%s)" % args.workload,
            "value": round(qps, 2),
            "unit": "queries/s",
            "n_gpus": n_dev,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 5),
            "repeats": args.repeats,
            "spread": {"min": round(args.steps / max(rep_elapsed), 2), "max": round(args.steps / min(rep_elapsed), 2),
                       "repeats": args.repeats, "what": "queries/s of each timed region of K steps; value = the median"},
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": DTYPE[bits],
            "data": "synthetic",
            "config": {
                "workload": "%s: %d x %d, %d-bit, %s, %s, 1 query per sweep" % (
                    args.workload, n_rows, dim, bits, "cosine" if metric else "euclidean",
                    "radius %.6g" % radius if radius > 0 else "k=%d" % k),
                "rows": n_rows, "dim": dim, "quantization": bits,
                "distance": "cosine" if metric else "euclidean", "k": k,
                "radius": radius if radius > 0 else None,
                "queries_per_sweep": 1,
                "mode": "inproc" if inproc else ("ranks" if dist_path else "single"),
                "parallelism": ("replicas: each of the %d GPUs holds all rows and answers its slice of the queries, "
                                "no collective in the data path" % n_dev) if replicas else
                "rows sharded over %d GPU(s)%s" % (
                    n_dev, (", one handle with %d device shards in one process" % n_dev) if inproc else
                    (", one process per GPU, 1 %s all-gather per %d queries (inside the library)" % (
                        "RCCL" if backend == "nccl" else backend, chunk) if dist_path else "")),
            },
            "roofline": rf,
            "host_us_per_query": round(host_us, 2),
            "host_us_breakdown": {"prepare": round(stats["host_prep_us"] / n_timed, 2),
                                  "enqueue_hip_calls": round(stats["host_enqueue_us"] / n_timed, 2),
                                  "assemble": round(stats["host_finish_us"] / n_timed, 2),
                                  "exchange_pack_merge": round(searcher.comm.stats()["host_us"] / n_timed, 2)
                                  if searcher is not None else 0.0},
            "escalations": int(stats["escalations"]),
            "full_replays": int(stats["full_replays"]),
        }
        if not dist_path:
            out["fixed_overhead_ms"] = round(1e3 * elapsed - rep_scan_ms[med], 4)  # timed region minus the sweeps
        if hits_per_query is not None:
            out["hits_per_query"] = round(hits_per_query, 1)
        if dist_path:
            out["rccl_ranks"] = rccl_ranks if backend == "nccl" else 0
            if searcher is not None:
                out["rccl_ranks_in_library"] = int(searcher.comm.stats()["rccl_ranks"])  # ncclCommCount of the library's communicator
                out["exchange_backend"] = backend
                out["exchange_transport"] = transport
            else:
                out["exchange_transport"] = "none (replicas)"
            out["fixed_overhead_ms"] = max(r["fixed_overhead_ms"] for r in per_rank)
            out["ranks"] = per_rank
        if inproc:
            out["roofline"]["note"] = "per device shard; the N shards sweep concurrently"

    # ---- N>1 parity: the oracle on every rank's own rows, merged on rank 0 -------------
    if dist_path and not replicas and not args.no_cpu and radius == 0 and args.verify > 0:
        import oracle as orc
        orc.build()
        nv = min(2, args.verify, len(qt))
        local = ix.read_rows(0, hi - lo)
        mine = []
        for i in range(nv):
            er, ed, _ = orc.search_exact(local, dim, bits, metric, qt[i], k=k)
            mine.append(((np.asarray(er, dtype=np.uint64) + np.uint64(lo)).tolist(), [float(x) for x in ed]))
        del local
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        if rank == 0:
            same = 0
            for i in range(nv):
                pairs = sorted((d, r) for part in everyone for r, d in zip(part[i][0], part[i][1]))[:k]
                same += int([int(r) for _, r in pairs] == [int(x) for x in res_rows[i]] and
                            [d for d, _ in pairs] == [float(x) for x in res_dist[i]])
            out["parity"] = {"queries_checked": nv, "ids_identical": same,
                             "recall_at_k": 1.0 if same == nv else None,
                             "how": "oracle top-k of every rank's own rows, merged by (distance, row) on rank 0"}
            if same != nv:
                log("bench.py: PARITY FAILURE: merged GPU result differs from the oracle's")

    # ---- shared multi-query sweep (MFMA), 32-bit cosine only, N=1 -----------------
    if world == 1 and not inproc and bits == 32 and metric == 1 and radius == 0 and not args.no_extras:
        qb = synth_vectors(seed + 2, 0, 1024, dim)  # its own fixed set, whatever --steps is
        ix.set_option("multi_query", 1)
        # untimed warm-up with the same call: the first full-size shared-sweep call after
        # allocation runs ~1.7x slower (buffer first use); steady state is what is reported
        ix.search_topk(qb, k)
        b_times = []
        for _ in range(5):                            # throughput: no per-kernel events; the MEDIAN of five calls
            t0 = time.perf_counter()                  # (one call is 6 ms: a single sample moved by 20 % from run to run)
            b_rows, _, _ = ix.search_topk(qb, k)
            b_times.append(time.perf_counter() - t0)
        b_elapsed = sorted(b_times)[2]
        ix.set_timing(True)
        ix.reset_stats()
        ix.search_topk(qb, k)                         # same call again with HIP events: per-sweep time
        bst = ix.stats()
        ix.set_timing(False)
        ix.set_option("multi_query", 0)
        s_rows, _, _ = ix.search_topk(qb[:64], k)     # the single-query path on the same queries
        sweep_ms = bst["scan_ms"] / max(bst["timed_launches"], 1)
        per_sweep = bst["mq_queries"] / max(bst["mq_launches"], 1)
        flops = 2.0 * n_rows * dim * per_sweep
        tf = flops / (sweep_ms * 1e-3) / 1e12
        gbps = n_rows * ix.row_bytes / (sweep_ms * 1e-3) / 1e9
        bf16 = bst.get("mq_bf16_sweeps", 0) > 0
        out["batched"] = {
            "queries": 1024,
            "queries_per_sweep": round(per_sweep, 2),
            "value": round(1024 / b_elapsed, 1), "unit": "queries/s",
            "spread": {"min": round(1024 / max(b_times), 1), "max": round(1024 / min(b_times), 1), "calls": 5},
            "avg_sweep_ms": round(sweep_ms, 5),
            "hbm_GBps": round(gbps, 1),
            "ids_identical_to_single_query_path": bool((b_rows[:64] == s_rows).all()),
        }
        if bf16:
            # rows and queries rounded to bfloat16 on the fly, v_mfma_f32_16x16x32_bf16: the sweep is a
            # stream of the rows (HBM-bound); candidates re-scored in float32, re-ranked in float64
            out["batched"].update({
                "kernel": "szg::mq_score_bf16s_kernel<6,cosine,collect,32> (v_mfma_f32_16x16x32_bf16)",
                "f32_equivalent_TFLOPs": round(tf, 2),
                "roofline": {"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(gbps / HBM_PEAK_GBS, 4),
                             "traffic": recorded_pass_traffic("mq_32bit", n_rows)},
                # the whole pipeline against the same roof: queries/s / queries per pass x bytes per pass
                "end_to_end_hbm_frac": round(1024 / b_elapsed / max(per_sweep, 1e-9) * n_rows * ix.row_bytes / 1e9 / HBM_PEAK_GBS, 4),
            })

    # ---- a lone Search: ONE query per call, calls one after the other (the unchanged Go API's shape) ---------------
    if world == 1 and not inproc and radius == 0 and not args.no_extras:
        ix.set_option("multi_query", 0)
        ql = synth_vectors(seed + 4, 0, 96, dim)
        for i in range(16):
            ix.search_topk(ql[i], k)
        lone = []
        for i in range(16, 96):
            t0 = time.perf_counter()
            ix.search_topk(ql[i], k)
            lone.append(time.perf_counter() - t0)
        lone.sort()
        l_med = lone[len(lone) // 2]
        out["lone_call"] = {
            "what": "sequential szg_search_topk calls of ONE query each (80 calls, median): prepare, upload, sweep, "
                    "merge, float64 re-rank, copy back, result assembly -- nothing to overlap with",
            "ms": round(1e3 * l_med, 4), "queries_per_s": round(1.0 / l_med, 1),
            "hbm_frac": round(n_rows * ix.row_bytes / l_med / 1e9 / HBM_PEAK_GBS, 4),
            "ms_min": round(1e3 * lone[0], 4), "ms_p90": round(1e3 * lone[int(len(lone) * 0.9)], 4)}

    # ---- 8-bit sketch pre-pass (optional path, off by default): the same queries, ONE per sweep --------
    if world == 1 and not inproc and bits == 32 and metric == 1 and radius == 0 and not args.no_extras and n_rows >= 65536:
        ix.set_option("multi_query", 0)
        ix.set_option("sketch", 1)
        qs = qt if len(qt) >= 256 else synth_vectors(seed + 3, 0, 256, dim)
        ix.search_topk(qs[:64], k)                    # builds the sketch (once per load) + warm-up
        ix.search_topk(qs[:64], k)
        t0 = time.perf_counter()
        k_rows, k_dist, _ = ix.search_topk(qs, k)
        k_elapsed = time.perf_counter() - t0
        ix.set_timing(True)
        ix.reset_stats()
        ix.search_topk(qs, k)
        kst = ix.stats()
        ix.set_timing(False)
        ix.set_option("sketch", 0)
        f_rows, f_dist, _ = ix.search_topk(qs[:32], k)   # the full-precision path on the same queries
        sk_row_bytes = ((dim + 15) // 16) * 16
        launch_ms = kst["scan_ms"] / max(kst["timed_launches"], 1)
        sweeps_per_launch = kst["scan_bytes"] / max(kst["scan_launches"], 1) / (n_rows * sk_row_bytes)
        gbps = kst["scan_bytes"] / max(kst["scan_launches"], 1) / (launch_ms * 1e-3) / 1e9 if launch_ms else 0.0
        out["sketch_prepass"] = {
            "what": "option sketch=1 (off by default): 8-bit sketch sweep -> float32/float64 re-rank, certified by the "
                    "triangle inequality of the angular distance; same answers (DESIGN.md 4.5)",
            "queries": int(len(qs)), "value": round(len(qs) / k_elapsed, 1), "unit": "queries/s",
            "settled_by_the_sketch": int(kst["sketch_queries"]), "handed_to_the_full_sweep": int(kst["sketch_fallbacks"]),
            "ids_and_distances_identical_to_full_precision_path":
                bool((k_rows[:32] == f_rows).all() and (k_dist[:32] == f_dist).all()),
            "kernel": "szg::scan_kernel<8,cosine,topk> over the sketch",
            "roofline": {"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbps / HBM_PEAK_GBS, 4), "traffic": None,
                         "bytes_per_launch": int(kst["scan_bytes"] / max(kst["scan_launches"], 1)),
                         "sweeps_per_launch": round(sweeps_per_launch, 2), "avg_launch_ms": round(launch_ms, 5)},
        }

    # ---- recall / parity spot check + CPU baseline (rank 0, N=1) -----------------
    if rank == 0 and (world == 1 or replicas) and not args.no_cpu:
        import oracle as orc
        orc.build()
        cores = host_cores()
        sample_rows = min(n_rows, 100_000)
        rows_host = ix.read_rows(0, sample_rows)  # the same bytes the GPU scans
        if not inproc and world == 1:
            t0 = time.time()
            kk = k if radius == 0 else 10
            # single-thread rate first, to size the all-core sample
            secs1, _ = orc.bench_topk(rows_host, dim, bits, metric, qt[:1], kk, 1)
            per_query = max(secs1, 1e-6)
            nq = int(max(cores, cores * args.cpu_seconds / per_query))
            nq = min(nq, 4096)
            q_cpu = qt if nq <= len(qt) else np.concatenate([qt, synth_vectors(seed + 5, 0, nq - len(qt), dim)])
            secs, cpu_rows = orc.bench_topk(rows_host, dim, bits, metric, q_cpu[:nq], kk, cores)
            scale = sample_rows / float(n_rows)
            out["cpu_baseline"] = {
                "value": round(nq / secs * scale, 3),
                "unit": "queries/s",
                "cores": cores,
                "kind": "port",
                "sample": "%d queries x first %d of %d rows, %d threads, %.1f s, scaled to %d rows; C port of the "
                          "reference's Go scan" % (nq, sample_rows, n_rows, cores, secs, n_rows),
                "single_thread": round(1.0 / per_query * scale, 4),
            }
            # the reference also CRCs and re-parses each record's span and allocates a decode
            # buffer on every visit (spanfile.go:757, collection.go:769): one query with that
            # per-record work included, single thread, on a smaller sample
            fs_rows = min(sample_rows, 20_000)
            fsecs, frows = orc.bench_topk_faithful(rows_host[:fs_rows], dim, bits, metric, qt[:1], kk)
            out["cpu_baseline"]["single_thread_with_span_crc_parse_alloc"] = round(
                1.0 / fsecs * (fs_rows / float(n_rows)), 4)
            log("cpu_baseline leg: %.1f s" % (time.time() - t0))
        # parity on the FULL corpus for a few queries: ids identical, distances bit-equal
        nv = min(args.verify, len(qt), len(res_rows))   # (replicas: rank 0 holds the answers of its own slice)
        if nv > 0 and radius == 0 and n_rows <= 2_000_000:
            t0 = time.time()
            rows_all = rows_host if sample_rows == n_rows else ix.read_rows(0, n_rows)
            _, ref_rows = orc.bench_topk(rows_all, dim, bits, metric, qt[:nv], k, min(cores, nv))
            same = sum(int((ref_rows[i] == res_rows[i]).all()) for i in range(nv))
            recall = float(np.mean([len(set(ref_rows[i]) & set(res_rows[i])) / float(k)
                                    for i in range(nv)]))
            out["parity"] = {"queries_checked": nv, "ids_identical": same, "recall_at_k": recall}
            log("parity leg: %.1f s" % (time.time() - t0))
            if same != nv:
                log("bench.py: PARITY FAILURE: GPU ids differ from the oracle's")
                emit(out)
                sys.exit(1)
    if searcher is not None:
        searcher.close()   # detaches the communicator from the handle, then destroys it (ncclCommDestroy)
    ix.close()

    # ---- the other BASELINE configs, one roofline object each (N=1) ----------------
    if rank == 0 and world == 1 and not inproc and not args.no_extras and not args.rows:
        extras = {}
        for name in ("cfg2", "cfg3", "cfg4", "cfg5"):
            if name == args.workload:
                continue
            try:
                extras[name] = side_workload(name, 48 if name != "cfg5" else 24, devices)
            except Exception as e:  # a side leg never takes the headline line down with it
                extras[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        out["other_workloads"] = extras
        if args.workload == "headline":
            bq = {}
            for b in (64, 16, 8, 4):
                try:
                    bq["%dbit" % b] = batched_leg(b, n_rows, dim, metric, k, devices, SEED + 40 + b)
                except Exception as e:
                    bq["%dbit" % b] = {"error": "%s: %s" % (type(e).__name__, e)}
            out["batched_quantized"] = bq

    if rank == 0:
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
