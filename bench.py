#!/usr/bin/env python3
"""bench.py -- queries/sec of the exact (brute-force) scan behind Collection.Search.

    python bench.py --gpus N --steps K --warmup W

A "step" is one exact top-k query over the whole synthetic corpus (one sweep of
the packed rows through the fused HIP scan, rerank and result assembly
included).  Default workload = BASELINE.json's headline: 1M x 768 float32,
cosine (angular) distance, k=10, one query per sweep.  With N>1 (launched by
torch.distributed.run, one rank per GPU) the corpus is sharded by rows over the
ranks, every query goes to every rank and the per-shard top-k lists are merged
after one RCCL all-gather per micro-batch ("scaling": "strong").

Rank 0 prints ONE JSON line: metric/value/unit/..., plus
  "roofline":     HBM roofline of the fused scan kernel, from HIP events recorded
                  on the library's scan stream inside the timed region;
  "cpu_baseline": the CPU oracle (a C port of the reference's Go scan; there is
                  no Go toolchain in this image) timed on this box's host cores on
                  a bounded sample of the same workload (N=1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: (rows, dim, bits, metric, k, radius)   metric 0 = Euclidean, 1 = Cosine
    "headline": (1_000_000, 768, 32, 1, 10, 0.0),      # BASELINE.json metric / north_star target
    "cfg2": (1_000_000, 384, 32, 1, 10, 0.0),
    "cfg3": (1_000_000, 768, 8, 1, 10, 0.0),
    "cfg4": (10_000_000, 768, 32, 0, 100, 0.0),
    "cfg5": (100_000_000, 384, 4, 1, 0, 0.42),
    "plumbing": (10_000, 128, 32, 1, 10, 0.0),
}
SEED = 0x53595A4700000000


# Libraries print to fd 1 (RCCL's version banner on communicator creation, for one): keep the real
# stdout for the JSON line and point fd 1 at stderr for everything else.
_REAL_STDOUT = os.fdopen(os.dup(1), "w")
os.dup2(2, 1)
sys.stdout = sys.stderr


def log(*a):
    print(*a, file=sys.stderr, flush=True)


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


def traffic_for(workload, rows_override, sweeps_per_launch):
    """HBM bytes per scan launch from the PMC passes recorded in profiles/traffic.json
    (FETCH_SIZE with the gfx950 x2 correction + WRITE_SIZE), scaled from the sweeps per
    launch of the PMC run to this run's; None when no pass exists for this exact workload."""
    if rows_override:
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            e = json.load(f)[workload]
        return int(e["hbm_bytes_per_launch"] / float(e.get("sweeps_per_launch", 1)) * sweeps_per_launch)
    except Exception:
        return None


def emit(obj):
    """The ONE JSON line, on the process's real stdout."""
    _REAL_STDOUT.write(json.dumps(obj) + "\n")
    _REAL_STDOUT.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap.add_argument("--rows", type=int, default=0, help="override the workload's row count")
    ap.add_argument("--exchange-every", type=int, default=256,
                    help="N>1: queries per all-gather micro-batch")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--settle-seconds", type=float, default=2.0,
                    help="untimed sweeps before the warm-up steps (clock / TLB settling)")
    ap.add_argument("--no-cpu", action="store_true", help="skip cpu_baseline and the recall check")
    ap.add_argument("--verify", type=int, default=4, help="queries re-checked against the oracle")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            log("bench.py: --gpus %d needs torch.distributed.run (WORLD_SIZE=%d)" % (args.gpus, world))
            sys.exit(2)
        args.gpus = world

    from syzgydb_amd import ScanIndex
    from syzgydb_amd.synth import synth_vectors
    from syzgydb_amd.sharded import ShardedSearcher, shard_range

    n_rows, dim, bits, metric, k, radius = WORKLOADS[args.workload]
    if args.rows:
        n_rows = args.rows
    if radius > 0:
        log("bench.py: radius workloads are exercised by tests; timing the top-k form with k=10")
        k = 10
    seed = SEED + sorted(WORKLOADS).index(args.workload)

    dist = None
    torch = None
    # rehearsal knobs (1-GPU box): SZG_BENCH_BACKEND=gloo exchanges on the CPU,
    # SZG_BENCH_ONE_GPU=1 puts every rank's shard on device 0
    backend = os.environ.get("SZG_BENCH_BACKEND", "nccl")
    if os.environ.get("SZG_BENCH_ONE_GPU") == "1":
        local_rank = 0
    # SZG_BENCH_FORCE_DIST=1: take the N>1 code path (process group, all-gather, merge)
    # with a single rank -- lets a 1-GPU box exercise the RCCL plumbing
    dist_path = world > 1 or os.environ.get("SZG_BENCH_FORCE_DIST") == "1"
    if dist_path:
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        try:
            import torch  # only for torch.cuda.synchronize() around the timed region
        except Exception:  # pragma: no cover
            torch = None

    lo, hi = shard_range(n_rows, rank, world)
    ix = ScanIndex(dim, bits, metric, devices=[local_rank])
    t0 = time.time()
    ix.synth(hi - lo, seed, first_row=lo)  # corpus generated in HBM, reference encoding rules
    ix.set_row_base(lo)
    log("[rank %d] corpus rows [%d, %d) x %d B resident in %.2f s" % (rank, lo, hi, ix.row_bytes,
                                                                     time.time() - t0))
    queries = synth_vectors(seed + 1, 0, args.warmup + args.steps, dim)
    qw, qt = queries[: args.warmup], queries[args.warmup:]

    if dist_path:
        searcher = ShardedSearcher(lambda q, kk: ix.search_topk(q, kk),
                                   device=torch.device("cuda", local_rank) if backend == "nccl" else None)

        def run(q):
            r, d, _, _ = searcher.search_stream(q, k, args.exchange_every)
            return r, d
    else:
        def run(q):
            r, d, _ = ix.search_topk(q, k)
            return r, d

    def sync():
        if torch is not None and backend == "nccl" and torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # the headline is one query per sweep (HBM roofline); the shared multi-query
    # sweep is measured separately below and reported under "batched"
    ix.set_option("multi_query", 0)
    # settle the card first (clocks, TLBs, allocator): a fresh process measures 3-5 % low for
    # its first second or two of sweeps.  Untimed, outside the W warm-up steps, same on every rank.
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < args.settle_seconds:
        ix.search_topk(qt[:64], k)
    if args.warmup:
        run(qw)
    ix.set_timing(True)
    ix.reset_stats()
    sync()
    t0 = time.perf_counter()
    res_rows, res_dist = run(qt)  # returns when every result is on the host
    sync()
    elapsed = time.perf_counter() - t0
    stats = ix.stats()
    ix.set_timing(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        qps = args.steps / elapsed
        scan_ms = stats["scan_ms"] / max(stats["timed_launches"], 1)
        bytes_per_launch = stats["scan_bytes"] / max(stats["scan_launches"], 1)
        achieved = bytes_per_launch / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        sweeps_per_launch = bytes_per_launch / float(max((hi - lo) * ix.row_bytes, 1))
        out = {
            "metric": "queries/sec, exact scan 1M x 768 cosine k=10" if args.workload == "headline"
            else "queries/sec, exact scan (%s)" % args.workload,
            "value": round(qps, 2),
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 5),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": {4: "u4", 8: "u8", 16: "u16", 32: "f32", 64: "f64"}[bits],
            "data": "synthetic",
            "config": {
                "workload": "%s: %d x %d, %d-bit, %s, k=%d, 1 query per sweep" % (
                    args.workload, n_rows, dim, bits, "cosine" if metric else "euclidean", k),
                "rows": n_rows, "dim": dim, "quantization": bits,
                "distance": "cosine" if metric else "euclidean", "k": k,
                "queries_per_sweep": 1,
                "parallelism": "rows sharded over %d GPU(s)%s" % (
                    world, ", 1 all-gather per %d queries" % args.exchange_every if world > 1 else ""),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic_for(args.workload, args.rows, sweeps_per_launch) if world == 1 else None,
                "kernel": "szg::scan_kernel<%d,%d,...>" % (bits, metric),
                "bytes_per_launch": int(bytes_per_launch),
                "sweeps_per_launch": round(sweeps_per_launch, 2),
                "avg_launch_ms": round(scan_ms, 5),
                "launches": int(stats["timed_launches"]),
            },
            "escalations": int(stats["escalations"]),
            "full_replays": int(stats["full_replays"]),
        }

    # ---- shared multi-query sweep (MFMA), 32-bit cosine only, N=1 -----------------
    if world == 1 and bits == 32 and metric == 1 and args.steps >= 64:
        ix.set_option("multi_query", 1)
        # untimed warm-up with the same call: the first full-size shared-sweep call after
        # allocation runs ~1.7x slower (buffer first use); steady state is what is reported
        ix.search_topk(qt, k)
        t0 = time.perf_counter()
        b_rows, _, _ = ix.search_topk(qt, k)          # throughput: no per-kernel events
        b_elapsed = time.perf_counter() - t0
        ix.set_timing(True)
        ix.reset_stats()
        ix.search_topk(qt, k)                         # same call again with HIP events: per-sweep time
        bst = ix.stats()
        ix.set_timing(False)
        ix.set_option("multi_query", 0)
        sweep_ms = bst["scan_ms"] / max(bst["timed_launches"], 1)
        qps_b = args.steps / b_elapsed
        per_sweep = bst["mq_queries"] / max(bst["mq_launches"], 1)
        flops = 2.0 * n_rows * dim * per_sweep
        out["batched"] = {
            "queries_per_sweep": round(per_sweep, 2),
            "value": round(qps_b, 1), "unit": "queries/s",
            "kernel": "szg::mq_score_kernel<3,32,cosine,collect> (v_mfma_f32_16x16x4_f32)",
            "avg_sweep_ms": round(sweep_ms, 5),
            "hbm_GBps": round(n_rows * ix.row_bytes / (sweep_ms * 1e-3) / 1e9, 1),
            "mfma_TFLOPs": round(flops / (sweep_ms * 1e-3) / 1e12, 2),
            "mfma_peak_TFLOPs": 157.3,
            "ids_identical_to_single_query_path": bool((b_rows == res_rows).all()),
        }

    # ---- recall / parity spot check + CPU baseline (rank 0, N=1) -----------------
    if rank == 0 and world == 1 and not args.no_cpu:
        import oracle as orc
        orc.build()
        cores = host_cores()
        t0 = time.time()
        sample_rows = min(n_rows, 100_000)
        rows_host = ix.read_rows(0, sample_rows)  # the same bytes the GPU scans
        # single-thread rate first, to size the all-core sample
        secs1, _ = orc.bench_topk(rows_host, dim, bits, metric, qt[:1], k, 1)
        per_query = max(secs1, 1e-6)
        nq = int(max(cores, min(len(qt), cores * args.cpu_seconds / per_query)))
        nq = min(nq, len(qt))
        secs, cpu_rows = orc.bench_topk(rows_host, dim, bits, metric, qt[:nq], k, cores)
        scale = sample_rows / float(n_rows)
        out["cpu_baseline"] = {
            "value": round(nq / secs * scale, 3),
            "unit": "queries/s",
            "cores": cores,
            "kind": "port",
            "sample": "%d queries x first %d of %d rows on %d threads (%.1f s), scaled linearly to "
                      "%d rows; C port of the reference's Go scan (no Go toolchain here)" % (
                          nq, sample_rows, n_rows, cores, secs, n_rows),
            "single_thread": round(1.0 / per_query * scale, 4),
        }
        # the reference also CRCs and re-parses each record's span and allocates a decode
        # buffer on every visit (spanfile.go:757, collection.go:769): one query with that
        # per-record work included, single thread, on a smaller sample
        fs_rows = min(sample_rows, 20_000)
        fsecs, frows = orc.bench_topk_faithful(rows_host[:fs_rows], dim, bits, metric, qt[:1], k)
        out["cpu_baseline"]["single_thread_with_span_crc_parse_alloc"] = round(
            1.0 / fsecs * (fs_rows / float(n_rows)), 4)
        log("cpu_baseline leg: %.1f s" % (time.time() - t0))
        # parity on the FULL corpus for a few queries: ids identical, distances bit-equal
        nv = min(args.verify, len(qt))
        if nv > 0:
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

    if rank == 0:
        emit(out)
    ix.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
