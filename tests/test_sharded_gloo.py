"""The N>1 path on CPU: two ranks over gloo exchange per-shard top-(k+1) lists
with one all-gather and merge them (syzgydb_amd/sharded.py).  The per-shard scan
is played by the oracle here (tests may use it); on GPUs it is ScanIndex."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, %(root)r)
import oracle as orc
from syzgydb_amd.sharded import ShardedSearcher, shard_range

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dim, bits, metric, n, k = 24, 8, 1, 3000, 10
rows = orc.synth_rows(11, 0, n, dim, bits)
Q = orc.synth_vectors(12, 0, 6, dim)
lo, hi = shard_range(n, rank, world)

def local_search(q, kk):
    R = np.full((q.shape[0], kk), np.iinfo(np.uint64).max, np.uint64)
    D = np.zeros((q.shape[0], kk)); C = np.zeros(q.shape[0], np.int32)
    for i in range(q.shape[0]):
        r, d, _ = orc.search_exact(rows[lo:hi], dim, bits, metric, q[i], k=kk)
        R[i, :len(r)] = r + lo; D[i, :len(r)] = d; C[i] = len(r)
    return R, D, C

s = ShardedSearcher(local_search)
r, d, c, hist = s.search(Q, k)
for i in range(Q.shape[0]):
    er, ed, _ = orc.search_exact(rows, dim, bits, metric, Q[i], k=k)
    assert [int(x) for x in r[i, :c[i]]] == [int(x) for x in er], (rank, i)
    assert (d[i, :c[i]] == ed).all()
r2, d2, c2, _ = s.search_stream(Q, k, 4)
assert (r2 == r).all() and (d2 == d).all() and (c2 == c).all()

# radius mode: counts + padded gather; order (incl. ties: 8-bit dim-2 corpus) is the reference's
def local_radius(q, radius, rows=rows, dim=dim, bits=bits, metric=metric):
    lo, hi = shard_range(rows.shape[0], rank, world)
    r, d, _ = orc.search_exact(rows[lo:hi], dim, bits, metric, q, radius=radius)
    return r + np.uint64(lo), d
for radius in (0.3, 0.42, 1e-9):
    r, d = s.search_radius(local_radius, Q[0], radius)
    er, ed, _ = orc.search_exact(rows, dim, bits, metric, Q[0], radius=radius)
    assert [int(x) for x in r] == [int(x) for x in er], (rank, radius, len(r), len(er))
    assert (d == ed).all()
trows = orc.synth_rows(13, 0, 500, 2, 4)   # many equal distances
tq = orc.synth_vectors(14, 0, 1, 2)[0]
def tie_radius(q, radius):
    return local_radius(q, radius, rows=trows, dim=2, bits=4, metric=0)
r, d = s.search_radius(tie_radius, tq, 0.8)
er, ed, _ = orc.search_exact(trows, 2, 4, 0, tq, radius=0.8)
assert len(er) > 50 and [int(x) for x in r] == [int(x) for x in er] and (d == ed).all()
dist.barrier()
if rank == 0:
    print("SHARDED_OK world=%%d" %% world)
dist.destroy_process_group()
'''


import pytest


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_sharded_search(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(29631 + world), str(script)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "SHARDED_OK world=%d" % world in p.stdout
