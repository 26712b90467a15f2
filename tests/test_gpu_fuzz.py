"""A short run of the differential fuzzer (scripts/fuzz_gpu.py): random shapes, masks,
tombstones, two-shard handles and tunables against the oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_short(seed):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_gpu.py"), "12", str(seed)],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "0 failures" in p.stdout


@pytest.mark.gpu
def test_collection_fuzz_short():
    """Stateful fuzz of the Collection mirror (add / re-add / remove / update / search / list)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_collection.py"), "10", "5"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "0 failures" in p.stdout
