"""GPU (-m gpu): a short run of the randomised parity sweep scripts/fuzz_gpu.py (random vocabulary / hidden /
batch / k / bias / seed patterns; fused vs oracle, unfused, shard + merge, bf16 fused vs unfused, exact bf16 vs
oracle -- all bit for bit).  The long sweeps (1000+ shapes, also under DAE_TOPK_LEAN / DAE_SAMPLE=strided / DAE_TOPK_THREADS=256) are
run by hand; this keeps a slice of them in every test run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_shapes_bit_exact(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_gpu.py"), "25", str(seed)],
                       capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-2000:]
    assert r.returncode == 0 and "25 cases, 0 bad" in r.stdout, tail
