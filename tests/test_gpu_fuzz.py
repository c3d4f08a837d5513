"""The structured differential fuzz (tools/gpu_fuzz.py: reduce / materialise / minimizers / quality masking / batched compat face against
the oracle, inputs built around lane, tile and chunk boundaries; a third of the iterations under forced launch geometries - few blocks, so
that every wave runs many tiles, the condition under which round 5's lost "scc" clobber showed) as part of every `-m gpu` run: 60 s over
all paths + 30 s on the fused minimizer builds (the seed changes when asked to).  Longer runs are recorded under profiles/."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_structured_fuzz_slice():
    seed = os.environ.get("NTK_FUZZ_SEED", "11")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz.py"), "--seconds", os.environ.get("NTK_FUZZ_SECONDS", "60"),
                        "--seed", seed], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all equal to the oracle" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.gpu
def test_structured_fuzz_minimizers():
    seed = os.environ.get("NTK_FUZZ_SEED", "12")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz.py"), "--seconds", os.environ.get("NTK_FUZZ_MIN_SECONDS", "30"),
                        "--seed", seed, "--minimizers-only"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all equal to the oracle" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
