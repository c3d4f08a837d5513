"""A short run of the structured differential fuzz (tools/gpu_fuzz.py: reduce / materialise / minimizers / quality masking / batched
compat face against the oracle, inputs built around lane, tile and chunk boundaries, forced launch geometries).  The long runs are
recorded under profiles/; this keeps a slice of it in every `-m gpu` run, with a seed that changes when asked to."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_structured_fuzz_slice():
    seed = os.environ.get("NTK_FUZZ_SEED", "11")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz.py"), "--seconds", os.environ.get("NTK_FUZZ_SECONDS", "12"),
                        "--seed", seed], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all equal to the oracle" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
