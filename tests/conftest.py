import os
import sys

import pytest

# the tests flip MM_* switches between runs of one process: the library must read them every time (mm_env.h)
os.environ.setdefault("MM_ENV_DYNAMIC", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import mm_oracle
    mm_oracle.lib()
    return mm_oracle


@pytest.fixture(scope="session")
def sm():
    """The product package; on a GPU box the HIP library must be present and is the only path."""
    import simd_minimizers_amd
    simd_minimizers_amd.lib()
    return simd_minimizers_amd


@pytest.fixture(scope="session")
def gpu(sm):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return sm.default_workspace(0)


@pytest.fixture(scope="session")
def exp_build(sm):
    """The cross-check kernels - the FASTA packers of rounds 2-4, the split path - and every MM_* switch that selects them
    exist in the EXPERIMENTS build of the library only (round 5): tests of them skip under the product
    and run in a child pytest that loads that build through MM_LIB_PATH
    (tests/test_gpu_round5.py::test_cross_checks_in_the_experiments_build)."""
    if not os.path.basename(sm.LIB_PATH).endswith("_exp.so"):
        pytest.skip("cross-check kernels live in the experiments build (run with MM_LIB_PATH=.../libsimd_minimizers_amd_exp.so)")
    return True
