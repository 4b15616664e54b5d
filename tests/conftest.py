import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tidal-wave_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no binaries (they are git-ignored): build them once, exactly as __graft_entry__.build()
    # does (hipcc for the product library, gcc for the oracle, g++ for the node addon).  Building is not a fallback:
    # a failing build fails the run.
    if not os.path.exists(os.path.join(ROOT, "tidal-wave_amd", "libtwflow.so")):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand from oracle/farneback_oracle.c."""
    import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def golden():
    import json
    import oracle as O
    with open(os.path.join(GOLDEN, "expected_responses.json")) as f:
        cases = json.load(f)
    for c in cases.values():
        c["expect_img"] = O.read_pgm(os.path.join(GOLDEN, c["expect"]))
        c["target_img"] = O.read_pgm(os.path.join(GOLDEN, c["target"]))
    return cases


@pytest.fixture(scope="session")
def twflow():
    import twflow as T
    T.lib()
    return T


@pytest.fixture(scope="session")
def engine(twflow):
    """Default-parameter engine on GPU 0; fails loudly (no fallback) if the HIP library or GPU is missing."""
    e = twflow.Engine(0, twflow.default_params(), slots=2)
    yield e
    e.close()


def planar(a):
    """OpenCV interleaved [h,w,c] -> planar [c,h,w]."""
    return np.ascontiguousarray(np.moveaxis(a, 2, 0))


def interleaved(a):
    return np.ascontiguousarray(np.moveaxis(a, 0, 2))
