import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


# The parity tests pair the HIP library with the oracle built for the SAME SPEC §5 block size; contexts made without an explicit
# block_envs would otherwise pick theirs from the env count (round 5). tests/test_gpu_block_geometry.py covers the other builds
# and the automatic choice.
os.environ.setdefault("SCG_BLOCK_ENVS", "256")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle_mod():
    import sc_oracle
    sc_oracle.build()
    return sc_oracle


@pytest.fixture(scope="session")
def pkg():
    import skill_chaining_with_graphs_amd as s
    return s
