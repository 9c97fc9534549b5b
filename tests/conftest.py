import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CRYO_HOST_TEST_HOOKS"] = "1"   # the host library build that exports cryo_host_set_codec_ops (codec double)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def codec():
    """A Codec handle on GPU 0.  No CPU fallback: fails loudly without the HIP library/GPU."""
    from pg_cryogen_amd import Codec
    c = Codec(0)
    yield c
    c.close()
