import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("ZJNI_DEBUG_LIVE_SWITCHES", "1")    # the library caches its ZJNI_* switches per process; tests flip them between calls (zj_env in zj_kernels.hip)

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def zj():
    """The product package (zstd-jni_amd/) with its HIP library built."""
    mod = entry.load_package()
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    return mod


@pytest.fixture(scope="session")
def oracle_ref():
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref/libzstd_ref.so not built (needs /root/reference at build time)")
    return ref


@pytest.fixture(scope="session")
def oracle_port():
    from oracle import port
    port.build()
    return port


GOLDEN = os.path.join(ROOT, "tests", "golden")
XML_SHA256_PREFIX = "0e82e54e695c1938"     # sha256 of the 5,345,280-byte Silesia xml


DOUBLED = {"xml-1x2.zst": "xml-1.zst", "xml-1-sizedx2.zst": "xml-1-sized.zst"}     # the reference's doubled streams = the frame twice (1.4 MB each: not stored a second time)


def golden(name):
    if name in DOUBLED:
        return golden(DOUBLED[name]) * 2
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()
