import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "you-can-not-recommend_amd", "python"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no libycnr_als.so (build products are not tracked): build it once,
    as __graft_entry__.build() does, when hipcc is here.  Without it the tests that load the
    library fail with the loader's own message -- there is no CPU fallback to fall back to."""
    import shutil
    import subprocess
    csrc = os.path.join(ROOT, "you-can-not-recommend_amd", "csrc")
    if not os.path.exists(os.path.join(csrc, "libycnr_als.so")) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.run(["make", "-C", csrc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    addon = os.path.join(ROOT, "you-can-not-recommend_amd", "addon")
    if not os.path.exists(os.path.join(addon, "ycnr_als.node")) and os.path.exists("/usr/include/node/node_api.h"):
        subprocess.run(["make", "-C", addon], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    o.lib()
    return o
