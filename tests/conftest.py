import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="session", autouse=True)
def _one_hip_runtime():
    """Test-order independence: PyTorch bundles its own HIP runtime, and whichever runtime initialises
    the GPU first owns it.  A test that dlopens a C front end (phaserotate.so -> libprot_hip.so) before
    any torch test ran would bind /opt/rocm's runtime, and every later torch.cuda call in the process
    then fails with "No HIP GPUs are available" (seen when test_host_frontends.py ran before
    test_gpu_parity.py).  Loading the library through the package first preloads torch's runtime
    (phaserotate.lv2_amd.__init__._share_hip_runtime_with_torch), whatever the order of the tests."""
    try:
        from __graft_entry__ import load_package
        load_package().load_library()
    except Exception:
        pass  # library not built: the tests that need it report that themselves
    yield


def xorshift_uniform(n, seed=1):
    """The documented synthetic input: 32-bit xorshift -> uniform [-0.5, 0.5) float32
    (SURVEY.md section 8d)."""
    import numpy as np
    out = np.empty(n, np.float32)
    s = np.uint32(0x9E3779B9 ^ seed)
    if s == 0:
        s = np.uint32(1)
    # vectorised in chunks: run 256 independent generators
    lanes = 256
    st = (np.arange(lanes, dtype=np.uint64) * 0x9E3779B1 + int(s)) & 0xFFFFFFFF
    st[st == 0] = 1
    st = st.astype(np.uint32)
    per = (n + lanes - 1) // lanes
    buf = np.empty((per, lanes), np.float32)
    for i in range(per):
        st ^= (st << np.uint32(13))
        st ^= (st >> np.uint32(17))
        st ^= (st << np.uint32(5))
        buf[i] = (st >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / (1 << 24)) - np.float32(0.5)
    return buf.reshape(-1)[:n].copy()


def harmonic_stack(n, rate=48000.0, f0=110.0, nh=32, peak=0.8):
    """Asymmetric 32-harmonic stack so the sweep has a real minimum."""
    import numpy as np
    t = np.arange(n, dtype=np.float64) / rate
    y = np.zeros(n)
    for k in range(1, nh + 1):
        y += np.cos(2 * np.pi * k * f0 * t + 0.3 * k) / k
    y *= peak / np.abs(y).max()
    return y.astype(np.float32)
