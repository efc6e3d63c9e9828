"""pytest configuration.

Markers:
  gpu — needs a real MI355X: these are the parity tests proper; they call the gfx950 build of
        libsealhip.so through the C ABI and fail (not skip) if the HIP extension is missing.
Everything else runs on CPU: the oracle against the reference's known answers and the golden vectors,
the host logic, the C-ABI export check, and — for index arithmetic only — the same kernel sources
executed by the fiber emulator in tests/hipemu (never used for a parity or performance claim).
"""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

EMU_LIB = os.path.join(HERE, "hipemu", "libsealhip_emu.so")
GPU_LIB = os.path.join(ROOT, "seal_amd", "lib", "libsealhip.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


def _build(target):
    subprocess.check_call(["make", "-s", "-j8", target], cwd=os.path.join(ROOT, "seal_amd", "csrc"))


@pytest.fixture(scope="session")
def emu():
    """seal_amd bound to the fiber-emulated build of the kernel sources (CPU, test only)."""
    import seal_amd
    if not os.path.exists(EMU_LIB):
        _build("emu")
    os.environ["SEALHIP_COMM_NO_RCCL"] = "1"  # no device here: the library's communicator runs as a one-rank loopback
    seal_amd.load(EMU_LIB)
    yield seal_amd
    seal_amd._native._lib = None
    seal_amd._native._lib_path = None


@pytest.fixture(scope="session")
def gpu():
    """seal_amd bound to the real gfx950 library; fails loudly when it or the device is missing."""
    import seal_amd
    seal_amd.load(GPU_LIB)
    name, cus, mem = seal_amd.device_info()
    assert cus > 0
    # an abort of this process (a device memory fault makes the ROCm runtime call abort(); pytest keeps stderr in a file that dies
    # with the process) leaves the aborting thread's call stack behind: gpurun_out/abort_trace_<pid>.txt, written only then
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        seal_amd.install_abort_trace(os.path.join(out, "abort_trace_%d.txt" % os.getpid()))
    except Exception:
        pass
    return seal_amd
