"""misti_amd/csrc/misti_multi.cpp built HOST-ONLY (g++, no GPU, no HIP runtime) against stand-ins for the single-device entry points
(tests/multi_host/stub_and_driver.cpp) and run under ThreadSanitizer and AddressSanitizer + UBSan: the persistent
worker dispatch, chain dealing, the per-object call lock, error hand-over from a worker that throws or fails, create / destroy cycles,
and the GATHERED device-resident form with D = 3 contexts on {0, 0, 0} through the RCCL double (tests/multi_host/fake_rccl.cpp,
-DFAKE_RCCL_HOST) - ragged and empty shards, NaN / -1 padding, every context's table complete (the same double runs
on the GPU in tests/test_gpu_multi.py).  GPU sanitizers are not available on the pool: this is where the host side gets them."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

SRC = [os.path.join(ROOT, "tests", "multi_host", "stub_and_driver.cpp"), os.path.join(ROOT, "misti_amd", "csrc", "misti_multi.cpp")]
FAKE = os.path.join(ROOT, "tests", "multi_host", "fake_rccl.cpp")
HIP_INCLUDE = "/opt/rocm/include"

pytestmark = pytest.mark.skipif(shutil.which("g++") is None or not os.path.exists(os.path.join(HIP_INCLUDE, "hip", "hip_runtime_api.h")),
                                reason="needs g++ and the HIP headers (types only; nothing of the runtime is linked)")


@pytest.fixture(scope="module")
def rccl_double(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl_host.so")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-DFAKE_RCCL_HOST", FAKE, "-o", out], check=True)
    return out


@pytest.mark.parametrize("sanitizer,env", [("thread", {"TSAN_OPTIONS": "halt_on_error=1 second_deadlock_stack=1"}),
                                           ("address,undefined", {"ASAN_OPTIONS": "detect_leaks=1", "UBSAN_OPTIONS": "halt_on_error=1"})],
                         ids=["tsan", "asan_ubsan"])
def test_multi_device_host_side_under_sanitizers(tmp_path, rccl_double, sanitizer, env):
    exe = str(tmp_path / "multi_host_driver")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=" + sanitizer, "-D__HIP_PLATFORM_AMD__", "-I" + HIP_INCLUDE]
                   + SRC + ["-o", exe, "-ldl", "-lpthread"], check=True)
    r = subprocess.run([exe, "150", rccl_double], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "0 failed checks" in r.stdout, r.stdout
    assert "Sanitizer" not in r.stderr, r.stderr[-4000:]


def test_lanes_host_side_under_sanitizers(tmp_path):
    """misti_amd/csrc/misti_lanes.cpp (ABI 6: the pool of contexts behind `misti_create_lanes`) built host-only against stand-ins for the single-context entry
    points and the HIP event calls (tests/multi_host/lanes_stub_and_driver.cpp), under AddressSanitizer + UBSan: round-robin and "an idle lane first" under
    MISTI_LANE_ANY, lane bounds, borrowed contexts, a context that fails in the middle of creation (everything made so far is released), no leak over 50 cycles."""
    exe = str(tmp_path / "lanes_host_driver")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-D__HIP_PLATFORM_AMD__", "-I" + HIP_INCLUDE,
                    os.path.join(ROOT, "tests", "multi_host", "lanes_stub_and_driver.cpp"), os.path.join(ROOT, "misti_amd", "csrc", "misti_lanes.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and "0 failed checks" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "Sanitizer" not in r.stderr, r.stderr[-3000:]
