"""The one-lane-per-block decoder (the large-batch decoder before the wave decoder got its batch path; kept for
comparison behind ZLZ4_DECOMP_LANE_MIN) through the same parity tests as the default decoder: a child pytest process
with ZLZ4_DECOMP_LANE_MIN=1 (the threshold is read once per process) runs every decompress / frame-decode test."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lane_decoder_passes_the_decoder_parity_tests(gpu):
    if os.environ.get("ZLZ4_DECOMP_LANE_MIN") == "1":
        pytest.skip("already inside the child run")
    # the knob exists only in the tuning build of the library (the shipped one reads no environment variable)
    tuning = os.path.join(ROOT, "zig-lz4_amd", "libzlz4_amd_tuning.so")
    assert os.path.exists(tuning), "make tuning"
    env = dict(os.environ, ZLZ4_DECOMP_LANE_MIN="1", ZLZ4_AMD_LIB=tuning)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu",
                        os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_frame.py"),
                        "-k", "decompress or batch_of or single_buffer or interop"],
                       env=env, capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout



def test_lane_copy_path_of_the_wave_decoder_passes_the_decoder_parity_tests(gpu):
    """The wave decoder moves short matches four bytes per lane only when a batch fills the chip (>= 6144 blocks), which
    the parity tests' small batches never do: a child pytest process with ZLZ4_DECOMP_SHORT=32 (tuning build) forces that
    path for every batch size and runs the decompress / frame-decode / fuzz tests through it."""
    if os.environ.get("ZLZ4_DECOMP_SHORT") == "32":
        pytest.skip("already inside the child run")
    tuning = os.path.join(ROOT, "zig-lz4_amd", "libzlz4_amd_tuning.so")
    assert os.path.exists(tuning), "make tuning"
    env = dict(os.environ, ZLZ4_DECOMP_SHORT="32", ZLZ4_AMD_LIB=tuning)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu",
                        os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_frame.py"),
                        os.path.join(ROOT, "tests", "test_gpu_fuzz.py"),
                        "-k", "decompress or batch_of or single_buffer or interop or fuzz or decoder"],
                       env=env, capture_output=True, text=True, cwd=ROOT, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
