"""Host-side table builders of the HIP kernels (packet-ELL matrix, element windows, RCM renumbering, explicit inverse of small systems), checked on the
CPU by a small C++ harness (tests/native/host_tables_check.cpp) against the plain constraint system."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_packet_window_and_rcm_tables(tmp_path):
    csrc = os.path.join(ROOT, "diffcloth_amd", "csrc")
    exe = str(tmp_path / "host_tables_check")
    srcs = [os.path.join(ROOT, "tests", "native", "host_tables_check.cpp")] + [os.path.join(csrc, f) for f in ("dc_system.cpp", "dc_windows.cpp", "dc_packets.cpp", "dc_dense.cpp")]
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", csrc, "-o", exe] + srcs)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0 and r.stdout.strip().endswith("ALL OK"), r.stdout + r.stderr


def test_deflation_builder_finds_the_lowest_eigenvectors(tmp_path):
    """csrc/dc_deflate.cpp on a synthetic badly graded strip (cells shrinking 100 x across the sheet): the Chebyshev-filtered subspace
    iteration returns orthonormal vectors whose eigen-residuals |A u - theta u| are small, (U^T A U)^-1 is consistent, in well under a
    second (tests/native/deflation_check.cpp)."""
    csrc = os.path.join(ROOT, "diffcloth_amd", "csrc")
    exe = str(tmp_path / "deflation_check")
    srcs = [os.path.join(ROOT, "tests", "native", "deflation_check.cpp")] + [os.path.join(csrc, f) for f in ("dc_system.cpp", "dc_deflate.cpp")]
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", csrc, "-o", exe] + srcs)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout + r.stderr
