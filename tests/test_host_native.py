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


def test_discretised_sphere_face_table_is_the_reference_mesh(tmp_path):
    """DC_PRIM_SPHERE_DISCRETIZED: the engine's face table (csrc/dc_spheremesh.cpp, written from the structure of the mesh) against the oracle's
    loop-by-loop restatement of Sphere::Sphere (Primitive.cpp:133-216): same faces, same ORDER (the contact code keeps the last face that
    qualifies), same corner order (it fixes the sign of the normal), bit-equal coordinates. A closed surface with outward normals."""
    import ctypes as C
    import numpy as np
    import orc
    csrc = os.path.join(ROOT, "diffcloth_amd", "csrc")
    exe = str(tmp_path / "spheremesh_dump")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", csrc, "-o", exe, os.path.join(ROOT, "tests", "native", "spheremesh_dump.cpp"), os.path.join(csrc, "dc_spheremesh.cpp")])
    L = orc.lib()
    L.orc_sphere_mesh.restype = C.c_int
    for radius, res in ((15.0, 40), (2.5, 7)):
        got = np.frombuffer(subprocess.run([exe, repr(radius), str(res)], capture_output=True, timeout=60, check=True).stdout, dtype=np.float64).reshape(-1, 12)
        want = np.zeros(12 * (2 * res * res + 16))
        n = L.orc_sphere_mesh(C.c_double(radius), C.c_int(res), want.ctypes.data_as(C.POINTER(C.c_double)), C.c_int(want.size // 12))
        want = want[:12 * n].reshape(n, 12)
        assert got.shape == want.shape and n == 2 * res * (res - 1)
        np.testing.assert_array_equal(got[:, :9], want[:, :9])
        np.testing.assert_allclose(got[:, 9:], want[:, 9:], rtol=0, atol=1e-15)
        cen = got[:, :9].reshape(n, 3, 3).mean(axis=1)
        assert ((got[:, 9:] * cen).sum(axis=1) > 0.9 * np.linalg.norm(cen, axis=1)).all()
