"""CPU-side checks of the product's host logic: the constraint-system builder inside libdiffcloth_hip.so
(diffcloth_amd/csrc/dc_system.cpp) must produce the same tables as the fp64 oracle's independent restatement
of Simulation::initializePrefactoredMatrices (reference Simulation.cpp:2969-3059), and the library must export
every symbol include/diffcloth_hip.h declares. No compute call is made here (no GPU needed).
"""
import ctypes
import os
import re

import numpy as np
import pytest

import meshes
import orc
from diffcloth_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = capi.load_library()
    header = open(os.path.join(ROOT, "include", "diffcloth_hip.h")).read()
    declared = set(re.findall(r"\b(dc_[a-z0-9_]+)\s*\(", header))
    assert declared == set(capi.EXPORTED_SYMBOLS)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.dc_version()


def test_struct_layouts_match_header():
    # sizes implied by the C declarations (x86-64 SysV): guards the ctypes mirrors in capi.py
    assert ctypes.sizeof(capi.dc_primitive) == 4 + 4 + 24 + 24 + 24 + 8 + 8 + 8 + 8
    assert ctypes.sizeof(capi.dc_step_stats) == 28 and ctypes.sizeof(capi.dc_bwd_stats) == 40      # (+ residual_verified, round 4; + workgroups, round 6)
    assert ctypes.sizeof(capi.dc_record) == 12 * 8                                                  # twelve pointers (dc_set_record)
    assert ctypes.sizeof(capi.dc_params) == 8 * 5 + 24 + 16 + 16 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8


@pytest.mark.parametrize("renumber", [False, True])
@pytest.mark.parametrize("shape", [(6, 5), (13, 13), (4, 9)])
def test_system_tables_match_oracle(shape, renumber, monkeypatch):
    """`renumber`: the engine's internal bandwidth-reducing vertex renumbering (reverse Cuthill-McKee) must be
    invisible at the C-ABI: matrices and per-vertex tables come back in the caller's numbering."""
    monkeypatch.setenv("DC_RENUMBER", "1" if renumber else "0")
    nx, ny = shape
    V, F = meshes.grid_cloth(nx, ny, 4.5, 3.5, "DOWN")
    rng = np.random.default_rng(5)
    V = V + 0.03 * rng.standard_normal(V.shape)     # irregular rest shape: non-trivial cotan weights
    att = [0, nx - 1]
    prm = dict(h=1 / 120, density=0.27, k_stretch=321.0, k_bend=0.4, k_att=5000.0)
    o = orc.Oracle(V, F, attachments=att, **prm).build()
    e = capi.Engine(device=-1)       # host-only context: system building / inspection
    e.set_mesh(V, F)
    e.set_attachments(att)
    e.set_params(time_step=prm["h"], density=prm["density"], k_stretch=prm["k_stretch"], k_bend=prm["k_bend"], k_att=prm["k_att"])
    e.build()
    assert (e.N, e.T, e.E, e.Af) == (o.N, o.T, o.E, o.Af)
    assert e.rows == 6 * o.T + 3 * o.E + 3 * o.Af
    ptr, col, val = e.system_matrix()
    optr, ocol, oval = o.P_csr()
    np.testing.assert_array_equal(ptr, optr)
    np.testing.assert_array_equal(col, ocol)
    np.testing.assert_allclose(val, oval, rtol=1e-11, atol=1e-14)
    m, a, r = e.vertex_data()
    om, oa, orad = o.vertex_data()
    np.testing.assert_allclose(m, om, rtol=1e-13)
    np.testing.assert_allclose(a, oa, rtol=1e-13)
    np.testing.assert_allclose(r, orad, rtol=1e-13)


def test_host_only_context_refuses_compute():
    V, F = meshes.grid_cloth(4, 4)
    e = capi.Engine(device=-1)
    e.set_mesh(V, F)
    e.build()
    with pytest.raises(capi.DcError, match="no CPU compute path"):
        e.alloc_batch(1, 1)


def test_bad_inputs_are_rejected():
    e = capi.Engine(device=-1)
    with pytest.raises(capi.DcError):
        e.set_mesh(np.zeros((3, 3)), [[0, 1, 5]])            # index out of range
    V, F = meshes.grid_cloth(3, 3)
    F2 = np.vstack([F, F[:1], F[:1]])                        # an edge shared by three triangles
    with pytest.raises(capi.DcError, match="non-manifold"):
        e.set_mesh(V, F2)
    with pytest.raises(capi.DcError):
        e.build()                                            # no valid mesh


@pytest.mark.parametrize("mesh,orient,dim,max_bw", [("tshirt", "BACK", 6.0, 511), ("hat", "FRONT", 6.0, 511), ("sock", "FRONT", 5.0, 511),
                                                    ("dress", "FRONT", 8.0, 511), ("dress7k", "FRONT", 8.0, 511)])
def test_shipped_garments_get_the_resident_kernel_set(mesh, orient, dim, max_bw):
    """dc_get_layout on a host-only context (no GPU): every garment mesh of the reference is renumbered to a system-matrix bandwidth
    the 10-bit packet deltas can hold, and gets the packet matrix and the LDS element windows — not the global-memory fallbacks."""
    import scenes
    V, F = scenes.load_mesh(mesh)
    P, _, _ = scenes.normalise_model(V, orient, dim)
    e = capi.Engine(-1)
    e.set_mesh(P, F); e.set_attachments([0])
    e.set_params(time_step=1.0 / 120, density=0.2, k_stretch=800.0, k_bend=0.05)
    e.set_primitives([]); e.build()
    lay = e.layout()
    assert lay["renumbered"] and 0 < lay["bandwidth"] <= max_bw, lay
    assert lay["packet_kernel"] and lay["element_windows"] and lay["windows"] >= 1, lay
    V2, F2 = meshes.grid_cloth(100, 100, 4.5, 4.5, "DOWN")
    g = capi.Engine(-1)
    g.set_mesh(V2, F2); g.set_attachments([]); g.set_params(time_step=1.0 / 180, density=0.3, k_stretch=150.0, k_bend=1e-5)
    g.set_primitives([]); g.build()
    lg = g.layout()
    assert not lg["renumbered"] and lg["packet_kernel"] and lg["element_windows"] and lg["windows"] == 10, lg



def test_deflation_space_is_built_for_the_irregular_garment_only():
    """dc_get_deflation on a host-only context: the probe solve (Jacobi-PCG to 1e-4 on a smooth right-hand side) decides — the reference's
    7 742-vertex dress needs > 300 iterations and gets its 16 lowest eigenvectors, the 3 634-vertex dress, the T-shirt and the bench cloth need
    15 ... 40 and get none; forward_deflation = 0 switches it off, > 0 forces it (csrc/dc_deflate.h)."""
    import scenes
    got = {}
    for mesh, orient, dim in (("dress7k", "FRONT", 8.0), ("dress", "FRONT", 8.0), ("tshirt", "BACK", 6.0)):
        V, F = scenes.load_mesh(mesh)
        P, _, _ = scenes.normalise_model(V, orient, dim)
        top = np.argsort(-P[:, 1])[:6].tolist()
        e = capi.Engine(-1)
        e.set_mesh(P, F); e.set_attachments(top)
        e.set_params(time_step=1.0 / 120, density=0.2, k_stretch=800.0, k_bend=0.05)
        e.set_primitives([]); e.build()
        got[mesh] = e.deflation()
    print(got)
    assert got["dress7k"][0] == 16 and got["dress7k"][1] > 200
    assert got["dress"][0] == 0 and got["dress"][1] <= 80 and got["tshirt"][0] == 0 and got["tshirt"][1] <= 80
    # the hat (579 vertices, k_bend 120, two clips): its forward step uses the explicit inverse, the space is the adjoint's coarse level
    V, F = scenes.load_mesh("hat")
    P, _, _ = scenes.normalise_model(V, scenes.HAT["orientation"], scenes.HAT["cloth_dim"])
    e = capi.Engine(-1)
    e.set_mesh(P, F); e.set_attachments(scenes.HAT["attachments"])
    e.set_params(time_step=scenes.HAT["h"], density=scenes.HAT["density"], k_stretch=scenes.HAT["k_stretch"], k_bend=scenes.HAT["k_bend"])
    e.set_primitives([]); e.build()
    assert e.deflation()[0] == 16 and e.deflation()[1] > 80
    V2, F2 = meshes.grid_cloth(100, 100, 4.5, 4.5, "DOWN")
    for want, expect in ((-1, 0), (1, 16)):
        g = capi.Engine(-1)
        g.set_mesh(V2, F2); g.set_attachments([]); g.set_params(time_step=1.0 / 180, density=0.3, k_stretch=150.0, k_bend=1e-5, forward_deflation=want)
        g.set_primitives([]); g.build()
        assert g.deflation()[0] == expect and g.deflation()[1] <= 40


def test_deflation_switch_is_read_at_every_build_and_the_space_is_cached(monkeypatch):
    """One rule for device and host-only contexts (csrc/dc_engine.hip: deflation_want): DC_DEFLATION is read at EVERY dc_build, so a test can
    toggle it between builds of one context; and a rebuild that leaves the system matrix P unchanged (same mesh, clips, stiffnesses, time step)
    takes the eigenvectors from the context's cache instead of repeating the probe solve and the Chebyshev subspace iteration — a
    rebuildSystem per optimisation iterate of the reference's loops (Simulation.cpp:1820-1860) costs what the tables cost."""
    import time
    import scenes
    V, F = scenes.load_mesh("dress7k")
    P, _, _ = scenes.normalise_model(V, "FRONT", 8.0)
    top = np.argsort(-P[:, 1])[:6].tolist()
    par = dict(time_step=1.0 / 120, density=0.2, k_stretch=800.0, k_bend=0.05)
    e = capi.Engine(-1)
    e.set_mesh(P, F); e.set_attachments(top); e.set_params(**par); e.set_primitives([])
    monkeypatch.setenv("DC_DEFLATION", "0")
    e.build()
    assert e.deflation() == (0, 0)
    monkeypatch.delenv("DC_DEFLATION")

    def timed_build():
        t = time.perf_counter(); e.build()
        return time.perf_counter() - t
    t_first = timed_build()
    first = e.deflation()
    assert first[0] == 16 and first[1] > 200
    t_again = timed_build()
    assert e.deflation() == first
    e.set_params(forward_tol=1e-6, **par)                  # a parameter that does not enter P
    t_same_p = timed_build()
    assert e.deflation() == first
    e.set_params(**dict(par, k_stretch=900.0))             # P changes: the space is rebuilt
    t_new_p = timed_build()
    assert e.deflation()[0] == 16 and e.deflation() != first
    print(f"\nbuild with the eigen-solve {t_first:.2f} s / {t_new_p:.2f} s, from the cache {t_again:.2f} s / {t_same_p:.2f} s")
    assert max(t_again, t_same_p) < 0.5 * min(t_first, t_new_p)
