"""GPU: behaviour at the edges of the C-ABI contract — call-order and range errors (status codes, no crash), the smallest
meshes, a state that is already a fixed point, capacity overflow of the self-contact list, and run-to-run determinism
(the kernels use no atomics: two contexts fed the same input must agree bit for bit)."""
import numpy as np
import pytest

import meshes
import orc
from diffcloth_amd import capi

pytestmark = pytest.mark.gpu


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def engine(V, F, att=(), prims=(), **params):
    e = capi.Engine(0)
    e.set_mesh(V, F)
    e.set_attachments(list(att))
    p = dict(time_step=1 / 120, density=0.3, k_stretch=150.0, k_bend=0.05, forward_tol=1e-9, backward_tol=1e-9, cg_rel_tol=1e-6,
             cg_max_iter=2000, gradient_clipping=0, selfcollision_enabled=0, adjoint_mode=1, adjoint_rel_tol=1e-8)
    p.update(params)
    e.set_params(**p)
    e.set_primitives(list(prims))
    e.build()
    return e


def test_call_order_and_range_errors_are_status_codes():
    V, F = meshes.grid_cloth(5, 5)
    V = f32(V)
    e = capi.Engine(0)
    with pytest.raises(capi.DcError):
        e.build()                                               # no mesh yet
    e.set_mesh(V, F)
    with pytest.raises(capi.DcError):
        e.set_attachments([0, 99]); e.build()                   # attachment vertex out of range
    e.set_attachments([])
    with pytest.raises(capi.DcError):
        e.set_params(time_step=0.0)                             # time_step must be > 0
    e.set_params(time_step=1 / 120)
    with pytest.raises(capi.DcError):
        e.set_primitives([dict(center=(0, 0, 0), radius=1.0)] * 9)      # more than 8 flattened primitives
    e.set_primitives([])
    e.build()
    e.B = 1
    with pytest.raises(capi.DcError, match="dc_alloc_batch has not been called"):
        e.lib.dc_sync(e.h); e.step_forward(0)
    with pytest.raises(capi.DcError):
        e.alloc_batch(0, 4)
    e.alloc_batch(2, 3)
    x = np.tile(V.reshape(-1), (2, 1))
    e.set_state(0, x, np.zeros_like(x))
    with pytest.raises(capi.DcError, match="out of range"):
        e.step_forward(3)                                       # would write slot 4 of a 3-step tape
    with pytest.raises(capi.DcError, match="out of range"):
        e.get_state(4)
    with pytest.raises(capi.DcError, match="slot 0 has no record"):
        e.step_backward(0, x, x)
    with pytest.raises(capi.DcError):
        e.rollout_backward(2, 3)                                # would run past slot 1
    with pytest.raises(ValueError):
        e.set_state(0, x[:1], x[:1])                            # wrong batch size caught by the wrapper
    # the context is still usable after the refusals
    st = e.step_forward(0)
    assert np.all(st["converged"] == 1)


@pytest.mark.parametrize("shape", [(2, 2), (3, 2)])
def test_smallest_meshes_match_the_oracle(shape):
    """2 x 2 vertices = two triangles and one bending flap; one rollout."""
    V, F = meshes.grid_cloth(shape[0], shape[1], 1.0, 1.0, "DOWN")
    V = f32(V)
    # the stopping rule |dx|_2 / N < tol is absolute: with N = 4..6 vertices 1e-9 lies below what fp32 positions resolve
    e = engine(V, F, forward_tol=1e-7)
    o = orc.Oracle(V, F, h=1 / 120, density=0.3, k_stretch=150.0, k_bend=0.05, fwd_tol=1e-7, bwd_tol=1e-9, selfcollision=False,
                   gradient_clipping=False)
    o.build()
    rng = np.random.default_rng(3)
    x = f32(V.reshape(-1) + 0.02 * rng.standard_normal(V.size)); v = f32(0.1 * rng.standard_normal(V.size))
    e.alloc_batch(1, 1)
    e.set_state(0, x[None], v[None])
    st = e.step_forward(0)
    ref = o.step(x, v)
    x1, v1 = e.get_state(1)
    assert st["converged"][0] == 1 and ref["converged"]
    np.testing.assert_allclose(x1[0], ref["x"], atol=2e-6)
    gx = f32(rng.standard_normal(V.size)); gv = f32(0.01 * rng.standard_normal(V.size))
    gb = e.step_backward(1, gx[None], gv[None])
    rb = o.step_backward(ref["id"], gx, gv, is_start=False, direct=True)
    assert np.linalg.norm(gb["dL_dx"][0] - rb["dL_dx"]) <= 1e-4 * np.linalg.norm(rb["dL_dx"])
    assert np.linalg.norm(gb["dL_dv"][0] - rb["dL_dv"]) <= 1e-4 * np.linalg.norm(rb["dL_dv"])


def test_rest_state_without_forces_is_a_fixed_point():
    """No gravity, no contact, zero velocity, rest shape: the step must return the input exactly after one PD iteration, and
    the backward step of a zero gradient is zero."""
    V, F = meshes.grid_cloth(12, 9, 3.0, 2.0, "DOWN")
    V = f32(V)
    e = engine(V, F, att=[0, 11], gravity_enabled=0, contact_enabled=0)
    e.alloc_batch(3, 1)
    x = np.tile(V.reshape(-1), (3, 1))
    e.set_state(0, x, np.zeros_like(x))
    st = e.step_forward(0, fixed_pts=np.tile(V[[0, 11]].reshape(-1), (3, 1)))
    x1, v1 = e.get_state(1)
    assert np.all(st["converged"] == 1) and np.all(st["pd_iters"] == 1) and np.all(st["prim_contacts"] == 0)
    np.testing.assert_allclose(x1, x, atol=1e-6)
    assert np.abs(v1).max() < 1e-4
    gb = e.step_backward(1, np.zeros_like(x), np.zeros_like(x))
    assert np.all(gb["dL_dx"] == 0) and np.all(gb["dL_dv"] == 0) and np.all(gb["adjoint_iters"] == 0)


def test_self_contact_list_overflow_fails_loudly():
    """More self contacts than max_self_contacts: the reference has no limit (Simulation.cpp:281-352), so a cut list is not the
    reference's step — the call reports DC_ERR_CAPACITY instead of truncating silently; with the default capacity (sized from
    the mesh) the same fold goes through and the list is complete."""
    V, F = meshes.grid_cloth(24, 24, 4.5, 4.5, "DOWN")
    V = f32(V)
    # fold the sheet onto itself: x -> |x| puts the two halves within the collision radii of each other
    X = V.copy()
    X[:, 0] = np.abs(X[:, 0] - X[:, 0].mean()) + 0.0
    X[:, 1] += np.where(V[:, 0] > V[:, 0].mean(), 0.02, 0.0)
    full = engine(V, F, selfcollision_enabled=1, forward_tol=1e-7)
    small = engine(V, F, selfcollision_enabled=1, forward_tol=1e-7, max_self_contacts=16)
    for e in (full, small):
        e.alloc_batch(1, 1)
        e.set_state(0, f32(X.reshape(-1))[None], np.zeros((1, X.size)))
    st = full.step_forward(0)
    sc = full.get_self_contacts(1, 0)
    assert st["self_contacts"][0] > 16 and st["self_overflow"][0] == 0 and sc["count"] == st["self_contacts"][0] == len(sc["pairs"])
    with pytest.raises(capi.DcError, match="self-contact list overflow"):
        small.step_forward(0)
    x1, _ = small.get_state(1)
    assert np.isfinite(x1).all()


def test_injected_record_with_a_vertex_twice_in_a_layer_is_refused():
    """dc_set_record applies the contacts of a layer in parallel: a record whose layer holds a vertex twice is not one contactSorting
    (Simulation.cpp:422-624) can produce — DC_ERR_INVALID; the same two contacts in successive layers are accepted."""
    V, F = meshes.grid_cloth(6, 6)
    V = f32(V)
    e = engine(V, F, selfcollision_enabled=1)
    e.alloc_batch(1, 1)
    x = V.reshape(-1)[None]
    e.set_state(0, x, np.zeros_like(x))
    n = np.tile([0.0, 1.0, 0.0], (2, 1))
    rec = dict(x=x, v=np.zeros_like(x), f=np.zeros_like(x), prim=-np.ones((1, e.N), dtype=np.int32), normal=np.zeros_like(x))
    bad = dict(pairs=[[3, 20], [3, 27]], layer=[0, 0], normal=n, d=np.zeros((2, 3)))
    with pytest.raises(capi.DcError, match="twice in self-contact layer 0"):
        e.set_record(1, self_contacts=[bad], **rec)
    e.set_record(1, self_contacts=[dict(bad, layer=[0, 1])], **rec)


@pytest.mark.parametrize("adjoint_mode", [1, 0])
def test_two_contexts_agree_bit_for_bit(adjoint_mode):
    """(This test found a missing barrier after the windowed adjoint operator: a fused multi-step sweep read a few entries
    of K u from the previous step's buffer — 7e-6 relative in the gradient, invisible at the parity tolerances.)"""
    V, F = meshes.grid_cloth(40, 40, 4.5, 4.5, "DOWN")
    V = f32(V)
    c = f32(meshes.sphere_scene_center(V, 2.0))
    rng = np.random.default_rng(9)
    B, K = 6, 4
    X0 = np.stack([f32(V.reshape(-1) + np.tile([rng.uniform(-0.3, 0.3), -0.05, rng.uniform(-0.3, 0.3)], len(V))) for _ in range(B)])
    mus = f32(rng.uniform(0.1, 0.9, (B, 1)))
    res = []
    for rep in range(2):
        e = engine(V, F, prims=[dict(kind=capi.DC_PRIM_SPHERE, group=0, center=c, radius=2.0, mu=0.5)], time_step=1 / 180, k_bend=1e-5,
                   forward_tol=1e-8, selfcollision_enabled=1, cg_rel_tol=1e-4, adjoint_rel_tol=1e-6, adjoint_mode=adjoint_mode)
        e.alloc_batch(B, K)
        e.set_mu(mus)
        e.set_state(0, X0, np.zeros_like(X0))
        e.rollout_forward(0, K)
        e.seed_gradient(K, None, 1e-3)
        e.rollout_backward(K, K)
        x, v = e.get_state(K)
        dx, dv, dmu = e.get_gradient()
        res.append((x, v, dx, dv, dmu))
        del e
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
    assert np.abs(res[0][2]).max() > 0


def test_two_contexts_agree_bit_for_bit_with_self_contacts():
    """Same check through self-collision detection, layering, the layered friction passes (LDS and global-memory versions)
    and their transposes: a folded sheet, fused sweeps."""
    V, F = meshes.grid_cloth(24, 24, 4.5, 4.5, "DOWN")
    V = f32(V)
    X = V.copy()
    X[:, 0] = np.abs(X[:, 0] - X[:, 0].mean())
    X[:, 1] += np.where(V[:, 0] > V[:, 0].mean(), 0.02, 0.0)
    rng = np.random.default_rng(10)
    B, K = 5, 3
    X0 = np.stack([f32(X.reshape(-1) + 0.001 * rng.standard_normal(X.size)) for _ in range(B)])
    V0 = np.stack([f32(0.05 * rng.standard_normal(X.size)) for _ in range(B)])
    res = []
    for rep in range(3):
        e = engine(V, F, selfcollision_enabled=1, forward_tol=1e-7, cg_rel_tol=1e-5, adjoint_rel_tol=1e-6)
        e.alloc_batch(B, K)
        e.set_state(0, X0, V0)
        e.rollout_forward(0, K)
        e.seed_gradient(K, None, 1e-3)
        e.rollout_backward(K, K)
        x, v = e.get_state(K)
        dx, dv, _ = e.get_gradient()
        nself = np.array([e.get_stats(s)[0]["self_contacts"] for s in range(1, K + 1)])
        res.append((x, v, dx, dv, nself))
        del e
    assert res[0][4].min() > 0, "every step of every rollout must see self contacts"
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert np.array_equal(a, b)
