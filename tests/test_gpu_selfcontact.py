"""GPU parity of the self-collision path: detection (Simulation.cpp:281-352), contactSorting layering (:422-624),
layered self friction (:655-678) and its Jacobian in the adjoint (:713-760), against the fp64 oracle (which is
itself pinned on the reference's golden frames that contain self contacts, tests/test_golden_tshirt.py)."""
import numpy as np
import pytest

import meshes
import orc
from diffcloth_amd import capi

pytestmark = pytest.mark.gpu


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def folded_pair(nx=13, gap=0.12, mu=0.3, with_sphere=True, fwd_tol=1e-9, seed=0, chain=False):
    """A grid cloth folded onto itself along x = 0: the two halves lie `gap` apart, closer than r_a + r_b."""
    V, F = meshes.grid_cloth(nx, nx, 4.5, 4.5, "DOWN")
    V = f32(V)
    c = f32(meshes.sphere_scene_center(V, 2.0))
    kw = dict(h=1 / 180, density=0.3, k_stretch=150.0, k_bend=0.05)
    o = orc.Oracle(V, F, fwd_tol=fwd_tol, bwd_tol=1e-9, selfcollision=True, contact=True, gradient_clipping=False, **kw)
    e = capi.Engine(0)
    e.set_mesh(V, F)
    e.set_params(time_step=kw["h"], density=kw["density"], k_stretch=kw["k_stretch"], k_bend=kw["k_bend"], forward_tol=fwd_tol,
                 backward_tol=1e-9, cg_rel_tol=1e-6, cg_max_iter=2000, gradient_clipping=0, selfcollision_enabled=1,
                 adjoint_mode=1, adjoint_rel_tol=1e-7)
    if with_sphere:
        o.add_sphere(c, 2.0, mu)
        e.set_primitives([dict(kind=capi.DC_PRIM_SPHERE, group=0, center=c, radius=2.0, mu=mu)])
    o.build()
    e.build()
    rng = np.random.default_rng(seed)
    X = V.copy()
    right = X[:, 0] > 1e-9
    X[right, 0] = -X[right, 0] + (0.07 if chain else 0.0)      # mirror the right half onto the left half
    X[right, 1] += gap
    X[:, 1] -= 0.12                                             # into the contact band of the sphere
    X += 0.004 * rng.standard_normal(X.shape)
    vel = 0.05 * rng.standard_normal(X.shape)
    vel[right, 1] -= 0.5                                        # the upper sheet moves towards the lower one
    return V, F, o, e, f32(X.reshape(-1)), f32(vel.reshape(-1))


def oracle_layers(o, rid):
    sc = o.self_contacts(rid)
    return sorted((int(l), int(a), int(b)) for l, a, b in zip(sc["layer"], sc["p1"], sc["p2"])), sc


@pytest.mark.parametrize("chain", [False, True])
def test_detection_and_layering_match_contactSorting(chain):
    V, F, o, e, x0, v0 = folded_pair(chain=chain)
    ref = o.step(x0, v0)
    assert ref["nself"] > 20 and ref["nlayers"] >= (2 if chain else 1)     # lonely pairs all land in layer 0; chains take one layer per contact
    e.alloc_batch(1, 1)
    e.set_state(0, x0, v0)
    st = e.step_forward(0)
    got = e.get_self_contacts(1)
    assert st["self_contacts"][0] == ref["nself"] == got["count"]
    want, sc = oracle_layers(o, ref["id"])
    have = sorted((int(l), int(a), int(b)) for l, (a, b) in zip(got["layer"], got["pairs"]))
    assert have == want                                       # same pairs AND the same layer for every pair
    assert got["layers"] == ref["nlayers"]
    # no vertex twice in a layer
    for l in range(got["layers"]):
        ids = got["pairs"][got["layer"] == l].reshape(-1)
        assert len(ids) == len(set(ids.tolist()))
    nmap = {(int(a), int(b)): n for a, b, n in zip(sc["p1"], sc["p2"], sc["normal"])}
    for (a, b), n in zip(got["pairs"], got["normal"]):
        np.testing.assert_allclose(n, nmap[(int(a), int(b))], atol=5e-6)


def test_forward_and_backward_with_self_contacts():
    V, F, o, e, x0, v0 = folded_pair(chain=True, seed=3)
    ref = o.step(x0, v0)
    assert ref["converged"] and ref["nself"] > 20 and ref["nprim"] > 0
    e.alloc_batch(1, 1)
    e.set_state(0, x0, v0)
    st = e.step_forward(0)
    x1, v1 = e.get_state(1)
    dx = np.abs(x1[0] - ref["x"]).max()
    f, r = e.get_record(1)
    rf, rr = o.record_fr(ref["id"])
    er = np.linalg.norm(r[0] - rr) / np.linalg.norm(rr)
    print(f"\n[self contact fwd] pd gpu {st['pd_iters'][0]} ref {ref['iters']} contacts self {st['self_contacts'][0]} prim {st['prim_contacts'][0]} "
          f"max|dx| {dx:.2e} rel err r {er:.2e}")
    assert dx <= 4.5e-5 and er <= 2e-3
    rng = np.random.default_rng(5)
    gx = f32(rng.standard_normal(x0.size)); gv = f32(rng.standard_normal(x0.size) * 0.01)
    rb = o.step_backward(ref["id"], gx, gv, is_start=False, direct=True)
    gb = e.step_backward(1, gx, gv, is_start=False)
    ex = np.linalg.norm(gb["dL_dx"][0] - rb["dL_dx"]) / np.linalg.norm(rb["dL_dx"])
    ev = np.linalg.norm(gb["dL_dv"][0] - rb["dL_dv"]) / np.linalg.norm(rb["dL_dv"])
    print(f"[self contact bwd] bicgstab iters {gb['adjoint_iters'][0]} rel err dx {ex:.2e} dv {ev:.2e}")
    assert ex <= 1e-4 and ev <= 1e-4


def test_no_self_contacts_is_a_noop():
    V, F, o, e, x0, v0 = folded_pair()
    x = f32(V.reshape(-1)); v = np.zeros_like(x)
    ref = o.step(x, v)
    e.alloc_batch(1, 1)
    e.set_state(0, x, v)
    st = e.step_forward(0)
    assert ref["nself"] == 0 and st["self_contacts"][0] == 0 and e.get_self_contacts(1)["count"] == 0
    x1, _ = e.get_state(1)
    assert np.abs(x1[0] - ref["x"]).max() <= 4.5e-5


def test_fused_rollout_with_self_contacts_equals_stepwise_calls():
    """dc_rollout_forward runs all steps of a rollout (detection + layering + step) in one launch; per-step calls launch
    the detection kernel and the step kernel separately. Same code, same inputs: bitwise identical states and contacts."""
    V, F, o, e, x0, v0 = folded_pair(chain=True, seed=7)
    S = 4
    e.alloc_batch(2, S)
    X = np.stack([x0, f32(x0 + 0.001)]); Vv = np.stack([v0, v0])
    e.set_state(0, X, Vv)
    e.rollout_forward(0, S)
    xa, va = e.get_state(S)
    ca = [e.get_self_contacts(s + 1) for s in range(S)]
    for s in range(S):
        st = e.step_forward(s)
    xb, vb = e.get_state(S)
    np.testing.assert_array_equal(xa, xb)
    np.testing.assert_array_equal(va, vb)
    for s in range(S):
        cb = e.get_self_contacts(s + 1)
        assert ca[s]["count"] == cb["count"] and ca[s]["layers"] == cb["layers"]
        np.testing.assert_array_equal(ca[s]["pairs"], cb["pairs"])
    assert ca[0]["count"] > 20
