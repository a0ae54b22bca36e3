"""GPU parity for the remaining analytic obstacles of the reference's isInContact family: the
finite plane (Plane::isInContact, Primitive.cpp:66-130 — the slope scenes) and the bowl (Bowl::isInContact,
Primitive.cpp:362-381 — the Y0PLANE scene), teacher-forced single steps against the fp64 oracle like test_gpu_parity.py."""
import numpy as np
import pytest

import meshes
import orc
import ledger
import records
from diffcloth_amd import capi

pytestmark = pytest.mark.gpu
H = 1.0 / 120


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def pair(V, F, prim_gpu, add_oracle, mu, k_bend=0.05):
    o = orc.Oracle(V, F, h=H, density=0.3, k_stretch=150.0, k_bend=k_bend, fwd_tol=1e-9, bwd_tol=1e-9, selfcollision=False,
                   gradient_clipping=False)
    add_oracle(o)
    o.build()
    e = capi.Engine(0)
    e.set_mesh(V, F)
    e.set_params(time_step=H, density=0.3, k_stretch=150.0, k_bend=k_bend, forward_tol=1e-9, backward_tol=1e-9, cg_rel_tol=1e-6,
                 cg_max_iter=2000, gradient_clipping=0, selfcollision_enabled=0, adjoint_mode=1, adjoint_rel_tol=1e-8)
    e.set_primitives([dict(group=0, mu=mu, **prim_gpu)])
    e.build()
    return o, e


def check_step(o, e, x, v, seed, min_contacts, some_free=True):
    e.alloc_batch(1, 1)
    e.set_state(0, x[None], v[None])
    st = e.step_forward(0)
    ref = o.step(x, v)
    x1, v1 = e.get_state(1)
    assert st["converged"][0] == 1 and ref["converged"]
    assert st["prim_contacts"][0] == ref["nprim"] >= min_contacts
    if some_free:
        assert ref["nprim"] < len(x) // 3, "some vertices must lie outside the obstacle's reach"
    grp, nrm = e.get_contacts(1)
    pc = o.prim_contacts(ref["id"])
    ids = np.nonzero(grp[0] >= 0)[0]
    order = np.argsort(pc["particle"])
    np.testing.assert_array_equal(ids, pc["particle"][order])                       # the same vertices are in contact
    np.testing.assert_allclose(nrm[0].reshape(-1, 3)[ids], pc["normal"][order], atol=2e-6)   # with the same normals
    dx = np.abs(x1[0] - ref["x"]).max()
    rng = np.random.default_rng(seed)
    gx = f32(rng.standard_normal(x.size)); gv = f32(0.01 * rng.standard_normal(x.size))
    gb = e.step_backward(1, gx[None], gv[None])
    rb = o.step_backward(ref["id"], gx, gv, is_start=False, direct=True)
    ex, ev, em = rel(gb["dL_dx"][0], rb["dL_dx"]), rel(gb["dL_dv"][0], rb["dL_dv"]), records.mu_err(gb["dL_dmu"][0], rb["dL_dmu"])
    # same record on both sides (tests/records.py): the oracle differentiates the engine's record; dL/dmu end to end is gated within the
    # oracle's own sensitivity to a float32 rounding of its x_new (see test_gpu_parity.py::test_backward_step_matches_oracle)
    o.override_record(ref["id"], x=f32(ref["x"]))
    sens = records.mu_err(o.step_backward(ref["id"], gx, gv, is_start=False, direct=True)["dL_dmu"], rb["dL_dmu"])
    records.oracle_adopts_gpu_record(o, ref["id"], e, 1, 0, x, x1[0], v1[0], e.get_record(1)[0][0], H, normals=nrm)
    rb3 = o.step_backward(ref["id"], gx, gv, is_start=False, direct=True)
    ea = max(rel(gb["dL_dx"][0], rb3["dL_dx"]), rel(gb["dL_dv"][0], rb3["dL_dv"]), records.mu_err(gb["dL_dmu"][0], rb3["dL_dmu"]))
    print(f"\n[primitive] contacts {ref['nprim']} PD iterations gpu {st['pd_iters'][0]} / oracle {ref['iters']}: max|dx| {dx:.2e}, "
          f"gradient rel err dx {ex:.2e} dv {ev:.2e} dmu {em:.2e}, dL/dmu gpu {gb['dL_dmu'][0, 0]:.6e} oracle {rb['dL_dmu'][0]:.6e}; same record (oracle adopts) {ea:.2e}; "
          f"dL/dmu sensitivity of the oracle to a float32 x_new {sens:.2e}")
    assert dx <= 5e-5
    assert ex <= 1e-4 and ev <= 1e-4
    assert ea <= 1e-4
    gate = max(1e-4, min(3 * sens, 1e-3))      # hard ceiling 1e-3 on dL/dmu end to end
    ledger.add("test_gpu_primitives.check_step", f"primitive-seed{seed}-contacts{ref['nprim']}", 0, em, sensitivity=sens, gate=gate, same_record_adopt=ea,
               note="dL_dmu (dx, dv gated flat 1e-4)")
    assert em <= gate
    return ref


@pytest.mark.parametrize("mu", [0.2, 0.7])
def test_finite_tilted_plane(mu):
    """A 3 x 3.06 rectangle tilted about x under a 4.5 x 4.5 cloth: vertices over the rectangle and within 0.4 of its plane
    are in contact, the rim of the cloth hangs free."""
    V, F = meshes.grid_cloth(22, 22, 4.5, 4.5, "DOWN")
    V = f32(V)
    c = f32(V.mean(axis=0) + np.array([0.13, -0.25, 0.07]))
    ul, ur = f32([-1.5, 0.3, -1.5]), f32([1.5, 0.3, -1.5])
    o, e = pair(V, F, dict(kind=capi.DC_PRIM_PLANE, center=c, top_offset=ul, corner2=ur, radius=0.0),
                lambda o: o.add_plane(c, ul, ur, mu), mu)
    x, v = f32(V.reshape(-1)), np.zeros(V.size)
    o.set(fwd_tol=1e-7); o.build()
    for _ in range(4):                                  # let the sheet start to slide down the slope
        out = o.step(x, v); x, v = f32(out["x"]), f32(out["v"])
    o.set(fwd_tol=1e-9); o.build()
    check_step(o, e, x, v, seed=21, min_contacts=50)


def test_bowl():
    """A sheet lying on the inside of the bowl's lower half (the shell is 0.01 thick: radius +- 0.005) and moving outwards."""
    V, F = meshes.grid_cloth(14, 14, 1.3, 1.3, "DOWN")
    V = f32(V)
    R = 1.2
    c = f32(np.array([V[:, 0].mean(), 0.0, V[:, 2].mean()]))
    X = V.copy()
    X[:, 1] = c[1] - np.sqrt(R * R - (X[:, 0] - c[0]) ** 2 - (X[:, 2] - c[2]) ** 2)
    X = f32(X)
    vel = np.zeros_like(X); vel[:, 1] = -0.3
    o, e = pair(X, F, dict(kind=capi.DC_PRIM_BOWL, center=c, radius=R), lambda o: o.add_bowl(c, R, 0.4), 0.4)
    check_step(o, e, f32(X.reshape(-1)), f32(vel.reshape(-1)), seed=22, min_contacts=100, some_free=False)


@pytest.mark.parametrize("mu", [0.0, 0.3])
def test_discretised_sphere(mu):
    """Sphere with discretized = true (the BIG_SPHERE scene of the reference: radius 15 at (-0.5, -16, 0), mu 0, Simulation.cpp:1905-1911): the
    contact normal of a vertex is the face normal of the sphere's own 40 x 40 mesh — the last face, in creation order, whose prism holds the
    sample point (Primitive.cpp:230-253) — piecewise constant instead of radial. A sheet lying on top of the sphere spans a dozen faces."""
    V, F = meshes.grid_cloth(22, 22, 4.5, 4.5, "DOWN")
    V = f32(V)
    R = 15.0
    c = np.array([-0.5, -16.0, 0.0])
    X = V.copy()
    X[:, 0] += c[0] - V[:, 0].mean() + 0.37; X[:, 2] += c[2] - V[:, 2].mean() - 0.21      # (off the mesh's symmetry lines)
    X[:, 1] = c[1] + np.sqrt(R * R - (X[:, 0] - c[0]) ** 2 - (X[:, 2] - c[2]) ** 2) + 0.02
    X = f32(X)
    vel = np.zeros_like(X); vel[:, 1] = -0.3; vel[:, 0] = 0.2
    o, e = pair(X, F, dict(kind=capi.DC_PRIM_SPHERE_DISCRETIZED, center=c, radius=R), lambda o: o.add_discretized_sphere(c, R, mu), mu)
    ref = check_step(o, e, f32(X.reshape(-1)), f32(vel.reshape(-1)), seed=23, min_contacts=400, some_free=False)
    pc = o.prim_contacts(ref["id"])
    radial = (X[pc["particle"]] - c) / np.linalg.norm(X[pc["particle"]] - c, axis=1)[:, None]
    faces = np.unique(np.round(pc["normal"], 9), axis=0)
    print(f"[discretised sphere] {len(pc['particle'])} contacts on {len(faces)} faces, largest angle between face normal and radial direction "
          f"{np.degrees(np.arccos(np.clip((pc['normal'] * radial).sum(axis=1), -1, 1))).max():.2f} degrees")
    assert 6 <= len(faces) <= 40 and np.abs(pc["normal"] - radial).max() > 1e-2
    assert ((pc["normal"] * radial).sum(axis=1) > 0.98).all()           # never the antipodal face
