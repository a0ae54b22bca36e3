"""Oracle self-checks: known-answer tests and finite-difference checks of every local Jacobian.

The reference ships no tests; these are the closed-form KATs of SURVEY.md §8(c)(4):
rest pose => zero constraint force, rigid-motion equivariance, FD of d(project)/dx (Triangle.cpp:354-451,
TriangleBending.cpp:154-172) and of calculatedri_dfi / calculatedri_dmu (Simulation.cpp:865-919).
"""
import numpy as np
import pytest

import meshes
import orc


@pytest.fixture(scope="module")
def small():
    V, F = meshes.grid_cloth(6, 5, 3.0, 2.5, "DOWN")
    o = orc.Oracle(V, F, h=1 / 90, density=0.3, k_stretch=150.0, k_bend=0.05, attachments=[0, 5]).build()
    return V, F, o


def test_counts_and_mass(small):
    V, F, o = small
    assert o.N == 30 and o.T == 2 * 5 * 4
    # interior edges of a (nx-1)x(ny-1) quad grid split into triangles: 3*qx*qy - qx - qy
    assert o.E == 3 * 5 * 4 - 5 - 4
    m, a, r = o.vertex_data()
    np.testing.assert_allclose(a.sum(), 3.0 * 2.5, rtol=1e-12)
    np.testing.assert_allclose(m, 0.3 * a, rtol=1e-14)
    assert (r > 0).all()


def test_P_is_spd_and_solve(small):
    V, F, o = small
    import scipy.sparse as sp
    ptr, col, val = o.P_csr()
    P = sp.csr_matrix((val, col, ptr), shape=(o.N, o.N))
    assert abs(P - P.T).max() < 1e-12
    w = np.linalg.eigvalsh(P.toarray())
    assert w.min() > 0
    rng = np.random.default_rng(0)
    rhs = rng.standard_normal(3 * o.N)
    sol = o.solveP(rhs)
    res = sp.kron(P, sp.identity(3)) @ sol - rhs
    assert np.abs(res).max() < 1e-10


def test_rest_pose_is_force_free(small):
    V, F, o = small
    x = V.reshape(-1).copy()
    # at rest F = [P, P_perp] is already an isometry: project(x) = w * vec(F), so A^T (p - A x) = 0
    for t in range(o.T):
        p = o.tri_project(t, x)
        J = o.tri_project_backward(t, x)
        assert np.isfinite(p).all() and np.isfinite(J).all()
        # |column| of an isometry is 1  ->  |p[:3]| = |p[3:]| = w
        np.testing.assert_allclose(np.linalg.norm(p[:3]), np.linalg.norm(p[3:]), rtol=1e-12)
        assert abs(p[:3] @ p[3:]) < 1e-10 * (p[:3] @ p[:3])
    idx, wv, n = o.bends()
    for e in range(o.E):
        p = o.bend_project(e, x)
        ee = (wv[e][:, None] * V[idx[e]]).sum(axis=0)
        # flat rest state: weighted sum has norm n (both ~0 for a planar grid)
        np.testing.assert_allclose(np.linalg.norm(ee), n[e], atol=1e-12)


def test_rigid_motion_equivariance(small):
    V, F, o = small
    rng = np.random.default_rng(1)
    x = (V + 0.05 * rng.standard_normal(V.shape))
    Q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(Q) < 0:
        Q[:, 0] *= -1
    xr = x @ Q.T + np.array([0.3, -2.0, 1.0])
    for t in range(0, o.T, 3):
        p = o.tri_project(t, x.reshape(-1)).reshape(2, 3)
        pr = o.tri_project(t, xr.reshape(-1)).reshape(2, 3)
        np.testing.assert_allclose(pr, p @ Q.T, atol=1e-10)
    for e in range(0, o.E, 3):
        p = o.bend_project(e, x.reshape(-1))
        pr = o.bend_project(e, xr.reshape(-1))
        np.testing.assert_allclose(pr, Q @ p, atol=1e-10)


def _fd_jac(fun, x, idx, eps=1e-6):
    cols = []
    for k in idx:
        xp = x.copy(); xp[k] += eps
        xm = x.copy(); xm[k] -= eps
        cols.append((fun(xp) - fun(xm)) / (2 * eps))
    return np.stack(cols, axis=1)


def test_triangle_jacobian_fd(small):
    V, F, o = small
    rng = np.random.default_rng(2)
    x = (V + 0.08 * rng.standard_normal(V.shape)).reshape(-1)
    for t in [0, 7, 19, 33]:
        vid = F[t]
        cols = [3 * v + d for v in vid for d in range(3)]
        Jfd = _fd_jac(lambda y: o.tri_project(t, y), x, cols)
        J = o.tri_project_backward(t, x)
        np.testing.assert_allclose(J, Jfd, atol=2e-6 * np.abs(Jfd).max())


def test_bending_jacobian_fd():
    # a non-flat rest shape so that the rest norm n is > 1e-6 and the constraint is active
    V, F = meshes.grid_cloth(5, 5, 2.0, 2.0, "DOWN")
    V = V.copy()
    V[:, 1] += 0.15 * np.sin(2.0 * V[:, 0]) * np.cos(1.5 * V[:, 2])
    o = orc.Oracle(V, F, k_bend=0.7).build()
    idx, wv, n = o.bends()
    assert (n > 1e-6).any()
    rng = np.random.default_rng(3)
    x = (V + 0.05 * rng.standard_normal(V.shape)).reshape(-1)
    checked = 0
    for e in range(o.E):
        if n[e] <= 1e-6:
            continue
        cols = [3 * v + d for v in idx[e] for d in range(3)]
        Jfd = _fd_jac(lambda y: o.bend_project(e, y), x, cols)
        J = o.bend_backward(e, x)
        np.testing.assert_allclose(J, Jfd, atol=2e-6 * max(np.abs(Jfd).max(), 1e-3))
        checked += 1
    assert checked > 5


@pytest.mark.parametrize("case", ["takeoff", "stick", "slide"])
def test_friction_branches_and_jacobian(case):
    n = np.array([0.2, 0.9, -0.3]); n /= np.linalg.norm(n)
    t1 = np.cross(n, [1.0, 0, 0]); t1 /= np.linalg.norm(t1)
    mu = 0.4
    if case == "takeoff":
        f = 2.0 * n + 0.5 * t1
    elif case == "stick":
        f = -2.0 * n + 0.5 * t1          # |f_T| = 0.5 <= mu*2 = 0.8
    else:
        f = -2.0 * n + 1.5 * t1          # |f_T| = 1.5 > 0.8
    r, ty, J, dmu = orc.friction(n, f, mu)
    if case == "takeoff":
        assert ty == 0 and np.allclose(r, 0) and np.allclose(J, 0)
    elif case == "stick":
        assert ty == 1
        np.testing.assert_allclose(r, -f, atol=1e-14)
        np.testing.assert_allclose(J, -np.eye(3), atol=1e-14)
        assert np.allclose(dmu, 0)
    else:
        assert ty == 2
        np.testing.assert_allclose(r, 2.0 * n - mu * 2.0 * t1, atol=1e-14)
    eps = 1e-6
    Jfd = np.zeros((3, 3))
    for k in range(3):
        fp = f.copy(); fp[k] += eps
        fm = f.copy(); fm[k] -= eps
        Jfd[:, k] = (orc.friction(n, fp, mu)[0] - orc.friction(n, fm, mu)[0]) / (2 * eps)
    np.testing.assert_allclose(J, Jfd, atol=1e-8)
    dfd = (orc.friction(n, f, mu + eps)[0] - orc.friction(n, f, mu - eps)[0]) / (2 * eps)
    np.testing.assert_allclose(dmu, dfd, atol=1e-8)


def _plane_contact_np(q, ul, ur):
    """Plane::isInContact (Primitive.cpp:66-130) re-stated with NumPy for a point q relative to the plane's centre."""
    eps, edge_tol = 0.4, 0.0005
    lr, ll = -ul, -ur
    if np.linalg.norm(q) > max(np.linalg.norm(ul), np.linalg.norm(ur)) + eps:
        return None
    n = np.cross(ur, ul); n /= np.linalg.norm(n)
    dist = n @ q
    if abs(dist) > eps:
        return None
    pp = q - n * (n @ q)

    def inside(a, b, c):
        AB, AC, AP = b - a, c - a, pp - a
        nn = np.cross(AB, AC); n2 = nn @ nn
        al, be = np.cross(AB, AP) @ nn / n2, np.cross(AP, AC) @ nn / n2
        ga = 1 - al - be
        return al >= 0 and be >= 0 and ga >= 0 and ga <= 1 and al <= 1 and be <= 1
    if inside(ul, ur, ll) or inside(ll, ur, lr):
        return n
    for a, b in ((ul, ur), (ur, lr), (ll, lr), (ul, ll)):
        AB = b - a
        P = a + AB * ((q - a) @ AB / (AB @ AB))
        t = np.linalg.norm(P - a) / np.linalg.norm(AB)
        if np.linalg.norm(P - b) > np.linalg.norm(AB):
            t = -t
        if np.linalg.norm(q - P) < edge_tol and -edge_tol < t < 1 + edge_tol:
            w = q - a if t < 0 else (q - b if t > 1 else q - P)
            return w / np.linalg.norm(w)
    return None


@pytest.mark.parametrize("kind", ["plane", "bowl"])
def test_plane_and_bowl_contact_sets_match_an_independent_restatement(kind):
    """The oracle's Plane / Bowl isInContact (orc_sim.cpp) against a NumPy restatement of Primitive.cpp:66-130 / :362-381 on a
    cloud of vertices around the obstacle (zero velocities: the three time samples of isInContactWithObstacle coincide)."""
    V, F = meshes.grid_cloth(17, 15, 3.0, 2.6, "DOWN")
    rng = np.random.default_rng(3)
    c = V.mean(axis=0) + np.array([0.1, -0.2, 0.05])
    if kind == "plane":
        ul, ur = np.array([-1.1, 0.25, -0.9]), np.array([1.1, 0.25, -0.9])
        X = V + np.array([0, 1, 0]) * rng.uniform(-0.9, 0.5, (len(V), 1)) + 0.02 * rng.standard_normal(V.shape)
        X[:6] = c + np.array([ur + (-ul - ur) * t for t in np.linspace(-0.0003, 1.0003, 6)]) + 1e-4 * rng.standard_normal((6, 3))   # on an edge
    else:
        R = 1.3
        dirs = rng.standard_normal(V.shape); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        X = c + dirs * (R + rng.choice([-0.02, -0.004, 0.0, 0.004, 0.02], (len(V), 1)))
    o = orc.Oracle(V, F, h=1 / 120, density=0.3, k_stretch=50.0, k_bend=0.01, fwd_tol=1e-3, bwd_tol=1e-3, selfcollision=False, pd_iter_cap=2)
    if kind == "plane":
        o.add_plane(c, ul, ur, 0.3)
    else:
        o.add_bowl(c, R, 0.3)
    o.build()
    out = o.step(X.reshape(-1), np.zeros(X.size))
    con = o.prim_contacts(out["id"])
    got = dict(zip(con["particle"].tolist(), con["normal"]))
    want = {}
    for i, p in enumerate(X):
        if kind == "plane":
            n = _plane_contact_np(p - c, ul, ur)
        else:
            d = np.linalg.norm(p - c)
            n = (c - p) / d if (d - R <= 0.005 and p[1] <= c[1] and d > R - 0.005) else None
        if n is not None:
            want[i] = n
    assert set(got) == set(want) and 10 < len(want) < len(X)
    for i in want:
        np.testing.assert_allclose(got[i], want[i], atol=1e-12)


def test_discretised_sphere_normals_are_face_normals_of_the_sphere_mesh():
    """Known answers for the oracle's discretized Sphere::isInContact (Primitive.cpp:230-253): vertices resting just above a radius-15
    sphere get, as contact normal, the normal of the mesh face whose prism holds them — worked out here a third time with numpy from the
    face table (barycentric weights of Primitive.h:176-190 in [0, 1]; the LAST such face in creation order whose swapped-weight projection is
    within one radius: that rules the antipodal face out). Piecewise constant: every vertex over one face gets the identical normal, a
    vertex over the neighbouring face a different one, and none is the radial direction."""
    import ctypes as C
    R, res = 15.0, 40
    c = np.array([-0.5, -16.0, 0.0])
    V, F = meshes.grid_cloth(9, 9, 3.2, 3.2, "DOWN")
    X = V.copy()
    X[:, 0] += c[0] - V[:, 0].mean() + 0.37; X[:, 2] += c[2] - V[:, 2].mean() - 0.21
    X[:, 1] = c[1] + np.sqrt(R * R - (X[:, 0] - c[0]) ** 2 - (X[:, 2] - c[2]) ** 2) + 0.03
    o = orc.Oracle(X, F, h=1 / 120, density=0.3, k_stretch=150.0, k_bend=0.05, selfcollision=False, gradient_clipping=False)
    o.add_discretized_sphere(c, R, 0.2, res)
    o.build()
    ref = o.step(X.reshape(-1), np.zeros(X.size))
    pc = o.prim_contacts(ref["id"])
    assert len(pc["particle"]) == len(X)                       # every vertex is within 0.1 of the surface
    L = orc.lib()
    L.orc_sphere_mesh.restype = C.c_int
    tab = np.zeros(12 * 2 * res * res)
    n = L.orc_sphere_mesh(C.c_double(R), C.c_int(res), tab.ctypes.data_as(C.POINTER(C.c_double)), C.c_int(tab.size // 12))
    tab = tab[:12 * n].reshape(n, 4, 3)
    p0, p1, p2, fn = tab[:, 0], tab[:, 1], tab[:, 2], tab[:, 3]
    AB, AC = p1 - p0, p2 - p0
    nn = np.cross(AB, AC); n2 = (nn * nn).sum(axis=1)
    want = np.zeros((len(X), 3)); face = -np.ones(len(X), dtype=int)
    for k, i in enumerate(pc["particle"]):
        q = X[i] - c                                           # (zero velocity: the three time samples coincide)
        AP = q - p0
        alpha = (np.cross(AB, AP) * nn).sum(axis=1) / n2
        beta = (np.cross(AP, AC) * nn).sum(axis=1) / n2
        gamma = 1 - alpha - beta
        inside = (alpha >= 0) & (beta >= 0) & (gamma >= 0) & (gamma <= 1) & (alpha <= 1) & (beta <= 1)
        proj = alpha[:, None] * p1 + beta[:, None] * p2 + gamma[:, None] * p0
        ok = inside & (np.linalg.norm(q - proj, axis=1) < R)
        hits = np.nonzero(ok)[0]
        assert len(hits) >= 1 and inside.sum() >= 2            # the face below the vertex and (at least) the antipodal one hold it in their prisms
        face[k] = hits[-1]
        want[k] = fn[hits[-1]]
    np.testing.assert_allclose(pc["normal"], want, rtol=0, atol=1e-15)
    radial = (X[pc["particle"]] - c) / np.linalg.norm(X[pc["particle"]] - c, axis=1)[:, None]
    assert ((pc["normal"] * radial).sum(axis=1) > 0.99).all() and np.abs(pc["normal"] - radial).max() > 1e-3
    assert 6 <= len(set(face.tolist())) <= 24                  # a 3.2 x 3.2 sheet on quads of 2.4 x 1.2, two faces each
    for f in set(face.tolist()):
        assert np.ptp(pc["normal"][face == f], axis=0).max() == 0.0
