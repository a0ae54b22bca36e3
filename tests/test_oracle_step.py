"""Oracle whole-step checks: an independent NumPy/SciPy restatement of one PD step
(Simulation.cpp:1043-1428) and central finite differences of the step against stepBackward
(Simulation.cpp:1455-1780) for dL/dx, dL/dv, dL/dx_fixed, dL/dmu, dL/dk, dL/ddensity.
"""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import meshes
import orc


def make_sphere_case(nx=9, tol=1e-13, mu=0.4, k_bend=0.05, attachments=(), selfcollision=False, density=0.3,
                     k_stretch=150.0, h=1 / 180):
    V, F = meshes.grid_cloth(nx, nx, 4.5, 4.5, "DOWN")
    o = orc.Oracle(V, F, h=h, density=density, k_stretch=k_stretch, k_bend=k_bend, fwd_tol=tol, bwd_tol=1e-13,
                   attachments=attachments, selfcollision=selfcollision, pd_iter_cap=20000, calc_atp=True)
    c = meshes.sphere_scene_center(V, 2.0)
    o.add_sphere(c, 2.0, mu)
    o.build()
    return V, F, o


def settle(o, V, steps, x_fixed=None):
    x = V.reshape(-1).copy()
    v = np.zeros_like(x)
    for s in range(steps):
        out = o.step(x, v, x_fixed, t_prev=s * o.params["h"])
        x, v = out["x"], out["v"]
    return x, v


# ---------------------------------------------------------------------------------------------
# independent single-step restatement with scipy (explicit A, P = M + h^2 A^T A, splu solves,
# batched numpy SVD for the triangle projection)
# ---------------------------------------------------------------------------------------------
def numpy_step(V, F, o, x_n, v_n, sphere_c, sphere_r, mu, iters):
    h = o.params["h"]; rho = o.params["density"]; ks = o.params["k_stretch"]; kb = o.params["k_bend"]
    N = V.shape[0]
    X0 = V[F[:, 0]]; E0 = V[F[:, 1]] - X0; E1 = V[F[:, 2]] - X0
    P0 = E0 / np.linalg.norm(E0, axis=1, keepdims=True)
    P1 = E1 - (E1 * P0).sum(1, keepdims=True) * P0
    P1 /= np.linalg.norm(P1, axis=1, keepdims=True)
    dUV = np.stack([np.stack([(P0 * E0).sum(1), (P0 * E1).sum(1)], 1), np.stack([(P1 * E0).sum(1), (P1 * E1).sum(1)], 1)], 1)
    D = np.linalg.inv(dUV)
    area = np.abs(np.linalg.det(dUV)) * 0.5
    wt = np.sqrt(area * ks)
    varea = np.zeros(N)
    for k in range(3):
        np.add.at(varea, F[:, k], area / 3)
    m = rho * varea
    bidx, bw, bn = o.bends()          # bending flap table (ordering/cotan weights checked in test_oracle_local)
    m_o, a_o, r_o = o.vertex_data()
    np.testing.assert_allclose(m, m_o, rtol=1e-12)
    A0 = np.zeros(len(bidx)); A1 = np.zeros(len(bidx))
    for e, q in enumerate(bidx):
        p = V[q]
        def heron(a, b, c):
            s = 0.5 * (a + b + c); return np.sqrt(s * (s - a) * (s - b) * (s - c))
        l01 = np.linalg.norm(p[1] - p[0]); l02 = np.linalg.norm(p[2] - p[0]); l03 = np.linalg.norm(p[3] - p[0])
        l12 = np.linalg.norm(p[1] - p[2]); l13 = np.linalg.norm(p[1] - p[3])
        A0[e] = heron(l01, l02, l12); A1[e] = heron(l01, l13, l03)
    wb = np.sqrt(kb * 3.0 / (A0 + A1))
    rows, cols, vals = [], [], []
    r = 0
    for t in range(len(F)):
        for i in range(2):
            for d in range(3):
                rows += [r + d + 3 * i] * 3
                cols += [3 * F[t, 0] + d, 3 * F[t, 1] + d, 3 * F[t, 2] + d]
                vals += [-wt[t] * (D[t, 0, i] + D[t, 1, i]), wt[t] * D[t, 0, i], wt[t] * D[t, 1, i]]
        r += 6
    for e in range(len(bidx)):
        for d in range(3):
            for i in range(4):
                rows.append(r + d); cols.append(3 * bidx[e, i] + d); vals.append(wb[e] * bw[e, i])
        r += 3
    A = sp.csr_matrix((vals, (rows, cols)), shape=(r, 3 * N))
    M = sp.diags(np.repeat(m, 3))
    C = (h * h) * (A.T @ A)
    Pm = (C + M).tocsc()
    lu = spla.splu(Pm)
    g = np.array(o.params["gravity"])
    f_ext = np.repeat(m, 3) * np.tile(g, N)
    s_n = x_n + h * v_n + h * h * f_ext / np.repeat(m, 3)
    x_now = s_n.copy(); v_now = (s_n - x_n) / h
    # contact set from x_n with the initial-guess velocity (first hit of t = 0, h/2, h)
    contacts = {}
    for i in range(N):
        for tt in (0.0, 0.5, 1.0):
            p = x_n[3 * i:3 * i + 3] + v_now[3 * i:3 * i + 3] * h * tt
            dist = np.linalg.norm(p - sphere_c) - sphere_r
            if dist < 0.1:
                contacts[i] = (p - sphere_c) / np.linalg.norm(p - sphere_c)
                break
    for it in range(iters):
        X = x_now.reshape(-1, 3)
        e0 = X[F[:, 1]] - X[F[:, 0]]; e1 = X[F[:, 2]] - X[F[:, 0]]
        Fm = np.stack([e0, e1], axis=2) @ D                     # T x 3 x 2
        q0 = Fm[:, :, 0] / np.linalg.norm(Fm[:, :, 0], axis=1, keepdims=True)
        q1 = Fm[:, :, 1] - (Fm[:, :, 1] * q0).sum(1, keepdims=True) * q0
        q1 /= np.linalg.norm(q1, axis=1, keepdims=True)
        Q = np.stack([q0, q1], axis=2)
        F2 = np.transpose(Q, (0, 2, 1)) @ Fm
        U, S, Vt = np.linalg.svd(F2)
        newF = Q @ (U @ Vt)
        p_tri = (np.concatenate([newF[:, :, 0], newF[:, :, 1]], axis=1) * wt[:, None]).reshape(-1)
        ee = (bw[:, :, None] * X[bidx]).sum(1)
        nrm = np.linalg.norm(ee, axis=1, keepdims=True)
        pb = np.where(bn[:, None] > 1e-6, ee / np.where(nrm > 0, nrm, 1) * bn[:, None], 0.0) * wb[:, None]
        p = np.concatenate([p_tri, pb.reshape(-1)])
        b = h * h * (A.T @ p) + M @ s_n
        b_t = (b - Pm @ x_n) / h
        f = b_t - C @ v_now
        rvec = np.zeros(3 * N)
        for i, n in contacts.items():
            d = f[3 * i:3 * i + 3]
            sd = d @ n
            if sd < 0:
                fN = sd * n; fT = d - fN
                ri = -fN
                if np.linalg.norm(fT) <= mu * abs(sd):
                    ri = ri - fT
                else:
                    ri = ri - mu * abs(sd) * fT / np.linalg.norm(fT)
                rvec[3 * i:3 * i + 3] = ri
        v_new = lu.solve(b_t + rvec)
        x_new = x_n + h * v_new
        x_now, v_now = x_new, v_new
    return x_now, v_now, len(contacts)


def test_numpy_cross_check_single_step():
    V, F, o = make_sphere_case(nx=8, tol=1e-30, mu=0.4)
    x0, v0 = settle_with_cap(o, V, 45, cap=400)
    K = 25
    o.set(cap=K)
    o.build()
    out = o.step(x0, v0)
    assert out["iters"] == K
    c = meshes.sphere_scene_center(V, 2.0)
    xn, vn, ncon = numpy_step(V, F, o, x0, v0, c, 2.0, 0.4, K)
    assert ncon == out["nprim"] and ncon > 5
    # with the cap reached the oracle reverts to its best iterate == last iterate for a contracting run
    np.testing.assert_allclose(out["x"], xn, atol=1e-10)
    np.testing.assert_allclose(out["v"], vn, atol=1e-8)


def settle_with_cap(o, V, steps, cap):
    o.set(cap=cap)
    o.build()
    return settle(o, V, steps)


# ---------------------------------------------------------------------------------------------
# finite differences of the whole step
# ---------------------------------------------------------------------------------------------
def _loss_setup(o, V, steps, att=()):
    xf = V.reshape(-1, 3)[list(att)].reshape(-1) + 0.02 if len(att) else None
    x0, v0 = settle(o, V, steps, xf)
    rng = np.random.default_rng(7)
    cx = rng.standard_normal(x0.size)
    cv = rng.standard_normal(x0.size) * 0.01
    return x0, v0, xf, cx, cv


def _L(o, x, v, xf, cx, cv, frozen=-1):
    # the reference differentiates with the contact set AND its normals held fixed (they are computed once per
    # step from x_n, Simulation.cpp:1254-1256), so the finite differences freeze them too
    out = o.step(x, v, xf, frozen=frozen)
    assert out["converged"]
    return cx @ out["x"] + cv @ out["v"], out


@pytest.mark.parametrize("mu", [0.05, 0.9])
def test_step_gradient_fd_sphere_contact(mu):
    V, F, o = make_sphere_case(nx=7, tol=1e-13, mu=mu, attachments=(0, 6))
    x0, v0, xf, cx, cv = _loss_setup(o, V, 45, att=(0, 6))
    L0, out = _L(o, x0, v0, xf, cx, cv)
    con = o.prim_contacts(out["id"])
    assert len(con["particle"]) > 3
    # reference convention (Simulation.cpp:1534,1608-1616): incoming dL_dx carries dL/dx_new + dL/dv_new / h
    h = o.params["h"]
    bw = o.step_backward(out["id"], cx + cv / h, cv, is_start=True, direct=True)
    rng = np.random.default_rng(11)
    eps = 1e-6
    for name, base, grad in (("x", x0, bw["dL_dx"]), ("v", v0, bw["dL_dv"]), ("xf", xf, bw["dL_dxfixed"])):
        for trial in range(3):
            d = rng.standard_normal(base.size)
            d /= np.linalg.norm(d)
            args = dict(x=x0, v=v0, xf=xf)
            args[name] = base + eps * d
            Lp, _ = _L(o, args["x"], args["v"], args["xf"], cx, cv, out["id"])
            args[name] = base - eps * d
            Lm, _ = _L(o, args["x"], args["v"], args["xf"], cx, cv, out["id"])
            fd = (Lp - Lm) / (2 * eps)
            an = grad @ d
            assert abs(fd - an) <= 2e-5 * max(abs(fd), abs(an), 1e-3), (name, fd, an)
    # dL/dmu
    o.set_mu(0, mu + eps); Lp, _ = _L(o, x0, v0, xf, cx, cv, out["id"])
    o.set_mu(0, mu - eps); Lm, _ = _L(o, x0, v0, xf, cx, cv, out["id"])
    o.set_mu(0, mu)
    fd = (Lp - Lm) / (2 * eps)
    assert abs(fd - bw["dL_dmu"][0]) <= 2e-5 * max(abs(fd), 1e-3), (fd, bw["dL_dmu"])
    types = set(con["type"].tolist())
    assert types & {1, 2}


def test_step_gradient_fd_params():
    V, F, o = make_sphere_case(nx=7, tol=1e-13, mu=0.3, k_bend=0.2)
    V2 = V.copy()
    x0, v0, xf, cx, cv = _loss_setup(o, V2, 45)
    L0, out = _L(o, x0, v0, xf, cx, cv)
    h = o.params["h"]
    bw = o.step_backward(out["id"], cx + cv / h, cv, is_start=True, direct=True)
    for key, gi, eps in (("k_stretch", 0, 1e-4), ("k_bend", 1, 1e-6)):
        base = o.params[key]
        o.set(**{key: base + eps}); o.build(); Lp, _ = _L(o, x0, v0, xf, cx, cv, out["id"])
        o.set(**{key: base - eps}); o.build(); Lm, _ = _L(o, x0, v0, xf, cx, cv, out["id"])
        o.set(**{key: base}); o.build()
        fd = (Lp - Lm) / (2 * eps)
        assert abs(fd - bw["dL_dk"][gi]) <= 1e-4 * max(abs(fd), 1e-3), (key, fd, bw["dL_dk"])
    base = o.params["density"]; eps = 1e-7
    o.set(density=base + eps); o.build(); Lp, _ = _L(o, x0, v0, xf, cx, cv, out["id"])
    o.set(density=base - eps); o.build(); Lm, _ = _L(o, x0, v0, xf, cx, cv, out["id"])
    o.set(density=base); o.build()
    fd = (Lp - Lm) / (2 * eps)
    assert abs(fd - bw["dL_ddensity"]) <= 1e-4 * max(abs(fd), 1e-3), (fd, bw["dL_ddensity"])


def test_iterative_adjoint_matches_direct():
    V, F, o = make_sphere_case(nx=7, tol=1e-12, mu=0.4)
    x0, v0, xf, cx, cv = _loss_setup(o, V, 45)
    L0, out = _L(o, x0, v0, xf, cx, cv)
    d = o.step_backward(out["id"], cx, cv, direct=True)
    it = o.step_backward(out["id"], cx, cv, direct=False)
    assert it["converged"] and not it["used_direct"] and it["iters"] > 1
    np.testing.assert_allclose(it["dL_dx"], d["dL_dx"], atol=1e-8 * np.abs(d["dL_dx"]).max())
    np.testing.assert_allclose(it["dL_dv"], d["dL_dv"], atol=1e-8 * np.abs(d["dL_dv"]).max())


def test_step_gradient_fd_force_field_and_wind_factors():
    """fillForces' per-vertex terms (Simulation.cpp:87-105) and their gradients (:1714-1760): wind with fall-off and a
    per-step factor, constant force field — dL_dfext_vec / dL_dwindtimestep / dL_dwind against central differences."""
    V, F, o = make_sphere_case(nx=7, tol=1e-13, mu=0.3, k_bend=0.2)
    x0, v0, xf, cx, cv = _loss_setup(o, V, 45)
    rng = np.random.default_rng(17)
    n3 = x0.size
    fall = np.repeat(rng.uniform(0.2, 1.0, n3 // 3), 3)
    field = 1e-3 * rng.standard_normal(n3)
    wdir = np.array([0.3, 0.1, 0.9]); wdir /= np.linalg.norm(wdir)
    h = o.params["h"]

    def run(factor, fld, frozen=-1, config=4, freq=9.0, phase=0.4, norm=0.02, w=wdir):
        o.set_wind(True, config, w, norm, freq, phase)
        o.set_force_extras(fall, fld, factor)
        return _L(o, x0, v0, xf, cx, cv, frozen)

    # per-step factor (WIND_FACTOR_PER_STEP) + force field
    L0, out = run(0.7, field)
    bw = o.step_backward(out["id"], cx + cv / h, cv, is_start=True, direct=True)
    eps = 1e-4
    fd = (run(0.7 + eps, field, out["id"])[0] - run(0.7 - eps, field, out["id"])[0]) / (2 * eps)
    assert abs(fd - bw["dL_dwindtimestep"]) <= 2e-4 * max(abs(fd), 1e-6), (fd, bw["dL_dwindtimestep"])   # (FD noise: L changes by 1e-10)
    for i in (5, 40, 100):
        e = np.zeros(n3); e[i] = 1e-5
        fd = (run(0.7, field + e, out["id"])[0] - run(0.7, field - e, out["id"])[0]) / 2e-5
        assert abs(fd - bw["dL_dfext_vec"][i]) <= 2e-4 * max(abs(fd), 1e-6), (i, fd, bw["dL_dfext_vec"][i])
    # sin wind with fall-off: the five wind parameters (force vector = direction * norm, frequency, phase)
    L0, out = run(1.0, None, config=3)
    bw = o.step_backward(out["id"], cx + cv / h, cv, is_start=True, direct=True)
    base = np.array([*(wdir * 0.02), 9.0, 0.4])
    for k, eps in ((0, 1e-6), (2, 1e-6), (3, 1e-4), (4, 1e-5)):
        vals = []
        for sgn in (1, -1):
            q = base.copy(); q[k] += sgn * eps
            n = np.linalg.norm(q[:3])
            vals.append(run(1.0, None, out["id"], config=3, freq=q[3], phase=q[4], norm=n, w=q[:3] / n)[0])
        fd = (vals[0] - vals[1]) / (2 * eps)
        assert abs(fd - bw["dL_dwind"][k]) <= 2e-3 * max(abs(fd), 1e-7), (k, fd, bw["dL_dwind"])     # (FD noise on 1e-6-sized slopes)
    o.set_wind(False, 0, wdir, 0.0, 0.0, 0.0)
    o.set_force_extras(None, None, 1.0)
