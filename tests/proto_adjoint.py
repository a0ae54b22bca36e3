"""Solver prototypes for the adjoint system K u = g (TEST INFRASTRUCTURE, CPU only: uses the fp64 oracle).

K = P - dP^T of a recorded step comes from the oracle as an explicit sparse matrix; the fp32 arithmetic of the device is emulated
with numpy float32 (matrix entries rounded to fp32 = the coefficient rounding of the matrix-free fp32 operator). Used to choose
the mixed-precision scheme of dc_adjoint.hip before spending GPU time:
  python tests/proto_adjoint.py c4|dress7k|dress|hat [options]
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc      # noqa: E402
import scenes   # noqa: E402


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def scene_c4(steps=6):
    import types
    import bench
    args = types.SimpleNamespace(grid=100, fold_rows=5, fold_gap=0.02, flap_force=2.0, h=1.0 / 180, fwd_tol=1e-8, bwd_tol=5e-4, cg_tol=1e-4, cg_max=500,
                                 adjoint_mode=1, adjoint_rel_tol=1e-6, block_precond=0, selfcollision=1, warmup=5, cpu_threads=0)
    V, F, V0, flap, center = bench.scene(args)
    o = orc.Oracle(V, F, h=args.h, density=0.3, k_stretch=150.0, k_bend=1e-5, fwd_tol=1e-8, bwd_tol=5e-4, selfcollision=True,
                   gradient_clipping=True, threads=min(os.cpu_count(), 32))
    X0, MU = bench.rollout_inputs(V0, np.arange(1))
    o.add_sphere(center, 2.0, float(f32(MU[0, 0])))
    o.build()
    m = o.vertex_data()[0]
    o.set_force_extras(None, bench.flap_force(args, m, flap), 1.0)
    x, v = X0[0], np.zeros_like(X0[0])
    for s in range(steps):
        out = o.step(x, v); x, v = out["x"], out["v"]
        print(f"  step {s}: PD {out['iters']} prim {out['nprim']} self {out['nself']}", flush=True)
    return o, out


def scene_dress(mesh="dress7k"):
    V, F = scenes.load_mesh(mesh)
    cfg = dict(h=1.0 / 120, density=0.2, k_stretch=800.0, k_bend=0.05)
    P, rmin, rmax = scenes.normalise_model(V, "FRONT", 8.0)
    P = f32(P)
    top = np.argsort(-P[:, 1])[:6].tolist()
    o = orc.Oracle(P, F, h=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], fwd_tol=1e-8,
                   bwd_tol=1e-9, attachments=top, selfcollision=True, contact=True, gradient_clipping=False, threads=min(os.cpu_count(), 32))
    o.build()
    rng = np.random.default_rng(8)
    X = P.copy(); X[:, 2] *= 0.9
    vel = np.zeros_like(X); vel[:, 2] = -0.1 * np.sign(P[:, 2])
    x0 = f32((X + 0.0005 * rng.standard_normal(X.shape)).reshape(-1))
    v0 = f32((vel + 0.005 * rng.standard_normal(X.shape)).reshape(-1))
    xf = f32(X[top].reshape(-1))
    out = o.step(x0, v0, xf)
    print(f"  dress {mesh}: PD {out['iters']} self {out['nself']} layers {out['nlayers']}", flush=True)
    return o, out


def scene_hat():
    cfg = scenes.HAT
    V, F = scenes.load_mesh("hat")
    P, rmin, rmax = scenes.normalise_model(V, cfg["orientation"], cfg["cloth_dim"])
    P = f32(P)
    center = f32(scenes.hat_head_center(rmin, rmax, cfg["sphere_radius"]))
    att = cfg["attachments"]
    o = orc.Oracle(P, F, h=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], fwd_tol=1e-8,
                   bwd_tol=1e-9, attachments=att, selfcollision=False, gradient_clipping=False)
    o.add_sphere(center, cfg["sphere_radius"], cfg["sphere_mu"])
    o.build()
    x, v = f32(P.reshape(-1)), np.zeros(P.size)
    xf = P[att].reshape(-1).copy()
    for s in range(13):
        xf = xf + np.tile([0.0, -0.05, -0.3], 2)
        out = o.step(x, v, f32(xf)); x, v = out["x"], out["v"]
    print(f"  hat: PD {out['iters']} prim {out['nprim']}", flush=True)
    return o, out


class Op32:
    """fp32 operator: entries of K rounded to fp32, products and sums in fp32"""
    def __init__(self, K):
        self.K = K.tocsr().astype(np.float32)
        self.n = 0

    def __call__(self, x):
        self.n += 1
        return self.K @ x.astype(np.float32)


def bicgstab32(A, b, minv, tol, maxit, x0=None):
    """right-preconditioned BiCGSTAB in fp32 (the device's loop, dc_adjoint.hip), dot products accumulated in fp64"""
    f = np.float32
    x = np.zeros_like(b, dtype=f) if x0 is None else x0.astype(f)
    r = b.astype(f) if x0 is None else (b.astype(f) - A(x))
    rhat = r.copy(); p = r.copy()
    rho = float(np.dot(r.astype(np.float64), r.astype(np.float64)))
    stop = tol * tol * float(np.dot(b.astype(np.float64), b.astype(np.float64)))
    hist = []
    for k in range(maxit):
        ph = minv(p)
        v = A(ph)
        rv = float(np.dot(rhat.astype(np.float64), v.astype(np.float64)))
        if abs(rv) < 1e-300: return x, k, "breakdown rv", hist
        alpha = f(rho / rv)
        s = r - alpha * v
        ss = float(np.dot(s.astype(np.float64), s.astype(np.float64)))
        if ss <= stop:
            return x + alpha * ph, k + 1, "ok", hist
        sh = minv(s)
        t = A(sh)
        ts = float(np.dot(t.astype(np.float64), s.astype(np.float64))); tt = float(np.dot(t.astype(np.float64), t.astype(np.float64)))
        if tt < 1e-300: return x, k, "breakdown tt", hist
        omega = f(ts / tt)
        x = x + alpha * ph + omega * sh
        r = s - omega * t
        rr = float(np.dot(r.astype(np.float64), r.astype(np.float64)))
        hist.append(np.sqrt(rr / (stop / tol / tol)))
        if rr <= stop: return x, k + 1, "ok", hist
        rho_new = float(np.dot(rhat.astype(np.float64), r.astype(np.float64)))
        if abs(rho_new) < 1e-300 or omega == 0: return x, k + 1, "breakdown rho", hist
        beta = f((rho_new / rho) * (float(alpha) / float(omega)))
        rho = rho_new
        p = r + beta * (p - omega * v)
    return x, maxit, "cap", hist


def block_jacobi(K):
    """inverse 3x3 diagonal blocks of K as a callable (fp32)"""
    n = K.shape[0] // 3
    Kc = K.tocsr()
    blocks = np.zeros((n, 3, 3))
    for a in range(3):
        for b in range(3):
            blocks[:, a, b] = Kc[np.arange(n) * 3 + a, np.arange(n) * 3 + b].A1 if hasattr(Kc[np.arange(n) * 3 + a, np.arange(n) * 3 + b], "A1") else np.asarray(Kc[np.arange(n) * 3 + a, np.arange(n) * 3 + b]).ravel()
    inv = np.linalg.inv(blocks).astype(np.float32)
    return lambda z: np.einsum("nab,nb->na", inv, z.reshape(n, 3)).reshape(-1).astype(np.float32)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c4"
    t0 = time.time()
    o, out = {"c4": scene_c4, "dress7k": lambda: scene_dress("dress7k"), "dress": lambda: scene_dress("dress"), "hat": scene_hat}[which]()
    K = o.adjoint_matrix(out["id"])
    n3 = K.shape[0]
    print(f"K: {n3} x {n3}, nnz {K.nnz}, built in {time.time() - t0:.1f} s", flush=True)
    rng = np.random.default_rng(4)
    g = f32(rng.standard_normal(n3))
    lu = spl.splu(K.tocsc())
    u_star = lu.solve(g)
    print(f"direct solve: residual {np.linalg.norm(K @ u_star - g) / np.linalg.norm(g):.1e}")
    Pd = sp.csr_matrix(o.P_csr()[::-1], shape=(o.N, o.N)).diagonal() if False else None
    ptr, col, val = o.P_csr()
    Ps = sp.csr_matrix((val, col, ptr), shape=(o.N, o.N))
    dinv = np.repeat(1.0 / Ps.diagonal(), 3).astype(np.float32)
    jac = lambda z: (z * dinv).astype(np.float32)
    blk = block_jacobi(K)
    for name, minv in (("diag(P)", jac), ("blocks of K", blk)):
        A = Op32(K)
        u1, it, status, hist = bicgstab32(A, g, minv, 1e-6, 4000)
        e1 = np.linalg.norm(u1 - u_star) / np.linalg.norm(u_star)
        r1 = np.linalg.norm(g - K @ u1.astype(np.float64)) / np.linalg.norm(g)
        print(f"[{name}] fp32 BiCGSTAB to 1e-6: {it} iterations ({status}), error vs direct {e1:.2e}, true fp64 residual {r1:.2e}, min recurrence residual {min(hist) if hist else 0:.1e}")
        # mixed-precision refinement: inner fp32 solves on the fp64 residual
        for inner_tol in (1e-2, 1e-3, 1e-4):
            u = np.zeros(n3); total = 0
            for cyc in range(12):
                r = g - K @ u
                rel = np.linalg.norm(r) / np.linalg.norm(g)
                if rel <= 1e-6: break
                scale = np.linalg.norm(r)
                A = Op32(K)
                d, it, status, hist = bicgstab32(A, r / scale, minv, max(inner_tol, 0.5e-6 / rel), 1000)
                total += it
                u = u + scale * d.astype(np.float64)
            r = g - K @ u
            print(f"    refinement inner tol {inner_tol:g}: {cyc} cycles, {total} inner iterations, final fp64 residual {np.linalg.norm(r) / np.linalg.norm(g):.1e}, "
                  f"error vs direct {np.linalg.norm(u - u_star) / np.linalg.norm(u_star):.2e}")


if __name__ == "__main__":
    main()
