"""Developer script (CPU only, scipy): OPERATOR APPLICATIONS different Krylov methods need for the adjoint system K u = g of the bench.py workload
(rollout 0, step 7 of the fp64 oracle; Jacobi diag(P) preconditioner as the engine's default) — to the inner tolerance of a correction solve
(1e-3) and to the full 1e-6. The engine's BiCGSTAB costs 2 applications + 172 B of vector passes per vertex and iteration; a method that needs
fewer applications for the same reduction is the one lever on the adjoint's iteration count that round 4's preconditioner study left open.
python tools/offline_adjoint_methods.py [bench | hat]  ->  profiles/r05_offline_adjoint_methods_{bench_config,hat}.txt"""
import os, sys, time, types
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench, orc, scenes
f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
SCENE = sys.argv[1] if len(sys.argv) > 1 else "bench"
if SCENE == "bench":
    args = types.SimpleNamespace(grid=100, fold_rows=5, fold_gap=0.02, flap_force=2.0, h=1.0 / 180, fwd_tol=1e-8, bwd_tol=5e-4, cg_tol=1e-4, cg_max=500,
                                 adjoint_mode=1, adjoint_rel_tol=1e-6, block_precond=0, selfcollision=1, warmup=5, cpu_threads=0)
    V, F, V0, flap, center = bench.scene(args)
    N = V.shape[0]
    o = orc.Oracle(V, F, h=args.h, density=0.3, k_stretch=150.0, k_bend=1e-5, fwd_tol=args.fwd_tol, bwd_tol=args.bwd_tol, selfcollision=True, gradient_clipping=True,
                   threads=min(os.cpu_count() or 1, 32))
    o.add_sphere(center, 2.0, 0.9); o.build()
    field = bench.flap_force(args, o.vertex_data()[0], flap)
    o.set_force_extras(None, field, 1.0)
    X0, MU = bench.rollout_inputs(V0, np.arange(2))
    o.set_mu(0, float(f32(MU[0, 0])))
    x, v = f32(X0[0]), np.zeros(3 * N)
    for s in range(7):
        t = time.time(); ref = o.step(x, v); x, v = f32(ref["x"]), f32(ref["v"])
        print("step", s, "PD", ref["iters"], "prim", ref["nprim"], "self", ref["nself"], f"{time.time() - t:.1f} s", flush=True)
    g = f32(x * (2.0 / (4 * N)))          # the loss gradient of bench.py's quadratic loss
elif SCENE == "hat":                      # the set-up of tools/offline_hat_adjoint.py: the hat lowered onto the head, 4 steps
    Vm, F = scenes.load_mesh("hat")
    cfg = scenes.HAT
    P, rmin, rmax = scenes.normalise_model(Vm, cfg["orientation"], cfg["cloth_dim"])
    N = P.shape[0]
    o = orc.Oracle(P, F, h=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], fwd_tol=1e-8, bwd_tol=1e-9,
                   attachments=cfg["attachments"], selfcollision=False, gradient_clipping=False)
    o.add_sphere(scenes.hat_head_center(rmin, rmax, cfg["sphere_radius"]), cfg["sphere_radius"], cfg["sphere_mu"])
    o.build()
    x = f32(P.reshape(-1)); v = np.zeros_like(x)
    a = f32(P[cfg["attachments"]].reshape(-1))
    for s in range(4):
        a = f32(a + np.array([0.02, -0.05, 0.01, 0.02, -0.05, 0.01]))
        ref = o.step(x, v, a); x, v = f32(ref["x"]), f32(ref["v"])
        print("step", s, "PD", ref["iters"], "prim", ref["nprim"], flush=True)
    g = f32(np.random.default_rng(1).standard_normal(3 * N) * 1e-2)
else:
    raise SystemExit("scene: bench | hat")
K = o.adjoint_matrix(ref["id"]).tocsr()
n3 = 3 * N
ptr, col, val = o.P_csr()
d = sp.csr_matrix((val, col, ptr), shape=(N, N)).diagonal()
# the oracle's vectors are planar or interleaved? K acts on the oracle's layout: use its own diagonal-of-P scaling through a probe
perm_test = K.diagonal()
Ji_inter = np.repeat(1.0 / d, 3)
Ji_planar = np.tile(1.0 / d, 3)
# pick the layout whose scaling makes diag(M^-1 K) closest to 1
Ji = Ji_inter if np.abs(perm_test * Ji_inter - 1).mean() < np.abs(perm_test * Ji_planar - 1).mean() else Ji_planar
gn = np.linalg.norm(g)
S = (K - K.T)
print(f"K: {K.shape[0]} rows, {K.nnz} non-zeros; |K - K^T|_F / |K|_F = {spla.norm(S) / spla.norm(K):.2e}; rows with a non-symmetric entry: {np.count_nonzero(np.abs(S).sum(axis=1))}", flush=True)

class Op:
    """right-preconditioned operator K M^-1 counting applications"""
    def __init__(self): self.n = 0
    def __call__(self, y): self.n += 1; return K @ (Ji * y)
def report(name, op, y, extra=""):
    u = Ji * y
    print(f"{name}: {op.n} operator applications, true residual {np.linalg.norm(g - K @ u) / gn:.1e} {extra}", flush=True)

def bicgstab(tol, maxit=20000):
    op = Op(); y = np.zeros(n3); r = g.copy(); rh = r.copy(); p = r.copy(); rho = rh @ r
    for k in range(maxit):
        vv = op(p); alpha = rho / (rh @ vv); s = r - alpha * vv
        if np.linalg.norm(s) <= tol * gn: y += alpha * p; break
        t = op(s); omega = (t @ s) / (t @ t); y += alpha * p + omega * s; r = s - omega * t
        if np.linalg.norm(r) <= tol * gn: break
        rho_n = rh @ r; beta = (rho_n / rho) * (alpha / omega); rho = rho_n; p = r + beta * (p - omega * vv)
    return op, y
def idr(s, tol, maxit=40000, seed=1):
    """IDR(s) (Sonneveld & van Gijzen 2008, the biorthogonal 'IDR(s)-biortho' prototype form)"""
    rng = np.random.default_rng(seed)
    op = Op(); y = np.zeros(n3); r = g.copy()
    Pm = np.linalg.qr(rng.standard_normal((n3, s)))[0]
    G = np.zeros((n3, s)); U = np.zeros((n3, s)); M = np.eye(s); om = 1.0
    it = 0
    while np.linalg.norm(r) > tol * gn and it < maxit:
        f = Pm.T @ r
        for k in range(s):
            c = np.linalg.solve(M[k:, k:], f[k:])
            vv = r - G[:, k:] @ c
            U[:, k] = U[:, k:] @ c + om * vv
            G[:, k] = op(U[:, k]); it += 1
            for i in range(k):
                a = (Pm[:, i] @ G[:, k]) / M[i, i]
                G[:, k] -= a * G[:, i]; U[:, k] -= a * U[:, i]
            M[k:, k] = Pm[:, k:].T @ G[:, k]
            b = f[k] / M[k, k]
            r = r - b * G[:, k]; y = y + b * U[:, k]
            if np.linalg.norm(r) <= tol * gn: return op, y
            if k + 1 < s: f[k + 1:] = f[k + 1:] - b * M[k + 1:, k]
        t = op(r); it += 1
        om = (t @ r) / (t @ t)
        # (the 'maintaining the convergence' safeguard of the paper, kappa = 0.7)
        rho = (t @ r) / (np.linalg.norm(t) * np.linalg.norm(r))
        if abs(rho) < 0.7: om *= 0.7 / abs(rho)
        y = y + om * r; r = r - om * t
    return op, y
def scipy_method(fn, tol, **kw):
    op = Op()
    A = spla.LinearOperator((n3, n3), matvec=op)
    y, info = fn(A, g, rtol=tol, atol=0, **kw)
    return op, y, info
def cg_plain(tol, maxit=20000):
    """CG as if K M^-1 were symmetric positive definite (it is not: contact rows) with symmetric Jacobi scaling"""
    sq = np.sqrt(Ji); op = Op()
    A = lambda z: sq * (K @ (sq * z))
    b = sq * g; z = np.zeros(n3); r = b.copy(); p = r.copy(); rz = r @ r; n = 0
    for k in range(maxit):
        Ap = A(p); n += 1; al = rz / (p @ Ap); z += al * p; r -= al * Ap
        if np.linalg.norm(r / sq) <= tol * gn: break
        rn = r @ r; p = r + (rn / rz) * p; rz = rn
    op.n = n
    return op, (sq * z) / Ji

for tol in (1e-3, 1e-6):
    print(f"--- relative residual {tol:g} ---", flush=True)
    op, y = bicgstab(tol); report("BiCGSTAB (the engine's method)", op, y)
    for s in (2, 4, 8):
        op, y = idr(s, tol); report(f"IDR({s})", op, y, f"[{s + 1} extra vectors of 3N; per application ~{2 + (s + 1) / 2:.1f} vector passes]")
    for m in (20, 50):
        op, y, info = scipy_method(spla.gmres, tol, restart=m, maxiter=1000); report(f"GMRES({m})", op, y, f"[info {info}; orthogonalisation against up to {m} vectors per application]")
    op, y, info = scipy_method(spla.tfqmr, tol, maxiter=20000); report("TFQMR", op, y, f"[info {info}]")
    op, y, info = scipy_method(spla.cgs, tol, maxiter=20000); report("CGS", op, y, f"[info {info}]")
    op, y = cg_plain(tol); report("CG on the symmetrically scaled K (not a valid method for this K: what its non-symmetry does to it)", op, y)

# ---- bench scene only: does CG keep working as the contact set evolves? Ten more steps of the rollout, the two-solve refinement of the engine
# (each correction solve reduces its right-hand side by 1e-3, the residual in between is the true one) with CG against BiCGSTAB
if SCENE == "bench":
    print("--- steps 7 .. 16 of the rollout: applications of two correction solves to 1e-3 each (CG | BiCGSTAB), smallest p.Kp / (|p||Kp|) seen by CG ---", flush=True)
    sq = np.sqrt(Ji)
    def cg_solve(K, b, tol):
        z = np.zeros(n3); r = sq * b; p = r.copy(); rz = r @ r; n = 0; bn = np.linalg.norm(b); worst = 1.0
        while n < 500:
            Ap = sq * (K @ (sq * p)); n += 1
            worst = min(worst, (p @ Ap) / (np.linalg.norm(p) * np.linalg.norm(Ap)))
            al = rz / (p @ Ap); z += al * p; r -= al * Ap
            if np.linalg.norm(r / sq) <= tol * bn: break
            rn = r @ r; p = r + (rn / rz) * p; rz = rn
        return sq * z, n, worst
    def bi_solve(K, b, tol):
        y = np.zeros(n3); r = b.copy(); rh = r.copy(); p = r.copy(); rho = rh @ r; n = 0; bn = np.linalg.norm(b)
        while n < 1000:
            vv = K @ (Ji * p); n += 1; alpha = rho / (rh @ vv); s_ = r - alpha * vv
            if np.linalg.norm(s_) <= tol * bn: y += alpha * p; break
            t = K @ (Ji * s_); n += 1; omega = (t @ s_) / (t @ t); y += alpha * p + omega * s_; r = s_ - omega * t
            if np.linalg.norm(r) <= tol * bn: break
            rho_n = rh @ r; beta = (rho_n / rho) * (alpha / omega); rho = rho_n; p = r + beta * (p - omega * vv)
        return Ji * y, n
    for s in range(7, 17):
        ref = o.step(x, v); x, v = f32(ref["x"]), f32(ref["v"])
        Ks = o.adjoint_matrix(ref["id"]).tocsr()
        gs = f32(x * (2.0 / (4 * N)))
        out = []
        for name in ("cg", "bi"):
            u = np.zeros(n3); total = 0; worst = 1.0
            for cyc in range(2):
                rres = gs - Ks @ u
                if name == "cg": du, n, w = cg_solve(Ks, rres, 1e-3); worst = min(worst, w)
                else: du, n = bi_solve(Ks, rres, 1e-3)
                u += du; total += n
            out.append((total, np.linalg.norm(gs - Ks @ u) / np.linalg.norm(gs), worst))
        print(f"step {s}: PD {ref['iters']}, prim {ref['nprim']}, self {ref['nself']}: CG {out[0][0]} applications (residual {out[0][1]:.1e}, min cos {out[0][2]:.2f}) | "
              f"BiCGSTAB {out[1][0]} (residual {out[1][1]:.1e})", flush=True)

# ---- the same two-solve scheme with the correction solves in float32 (vectors and operator in fp32, scalars accumulated in fp64 as the kernels
# do), the residual between them in fp64: does CG on this K survive single precision?
if SCENE == "bench":
    print("--- last step above, correction solves in float32 ---", flush=True)
    K32 = Ks.astype(np.float32); J32 = Ji.astype(np.float32); sq32 = np.sqrt(Ji).astype(np.float32)
    dot = lambda a, b: float(np.dot(a.astype(np.float64), b.astype(np.float64)))
    def cg32(b, tol):
        z = np.zeros(n3, np.float32); r = sq32 * b.astype(np.float32); p = r.copy(); rz = dot(r, r); n = 0; bn = np.linalg.norm(b)
        while n < 500:
            Ap = sq32 * (K32 @ (sq32 * p)); n += 1
            al = np.float32(rz / dot(p, Ap)); z += al * p; r -= al * Ap
            if np.sqrt(dot(r / sq32, r / sq32)) <= tol * bn: break
            rn = dot(r, r); p = r + np.float32(rn / rz) * p; rz = rn
        return (sq32 * z).astype(np.float64), n
    def bi32(b, tol):
        y = np.zeros(n3, np.float32); r = b.astype(np.float32); rh = r.copy(); p = r.copy(); rho = dot(rh, r); n = 0; bn = np.linalg.norm(b)
        while n < 1000:
            vv = K32 @ (J32 * p); n += 1; alpha = np.float32(rho / dot(rh, vv)); s_ = r - alpha * vv
            if np.sqrt(dot(s_, s_)) <= tol * bn: y += alpha * p; break
            t = K32 @ (J32 * s_); n += 1; omega = np.float32(dot(t, s_) / dot(t, t)); y += alpha * p + omega * s_; r = s_ - omega * t
            if np.sqrt(dot(r, r)) <= tol * bn: break
            rho_n = dot(rh, r); beta = np.float32((rho_n / rho) * (float(alpha) / float(omega))); rho = rho_n; p = r + beta * (p - omega * vv)
        return (J32 * y).astype(np.float64), n
    for name, solve in (("CG", cg32), ("BiCGSTAB", bi32)):
        u = np.zeros(n3); total = 0; hist = []
        for cyc in range(3):
            rres = gs - Ks @ u
            hist.append(np.linalg.norm(rres) / np.linalg.norm(gs))
            if hist[-1] <= 1e-6: break
            du, n = solve(rres, 1e-3); u += du; total += n
        print(f"{name} in float32: {total} applications, fp64 residual after each solve {['%.1e' % h for h in hist[1:]] + ['%.1e' % (np.linalg.norm(gs - Ks @ u) / np.linalg.norm(gs))]}", flush=True)
