"""Developer script (CPU only, scipy): the same study on the hat (579 vertices), profiles/r04_offline_hat_adjoint.txt
"""
import os, sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import scenes, orc
f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
V, F = scenes.load_mesh("hat")
cfg = scenes.HAT
P, rmin, rmax = scenes.normalise_model(V, cfg["orientation"], cfg["cloth_dim"])
N = P.shape[0]
o = orc.Oracle(P, F, h=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], fwd_tol=1e-8, bwd_tol=1e-9,
               attachments=cfg["attachments"], selfcollision=False, gradient_clipping=False)
o.add_sphere(scenes.hat_head_center(rmin, rmax, cfg["sphere_radius"]), cfg["sphere_radius"], cfg["sphere_mu"])
o.build()
x = f32(P.reshape(-1)); v = np.zeros_like(x)
a = f32(P[cfg["attachments"]].reshape(-1))
nst = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for s in range(nst):
    a = f32(a + np.array([0.02, -0.05, 0.01, 0.02, -0.05, 0.01]))
    ref = o.step(x, v, a); x, v = f32(ref["x"]), f32(ref["v"])
    print("step", s, "iters", ref["iters"], "nprim", ref["nprim"], flush=True)
K = o.adjoint_matrix(ref["id"]).tocsr()
n3 = 3 * N
ptr, col, val = o.P_csr()
Pm = sp.csr_matrix((val, col, ptr), shape=(N, N))
d = Pm.diagonal(); sq = 1 / np.sqrt(d)
Ah = (sp.diags(sq) @ Pm @ sp.diags(sq)).tocsc()
w, U = spla.eigsh(Ah, k=32, sigma=0, which='LM'); print("lowest eigenvalues of scaled P", w[:8], "... largest", spla.eigsh(Ah, k=1, which='LA')[0])
Kb = K.tobsr(blocksize=(3, 3)); blocks = np.zeros((N, 3, 3))
for i in range(N):
    for jj in range(Kb.indptr[i], Kb.indptr[i + 1]):
        if Kb.indices[jj] == i: blocks[i] = Kb.data[jj]
binv = np.linalg.inv(blocks)
def Bi(r): return np.einsum('nij,nj->ni', binv, r.reshape(N, 3)).reshape(-1)
rng = np.random.default_rng(1)
g = f32(rng.standard_normal(n3) * 1e-2)
def run(name, M, tol=1e-6):
    cnt = [0]
    def cb(xk): cnt[0] += 1
    u, info = spla.bicgstab(K, g, rtol=tol, atol=0, maxiter=20000, M=spla.LinearOperator((n3, n3), matvec=M), callback=cb)
    print(f"{name}: iters {cnt[0]} info {info} true res {np.linalg.norm(g - K @ u) / np.linalg.norm(g):.2e}", flush=True)
run("block-Jacobi", Bi)
run("Jacobi diag(P)", lambda r: (r.reshape(N, 3) / d[:, None]).reshape(-1))
for k in (8, 16, 32):
    Z = sq[:, None] * U[:, :k]; G = np.linalg.inv(Z.T @ (Pm @ Z))
    run(f"block-Jacobi + P-coarse k={k}", lambda r, Z=Z, G=G: Bi(r) + (Z @ (G @ (Z.T @ r.reshape(N, 3)))).reshape(-1))
# P^-1 itself as preconditioner (the reference's iteration as M^-1): an upper bound on what any P-based preconditioner gives
lu = spla.splu(Pm.tocsc())
run("P^-1 (exact)", lambda r: lu.solve(r.reshape(N, 3)).reshape(-1))
