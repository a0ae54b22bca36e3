"""Developer script (CPU only, scipy): BiCGSTAB on the explicit adjoint matrix of the fp64 oracle for the 7 742-vertex dress (argument `squash`: the step of
tests/test_gpu_configs.py; `twirl N`: N steps of the rim twirl) with K's 3 x 3 blocks alone and with the coarse level over the k lowest eigenvectors of the
scaled P (Galerkin operator of P, or of K itself). Numbers quoted in DESIGN.md section 4.1; outputs in profiles/r04_offline_dress7742_adjoint_*.txt
"""
import os, sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import scenes, orc
f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
V, F = scenes.load_mesh("dress7k")
P, rmin, rmax = scenes.normalise_model(V, "FRONT", 8.0); P = f32(P)
mode = sys.argv[1] if len(sys.argv) > 1 else "squash"
N = P.shape[0]
if mode == "squash":
    top = np.argsort(-P[:, 1])[:6].tolist()
else:
    top = np.where(P[:, 1] >= np.quantile(P[:, 1], 0.995))[0].tolist()
o = orc.Oracle(P, F, h=1/120, density=0.2, k_stretch=800.0, k_bend=0.05, fwd_tol=1e-8, bwd_tol=1e-9, attachments=top, selfcollision=True, contact=True,
               gradient_clipping=False, threads=min(os.cpu_count() or 1, 32))
o.build()
rng = np.random.default_rng(8)
if mode == "squash":
    X = P.copy(); X[:, 2] *= 0.9
    vel = np.zeros_like(X); vel[:, 2] = -0.1 * np.sign(P[:, 2])
    x0 = f32((X + 0.0005 * rng.standard_normal(X.shape)).reshape(-1)); v0 = f32((vel + 0.005 * rng.standard_normal(X.shape)).reshape(-1))
    ref = o.step(x0, v0, f32(X[top].reshape(-1)))
else:
    nst = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    mid = 0.5 * (rmin + rmax)
    x = f32(P.reshape(-1) + 0.001 * rng.standard_normal(P.size)); v = np.zeros_like(x)
    for s in range(nst):
        a = 0.02 * (s + 1); c, sn = np.cos(a), np.sin(a)
        q = P[top].copy(); r = q - mid
        q[:, 0] = mid[0] + c * r[:, 0] + sn * r[:, 2]; q[:, 2] = mid[2] - sn * r[:, 0] + c * r[:, 2]
        ref = o.step(x, v, f32(q.reshape(-1)))
        x, v = ref["x"], ref["v"]
        print("step", s, "iters", ref["iters"], "nself", ref["nself"], flush=True)
K = o.adjoint_matrix(ref["id"]).tocsr()
n3 = 3 * N
ptr, col, val = o.P_csr()
Pm = sp.csr_matrix((val, col, ptr), shape=(N, N))
d = Pm.diagonal(); sq = 1 / np.sqrt(d)
Ah = sp.diags(sq) @ Pm @ sp.diags(sq)
t = time.time(); w, U = spla.eigsh(Ah.tocsc(), k=32, sigma=0, which='LM'); print("eigsh", time.time() - t, w[:16])
sym = abs(K - K.T).max() / abs(K).max(); print("K asym", sym)
# block-Jacobi of K
Kb = K.tobsr(blocksize=(3, 3))
blocks = np.zeros((N, 3, 3))
for i in range(N):
    for jj in range(Kb.indptr[i], Kb.indptr[i + 1]):
        if Kb.indices[jj] == i: blocks[i] = Kb.data[jj]
binv = np.linalg.inv(blocks)
def Bi(r): return np.einsum('nij,nj->ni', binv, r.reshape(N, 3)).reshape(-1)
g = f32(rng.standard_normal(n3))
def run(name, M, tol=1e-7, maxit=20000):
    cnt = [0]
    def cb(xk): cnt[0] += 1
    t = time.time()
    u, info = spla.bicgstab(K, g, rtol=tol, atol=0, maxiter=maxit, M=spla.LinearOperator((n3, n3), matvec=M), callback=cb)
    print(f"{name}: iters {cnt[0]} info {info} true res {np.linalg.norm(g - K @ u) / np.linalg.norm(g):.2e} ({time.time() - t:.1f} s)", flush=True)
for k in (16, 32):
    Z = sq[:, None] * U[:, :k]                     # coarse basis, per coordinate
    G = np.linalg.inv(Z.T @ (Pm @ Z))
    def Mp(r, Z=Z, G=G):
        R = r.reshape(N, 3)
        return Bi(r) + (Z @ (G @ (Z.T @ R))).reshape(-1)
    run(f"block-Jacobi + P-coarse k={k}", Mp)
    # true Galerkin coarse operator of K on Z (x) I3
    Z3 = sp.kron(sp.csr_matrix(Z), sp.identity(3)).tocsr()
    E = (Z3.T @ (K @ Z3)).toarray(); Ei = np.linalg.inv(E)
    def Mk(r, Z3=Z3, Ei=Ei): return Bi(r) + Z3 @ (Ei @ (Z3.T @ r))
    run(f"block-Jacobi + K-coarse k={k}", Mk)
run("block-Jacobi", Bi)
