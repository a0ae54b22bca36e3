"""Developer script (CPU only, scipy): BiCGSTAB iteration counts of the adjoint system K u = g of the bench.py workload (rollout 0, step 7 of the
fp64 oracle) under different preconditioners — what a better preconditioner could buy at most. Output: profiles/r04_offline_adjoint_preconditioners_bench_config.txt
"""
import os, sys, time, types
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench, orc
f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
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
K = o.adjoint_matrix(ref["id"]).tocsr()
n3 = 3 * N
ptr, col, val = o.P_csr()
Pm = sp.csr_matrix((val, col, ptr), shape=(N, N))
d = Pm.diagonal(); sq = 1 / np.sqrt(d)
Ah = (sp.diags(sq) @ Pm @ sp.diags(sq)).tocsc()
w, U = spla.eigsh(Ah, k=32, sigma=0, which='LM'); wmax = spla.eigsh(Ah, k=1, which='LA')[0][0]
print("scaled P: lowest", w[:6], "largest", wmax, flush=True)
Kb = K.tobsr(blocksize=(3, 3)); blocks = np.zeros((N, 3, 3))
for i in range(N):
    for jj in range(Kb.indptr[i], Kb.indptr[i + 1]):
        if Kb.indices[jj] == i: blocks[i] = Kb.data[jj]
binv = np.linalg.inv(blocks)
Bi = lambda r: np.einsum('nij,nj->ni', binv, r.reshape(N, 3)).reshape(-1)
Ji = lambda r: (r.reshape(N, 3) / d[:, None]).reshape(-1)
g = f32(x * (2.0 / (4 * N)))          # the loss gradient of bench.py's quadratic loss
def run(name, M, tol=1e-6, nmv=1):
    cnt = [0]
    def cb(xk): cnt[0] += 1
    t = time.time()
    u, info = spla.bicgstab(K, g, rtol=tol, atol=0, maxiter=5000, M=spla.LinearOperator((n3, n3), matvec=M), callback=cb)
    print(f"{name}: {cnt[0]} iterations (info {info}, true res {np.linalg.norm(g - K @ u) / np.linalg.norm(g):.1e}), products with P-like matrices per application {nmv}, {time.time() - t:.1f} s", flush=True)
run("Jacobi diag(P)  [bench.py default]", Ji)
run("block-Jacobi of K", Bi)
for k in (16, 32):
    Z = sq[:, None] * U[:, :k]; G = np.linalg.inv(Z.T @ (Pm @ Z))
    run(f"block-Jacobi + P-coarse k={k}", lambda r, Z=Z, G=G: Bi(r) + (Z @ (G @ (Z.T @ r.reshape(N, 3)))).reshape(-1))
# m steps of Chebyshev iteration on P (Jacobi-scaled) as a fixed polynomial preconditioner: M^-1 = p_m(D^-1 P) D^-1
lmin, lmax = w[0], wmax
def cheb(m, lo):
    th, de = (lmax + lo) / 2, (lmax - lo) / 2
    def M(r):
        R = r.reshape(N, 3)
        b = R / d[:, None]                      # Jacobi-scaled rhs, system (D^-1 P) y = D^-1 r
        y = b / th; rr = b - (Pm @ y) / d[:, None]
        sig = th / de; rho = 1 / sig; dvec = y.copy()
        for _ in range(m - 1):
            rho_n = 1 / (2 * sig - rho)
            dvec = rho_n * rho * dvec + (2 * rho_n / de) * rr
            y = y + dvec; rr = b - (Pm @ y) / d[:, None]
            rho = rho_n
        return y.reshape(-1)
    return M
for m in (2, 3, 4, 6):
    for lo in (lmax / 10, lmax / 30):
        run(f"Chebyshev({m}) on P, interval [lmax/{lmax / lo:.0f}, lmax]", cheb(m, lo), nmv=m - 1)
lu = spla.splu(Pm.tocsc())
run("P^-1 exact", lambda r: lu.solve(r.reshape(N, 3)).reshape(-1))
ilu = spla.spilu(K.tocsc(), drop_tol=0, fill_factor=1)
run("ILU(0)-like of K", ilu.solve)
