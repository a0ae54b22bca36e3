"""Developer script (CPU only, scipy): would a multigrid preconditioner pay for the forward solve of the bench.py workload? CG on the system
matrix P of the 100 x 100 cloth (cond 61 after Jacobi scaling) to 1e-4 with (a) Jacobi, what the resident kernel runs, (b) a two-grid cycle
with an EXACT coarse solve (the best any V-cycle can do): bilinear interpolation from the 50 x 50 grid, Galerkin coarse operator, nu damped-Jacobi
sweeps before and after. Counted in products with P per solve (a cycle costs 2 nu smoothing products + 1 residual, coarse work not counted).
Output: profiles/r04_offline_forward_multigrid.txt"""
import os, sys, types
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench, orc
args = types.SimpleNamespace(grid=100, fold_rows=5, fold_gap=0.02, flap_force=2.0, h=1.0 / 180, fwd_tol=1e-8, bwd_tol=5e-4, cg_tol=1e-4, cg_max=500,
                             adjoint_mode=1, adjoint_rel_tol=1e-6, block_precond=0, selfcollision=1, warmup=5, cpu_threads=0)
V, F, V0, flap, center = bench.scene(args)
N = V.shape[0]; n = 100
o = orc.Oracle(V, F, h=args.h, density=0.3, k_stretch=150.0, k_bend=1e-5, fwd_tol=1e-8, bwd_tol=5e-4, selfcollision=False).build()
ptr, col, val = o.P_csr()
P = sp.csr_matrix((val, col, ptr), shape=(N, N))
d = P.diagonal()
# bilinear prolongation from the (n/2 x n/2) grid of every second vertex (vertex id = row * n + column in bench.scene's grid)
nc = n // 2
rows, cols, vals = [], [], []
for i in range(n):
    for j in range(n):
        fi, fj = min(i / 2.0, nc - 1), min(j / 2.0, nc - 1)
        i0, j0 = int(np.floor(fi)), int(np.floor(fj)); i1, j1 = min(i0 + 1, nc - 1), min(j0 + 1, nc - 1)
        a, b = fi - i0, fj - j0
        for (ci, cj, w) in ((i0, j0, (1 - a) * (1 - b)), (i1, j0, a * (1 - b)), (i0, j1, (1 - a) * b), (i1, j1, a * b)):
            if w > 0: rows.append(i * n + j); cols.append(ci * nc + cj); vals.append(w)
Pr = sp.csr_matrix((vals, (rows, cols)), shape=(N, nc * nc))
Ac = (Pr.T @ P @ Pr).tocsc(); lu = spla.splu(Ac)
rng = np.random.default_rng(0)
grid = np.stack(np.meshgrid(np.linspace(0, 1, n), np.linspace(0, 1, n), indexing="ij"), -1).reshape(N, 2)
rhs = {"smooth": d * np.sin(3 * grid[:, 0]) * np.cos(2 * grid[:, 1]), "rough": d * rng.standard_normal(N), "mixed": d * (np.sin(3 * grid[:, 0]) + 0.1 * rng.standard_normal(N))}
def cg(b, M, tol=1e-4):
    x = np.zeros_like(b); r = b.copy(); z = M(r); p = z.copy(); rz = r @ z; r0 = np.sqrt(rz); it = 0
    while np.sqrt(rz) > tol * r0 and it < 500:
        Ap = P @ p; al = rz / (p @ Ap); x += al * p; r -= al * Ap; z = M(r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn; it += 1
    return it
def twogrid(nu, omega=0.7):
    def M(r):
        x = np.zeros_like(r)
        for _ in range(nu): x += omega * (r - P @ x) / d
        x += Pr @ lu.solve(Pr.T @ (r - P @ x))
        for _ in range(nu): x += omega * (r - P @ x) / d
        return x
    return M
for name, b in rhs.items():
    j = cg(b, lambda r: r / d)
    line = f"{name} right-hand side: Jacobi-CG {j} iterations = {j} products"
    for nu in (1, 2):
        it = cg(b, twogrid(nu))
        line += f"; two-grid({nu},{nu}) CG {it} iterations = {it * (2 * nu + 1 + 1)} products (+ {it} exact coarse solves)"
    print(line, flush=True)
