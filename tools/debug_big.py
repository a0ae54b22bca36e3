"""Developer script: N = 16384, compare the split adjoint (K = 8) with K = 1 vertex by vertex."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fullsize as T

def run(K, B, fused):
    os.environ["DC_CLUSTER"] = str(K)
    V, F, e, o = T.scene(128, selfcollision=False, fwd_tol=1e-8)
    X0, MU = T.start_states(V, B, [])
    W = 4
    e.alloc_batch(B, W + 1)
    e.set_mu(MU); e.set_state(0, X0, np.zeros_like(X0))
    if fused: e.rollout_forward(0, W)
    else:
        for s in range(W): e.step_forward(s)
    st = e.step_forward(W)
    rng = np.random.default_rng(12)
    gx = T.f32(rng.standard_normal(X0.shape)); gv = T.f32(0.01 * rng.standard_normal(X0.shape))
    gb = e.step_backward(W + 1, gx, gv, is_start=False)
    x1, _ = e.get_state(W + 1)
    return e.cluster(), x1, gb, st

ref = run(1, 3, False)
for K, B, fused in [(8, 3, False), (8, 3, True), (8, 2, False), (4, 3, False)]:
    k, x1, gb, st = run(K, B, fused)
    for b in range(min(B, 3)):
        d = (gb["dL_dx"][b] - ref[2]["dL_dx"][b]).reshape(-1, 3)
        nv = np.linalg.norm(d, axis=1); tot = np.linalg.norm(ref[2]["dL_dx"][b])
        top = np.argsort(-nv)[:6]
        print(f"K={k} B={B} fused={fused} rollout {b}: |dx state| {np.abs(x1[b]-ref[1][b]).max():.1e} grad rel diff {np.linalg.norm(d)/tot:.2e} iters {gb['adjoint_iters'][b]}/{ref[2]['adjoint_iters'][b]} "
              f"conv {gb['converged'][b]} udiff {gb['last_udiff'][b]:.1e}; worst vertices {top.tolist()} (rows of 128: {(top//128).tolist()}) {np.array2string(nv[top]/tot, precision=1)}", flush=True)
