"""Developer script: fwd+bwd throughput of the other BASELINE.json configs (C2 T-shirt, C3 hat, C5 sock, dress mesh) with
device-resident fused rollouts. Not the headline metric (bench.py is); numbers go into DESIGN.md §6."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes, meshes
from diffcloth_amd import capi

def f32(a): return np.asarray(a, dtype=np.float32).astype(np.float64)

ONLY = sys.argv[1:]          # optional: substrings of the configuration names to run

def run(name, B, K, cfg, prims_fn, att, selfc, fwd_tol, orient="FRONT", dim=6.0, adjoint_mode=1, bwd_tol=5e-4):
    if ONLY and not any(k in name for k in ONLY):
        return
    V, F = scenes.load_mesh(cfg["mesh"])
    if cfg.get("raw"):      # the mesh file's own coordinates (the slope fabric lies on its plane as shipped)
        P, rmin, rmax = V, V.min(axis=0), V.max(axis=0)
    else:
        P, rmin, rmax = scenes.normalise_model(V, orient, dim)
    P = f32(P)
    e = capi.Engine(0)
    e.set_mesh(P, F); e.set_attachments(att)
    e.set_params(time_step=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], forward_tol=fwd_tol,
                 backward_tol=bwd_tol, cg_rel_tol=1e-4, cg_max_iter=2000, gradient_clipping=1, selfcollision_enabled=int(selfc),
                 adjoint_mode=adjoint_mode, adjoint_rel_tol=1e-6)
    e.set_primitives(prims_fn(rmin, rmax))
    e.build()
    e.alloc_batch(B, K + 2)
    rng = np.random.default_rng(0)
    X = np.stack([f32(P.reshape(-1) + 0.001 * rng.standard_normal(P.size)) for _ in range(B)])
    e.set_state(0, X, np.zeros_like(X))
    e.rollout_forward(0, 2)
    e.seed_gradient(2, None, 1e-4); e.rollout_backward(2, 1); e.sync(); e.kernel_times(reset=True)
    t0 = time.perf_counter()
    e.rollout_forward(2, K); e.seed_gradient(2 + K, None, 2.0 / ((K + 1) * e.N)); e.rollout_backward(2 + K, K); e.sync()
    dt = time.perf_counter() - t0
    kt = e.kernel_times()
    pd = np.mean([e.get_stats(s)[0]["pd_iters"].mean() for s in range(3, 3 + K)])
    sc = np.mean([e.get_stats(s)[0]["self_contacts"].mean() for s in range(3, 3 + K)])
    adj = np.mean([e.get_stats(s)[1]["adjoint_iters"].mean() for s in range(3, 3 + K)])
    cg = np.mean([e.get_stats(s)[0]["cg_iters"].mean() for s in range(3, 3 + K)])
    cyc = np.mean([e.get_stats(s)[1]["refine_cycles"].mean() for s in range(3, 3 + K)]); f64 = np.mean([e.get_stats(s)[1]["fp64_iters"].mean() for s in range(3, 3 + K)])
    conv = np.mean([(e.get_stats(s)[1]["converged"] != 0).mean() for s in range(3, 3 + K)])
    per_rollout = np.sum([e.get_stats(s)[1]["adjoint_iters"] for s in range(3, 3 + K)], axis=0) / K      # (the launch waits for the slowest rollout)
    print(f"{name}: N={e.N} B={B} K={K}: {B * K / dt:.0f} rollout-steps/s, {dt / K * 1e3:.2f} ms per batch step "
          f"(fwd {kt['fwd_ms'] / K:.2f} ms, bwd {kt['bwd_ms'] / K:.2f} ms), mean PD iters {pd:.0f} (PCG {cg / max(pd, 1):.0f} each), adjoint iters {adj:.0f} (slowest rollout {per_rollout.max():.0f}) in {cyc:.1f} fp32 solves (+ {f64:.0f} fp64 fall-back iterations, converged {conv:.2f}), self contacts {sc:.0f}")

hat = lambda rmin, rmax: [dict(kind=capi.DC_PRIM_SPHERE, group=0, center=f32(scenes.hat_head_center(rmin, rmax, 2.1)), radius=2.1, mu=0.1)]
none = lambda rmin, rmax: []
def leg(rmin, rmax):
    c, ch = scenes.sock_leg(rmin, rmax)
    return [dict(kind=capi.DC_PRIM_SPHERE if k == 0 else capi.DC_PRIM_CAPSULE, group=0, center=f32(c + c0), radius=float(r), mu=0.4, top_offset=f32(t), length=float(l))
            for k, c0, t, r, l in ch]
T = scenes.TSHIRT
run("C2 tshirt (wind off, free fall, self-collision on)", 1, 10, T, none, [], True, 1e-8, "BACK")
run("C2 tshirt x256", 256, 10, T, none, [], True, 1e-8, "BACK")
run("C3 hat", 64, 10, scenes.HAT, hat, scenes.HAT["attachments"], False, 1e-8)
# the reference's adjoint iteration (mode 0). Its stopping rule |du| / N < tol is absolute: at the scene's own 5e-4 and this loss
# scale it stops after the first iteration; 1e-9 makes it work (cap 400 iterations, then the direct solve, Simulation.cpp:1589-1594)
run("C3 hat, reference adjoint iteration, tol 5e-4", 64, 10, scenes.HAT, hat, scenes.HAT["attachments"], False, 1e-8, adjoint_mode=0)
run("C3 hat, reference adjoint iteration, tol 1e-9", 64, 10, scenes.HAT, hat, scenes.HAT["attachments"], False, 1e-8, adjoint_mode=0, bwd_tol=1e-9)
run("C5 sock", 512, 10, scenes.SOCK, leg, scenes.SOCK["attachments"], False, 1e-9, "CUSTOM", 5.0)
# the reference's own 10k-class meshes (tests/test_gpu_garments10k.py holds their parity tests)
def slope(rmin, rmax):
    V, _ = scenes.load_mesh("perf96")
    P = f32(V); c0 = P.mean(axis=0)
    n = np.linalg.svd(P - c0)[2][2]; n = -n if n[1] < 0 else n
    ex = np.array([1.0, 0.0, 0.0]); ex = ex - n * (ex @ n); ex /= np.linalg.norm(ex); es = np.cross(n, ex)
    return [dict(kind=capi.DC_PRIM_PLANE, group=0, center=f32(c0 - 0.02 * n), top_offset=f32(-3.6 * ex + 3.6 * es), corner2=f32(3.6 * ex + 3.6 * es), radius=0.0, mu=0.2)]
run("perfFabric 96x96 sliding on the slope plane (9216 vertices, all in contact)", 256, 10, dict(mesh="perf96", raw=True, h=1.0 / 100, density=0.2, k_stretch=50.0, k_bend=1e-5), slope, [], False, 1e-8)
# (the 17 562-vertex dress has no throughput line: bandwidth 647 after renumbering puts it on the general global-memory kernels — 355 PCG iterations
#  per PD iteration, seconds per step; tests/test_gpu_garments10k.py holds its parity test)
D = dict(mesh="dress", h=1.0 / 120, density=0.2, k_stretch=800.0, k_bend=0.05)
run("dress (3634 vertices, self-collision on)", 256, 5, D, none, [0, 1, 2, 3, 4, 5], True, 1e-8, "FRONT", 8.0)
