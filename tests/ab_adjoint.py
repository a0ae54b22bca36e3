"""A/B harness for the adjoint solve (TEST INFRASTRUCTURE: compares the HIP path with the fp64 oracle; run on a GPU box).

  DC_LIB=<build of libdiffcloth_hip.so> python tests/ab_adjoint.py c4,hat,dress7k,params [--fp32-only]

Prints, per scene: gradient errors of sampled rollouts against the oracle's direct (fp64) adjoint, solver statistics
(fp32 BiCGSTAB iterations, refinement cycles, fp64 fall-back iterations, true relative residual) and the kernel time of the
backward sweep. Used to choose between builds / settings within one GPU session (round 3: mixed-precision adjoint).
"""
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench                      # noqa: E402
import orc                        # noqa: E402
import scenes                     # noqa: E402
from diffcloth_amd import capi    # noqa: E402


DUMP = None      # --dump DIR: write the HIP path's tape of the sampled rollouts (tests/analyze_dump.py reads them on a CPU box)


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def bstat(bs, b):
    return (f"converged {bs['converged'][b]} bicgstab32 {bs['adjoint_iters'][b]} cycles {bs['refine_cycles'][b]} fp64 iters {bs['fp64_iters'][b]} "
            f"rel res {bs['last_udiff'][b]:.1e}")


def run_c4(fp32_only, B=256, sample=(0, 101, 255), S=2, W=5):
    args = types.SimpleNamespace(grid=100, fold_rows=5, fold_gap=0.02, flap_force=2.0, h=1.0 / 180, fwd_tol=1e-8, bwd_tol=5e-4, cg_tol=1e-4,
                                 cg_max=500, adjoint_mode=1, adjoint_rel_tol=1e-6, block_precond=0, selfcollision=1, warmup=5, cpu_threads=0)
    V, F, V0, flap, center = bench.scene(args)
    e = bench.make_engine(0, args, V, F, center)
    if fp32_only:
        e.set_params(adjoint_fp32_only=1); e.build()
    e.alloc_batch(B, W + S)
    X0, MU = bench.rollout_inputs(V0, np.arange(B))
    e.set_mu(MU)
    e.set_state(0, X0, np.zeros_like(X0))
    field = bench.flap_force(args, e.vertex_data()[0], flap)
    e.set_vertex_forces(np.tile(field, (B, 1)))
    e.rollout_forward(0, W + S)
    states = [e.get_state(W + s) for s in range(S + 1)]
    e.seed_gradient(W + S, None, 2.0 / ((S + 1) * e.N))
    carried = [e.get_gradient()[:2]]
    times = []
    for s in range(S):
        e.sync(); t0 = time.perf_counter()
        e.rollout_backward(W + S - s, 1)
        e.sync(); times.append(time.perf_counter() - t0)
        carried.append(e.get_gradient()[:2])
    stats = [e.get_stats(W + s + 1) for s in range(S)]
    bs_all = stats[-1][1]
    print(f"[c4] B={B} cluster {e.cluster()} backward step wall {['%.2f ms' % (1e3 * t) for t in times]}; bicgstab32 iters mean {bs_all['adjoint_iters'].mean():.1f} "
          f"cycles mean {bs_all['refine_cycles'].mean():.2f} max {bs_all['refine_cycles'].max()} fp64 iters max {bs_all['fp64_iters'].max()} "
          f"rel res max {bs_all['last_udiff'].max():.1e} converged {np.bincount(bs_all['converged'], minlength=3)}", flush=True)
    o = orc.Oracle(V, F, h=args.h, density=0.3, k_stretch=150.0, k_bend=1e-5, fwd_tol=args.fwd_tol, bwd_tol=args.bwd_tol,
                   selfcollision=True, gradient_clipping=True, threads=min(os.cpu_count() or 1, 32))
    o.add_sphere(center, 2.0, 0.9)
    o.build()
    o.set_force_extras(None, field, 1.0)
    errs = []
    recs = [e.get_record(W + s + 1) for s in range(S)] if DUMP else None
    for b in sample:
        o.set_mu(0, float(f32(MU[b, 0])))
        o.clear_records()
        for s in range(S):
            xs, vs = states[s]
            if DUMP:
                np.savez_compressed(os.path.join(DUMP, f"c4_b{b}_s{s}.npz"), x0=xs[b], v0=vs[b], x1=states[s + 1][0][b], v1=states[s + 1][1][b], f=recs[s][0][b],
                                    r=recs[s][1][b], gin_x=carried[S - 1 - s][0][b], gin_v=carried[S - 1 - s][1][b], gout_x=carried[S - s][0][b],
                                    gout_v=carried[S - s][1][b], mu=f32(MU[b, 0]))
            ref = o.step(xs[b], vs[b])
            fs, bs = stats[s]
            gin, gout = carried[S - 1 - s], carried[S - s]
            rb = o.step_backward(ref["id"], gin[0][b], gin[1][b], is_start=False, direct=True)
            ex, ev = rel(gout[0][b], rb["dL_dx"]), rel(gout[1][b], rb["dL_dv"])
            errs.append(max(ex, ev))
            print(f"[c4] rollout {b} step {W + s}: PD gpu {fs['pd_iters'][b]} / oracle {ref['iters']}, self {ref['nself']}, max|dx| {np.abs(states[s + 1][0][b] - ref['x']).max():.1e}, "
                  f"grad rel err dx {ex:.2e} dv {ev:.2e} | {bstat(bs, b)}", flush=True)
    print(f"[c4] worst {max(errs):.2e} median {np.median(errs):.2e}", flush=True)


def run_hat(fp32_only, B=64, sample=(0, 17, 63)):
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
    e = capi.Engine(0)
    e.set_mesh(P, F); e.set_attachments(att)
    e.set_params(time_step=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], forward_tol=1e-8, backward_tol=1e-9,
                 cg_rel_tol=1e-6, cg_max_iter=3000, gradient_clipping=0, selfcollision_enabled=0, adjoint_mode=1, adjoint_rel_tol=1e-7,
                 adjoint_fp32_only=int(fp32_only))
    e.set_primitives([dict(kind=capi.DC_PRIM_SPHERE, group=0, center=center, radius=cfg["sphere_radius"], mu=cfg["sphere_mu"])])
    e.build()
    rng = np.random.default_rng(2)
    x, v = f32(P.reshape(-1)), np.zeros(P.size)
    xf = P[att].reshape(-1).copy()
    for s in range(12):
        xf = xf + np.tile([0.0, -0.05, -0.3], 2)
        out = o.step(x, v, f32(xf)); x, v = out["x"], out["v"]
    X0 = np.stack([f32(x + 0.002 * rng.standard_normal(x.size)) for _ in range(B)])
    V0 = np.stack([f32(v + 0.01 * rng.standard_normal(x.size)) for _ in range(B)])
    XF = np.stack([f32(xf + np.tile([0.0, -0.05, -0.3], 2) + 0.02 * rng.standard_normal(6)) for _ in range(B)])
    mus = f32(rng.uniform(0.05, 0.6, (B, 1)))
    e.alloc_batch(B, 1)
    e.set_mu(mus)
    e.set_state(0, X0, V0)
    st = e.step_forward(0, fixed_pts=XF)
    rng = np.random.default_rng(4)
    gx = f32(rng.standard_normal(X0.shape)); gv = f32(rng.standard_normal(X0.shape) * 0.01)
    e.sync(); t0 = time.perf_counter()
    gb = e.step_backward(1, gx, gv, is_start=False)
    t1 = time.perf_counter() - t0
    print(f"[hat] B={B} backward wall {1e3 * t1:.1f} ms (incl. transfers); bicgstab32 mean {gb['adjoint_iters'].mean():.0f} cycles mean {gb['refine_cycles'].mean():.2f} "
          f"max {gb['refine_cycles'].max()} fp64 max {gb['fp64_iters'].max()} rel res max {gb['last_udiff'].max():.1e} converged {np.bincount(gb['converged'], minlength=3)}", flush=True)
    if DUMP:
        x1, v1 = e.get_state(1); fr = e.get_record(1)
        for b in sample:
            np.savez_compressed(os.path.join(DUMP, f"hat_b{b}.npz"), x0=X0[b], v0=V0[b], xf=XF[b], x1=x1[b], v1=v1[b], f=fr[0][b], r=fr[1][b], gin_x=gx[b], gin_v=gv[b],
                                gout_x=gb["dL_dx"][b], gout_v=gb["dL_dv"][b], gout_xf=gb["dL_dxfixed"][b], mu=mus[b, 0])
    for b in sample:
        o.set_mu(0, float(mus[b, 0]))
        ref = o.step(X0[b], V0[b], XF[b])
        rb = o.step_backward(ref["id"], gx[b], gv[b], is_start=False, direct=True)
        print(f"[hat] rollout {b}: PD gpu {st['pd_iters'][b]} / oracle {ref['iters']} contacts {ref['nprim']}, grad rel err dx {rel(gb['dL_dx'][b], rb['dL_dx']):.2e} "
              f"dv {rel(gb['dL_dv'][b], rb['dL_dv']):.2e} dxfixed {rel(gb['dL_dxfixed'][b], rb['dL_dxfixed']):.2e} | {bstat(gb, b)}", flush=True)


def run_dress7k(fp32_only, mesh="dress7k"):
    V, F = scenes.load_mesh(mesh)
    cfg = dict(h=1.0 / 120, density=0.2, k_stretch=800.0, k_bend=0.05)
    P, rmin, rmax = scenes.normalise_model(V, "FRONT", 8.0)
    P = f32(P)
    top = np.argsort(-P[:, 1])[:6].tolist()
    o = orc.Oracle(P, F, h=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], fwd_tol=1e-8,
                   bwd_tol=1e-9, attachments=top, selfcollision=True, contact=True, gradient_clipping=False, threads=min(os.cpu_count() or 1, 32))
    o.build()
    e = capi.Engine(0)
    e.set_mesh(P, F); e.set_attachments(top)
    e.set_params(time_step=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], forward_tol=1e-8, backward_tol=1e-9,
                 cg_rel_tol=1e-6, cg_max_iter=3000, gradient_clipping=0, selfcollision_enabled=1, adjoint_mode=1, adjoint_rel_tol=1e-7,
                 adjoint_fp32_only=int(fp32_only))
    e.set_primitives([])
    e.build()
    rng = np.random.default_rng(8)
    X = P.copy(); X[:, 2] *= 0.9
    vel = np.zeros_like(X); vel[:, 2] = -0.1 * np.sign(P[:, 2])
    x0 = f32((X + 0.0005 * rng.standard_normal(X.shape)).reshape(-1))[None, :]
    v0 = f32((vel + 0.005 * rng.standard_normal(X.shape)).reshape(-1))[None, :]
    xf = f32(X[top].reshape(-1))[None, :]
    e.alloc_batch(1, 1)
    e.set_state(0, x0, v0)
    st = e.step_forward(0, fixed_pts=xf)
    x1, v1 = e.get_state(1)
    ref = o.step(x0[0], v0[0], xf[0])
    print(f"[{mesh}] cluster {e.cluster()} self {st['self_contacts'][0]} / {ref['nself']} PD {st['pd_iters'][0]} / {ref['iters']} max|dx| {np.abs(x1[0] - ref['x']).max():.1e}", flush=True)
    gx = f32(rng.standard_normal(x0.shape)); gv = f32(0.01 * rng.standard_normal(x0.shape))
    e.sync(); t0 = time.perf_counter()
    gb = e.step_backward(1, gx, gv, is_start=False)
    t1 = time.perf_counter() - t0
    if DUMP:
        fr = e.get_record(1)
        np.savez_compressed(os.path.join(DUMP, f"{mesh}.npz"), x0=x0[0], v0=v0[0], xf=xf[0], x1=x1[0], v1=v1[0], f=fr[0][0], r=fr[1][0], gin_x=gx[0], gin_v=gv[0],
                            gout_x=gb["dL_dx"][0], gout_v=gb["dL_dv"][0], gout_xf=gb["dL_dxfixed"][0])
    rb = o.step_backward(ref["id"], gx[0], gv[0], is_start=False, direct=True)
    print(f"[{mesh}] backward wall {t1:.2f} s | {bstat(gb, 0)} | grad rel err dx {rel(gb['dL_dx'][0], rb['dL_dx']):.2e} dv {rel(gb['dL_dv'][0], rb['dL_dv']):.2e} "
          f"dxfixed {rel(gb['dL_dxfixed'][0], rb['dL_dxfixed']):.2e}", flush=True)


def main():
    what = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c4"]
    fp32_only = "--fp32-only" in sys.argv
    global DUMP
    if "--dump" in sys.argv:
        DUMP = sys.argv[sys.argv.index("--dump") + 1]
        os.makedirs(DUMP, exist_ok=True)
    print(f"lib {os.environ.get('DC_LIB', 'default')} fp32_only {fp32_only} DC_CLUSTER {os.environ.get('DC_CLUSTER')}", flush=True)
    for w in what:
        t0 = time.time()
        {"c4": run_c4, "hat": run_hat, "dress7k": run_dress7k, "dress": lambda f: run_dress7k(f, "dress")}[w](fp32_only)
        print(f"[{w}] done in {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
