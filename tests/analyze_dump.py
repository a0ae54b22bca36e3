"""Which recorded quantity does a GPU-vs-oracle gradient difference come from? (TEST INFRASTRUCTURE, CPU only.)

Reads the HIP path's tape of sampled rollouts (tests/ab_adjoint.py --dump DIR), repeats the step with the fp64 oracle and
differentiates it four times: with its own record, with the GPU's x_new, with the GPU's f (contact vectors d, r re-derived), with
both. If "both" reproduces the GPU's gradient, the adjoint SOLVE is exact and what differs is the forward record.
  python tests/analyze_dump.py DIR c4|hat|hat6|dress7k [file pattern]
"""
import glob
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench    # noqa: E402
import orc      # noqa: E402
import scenes   # noqa: E402


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def oracle_c4():
    args = types.SimpleNamespace(grid=100, fold_rows=5, fold_gap=0.02, flap_force=2.0, h=1.0 / 180, fwd_tol=1e-8, bwd_tol=5e-4)
    V, F, V0, flap, center = bench.scene(args)
    o = orc.Oracle(V, F, h=args.h, density=0.3, k_stretch=150.0, k_bend=1e-5, fwd_tol=args.fwd_tol, bwd_tol=args.bwd_tol,
                   selfcollision=True, gradient_clipping=True, threads=min(os.cpu_count() or 1, 32))
    o.add_sphere(center, 2.0, 0.9)
    o.build()
    o.set_force_extras(None, bench.flap_force(args, o.vertex_data()[0], flap), 1.0)
    return o


def oracle_hat(fwd_tol=1e-8):
    cfg = scenes.HAT
    V, F = scenes.load_mesh("hat")
    P, rmin, rmax = scenes.normalise_model(V, cfg["orientation"], cfg["cloth_dim"])
    P = f32(P)
    center = f32(scenes.hat_head_center(rmin, rmax, cfg["sphere_radius"]))
    o = orc.Oracle(P, F, h=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], fwd_tol=fwd_tol,
                   bwd_tol=1e-9, attachments=cfg["attachments"], selfcollision=False, gradient_clipping=False)
    o.add_sphere(center, cfg["sphere_radius"], cfg["sphere_mu"])
    o.build()
    return o


def oracle_dress(mesh):
    V, F = scenes.load_mesh(mesh)
    P, rmin, rmax = scenes.normalise_model(V, "FRONT", 8.0)
    P = f32(P)
    top = np.argsort(-P[:, 1])[:6].tolist()
    o = orc.Oracle(P, F, h=1.0 / 120, density=0.2, k_stretch=800.0, k_bend=0.05, fwd_tol=1e-8, bwd_tol=1e-9, attachments=top,
                   selfcollision=True, contact=True, gradient_clipping=False, threads=min(os.cpu_count() or 1, 32))
    o.build()
    return o


def main():
    d, which = sys.argv[1], sys.argv[2]
    o = {"c4": oracle_c4, "hat": oracle_hat, "hat6": lambda: oracle_hat(1e-6), "dress7k": lambda: oracle_dress("dress7k")}[which]()
    pattern = sys.argv[3] if len(sys.argv) > 3 else which + "*.npz"      # e.g. "cfg_B64_N579_*.npz": the DC_DUMP_DIR files of tests/test_gpu_configs.py
    for fn in sorted(glob.glob(os.path.join(d, pattern))):
        z = np.load(fn)
        if "mu" in z.files:
            o.set_mu(0, float(z["mu"]))
        o.clear_records()
        ref = o.step(z["x0"], z["v0"], z["xf"] if "xf" in z.files else None)
        fo, ro = o.record_fr(ref["id"])
        print(f"{os.path.basename(fn)}: PD {ref['iters']} prim {ref['nprim']} self {ref['nself']} | GPU vs oracle record: max|dx| {np.abs(z['x1'] - ref['x']).max():.1e} "
              f"rel |df| {rel(z['f'], fo):.1e} rel |dr| {rel(z['r'], ro):.1e}")
        outs = {}
        for name, kw in (("own record", {}), ("GPU x_new", dict(x=z["x1"])), ("GPU f", dict(f=z["f"])), ("GPU x_new + f", dict(x=z["x1"], f=z["f"]))):
            o.override_record(ref["id"], x=ref["x"], f=fo)
            if kw:
                o.override_record(ref["id"], **kw)
            rb = o.step_backward(ref["id"], z["gin_x"], z["gin_v"], is_start=False, direct=True)
            outs[name] = rb
            print(f"    oracle adjoint with {name:14s}: GPU gradient vs it: dx {rel(z['gout_x'], rb['dL_dx']):.2e} dv {rel(z['gout_v'], rb['dL_dv']):.2e}"
                  + (f" dxfixed {rel(z['gout_xf'], rb['dL_dxfixed']):.2e}" if "gout_xf" in z.files else "")
                  + f" | it vs own record: dx {rel(rb['dL_dx'], outs['own record']['dL_dx']):.2e}")


if __name__ == "__main__":
    main()
