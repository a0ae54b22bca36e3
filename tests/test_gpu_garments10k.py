"""The reference's own 10k-class meshes (SURVEY.md section 8d names them as the alternatives for the "10k-vertex" metric; VERDICT r04 "missing" 1):

  * `Slope/perfFabric4-96x96-onPlane.obj` — 9 216 vertices, the 96 x 96 performance fabric lying on the slope plane: every vertex in sliding
    Coulomb contact with a Plane primitive (Primitive.cpp:67-130), the one-workgroup kernels at 20 rows per thread;
  * `dress-v17k-f34k.obj` — 17 562 vertices, 34 099 triangles, self contacts, clips: matrix bandwidth 647 after renumbering, i.e. beyond the
    packet tables — the general (global-memory) forward kernel and the one-workgroup adjoint with its two-level fp64 fall-back.

Each: one forward step against the fp64 oracle (contact sets, PD iteration count, positions) and its adjoint — END TO END and on the SAME
RECORD in both directions (tests/records.py), flat 1e-4 on the same record. The oracle's adjoint system is solved with a sparse LU of the
explicit K (the reference's solveDirect, Simulation.cpp:1431-1440), residual asserted."""
import os

import numpy as np
import pytest

import meshes
import orc
import ledger
import records
import scenes
from diffcloth_amd import capi

pytestmark = pytest.mark.gpu


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def slope_plane(P):
    """The plane the fabric lies on, as the reference's Plane(center, upperLeft, upperRight) (corners relative to the centre, normal =
    upperRight x upperLeft, Primitive.cpp:13-40): fitted through the vertices, a 7.2 x 7.2 rectangle 0.02 under the 6 x 6 sheet."""
    c0 = P.mean(axis=0)
    n = np.linalg.svd(P - c0)[2][2]
    if n[1] < 0:
        n = -n
    ex = np.array([1.0, 0.0, 0.0])
    ex = ex - n * (ex @ n); ex /= np.linalg.norm(ex)
    es = np.cross(n, ex)                         # ex x es = n
    flat = float(np.abs((P - c0) @ n).max())
    c = c0 - 0.02 * n
    return f32(c), f32(-3.6 * ex + 3.6 * es), f32(3.6 * ex + 3.6 * es), n, flat


def compare_step(o, e, x0, v0, xf, tag, pos_tol, min_prim=0, min_self=0):
    """forward + backward of ONE rollout (batch of 1) against the oracle; returns the engine's forward statistics"""
    h = float(o.params["h"])
    e.alloc_batch(1, 1)
    e.set_state(0, x0[None, :], v0[None, :])
    st = e.step_forward(0, fixed_pts=None if xf is None else xf[None, :])
    x1, v1 = e.get_state(1)
    ref = o.step(x0, v0, xf)
    dx = np.abs(x1[0] - ref["x"]).max()
    print(f"\n[{tag}] N = {e.N}, {e.cluster()} workgroup(s), deflation vectors {e.deflation()[0]} (probe {e.deflation()[1]} iterations): contacts prim "
          f"{st['prim_contacts'][0]} / {ref['nprim']}, self {st['self_contacts'][0]} / {ref['nself']}, PD iterations {st['pd_iters'][0]} / {ref['iters']}, "
          f"PCG per PD iteration {st['cg_iters'][0] / max(st['pd_iters'][0], 1):.0f}, max|dx| {dx:.2e}")
    assert st["converged"][0] == 1 and ref["converged"]
    assert st["prim_contacts"][0] == ref["nprim"] >= min_prim and st["self_contacts"][0] == ref["nself"] >= min_self
    assert abs(int(st["pd_iters"][0]) - int(ref["iters"])) <= 2
    assert dx <= pos_tol
    rng = np.random.default_rng(4)
    gx = f32(rng.standard_normal(x0.shape)); gv = f32(0.01 * rng.standard_normal(x0.shape))
    gb = e.step_backward(1, gx[None, :], gv[None, :], is_start=False)
    assert gb["converged"][0] == 1 and np.isfinite(gb["dL_dx"]).all()
    rb = o.step_backward_lu(ref["id"], gx, gv, is_start=False)
    assert rb["lu_residual"] <= 1e-10

    def errs(g, r):
        out = [rel(g["dL_dx"][0], r["dL_dx"]), rel(g["dL_dv"][0], r["dL_dv"])]
        if xf is not None:
            out.append(rel(g["dL_dxfixed"][0], r["dL_dxfixed"]))
        if o.nprim > 0:
            out.append(records.mu_err(g["dL_dmu"][0], r["dL_dmu"][:g["dL_dmu"].shape[1]]))
        return max(out)
    ee = errs(gb, rb)
    rec = records.oracle_record(o, ref)                       # (before the adoption overrides the oracle's record)
    fr = e.get_record(1)
    nrm = e.get_contacts(1)[1]
    matched = records.oracle_adopts_gpu_record(o, ref["id"], e, 1, 0, x0, x1[0], v1[0], fr[0][0], h, normals=nrm)
    assert matched == ref["nself"]
    ea = errs(gb, o.step_backward_lu(ref["id"], gx, gv, is_start=False))
    records.upload_oracle_records(e, 1, [rec], x_fixed=None if xf is None else xf[None, :])
    gt = e.step_backward(1, gx[None, :], gv[None, :], is_start=False)
    assert gt["converged"][0] == 1
    et = errs(gt, rb)
    # the oracle's own sensitivity to a float32 rounding of its x_new: what END TO END can mean on this step
    o.override_record(ref["id"], x=f32(ref["x"]))
    sens = errs({k: np.asarray(v)[None] for k, v in o.step_backward_lu(ref["id"], gx, gv, is_start=False).items() if k.startswith("dL_")}, rb)
    print(f"[{tag}] adjoint: BiCGSTAB {gb['adjoint_iters'][0]} in {gb['refine_cycles'][0]} fp32 solves (+ {gb['fp64_iters'][0]} fp64 iterations), residual "
          f"{gb['last_udiff'][0]:.1e}; gradient rel err END TO END {ee:.2e} (the oracle's own under a float32 rounding of its x_new: {sens:.2e}) | SAME RECORD: "
          f"oracle adopts the engine's {ea:.2e}, engine differentiates the oracle's {et:.2e}")
    assert ea <= 1e-4 and et <= 1e-4
    gate = max(1e-4, min(3 * sens, 8e-3))      # (the conditioning rule and hard ceiling of tests/test_gpu_configs.py::check_rollouts)
    ledger.add("test_gpu_garments10k.check_step", tag, 0, ee, sensitivity=sens, gate=gate, same_record_adopt=ea, same_record_forced=et)
    assert ee <= gate
    return st


def test_perf_fabric_96x96_sliding_on_the_slope_plane():
    """9 216 vertices, all of them in contact with the slope (42 degrees, mu = 0.2 as Simulation.cpp:1958: tan 42 = 0.9 > mu, the sheet slides)."""
    V, F = scenes.load_mesh("perf96")
    P = f32(V)
    c, ul, ur, n, flat = slope_plane(P)
    assert flat < 0.1 and 0.6 < n[1] < 0.9          # a (slightly relaxed) flat sheet on a steep slope
    cfg = dict(h=1.0 / 100, density=0.2, k_stretch=50.0, k_bend=1e-5)      # slopeFabricRestOnPlane (OptimizationTaskConfigurations.cpp:98-111), slope scene time step
    mu = 0.2
    o = orc.Oracle(P, F, h=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], fwd_tol=1e-8, bwd_tol=1e-9,
                   selfcollision=False, contact=True, gradient_clipping=False, threads=min(os.cpu_count() or 1, 32))
    o.add_plane(c, ul, ur, mu)
    o.build()
    e = capi.Engine(0)
    e.set_mesh(P, F)
    e.set_params(time_step=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], forward_tol=1e-8, backward_tol=1e-9,
                 cg_rel_tol=1e-6, cg_max_iter=3000, gradient_clipping=0, selfcollision_enabled=0, adjoint_mode=1, adjoint_rel_tol=1e-7)
    e.set_primitives([dict(kind=capi.DC_PRIM_PLANE, group=0, center=c, top_offset=ul, corner2=ur, radius=0.0, mu=mu)])
    e.build()
    lay = e.layout()
    assert lay["packet_kernel"] and lay["element_windows"]
    # two steps of the oracle first: the sheet is moving down the slope when the compared step starts
    x, v = P.reshape(-1).copy(), np.zeros(P.size)
    for _ in range(2):
        out = o.step(x, v); x, v = f32(out["x"]), f32(out["v"])
    os.environ["DC_CLUSTER"] = "1"      # one workgroup per rollout: what a batch of 256 gets
    try:
        compare_step(o, e, x, v, None, "perfFabric 96x96 on the slope", pos_tol=5e-5, min_prim=9000)
    finally:
        del os.environ["DC_CLUSTER"]


def test_dress_17562_vertices_self_contacts_and_clips():
    """The reference's largest garment. After the engine's reverse Cuthill-McKee renumbering its matrix bandwidth is 647 (rings of ~320 vertices, the
    bending stencil spans two): beyond the +-511 of the packet tables' column deltas, so this mesh has NO packet / split kernels and runs the
    general ones — forward step in global memory (Jacobi-PCG: 355 iterations per PD iteration until round 6 gave this kernel the deflation
    projection as well), adjoint through the element windows on one
    workgroup, its fp64 fall-back preconditioned on two levels (the deflation space is built for the adjoint alone on such a mesh). Parity,
    not throughput: ~650 self contacts at the garment's fine regions, six clips, squashed pose (z scaled by 0.97, sheets closing at 0.1),
    152 PD iterations."""
    V, F = scenes.load_mesh("dress17k")
    cfg = dict(h=1.0 / 120, density=0.3, k_stretch=3000.0, k_bend=0.3)      # dressScene's fabric (OptimizationTaskConfigurations.cpp:115-129), clothDim 13
    P, rmin, rmax = scenes.normalise_model(V, "FRONT", 13.0)
    P = f32(P)
    top = np.argsort(-P[:, 1])[:6].tolist()
    o = orc.Oracle(P, F, h=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], fwd_tol=1e-8, bwd_tol=1e-9,
                   attachments=top, selfcollision=True, contact=True, gradient_clipping=False, threads=min(os.cpu_count() or 1, 32))
    o.build()
    e = capi.Engine(0)
    e.set_mesh(P, F); e.set_attachments(top)
    e.set_params(time_step=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], forward_tol=1e-8, backward_tol=1e-9,
                 cg_rel_tol=1e-6, cg_max_iter=3000, gradient_clipping=0, selfcollision_enabled=1, adjoint_mode=1, adjoint_rel_tol=1e-7)
    e.set_primitives([]); e.build()
    lay = e.layout()
    assert lay["renumbered"] and lay["bandwidth"] > 511 and not lay["packet_kernel"] and lay["element_windows"]
    assert e.deflation()[0] == 16          # for the adjoint's coarse level
    X = P.copy(); X[:, 2] *= 0.97
    vel = np.zeros_like(X); vel[:, 2] = -0.1 * np.sign(P[:, 2])
    x0, v0, xf = f32(X.reshape(-1)), f32(vel.reshape(-1)), f32(X[top].reshape(-1))
    st = compare_step(o, e, x0, v0, xf, "dress 17 562", pos_tol=1e-4, min_self=300)
    # round 6: the global-memory forward kernel projects onto the deflation space too (dc_devlib.h: deflate_global) — before, Jacobi-PCG needed 355
    # iterations per PD iteration on this mesh
    per_pd = st["cg_iters"][0] / st["pd_iters"][0]
    print(f"[dress 17 562] PCG iterations per PD iteration with the 16-vector deflation space: {per_pd:.0f} (355 without)")
    assert per_pd <= 150
    # a forward throughput figure (8 copies of the compared state — perturbed copies of this garment's squashed pose blow up, in the oracle as well:
    # DESIGN.md section 8 — one workgroup per rollout: this mesh has no split kernels)
    B = 8
    e.alloc_batch(B, 1)
    e.set_state(0, np.tile(x0, (B, 1)), np.tile(v0, (B, 1)))
    e.timer_start()
    q = e.step_forward(0, fixed_pts=np.tile(xf, (B, 1)))
    ms = e.timer_stop()
    print(f"[dress 17 562] {B} rollouts: forward step {ms:.0f} ms ({q['pd_iters'].mean():.0f} PD iterations of {q['cg_iters'].sum() / q['pd_iters'].sum():.0f} PCG) "
          f"-> {B / (ms * 1e-3):.1f} rollout-steps/s forward on {B} of 256 CUs")
    assert np.all(q["converged"] == 1) and np.all(q["pd_iters"] == st["pd_iters"][0])
