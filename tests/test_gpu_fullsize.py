"""GPU parity at BASELINE.json's full size (config C4: 100 x 100 grid cloth, N = 10 000, 256 rollouts on one GPU — the
bench.py workload) and beyond the packet kernels' size limit (N = 16 384 > 10 240: the global-memory kernels).

At these sizes the fp64 oracle checks two sampled rollouts; the whole batch is covered by size-independent properties:
  * rollouts are independent and the kernels deterministic: two batch slots fed the same input give bit-identical
    states and gradients, whatever their neighbours do;
  * the backward step is linear in the incoming gradient (adjoint of the linearised step, Simulation.cpp:1455-1780);
  * every rollout converges and reports finite results.
"""
import numpy as np
import pytest

import meshes
import orc
from diffcloth_amd import capi

pytestmark = pytest.mark.gpu

H = 1.0 / 180
FABRIC = dict(density=0.3, k_stretch=150.0, k_bend=1e-5)      # sphereFabric, OptimizationTaskConfigurations.cpp:81-96


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def scene(nx, selfcollision, fwd_tol):
    V, F = meshes.grid_cloth(nx, nx, 4.5, 4.5, "DOWN")
    V = f32(V)
    c = f32(meshes.sphere_scene_center(V, 2.0))
    e = capi.Engine(0)
    e.set_mesh(V, F)
    e.set_params(time_step=H, forward_tol=fwd_tol, backward_tol=1e-9, cg_rel_tol=1e-6, cg_max_iter=3000, gradient_clipping=0,
                 selfcollision_enabled=int(selfcollision), adjoint_mode=1, adjoint_rel_tol=1e-8, **FABRIC)
    e.set_primitives([dict(kind=capi.DC_PRIM_SPHERE, group=0, center=c, radius=2.0, mu=0.9)])
    e.build()
    o = orc.Oracle(V, F, h=H, fwd_tol=fwd_tol, bwd_tol=1e-9, selfcollision=bool(selfcollision), gradient_clipping=False, **FABRIC)
    o.add_sphere(c, 2.0, 0.9)
    o.build()
    return V, F, e, o


def start_states(V, B, twins):
    """bench.py's per-rollout start: the cloth shifted over the sphere, own friction coefficient; `twins` = pairs of
    batch slots that get the same input."""
    X = np.empty((B, V.size)); MU = np.empty((B, 1))
    for b in range(B):
        rng = np.random.default_rng(1000 + b)
        shift = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.09, -0.02), rng.uniform(-0.5, 0.5)])
        X[b] = f32((V + shift).reshape(-1)); MU[b, 0] = rng.uniform(0.1, 0.9)
    for a, b in twins:
        X[b] = X[a]; MU[b] = MU[a]
    return X, f32(MU)


def check_against_oracle(o, e, step, sample, MU, gx, gv, st, gb, pos_tol, grad_tol):
    xs, vs = e.get_state(step)
    x1, v1 = e.get_state(step + 1)
    for b in sample:
        o.set_mu(0, float(MU[b, 0]))
        ref = o.step(xs[b], vs[b])
        assert ref["converged"] and st["prim_contacts"][b] == ref["nprim"] and st["self_contacts"][b] == ref["nself"]
        rb = o.step_backward(ref["id"], gx[b], gv[b], is_start=False, direct=True)
        dx = np.abs(x1[b] - ref["x"]).max()
        ex, ev = rel(gb["dL_dx"][b], rb["dL_dx"]), rel(gb["dL_dv"][b], rb["dL_dv"])
        print(f"\n[full size] rollout {b}: contacts {ref['nprim']}, PD iterations gpu {st['pd_iters'][b]} / oracle {ref['iters']}, "
              f"max|dx| {dx:.2e}, gradient rel err dx {ex:.2e} dv {ev:.2e}")
        assert dx <= pos_tol
        assert ex <= grad_tol and ev <= grad_tol, (b, ex, ev)      # plain gate (round 2 excused the 8 worst vertices up to 5e-3 here)


def test_c4_10k_vertices_batch_256():
    B, W = 256, 5
    V, F, e, o = scene(100, selfcollision=True, fwd_tol=1e-8)
    assert e.N == 10000
    twins = [(3, 200), (0, 255)]
    X0, MU = start_states(V, B, twins)
    e.alloc_batch(B, W + 1)
    e.set_mu(MU)
    e.set_state(0, X0, np.zeros_like(X0))
    e.rollout_forward(0, W)                                   # contact onset, all steps of a rollout in one launch
    st = e.step_forward(W)
    rng = np.random.default_rng(11)
    g1x = f32(rng.standard_normal(X0.shape)); g1v = f32(0.01 * rng.standard_normal(X0.shape))
    g2x = f32(rng.standard_normal(X0.shape)); g2v = f32(0.01 * rng.standard_normal(X0.shape))
    for a, b in twins:
        for g in (g1x, g1v, g2x, g2v):
            g[b] = g[a]
    gb1 = e.step_backward(W + 1, g1x, g1v, is_start=False)
    gb2 = e.step_backward(W + 1, g2x, g2v, is_start=False)
    gb3 = e.step_backward(W + 1, f32(g1x + 2 * g2x), f32(g1v + 2 * g2v), is_start=False)
    assert np.all(np.isin(st["converged"], (1, 2))) and np.all(np.isin(gb1["converged"], (1, 2)))
    assert st["prim_contacts"].min() > 0, "every rollout touches the sphere after the warm-up steps"
    x1, v1 = e.get_state(W + 1)
    assert np.isfinite(x1).all() and np.isfinite(v1).all() and np.isfinite(gb1["dL_dx"]).all()
    # batch slots are independent and deterministic
    for a, b in twins:
        assert np.array_equal(x1[a], x1[b]) and np.array_equal(v1[a], v1[b])
        assert np.array_equal(gb1["dL_dx"][a], gb1["dL_dx"][b]) and np.array_equal(gb1["dL_dv"][a], gb1["dL_dv"][b])
        assert st["pd_iters"][a] == st["pd_iters"][b] and st["cg_iters"][a] == st["cg_iters"][b]
    # linearity of the adjoint step over the whole batch (seed g1 + 2 g2 is rounded to fp32: 1e-7 relative)
    worst = 0.0
    for key in ("dL_dx", "dL_dv"):
        lin = gb1[key] + 2 * gb2[key]
        err = np.linalg.norm(gb3[key] - lin, axis=1) / np.linalg.norm(lin, axis=1)
        worst = max(worst, err.max())
    print(f"\n[full size] B={B} N={e.N}: PD iterations {st['pd_iters'].min()}..{st['pd_iters'].max()}, contacts "
          f"{st['prim_contacts'].min()}..{st['prim_contacts'].max()}, adjoint linearity worst rel err {worst:.2e}")
    assert worst <= 2e-5
    check_against_oracle(o, e, W, (0, 137), MU, g1x, g1v, st, gb1, pos_tol=5e-5, grad_tol=1e-4)


def test_mesh_beyond_the_packet_kernel_limit():
    """128 x 128 grid: N = 16 384 vertices do not fit the LDS-resident kernels (N <= 10 240); the engine must take its
    global-memory kernels by itself and stay within the same tolerances."""
    B, W = 3, 4
    V, F, e, o = scene(128, selfcollision=False, fwd_tol=1e-8)
    assert e.N == 16384
    X0, MU = start_states(V, B, [])
    e.alloc_batch(B, W + 1)
    e.set_mu(MU)
    e.set_state(0, X0, np.zeros_like(X0))
    for s in range(W):
        e.step_forward(s)
    st = e.step_forward(W)
    rng = np.random.default_rng(12)
    gx = f32(rng.standard_normal(X0.shape)); gv = f32(0.01 * rng.standard_normal(X0.shape))
    gb = e.step_backward(W + 1, gx, gv, is_start=False)
    assert np.all(np.isin(st["converged"], (1, 2))) and np.all(np.isin(gb["converged"], (1, 2)))
    assert st["prim_contacts"].min() > 0
    check_against_oracle(o, e, W, (1,), MU, gx, gv, st, gb, pos_tol=5e-5, grad_tol=1e-4)


@pytest.mark.parametrize("nx,vpt", [(22, 1), (32, 2), (39, 3), (45, 4), (55, 6), (64, 8), (71, 10), (78, 12), (90, 16)])
def test_every_rows_per_thread_instance_of_the_packet_kernel(nx, vpt, monkeypatch):
    """k_pd_step_pk<512, VPT, XL> is instantiated for 1 ... 20 rows per thread (N <= 512 VPT); small batches of large meshes are
    split over several workgroups by default, so the one-workgroup instances between the demo meshes' sizes and N = 10 000 are
    pinned here (DC_CLUSTER=1): one step forward + backward of a draped, contacting cloth against the fp64 oracle."""
    monkeypatch.setenv("DC_CLUSTER", "1")
    monkeypatch.setenv("DC_DENSE_MAX_N", "0")            # the packet solve itself, not the explicit inverse of the smallest meshes
    V, F, e, o = scene(nx, False, 1e-8)
    prev = {1: 0, 2: 1, 3: 2, 4: 3, 6: 4, 8: 6, 10: 8, 12: 10, 16: 12}[vpt]
    assert 512 * prev < e.N <= 512 * vpt
    lay = e.layout()
    assert lay["packet_kernel"] and lay["element_windows"]
    X, MU = start_states(V, 2, [])
    X[:, 1::3] -= 0.12                                    # lower the sheet onto the sphere: contacts in the very first step
    X = f32(X)
    e.alloc_batch(2, 1)
    assert e.cluster() == 1
    e.set_mu(MU)
    e.set_state(0, X, np.zeros_like(X))
    st = e.step_forward(0)
    x1, v1 = e.get_state(1)
    rng = np.random.default_rng(nx)
    gx = f32(rng.standard_normal(X.shape)); gv = f32(0.01 * rng.standard_normal(X.shape))
    gb = e.step_backward(1, gx, gv, is_start=True)
    o.set_mu(0, float(MU[1, 0]))
    ref = o.step(X[1], np.zeros_like(X[1]))
    rb = o.step_backward(ref["id"], gx[1], gv[1], is_start=True, direct=True)
    dx = np.abs(x1[1] - ref["x"]).max()
    print(f"\n[pk VPT={vpt}] N={e.N}: contacts {st['prim_contacts'][1]} / {ref['nprim']}, PD iterations {st['pd_iters'][1]} / {ref['iters']}, PCG per PD iteration "
          f"{st['cg_iters'][1] / max(st['pd_iters'][1], 1):.1f}, max|dx| {dx:.2e}, gradient rel err dx {rel(gb['dL_dx'][1], rb['dL_dx']):.2e} dv {rel(gb['dL_dv'][1], rb['dL_dv']):.2e}")
    assert st["converged"][1] == 1 and ref["converged"] and st["prim_contacts"][1] == ref["nprim"] and ref["nprim"] > 0
    assert abs(int(st["pd_iters"][1]) - ref["iters"]) <= 1
    assert dx <= 4.5e-5
    assert rel(gb["dL_dx"][1], rb["dL_dx"]) <= 1e-4 and rel(gb["dL_dv"][1], rb["dL_dv"]) <= 1e-4


@pytest.mark.parametrize("B,split", [(8, False), (8, True)], ids=["one-workgroup", "split"])
def test_cg_first_and_bicgstab_correction_solves_give_the_same_adjoint(monkeypatch, B, split):
    """Round 6: the correction solves of the mixed-precision direct adjoint solve are CG first (diag(P) preconditioner instances), BiCGSTAB once a CG
    cycle stalls or fails to contract the fp64 residual (dc_adjoint.hip: cg32_solve; the split kernel: dc_adjoint_cl.hip). What makes CG admissible on
    the slightly non-symmetric K is the refinement around it — so the RESULT must not depend on the inner method: same step, DC_ADJ_CG=1 against
    DC_ADJ_CG=0, both converged to 1e-8 in the fp64-evaluated residual, gradients equal to 1e-6; the statistics say which method ran."""
    monkeypatch.setenv("DC_CLUSTER", "8" if split else "1")
    V, F, e, o = scene(64, selfcollision=False, fwd_tol=1e-8)
    e.set_params(adjoint_block_precond=0); e.build()
    X, MU = start_states(V, B, twins=())
    e.alloc_batch(B, 6)
    assert e.cluster() == (8 if split else 1)
    e.set_mu(MU)
    e.set_state(0, X, np.zeros_like(X))
    e.rollout_forward(0, 6)
    rng = np.random.default_rng(11)
    gx = f32(rng.standard_normal(X.shape)); gv = f32(0.01 * rng.standard_normal(X.shape))
    out = {}
    for cg in ("1", "0"):
        monkeypatch.setenv("DC_ADJ_CG", cg)
        out[cg] = e.step_backward(6, gx, gv, is_start=False)
        assert np.all(out[cg]["converged"] == 1) and out[cg]["last_udiff"].max() <= 1.01e-8
    a, b = out["1"], out["0"]
    assert a["cg_iters"].min() > 0 and b["cg_iters"].max() == 0 and b["adjoint_iters"].min() > 0
    apps_cg = (a["cg_iters"] + 2 * a["adjoint_iters"]).mean(); apps_bi = (2 * b["adjoint_iters"]).mean()
    ex, ev = rel(a["dL_dx"], b["dL_dx"]), rel(a["dL_dv"], b["dL_dv"])
    print(f"\n[CG-first vs BiCGSTAB, {'split x 8' if split else 'one workgroup'}] operator applications per step {apps_cg:.1f} / {apps_bi:.1f}; "
          f"gradients differ by dx {ex:.2e} dv {ev:.2e}; workgroups {a['workgroups'][0]}")
    assert ex <= 1e-6 and ev <= 1e-6
    assert np.all(a["workgroups"] == (8 if split else 1))
