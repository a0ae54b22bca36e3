"""Device-resident schedules (dc_set_fixed_point_schedule / dc_set_force_schedule / dc_set_seed_schedule): what the host loop of
Simulation::runBackwardTask feeds into every step (stepFixPoints targets, fillForces terms, per-frame loss gradients; reference
Simulation.cpp:55-116, 964-1018, 3938-3952) uploaded once, so that a loss + gradient evaluation is two launches. The fused,
scheduled sweeps must reproduce the per-step calls that receive the same values as arguments: bitwise in the forward direction
(same kernels, same fp32 inputs), to solver tolerance in the backward direction.
"""
import numpy as np
import pytest
import torch          # before the engine's library: both must share ONE HIP runtime (torch's), see tests/test_gpu_functional.py

import meshes
from diffcloth_amd import capi

pytestmark = pytest.mark.gpu


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-30)


def scene(nx, att, selfcollision=0):
    V, F = meshes.grid_cloth(nx, nx, 4.5, 4.5, "DOWN")
    V = f32(V)
    c = f32(meshes.sphere_scene_center(V, 2.0))
    e = capi.Engine(0)
    e.set_mesh(V, F)
    e.set_attachments(att)
    e.set_params(time_step=1 / 180, density=0.3, k_stretch=150.0, k_bend=0.05, forward_tol=1e-7, backward_tol=1e-7, cg_rel_tol=1e-5,
                 cg_max_iter=2000, gradient_clipping=0, selfcollision_enabled=selfcollision, adjoint_mode=1, adjoint_rel_tol=1e-7)
    e.set_primitives([dict(kind=capi.DC_PRIM_SPHERE, group=0, center=c, radius=2.0, mu=0.4)])
    e.build()
    return V, F, e


def streams(V, att, B, S, rng):
    """per-step fixed-point targets (moving clips), uniform forces and factors on the per-vertex force (exact in fp32)"""
    N = V.shape[0]
    top = V[list(att)]
    XF = np.stack([np.stack([f32((top + np.array([0.01 * (s + 1) * (b + 1), 0.02 * (s + 1), 0.0])).reshape(-1)) for b in range(B)]) for s in range(S)])
    FU = f32(0.02 * rng.standard_normal((S, B, 3)))
    FV = f32(0.001 * rng.standard_normal((B, 3 * N)))
    FVS = np.array([[0.5, 1.0, 0.25, 2.0][(s + b) % 4] for s in range(S) for b in range(B)]).reshape(S, B)     # powers of two: exact under FMA contraction
    return XF, FU, FV, FVS


@pytest.mark.parametrize("nx,cluster", [(17, 0), (48, 4)])
def test_scheduled_rollout_equals_per_step_calls(nx, cluster, monkeypatch):
    """forward: fixed points + uniform force + scaled per-vertex force per step; backward: per-frame loss seeds, dL_dxfixed per slot"""
    if cluster:
        monkeypatch.setenv("DC_CLUSTER", str(cluster))
    rng = np.random.default_rng(5)
    att = (0, nx - 1)
    B, S = 2, 5
    V, F, e = scene(nx, att)
    N = V.shape[0]
    e.alloc_batch(B, S)
    assert e.cluster() == (cluster if cluster else 1)
    XF, FU, FV, FVS = streams(V, att, B, S, rng)
    X0 = np.stack([f32(V.reshape(-1)), f32(V.reshape(-1) + np.tile([0.05, 0.0, 0.02], N))])
    V0 = np.zeros_like(X0)

    # a constant per-vertex force field next to the scaled per-vertex term: the SECOND per-vertex term of fillForces (Simulation.cpp:91-93),
    # factor 1 in every step, per step and inside the fused rollout (dc_set_vertex_force_field)
    FF = f32(0.0005 * rng.standard_normal((B, 3 * N)))
    e.set_vertex_force_field(FF)
    # ---- (a) per-step calls: every value handed over as an argument of its step
    e.set_state(0, X0, V0)
    for s in range(S):
        e.set_uniform_force(FU[s])
        e.set_vertex_forces(FV * FVS[s][:, None])
        e.step_forward(s, fixed_pts=XF[s])
    xa, va = e.get_states(0, S + 1)
    seeds = f32(1e-3 * rng.standard_normal((S + 1, B, 3 * N)))          # loss gradient w.r.t. the state of every slot
    gx, gv = seeds[S].copy(), np.zeros((B, 3 * N))
    dxf_a, par_a = {}, {}
    dmu_a = np.zeros((B, 1))
    for s in range(S, 0, -1):
        out = e.step_backward(s, gx, gv, dL_dxinit=seeds[s - 1], dL_dvinit=np.zeros((B, 3 * N)), is_start=(s == 1))
        gx, gv = out["dL_dx"], out["dL_dv"]
        dxf_a[s] = out["dL_dxfixed"].copy(); par_a[s] = e.get_param_gradients(s); dmu_a += out["dL_dmu"]

    # ---- (b) schedules + two launches
    e.set_uniform_force(None)
    e.set_vertex_forces(FV)                                             # the factor-free field; the schedule carries the factors
    e.set_state(0, X0, V0)
    e.set_fixed_point_schedule(0, XF)
    e.set_force_schedule(0, S, fu=FU, fv_scale=FVS)
    e.rollout_forward(0, S)
    xb, vb = e.get_states(0, S + 1)
    np.testing.assert_array_equal(xa, xb)
    np.testing.assert_array_equal(va, vb)
    e.set_seed_schedule(0, seeds[:S])
    e.set_gradient(seeds[S], np.zeros((B, 3 * N)))
    e.rollout_backward(S, S)
    dx, dv, dmu = e.get_gradient()
    dxf_b = e.get_dxfixed(1, S)
    print(f"\n[schedules nx={nx} K={e.cluster()}] fused vs per-step: dL_dx {rel(dx, gx):.2e} dL_dv {rel(dv, gv):.2e} dmu {rel(dmu, dmu_a):.2e} "
          f"dxfixed {max(rel(dxf_b[s - 1], dxf_a[s]) for s in range(1, S + 1)):.2e}")
    assert rel(dx, gx) <= 2e-6 and rel(dv, gv) <= 2e-6
    np.testing.assert_allclose(dmu, dmu_a, rtol=1e-4, atol=1e-9)
    for s in range(1, S + 1):
        assert rel(dxf_b[s - 1], dxf_a[s]) <= 2e-6
        pb = e.get_param_gradients(s)
        for key in ("dL_dk", "dL_ddensity", "sum_dfext"):
            np.testing.assert_allclose(pb[key], par_a[s][key], rtol=1e-4, atol=1e-10)

    # ---- a schedule that covers only part of a fused sweep is an error, not a silent mix
    e.clear_schedules()
    e.set_force_schedule(0, 2, fu=FU[:2])
    with pytest.raises(capi.DcError, match="covers only part"):
        e.rollout_forward(0, S)
    e.clear_schedules()
    e.set_state(0, X0, V0)
    e.rollout_forward(0, S)                                             # and without schedules the current values apply again
    xc, _ = e.get_states(S, 1)
    assert np.abs(xc[0] - xa[S]).max() > 1e-6
    e.set_vertex_force_field(None)                                      # ... and the field is a term of its own: without it the rollout differs
    e.set_state(0, X0, V0)
    e.rollout_forward(0, S)
    xd, _ = e.get_states(S, 1)
    assert np.abs(xd[0] - xc[0]).max() > 1e-7


def test_schedules_with_self_collision_and_explicit_fixed_points_override():
    V, F, e = scene(24, (0, 23), selfcollision=1)
    rng = np.random.default_rng(2)
    B, S = 1, 3
    e.alloc_batch(B, S)
    XF, FU, FV, FVS = streams(V, (0, 23), B, S, rng)
    X0 = f32(V.reshape(-1))[None, :]
    e.set_state(0, X0, np.zeros_like(X0))
    e.set_fixed_point_schedule(0, XF)
    e.set_force_schedule(0, S, fu=FU)
    e.rollout_forward(0, S)
    xa, _ = e.get_states(0, S + 1)
    # the per-step call of a scheduled slot uses the schedule, explicit fixed points win over it
    e.set_state(0, X0, np.zeros_like(X0))
    for s in range(S):
        e.step_forward(s)
    xb, _ = e.get_states(0, S + 1)
    np.testing.assert_array_equal(xa, xb)
    e.step_forward(S - 1, fixed_pts=XF[0])
    xc, _ = e.get_states(S, 1)
    assert np.abs(xc[0] - xa[S]).max() > 1e-6


@pytest.mark.parametrize("nx,cluster", [(17, 0), (48, 4)])
def test_fused_backward_sweep_keeps_the_force_gradient_of_every_step(nx, cluster, monkeypatch):
    """dL_dfext_vec = h^2 (I + dr_df)^T u* of EVERY step of a fused sweep (dc_keep_force_gradients / dc_get_force_gradients) — what
    Simulation::stepBackward forms dL_dconstantForceField, dL_dwindtimestep and the fall-off wind gradients from, step by step
    (Simulation.cpp:1700-1764) — against dc_get_force_gradient after each per-step backward call: same kernels, same inputs, same bits."""
    if cluster:
        monkeypatch.setenv("DC_CLUSTER", str(cluster))
    rng = np.random.default_rng(11)
    B, S = 2, 4
    V, F, e = scene(nx, (0, nx - 1))
    e.alloc_batch(B, S)
    X0 = np.stack([f32(V.reshape(-1) + np.tile([0.01 * b, -0.05, 0.0], V.shape[0])) for b in range(B)])
    e.set_state(0, X0, np.zeros_like(X0))
    e.rollout_forward(0, S)
    gx = f32(rng.standard_normal(X0.shape)); gv = f32(0.01 * rng.standard_normal(X0.shape))
    e.keep_force_gradients(True)
    e.set_gradient(gx, gv)
    e.rollout_backward(S, S)                                  # ONE launch for the S steps
    kept = e.get_force_gradients(1, S)
    dx_f, dv_f, _ = e.get_gradient()
    e.keep_force_gradients(False)
    e.set_gradient(gx, gv)
    for s in range(S, 0, -1):
        e.rollout_backward(s, 1)
        np.testing.assert_array_equal(kept[s - 1], e.get_force_gradient())
    dx_s, dv_s, _ = e.get_gradient()
    np.testing.assert_array_equal(dx_f, dx_s); np.testing.assert_array_equal(dv_f, dv_s)
    assert np.abs(kept).max() > 0


def test_several_attachment_sets_in_fused_batched_rollouts_by_composing_contexts():
    """Several attachment sets (SceneConfiguration::customAttachmentVertexIdx with more than one entry, Simulation.cpp:1053-1068, :2371-2393: one
    SystemMatrix per set, set i takes over at a given record) in the BATCHED, fused form: every set is its own system matrix, i.e. its own context,
    and a rollout that switches sets is a chain of fused segments — the state is handed from context to context ON THE DEVICE (dc_get_state_dev /
    dc_set_state_dev: no host copy), the carried gradient on the way back. The host class does the same per step for one rollout
    (tests/test_gpu_pymodule.py::test_several_attachment_sets_switch_the_system_matrix, against two oracles); here: a batch of 4 rollouts, 2 steps
    with set A then 3 steps with set B, each segment ONE launch per direction with its clip targets as a schedule — against the same chain driven
    by per-step calls (forward bitwise, backward to solver tolerance) and with the segments' record types intact (dL_dxfixed of a step has the size
    of ITS set)."""
    nx = 24
    setA, setB = (0, nx - 1), (nx * (nx - 1), nx * nx - 1, nx * nx // 2)
    rng = np.random.default_rng(9)
    B, SA, SB = 4, 2, 3
    V, F, eA = scene(nx, setA)
    _, _, eB = scene(nx, setB)
    N = V.shape[0]
    X0 = np.stack([f32((V + np.array([0.03 * b, -0.05, 0.02 * b])).reshape(-1)) for b in range(B)])
    XFA = np.stack([np.stack([f32((V[list(setA)] + np.array([0.0, 0.01 * (s + 1), 0.005 * b])).reshape(-1)) for b in range(B)]) for s in range(SA)])
    XFB = np.stack([np.stack([f32((V[list(setB)] + np.array([0.004 * b, 0.01 * (s + 1), 0.0])).reshape(-1)) for b in range(B)]) for s in range(SB)])
    gx = f32(rng.standard_normal(X0.shape)); gv = f32(0.01 * rng.standard_normal(X0.shape))
    dev = torch.device("cuda", 0)

    def hand_over(src, slot, dst):
        x = torch.empty(B * 3 * N, dtype=torch.float64, device=dev); v = torch.empty_like(x)
        src.get_state_dev(slot, x, v)
        dst.set_state_dev(0, x, v)
        torch.cuda.synchronize()

    def run(fused):
        eA.alloc_batch(B, SA); eB.alloc_batch(B, SB)
        eB.set_trajectory_start(-1)                # set B's tape holds the SECOND segment: none of its steps is the trajectory's isStart step
        eA.set_state(0, X0, np.zeros_like(X0))
        if fused:
            eA.set_fixed_point_schedule(0, XFA); eA.rollout_forward(0, SA)
        else:
            for s in range(SA):
                eA.step_forward(s, fixed_pts=XFA[s])
        eA.sync()
        hand_over(eA, SA, eB)
        if fused:
            eB.set_fixed_point_schedule(0, XFB); eB.rollout_forward(0, SB)
        else:
            for s in range(SB):
                eB.step_forward(s, fixed_pts=XFB[s])
        xe, ve = eB.get_state(SB)
        stats = [eA.get_stats(s + 1)[0] for s in range(SA)] + [eB.get_stats(s + 1)[0] for s in range(SB)]
        # backward: set B's segment, the carried gradient to set A's context, set A's segment
        if fused:
            eB.set_gradient(gx, gv); eB.rollout_backward(SB, SB)
            dxB, dvB, _ = eB.get_gradient()
            dxfB = eB.get_dxfixed(1, SB)
            eA.set_gradient(dxB, dvB); eA.rollout_backward(SA, SA)
            dx, dv, _ = eA.get_gradient()
            dxfA = eA.get_dxfixed(1, SA)
        else:
            cx, cv = gx, gv
            dxfB = np.zeros((SB, B, 3 * len(setB))); dxfA = np.zeros((SA, B, 3 * len(setA)))
            for s in range(SB, 0, -1):
                gb = eB.step_backward(s, cx, cv, is_start=False)
                assert np.all(gb["converged"] == 1)
                cx, cv = gb["dL_dx"], gb["dL_dv"]; dxfB[s - 1] = gb["dL_dxfixed"]
            for s in range(SA, 0, -1):
                gb = eA.step_backward(s, cx, cv, is_start=(s == 1))
                assert np.all(gb["converged"] == 1)
                cx, cv = gb["dL_dx"], gb["dL_dv"]; dxfA[s - 1] = gb["dL_dxfixed"]
            dx, dv = cx, cv
        eA.clear_schedules(); eB.clear_schedules()
        return dict(x=xe, v=ve, dx=dx, dv=dv, dxfA=dxfA, dxfB=dxfB, conv=[np.all(st["converged"] == 1) for st in stats])

    a, b = run(True), run(False)
    assert all(a["conv"]) and all(b["conv"])
    np.testing.assert_array_equal(a["x"], b["x"]); np.testing.assert_array_equal(a["v"], b["v"])      # fused = per-step, bit for bit, across the switch
    assert a["dxfA"].shape == (SA, B, 6) and a["dxfB"].shape == (SB, B, 9)
    e_dx, e_dv = rel(a["dx"], b["dx"]), rel(a["dv"], b["dv"])
    e_fa, e_fb = rel(a["dxfA"], b["dxfA"]), rel(a["dxfB"], b["dxfB"])
    print(f"\n[attachment sets, fused segments vs per-step calls] dL_dx {e_dx:.2e} dL_dv {e_dv:.2e} dL_dxfixed set A {e_fa:.2e} set B {e_fb:.2e}")
    assert max(e_dx, e_dv, e_fa, e_fb) <= 1e-5 and np.linalg.norm(a["dxfA"]) > 0 and np.linalg.norm(a["dxfB"]) > 0
