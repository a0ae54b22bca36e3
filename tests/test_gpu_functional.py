"""GPU: the batched torch.autograd step (diffcloth_amd/functional.py) against the reference-shaped per-rollout path — the
SimFunction of src/python_code/pySim/functional.py re-stated here over diffcloth_py.Simulation (stepNN / stepBackwardNN),
one rollout at a time, on the hat scene: states and the gradients w.r.t. the start state and the per-step actions."""
import os
import sys

import numpy as np
import pytest
import torch

import scenes
from diffcloth_amd import capi
from diffcloth_amd.functional import BatchedSim, sim_step

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffcloth_amd", "lib"))


class RefSimFunction(torch.autograd.Function):       # functional.py:18-106, unchanged in behaviour
    @staticmethod
    def forward(ctx, x, v, a, cppSim, helper):
        ctx.helper = helper; ctx.simulation = cppSim
        past = cppSim.getStateInfo()
        cppSim.stepNN(past.stepIdx + 1, np.float64(x.detach().numpy()), np.float64(v.detach().numpy()), np.float64(a.detach().numpy()))
        ctx.newRecord = cppSim.getStateInfo()
        return torch.as_tensor(ctx.newRecord.x), torch.as_tensor(ctx.newRecord.v)

    @staticmethod
    def backward(ctx, gx, gv):
        sim = ctx.simulation
        gx = gx.detach().numpy(); gv = gv.detach().numpy()
        rec = ctx.newRecord
        if rec.stepIdx == sim.sceneConfig.stepNum:
            back = sim.stepBackwardNN(ctx.helper.taskInfo, np.zeros_like(gx), np.zeros_like(gv), rec, rec.stepIdx == 1, gx, gv)
        else:
            back = sim.stepBackwardNN(ctx.helper.taskInfo, gx, gv, rec, rec.stepIdx == 1, np.zeros_like(gx), np.zeros_like(gv))
        da = np.array(back.dL_dxfixed)
        n = np.linalg.norm(da)
        if n > 1e-7:
            da = da * (max(min(da.shape[0] * 4.0, n), 0.05) / n)
        return torch.as_tensor(back.dL_dx), torch.as_tensor(back.dL_dv), torch.as_tensor(da), None, None


def test_batched_autograd_step_matches_the_per_rollout_reference_path():
    import diffcloth_py as d
    V, F = scenes.load_mesh("hat")
    cfg = scenes.HAT
    K, B = 3, 3
    sim = d.makeSimFromMesh("wear_hat", V.reshape(-1), F.reshape(-1).tolist())
    helper = d.makeOptimizeHelperWithSim("wear_hat", sim)
    d.Simulation.forwardConvergenceThreshold = 1e-8
    P = np.array(sim.getRestPositions()).reshape(-1, 3)
    att = sim.getAttachmentVertices()
    # the same system through the C-ABI, batch of B rollouts (reference adjoint iteration, the host class's settings)
    e = capi.Engine(0)
    e.set_mesh(P, F); e.set_attachments(att)
    e.set_params(time_step=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], forward_tol=1e-8,
                 backward_tol=d.Simulation.backwardConvergenceThreshold, gradient_clipping=int(sim.gradientClipping),
                 gradient_clipping_threshold=sim.gradientClippingThreshold, selfcollision_enabled=1, adjoint_mode=0)
    c = np.array(sim.primitives[0].center)
    e.set_primitives([dict(kind=capi.DC_PRIM_SPHERE, group=0, center=c, radius=cfg["sphere_radius"], mu=cfg["sphere_mu"])])
    e.build()
    e.alloc_batch(B, K)
    bs = BatchedSim(e, sim.sceneConfig.stepNum)      # 400: none of the K steps is the episode's last one, in either path
    rng = np.random.default_rng(41)
    X0 = np.stack([(P + 0.002 * rng.standard_normal(P.shape)).reshape(-1) for _ in range(B)]).astype(np.float32).astype(np.float64)
    A = np.stack([[P[att].reshape(-1) + (k + 1) * np.tile([0.0, -0.04, -0.25], len(att)) + 0.01 * rng.standard_normal(3 * len(att))
                   for k in range(K)] for _ in range(B)]).astype(np.float32).astype(np.float64)        # [B, K, 3 Af]
    W = rng.standard_normal(P.size)

    def loss_of(xs):                                # something that touches every step's state
        return sum((k + 1) * (xs[k] * torch.as_tensor(W)).sum(-1) for k in range(len(xs))).sum() * 1e-3

    # batched path
    x, v = bs.reset(X0)
    x = x.double().requires_grad_(True); v = v.double().requires_grad_(True)
    acts = [torch.tensor(A[:, k], dtype=torch.float64, requires_grad=True) for k in range(K)]
    xs, xk, vk = [], x, v
    for k in range(K):
        xk, vk = sim_step(bs, xk, vk, acts[k])
        xs.append(xk)
    loss_of(xs).backward()
    # per-rollout reference-shaped path
    for b in range(B):
        sim.resetSystem()
        xr = torch.tensor(X0[b], dtype=torch.float64, requires_grad=True); vr = torch.zeros(P.size, dtype=torch.float64, requires_grad=True)
        ar = [torch.tensor(A[b, k], dtype=torch.float64, requires_grad=True) for k in range(K)]
        ys, yk, wk = [], xr, vr
        for k in range(K):
            yk, wk = RefSimFunction.apply(yk, wk, ar[k], sim, helper)
            ys.append(yk)
        loss_of([y[None] for y in ys]).backward()
        for k in range(K):
            np.testing.assert_allclose(xs[k][b].detach().numpy(), ys[k].detach().numpy(), atol=2e-6)
        scale = np.abs(xr.grad.numpy()).max()
        np.testing.assert_allclose(x.grad[b].numpy(), xr.grad.numpy(), atol=2e-4 * scale)
        np.testing.assert_allclose(v.grad[b].numpy(), vr.grad.numpy(), atol=2e-4 * np.abs(vr.grad.numpy()).max())
        for k in range(K):
            np.testing.assert_allclose(acts[k].grad[b].numpy(), ar[k].grad.numpy(), atol=2e-4 * max(np.abs(ar[k].grad.numpy()).max(), 1e-12))
    assert np.abs(x.grad.numpy()).max() > 0 and np.abs(acts[0].grad.numpy()).max() > 0


def test_last_step_of_an_episode_is_an_identity_in_the_adjoint():
    """functional.py:66-75: at stepIdx == stepNum the incoming gradients are handed over as dL_dxinit / dL_dvinit and the adjoint
    sees zeros — the step contributes dL_dx = gx + gv / h ... exactly what Simulation::stepBackward adds for its init terms."""
    V, F = scenes.load_mesh("hat")
    cfg = scenes.HAT
    P, rmin, rmax = scenes.normalise_model(V, cfg["orientation"], cfg["cloth_dim"])
    P = P.astype(np.float32).astype(np.float64)
    e = capi.Engine(0)
    e.set_mesh(P, F); e.set_attachments(cfg["attachments"])
    e.set_params(time_step=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], forward_tol=1e-7,
                 gradient_clipping=0, selfcollision_enabled=0, adjoint_mode=0)
    e.set_primitives([])
    e.build()
    e.alloc_batch(2, 2)
    bs = BatchedSim(e, 2)
    x, v = bs.reset(np.stack([P.reshape(-1)] * 2))
    x = x.double().requires_grad_(True); v = v.double().requires_grad_(True)
    a = torch.tensor(np.stack([P[cfg["attachments"]].reshape(-1)] * 2), dtype=torch.float64, requires_grad=True)
    x1, v1 = sim_step(bs, x, v, a)
    x2, v2 = sim_step(bs, x1, v1, a)              # step 2 == step_num: the "last" branch
    rng = np.random.default_rng(5)
    gx = torch.tensor(rng.standard_normal(x2.shape)); gv = torch.tensor(0.01 * rng.standard_normal(x2.shape))
    g1x, g1v = torch.autograd.grad([x2, v2], [x1, v1], [gx, gv])
    h = cfg["h"]
    np.testing.assert_allclose(g1x.numpy(), (gx + gv / h).numpy(), rtol=1e-6, atol=2e-6)     # Simulation.cpp:1534-1540 with u* = 0
    np.testing.assert_allclose(g1v.numpy(), gv.numpy(), rtol=1e-6, atol=1e-8)


def test_device_pointer_path_equals_the_host_path():
    """CUDA tensors go through the device-pointer boundary (dc_*_dev: no host copy, torch's stream); the same episode with CPU
    tensors goes through the host path. Same kernels, same fp32 device state: states and every gradient must agree to the last bit
    of the fp64 tensors handed back; an fp32 CUDA episode must agree to fp32 rounding."""
    V, F = scenes.load_mesh("hat")
    cfg = scenes.HAT
    P, rmin, rmax = scenes.normalise_model(V, cfg["orientation"], cfg["cloth_dim"])
    P = P.astype(np.float32).astype(np.float64)
    att = cfg["attachments"]
    K, B = 3, 4

    def run(device, dtype):
        e = capi.Engine(0)
        e.set_mesh(P, F); e.set_attachments(att)
        e.set_params(time_step=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], forward_tol=1e-7,
                     gradient_clipping=0, selfcollision_enabled=0, adjoint_mode=1, adjoint_rel_tol=1e-7)
        e.set_primitives([])
        e.build()
        e.alloc_batch(B, K)
        bs = BatchedSim(e, 400)
        rng = np.random.default_rng(3)
        X0 = np.stack([(P + 0.002 * rng.standard_normal(P.shape)).reshape(-1) for _ in range(B)]).astype(np.float32).astype(np.float64)
        A = np.stack([[P[att].reshape(-1) + (k + 1) * np.tile([0.0, -0.04, -0.2], len(att)) for k in range(K)] for _ in range(B)]).astype(np.float32)
        W = torch.tensor(rng.standard_normal(P.size), dtype=dtype, device=device)
        x = torch.tensor(X0, dtype=dtype, device=device, requires_grad=True)
        v = torch.zeros_like(x, requires_grad=True)
        acts = [torch.tensor(A[:, k], dtype=dtype, device=device, requires_grad=True) for k in range(K)]
        bs.reset(X0)
        xs, xk, vk = [], x, v
        for k in range(K):
            xk, vk = sim_step(bs, xk, vk, acts[k])
            xs.append(xk)
        loss = sum((k + 1) * (xs[k] * W).sum() for k in range(K)) * 1e-3 + (vk ** 2).sum() * 1e-4
        loss.backward()
        return [t.detach().cpu().double().numpy() for t in xs] + [x.grad.cpu().double().numpy(), v.grad.cpu().double().numpy()] + [a.grad.cpu().double().numpy() for a in acts]

    host = run("cpu", torch.float64)
    dev64 = run("cuda", torch.float64)
    dev32 = run("cuda", torch.float32)
    for h, d in zip(host, dev64):
        np.testing.assert_array_equal(h, d)
    for h, d in zip(host, dev32):
        np.testing.assert_allclose(d, h, rtol=2e-5, atol=2e-6 * max(np.abs(h).max(), 1e-12))
    assert np.abs(host[-1]).max() > 0
