"""Batched counterpart of the reference's `pySim/functional.py` (SimFunction, src/python_code/pySim/functional.py:18-106):
a `torch.autograd.Function` that advances B independent rollouts by one time step on the GPU and back-propagates through
it — the building block of the controller training loops (hatController.py) for a whole batch of rollouts at once.

The reference's function wraps ONE `diffcloth_py.Simulation` (stepNN / stepBackwardNN) and is called once per rollout and
step; here `x`, `v`, `a` carry a leading batch dimension and the step runs through the C-ABI of libdiffcloth_hip.so
(`diffcloth_amd.capi.Engine`). Behaviours kept from the reference function:
  * teacher forcing: the state given to `forward` replaces the stored one (stepNN, Simulation.cpp:1020-1042);
  * the step that reaches `step_num` back-propagates with zero incoming gradients and the loss gradient passed as
    dL_dxinit / dL_dvinit (functional.py:66-75) — for that step the adjoint is the identity;
  * the gradient w.r.t. the action (fixed-point targets) is rescaled to a norm within [0.05, 4 * dim] per rollout
    (functional.py:88-97).
Tensors cross the boundary as float64 host arrays, as in the reference (numpy <-> Eigen by copy there).
"""
import numpy as np
import torch


class BatchedSim:
    """B rollouts of one scene on one GPU: a `capi.Engine` with an allocated batch, the step counter of the episode and
    the episode length `step_num` (sceneConfig.stepNum of the reference)."""

    def __init__(self, engine, step_num):
        if engine.B <= 0:
            raise ValueError("the engine needs alloc_batch(B, tape) before it is wrapped (tape >= the steps of an episode)")
        self.engine = engine
        self.step_num = int(step_num)
        self.step_idx = 0

    def reset(self, x0, v0=None):
        """Start a new episode from the given states ([B, 3N]); returns them as float32 tensors like getStateInfo()."""
        x0 = np.asarray(x0, dtype=np.float64).reshape(self.engine.B, -1)
        v0 = np.zeros_like(x0) if v0 is None else np.asarray(v0, dtype=np.float64).reshape(self.engine.B, -1)
        self.engine.set_state(0, x0, v0)
        self.step_idx = 0
        return torch.as_tensor(x0).float(), torch.as_tensor(v0).float()


class BatchedSimFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, v, a, sim):
        e = sim.engine
        slot = sim.step_idx
        if slot >= e.tape:
            raise RuntimeError("BatchedSimFunction: tape exhausted, call BatchedSim.reset()")
        e.set_state(slot, np.float64(x.contiguous().detach().cpu().numpy()), np.float64(v.contiguous().detach().cpu().numpy()))
        act = None if e.Af == 0 else np.float64(a.contiguous().detach().cpu().numpy())
        e.step_forward(slot, fixed_pts=act, want_stats=False)
        sim.step_idx = slot + 1
        ctx.sim = sim
        ctx.slot = slot + 1
        xn, vn = e.get_state(slot + 1)
        return torch.as_tensor(xn).to(x.dtype), torch.as_tensor(vn).to(v.dtype)

    @staticmethod
    def backward(ctx, dL_dx_next, dL_dv_next):
        sim, slot = ctx.sim, ctx.slot
        e = sim.engine
        gx = np.float64(dL_dx_next.contiguous().detach().cpu().numpy())
        gv = np.float64(dL_dv_next.contiguous().detach().cpu().numpy())
        if slot == sim.step_num:            # functional.py:66-75
            out = e.step_backward(slot, np.zeros_like(gx), np.zeros_like(gv), dL_dxinit=gx, dL_dvinit=gv, is_start=(slot == 1))
        else:
            out = e.step_backward(slot, gx, gv, is_start=(slot == 1))
        da = out["dL_dxfixed"].copy()
        for b in range(da.shape[0]):        # functional.py:88-97, per rollout
            n = np.linalg.norm(da[b])
            if n > 1e-7:
                da[b] *= max(min(da.shape[1] * 4.0, n), 0.05) / n
        dt = dL_dx_next.dtype
        return torch.as_tensor(out["dL_dx"]).to(dt), torch.as_tensor(out["dL_dv"]).to(dt), torch.as_tensor(da).to(dt), None


def sim_step(sim, x, v, a):
    """One differentiable time step of all rollouts: (x', v') = step(x, v; a). x, v: [B, 3N]; a: [B, 3 Af] clip targets."""
    return BatchedSimFunction.apply(x, v, a, sim)
