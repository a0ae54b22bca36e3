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
CUDA tensors (fp32 or fp64) never leave the GPU: states, actions and gradients cross the boundary as device pointers
(dc_*_dev of include/diffcloth_hip.h), the step is enqueued on torch's current stream and nothing synchronises — the
reference copies every tensor through numpy on the host (functional.py:30-34, 60-64), which for B = 256 rollouts of 10 000
vertices is 246 MB over PCIe per step. CPU tensors still take the host path (float64 arrays, as in the reference).
"""
import numpy as np
import torch


class BatchedSim:
    """B rollouts of one scene on one GPU: a `capi.Engine` with an allocated batch, the step counter of the episode and
    the episode length `step_num` (sceneConfig.stepNum of the reference)."""

    def __init__(self, engine, step_num, strict=False):
        """strict: an adjoint solve that did not converge raises at the end of the episode's backward sweep. Off by default: the reference
        does not stop on non-convergence (it prints and goes on, Simulation.cpp:1589-1600) and controller loops that tolerate an occasional
        unconverged step keep running — they get ONE warning per episode instead, with the number of unconverged (step, rollout) pairs and the worst residual
        (also left in `self.unconverged`, so a training loop can act on it without parsing warnings). Hard errors (a timed-out exchange of the split kernels, a
        self-contact list overflow) always raise."""
        if engine.B <= 0:
            raise ValueError("the engine needs alloc_batch(B, tape) before it is wrapped (tape >= the steps of an episode)")
        self.engine = engine
        self.step_num = int(step_num)
        self.step_idx = 0
        self._stream = None
        self._bwd_slots = set()
        self.strict = bool(strict)
        self.unconverged = 0

    def on_current_stream(self, device=None):
        """order the engine's work with torch's current CUDA stream of the tensors' device (once per stream change). torch's default
        stream has handle 0, which dc_use_stream reads as "the context's own stream": ordering with torch's kernels on the legacy
        default stream then rests on its implicit synchronisation with blocking streams (the context's stream is a blocking one)."""
        st = torch.cuda.current_stream(device)
        if self._stream is None or self._stream.cuda_stream != st.cuda_stream:
            self.engine.use_stream(st)
            self._stream = st

    def check_episode(self):
        """Surface engine errors of the episode so far: the device-pointer calls (dc_*_dev) only enqueue work and report nothing, so a
        timed-out exchange of the split kernels (sticky error word) or a self-contact list overflow would otherwise flow into the optimiser as
        garbage states / gradients. One synchronisation + the statistics of the recorded steps; called at the end of an episode's backward
        sweep (slot 1, host and device path alike) and by reset(). Raises capi.DcError; an adjoint solve that did not converge raises
        RuntimeError when `strict`, else warns once per episode (the reference goes on as well)."""
        import warnings
        e = self.engine
        e.sync()                                     # raises on a timed-out exchange
        bad_pairs, worst, first = 0, 0.0, None
        for slot in range(1, self.step_idx + 1):
            fwd, bwd = e.get_stats(slot)             # raises DC_ERR_CAPACITY on a self-contact overflow of that step
            if slot in self._bwd_slots and (bwd["converged"] == 0).any():
                bad = np.nonzero(bwd["converged"] == 0)[0]
                bad_pairs += len(bad)
                w = int(bad[np.argmax(bwd["last_udiff"][bad])])
                if first is None:
                    first = (slot, int(bad[0]))
                if float(bwd["last_udiff"][w]) >= worst:
                    worst, worst_at = float(bwd["last_udiff"][w]), (slot, w)
        self.unconverged = bad_pairs                 # (step, rollout) pairs of the episode whose adjoint solve did not converge
        if bad_pairs:
            msg = (f"BatchedSim: {bad_pairs} adjoint solve(s) of this episode did not converge (first: step {first[0]}, rollout {first[1]}; "
                   f"worst relative residual {worst:.2e} at step {worst_at[0]}, rollout {worst_at[1]}): the gradients of those rollouts are not converged")
            if self.strict:
                raise RuntimeError(msg)
            warnings.warn(msg, RuntimeWarning, stacklevel=2)

    def reset(self, x0, v0=None):
        """Start a new episode from the given states ([B, 3N]); returns them as float32 tensors like getStateInfo()."""
        if self.step_idx > 0:
            self.check_episode()
        self._bwd_slots = set()
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
        ctx.sim = sim
        ctx.slot = slot + 1
        if x.is_cuda:           # device path: pointers in, pointers out, torch's stream
            sim.on_current_stream(x.device)
            xd, vd = x.detach().contiguous(), v.detach().to(x.dtype).contiguous()
            e.set_state_dev(slot, xd, vd)
            e.step_forward_dev(slot, None if e.Af == 0 else a.detach().to(x.dtype).contiguous())
            xn, vn = torch.empty_like(xd), torch.empty_like(vd)
            e.get_state_dev(slot + 1, xn, vn)
            sim.step_idx = slot + 1            # (only once the step is enqueued: a raised error leaves the tape index where it was)
            return xn, vn
        e.set_state(slot, np.float64(x.contiguous().detach().numpy()), np.float64(v.contiguous().detach().numpy()))
        act = None if e.Af == 0 else np.float64(a.contiguous().detach().numpy())
        e.step_forward(slot, fixed_pts=act, want_stats=False)
        xn, vn = e.get_state(slot + 1)
        sim.step_idx = slot + 1
        return torch.as_tensor(xn).to(x.dtype), torch.as_tensor(vn).to(v.dtype)

    @staticmethod
    def backward(ctx, dL_dx_next, dL_dv_next):
        sim, slot = ctx.sim, ctx.slot
        e = sim.engine
        last = slot == sim.step_num            # functional.py:66-75
        sim._bwd_slots.add(slot)
        if dL_dx_next.is_cuda:
            sim.on_current_stream(dL_dx_next.device)
            gx = dL_dx_next.detach().contiguous(); gv = dL_dv_next.detach().to(gx.dtype).contiguous()
            dx, dv = torch.empty_like(gx), torch.empty_like(gx)
            da = torch.zeros((e.B, max(3 * e.Af, 1)), dtype=gx.dtype, device=gx.device)[:, :3 * e.Af].contiguous()
            zero = torch.zeros_like(gx) if last else None
            if last:
                e.step_backward_dev(slot, zero, zero, dx, dv, dxfixed=da if e.Af else None, ix=gx, iv=gv, is_start=(slot == 1))
            else:
                e.step_backward_dev(slot, gx, gv, dx, dv, dxfixed=da if e.Af else None, is_start=(slot == 1))
            if e.Af:                           # functional.py:88-97, per rollout, on the device
                n = da.norm(dim=1, keepdim=True)
                scale = torch.where(n > 1e-7, n.clamp(min=0.05, max=4.0 * da.shape[1]) / n.clamp(min=1e-30), torch.ones_like(n))
                da = da * scale
            if slot == 1:
                sim.check_episode()            # end of the episode's backward sweep: one synchronisation, errors surface here
            return dx, dv, da, None
        gx = np.float64(dL_dx_next.contiguous().detach().numpy())
        gv = np.float64(dL_dv_next.contiguous().detach().numpy())
        if last:
            out = e.step_backward(slot, np.zeros_like(gx), np.zeros_like(gv), dL_dxinit=gx, dL_dvinit=gv, is_start=(slot == 1))
        else:
            out = e.step_backward(slot, gx, gv, is_start=(slot == 1))
        if slot == 1:
            sim.check_episode()                # (the host path checks where the device path does)
        da = out["dL_dxfixed"].copy()
        for b in range(da.shape[0]):        # functional.py:88-97, per rollout
            n = np.linalg.norm(da[b])
            if n > 1e-7:
                da[b] *= max(min(da.shape[1] * 4.0, n), 0.05) / n
        dt = dL_dx_next.dtype
        return torch.as_tensor(out["dL_dx"]).to(dt), torch.as_tensor(out["dL_dv"]).to(dt), torch.as_tensor(da).to(dt), None


def sim_step(sim, x, v, a):
    """One differentiable time step of all rollouts: (x', v') = step(x, v; a). x, v: [B, 3N]; a: [B, 3 Af] clip targets.
    CUDA tensors stay on the GPU (device-pointer boundary); CPU tensors go through the host path."""
    return BatchedSimFunction.apply(x, v, a, sim)
