"""The reference's OWN PyTorch caller layer (src/python_code/pySim/functional.py + pySim.py, frozen byte for byte under
tests/golden/reference_callers/) imported and run UNMODIFIED against this repository's diffcloth_py module: three stepNN calls
through torch.autograd (`SimFunction.apply`), a loss on the last state, `loss.backward()` through three stepBackwardNN calls —
the exact sequence hatController.py's training loop drives (hatController.py:257-300). Checked: it runs, every gradient it hands
back to torch (dL/dx0, dL/dv0, dL/da per step) is finite and equals what the same chain gives when the module's stepBackwardNN
is called by hand with the reference's conventions (functional.py:66-101)."""
import os
import sys

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffcloth_amd", "lib"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden", "reference_callers"))


def test_reference_functional_py_runs_unmodified():
    torch = pytest.importorskip("torch")
    d = pytest.importorskip("diffcloth_py")
    from pySim.pySim import pySim            # the reference's files, unmodified
    from pySim.functional import SimFunction  # noqa: F401
    V, F = scenes.load_mesh("hat")
    sim = d.makeSimFromMesh("wear_hat", V.reshape(-1), F.reshape(-1).tolist())
    helper = d.makeOptimizeHelperWithSim("wear_hat", sim)
    d.Simulation.forwardConvergenceThreshold = 1e-6          # hatController.py validates at 1e-6
    sim.gradientClipping = False
    sim.resetSystem()
    module = pySim(sim, helper, True)                        # sets useCustomRLFixedPoint, as hatController.py:99 does
    assert sim.useCustomRLFixedPoint
    rec0 = sim.getStateInfo()
    x0 = torch.tensor(np.asarray(rec0.x), dtype=torch.float32, requires_grad=True)
    v0 = torch.tensor(np.asarray(rec0.v), dtype=torch.float32, requires_grad=True)
    a_base = np.asarray(rec0.x_fixedpoints, dtype=np.float64)
    actions = [torch.tensor(a_base + (s + 1) * np.array([0.02, -0.05, 0.01, 0.02, -0.05, 0.01]), dtype=torch.float32, requires_grad=True)
               for s in range(3)]
    x, v = x0, v0
    for a in actions:
        x, v = module(x, v, a)
    assert sim.getStateInfo().stepIdx == 3
    target = torch.tensor(np.asarray(rec0.x), dtype=torch.float32)          # pull back to the start shape
    loss = ((x - target) ** 2).sum() * 1e-3 + (v ** 2).sum() * 1e-5
    loss.backward()
    grads = [x0.grad, v0.grad] + [a.grad for a in actions]
    assert all(g is not None and torch.isfinite(g).all() for g in grads)
    assert float(x0.grad.abs().max()) > 0 and float(actions[0].grad.abs().max()) > 0
    # the same chain by hand, with functional.py's conventions (not "isLast": incoming gradient in the first pair of arguments,
    # zeros as dL_dxinit / dL_dvinit; dL/da rescaled when its norm exceeds 1e-7)
    recs = list(sim.forwardRecords)
    gx = (2e-3 * (x.detach() - target)).numpy().astype(np.float32); gv = (2e-5 * v.detach()).numpy().astype(np.float32)
    das = []
    for s in (3, 2, 1):
        z = np.zeros_like(gx)
        back = sim.stepBackwardNN(helper.taskInfo, gx, gv, recs[s], recs[s].stepIdx == 1, z, z)
        da = np.asarray(back.dL_dxfixed); n = np.linalg.norm(da)
        if n > 1e-7:
            da = da * (max(min(da.shape[0] * 4.0, n), 0.05) / n)
        das.append(da)
        gx, gv = np.asarray(back.dL_dx), np.asarray(back.dL_dv)
    np.testing.assert_allclose(x0.grad.numpy(), gx, rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(v0.grad.numpy(), gv, rtol=1e-5, atol=1e-9)
    for a, da in zip(actions, das[::-1]):
        np.testing.assert_allclose(a.grad.numpy(), da, rtol=1e-5, atol=1e-9)


# (hatController.py itself, unmodified, for one epoch: tests/test_gpu_batched_epoch.py runs it and compares the batched epoch with it)
