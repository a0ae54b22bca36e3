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


def test_hatController_trains_one_epoch_unmodified(tmp_path):
    """The controller-training script BASELINE.json's north star names — src/python_code/hatController.py with common.py, utils.py,
    clothNN/ and pySim/, frozen byte for byte under tests/golden/reference_callers/ — run UNMODIFIED for one epoch
    (`python hatController.py --epochNum 1`) against this repository's diffcloth_py: makeSim("wear_hat"), makeOptimizeHelper, 20
    training rollouts of 400 stepNN steps through torch.autograd with loss.backward() through 400 stepBackwardNN calls each
    (hatController.py:78-105), Adam step, 9 validation rollouts, checkpoints and plots. What the test adds around it is environment
    only: an assets directory holding the hat mesh (from tests/golden/meshes.npz) and hat_target.txt, a scratch working directory
    (the script writes experiments/ next to itself), a headless matplotlib backend and a three-line stand-in for the `colorama`
    package, which this image does not have."""
    import shutil
    import subprocess
    import time
    pytest.importorskip("torch"); pytest.importorskip("matplotlib")
    src = os.path.join(ROOT, "tests", "golden", "reference_callers")
    work = tmp_path / "python_code"
    shutil.copytree(src, work)
    (work / "colorama.py").write_text("class _C:\n    def __getattr__(self, k):\n        return ''\nFore = Style = _C()\n")
    assets = tmp_path / "assets"
    (assets / "remeshed" / "Hat").mkdir(parents=True)
    V, F = scenes.load_mesh("hat")
    with open(assets / "remeshed" / "agenthat2-579-rotated.obj", "w") as f:
        for p in np.asarray(V).reshape(-1, 3):
            f.write(f"v {p[0]:.17g} {p[1]:.17g} {p[2]:.17g}\n")
        for t in np.asarray(F).reshape(-1, 3):
            f.write(f"f {t[0] + 1} {t[1] + 1} {t[2] + 1}\n")
    shutil.copyfile(os.path.join(src, "hat_target.txt"), assets / "remeshed" / "Hat" / "hat_target.txt")
    env = dict(os.environ, DIFFCLOTH_ASSETS=str(assets), MPLBACKEND="Agg",
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "diffcloth_amd", "lib"), os.environ.get("PYTHONPATH", "")]))
    t0 = time.perf_counter()
    run = subprocess.run([sys.executable, "hatController.py", "--epochNum", "1", "--randSeed", "2"], cwd=work, env=env, capture_output=True, text=True, timeout=1500)
    dt = time.perf_counter() - t0
    tail = "\n".join((run.stdout + run.stderr).splitlines()[-25:])
    assert run.returncode == 0, tail
    logs = list((work / "experiments" / "wear_hat").glob("*/log.txt"))
    assert len(logs) == 1, tail
    text = logs[0].read_text()
    train = [ln for ln in text.splitlines() if ln.startswith("Train: loss:")]
    test = [ln for ln in text.splitlines() if ln.startswith("Test: loss:")]
    assert len(train) == 1 and len(test) == 1, text
    loss = float(train[0].split()[2])
    assert np.isfinite(loss) and loss > 0
    ckpts = sorted(p.name for p in logs[0].parent.glob("*.pth"))
    assert "0.pth" in ckpts and "trainBestEpoch.pth" in ckpts, ckpts
    steps = 20 * 400 * 2 + 9 * 400          # forward + backward steps of the training rollouts, forward steps of the validation
    print(f"\n[hatController.py, unmodified, 1 epoch] {dt:.1f} s wall for 20 training rollouts (400 stepNN + 400 stepBackwardNN each) + 9 validation "
          f"rollouts = {steps / dt:.0f} single-rollout steps/s including torch and the script's plotting; {train[0].strip()} | {test[0].strip()}")
