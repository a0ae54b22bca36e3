"""Pins the fp64 oracle against the reference's ONLY golden data: output/tshirt-exampleopt/iter0.

The frames are OBJ dumps at ~6 significant digits, the parameters (iter0/param.txt) are truncated to 6 decimals,
the run used OpenMP (self-contact order not reproducible) and a 250-step cloth rollout is chaotic, so this is a
weak pin by nature: frame 0 checks the mesh normalisation exactly (to the file precision), the next frames check
the dynamics (gravity, sinusoidal wind, two corner attachments, stretch/bend constraints) to 2e-3 of the cloth size.
"""
import os

import numpy as np
import pytest

import orc
import scenes


@pytest.fixture(scope="module")
def tshirt():
    g = np.load(os.path.join(scenes.GOLDEN, "tshirt_golden.npz"))
    V, F = scenes.load_mesh("tshirt")
    cfg = scenes.TSHIRT
    P, rmin, rmax = scenes.normalise_model(V, cfg["orientation"], cfg["cloth_dim"])
    att = scenes.corner_attachments(P, rmin, rmax)
    o = orc.Oracle(P, F, h=cfg["h"], density=cfg["density"], k_stretch=float(g["k_stretch"]), k_bend=cfg["k_bend"],
                   fwd_tol=cfg["fwd_tol"], bwd_tol=cfg["bwd_tol"], attachments=att, contact=True, selfcollision=True,
                   gradient_clipping=True, threads=4)
    fw = g["f_wind"]
    o.set_wind(True, 2, fw[0:3] / np.linalg.norm(fw[0:3]), float(np.linalg.norm(fw[0:3])), float(fw[3]), float(fw[4]))
    o.build()
    return g, P, F, att, o


def test_rest_shape_matches_frame0(tshirt):
    g, P, F, att, o = tshirt
    assert P.shape == g["frames"][0].shape
    np.testing.assert_allclose(P, g["frames"][0], atol=2e-5)      # file precision: 6 significant digits


def test_first_frames_match_reference_rollout(tshirt):
    g, P, F, att, o = tshirt
    x = P.reshape(-1).copy()
    v = np.zeros_like(x)
    xf = P[att].reshape(-1)
    errs = []
    iters = 0
    nself = 0
    K = 40
    for k in range(1, K + 1):
        out = o.step(x, v, xf, t_prev=(k - 1) * scenes.TSHIRT["h"])
        x, v = out["x"], out["v"]
        iters += out["iters"]
        nself += out["nself"]
        errs.append(np.abs(x.reshape(-1, 3) - g["frames"][k]).max())
    print(f"\n[golden tshirt] max |x - frame_k| k=1..{K}:", " ".join(f"{e:.1e}" for e in errs[4::5]), "| mean PD iters", iters / K,
          "| self contacts seen", nself)
    # measured: 5e-6 (= the 6-significant-digit file precision) growing to 8e-6 at frame 40, self-contacts included
    assert max(errs[:10]) < 1e-5
    assert max(errs) < 5e-5
    assert nself > 0                      # the last frames exercise collisionDetection / contactSorting / self friction
    # the reference averaged 201 PD iterations per step over the whole 250-step run (forwardLog.txt: 50295 / 250)
    assert 50 < iters / K < 600


def test_loss_of_logged_evaluation_0_matches_the_reference_log(tshirt):
    """forwardLog.txt record 0: MATCH_TRAJECTORY loss 9.52254 of the 250-step rollout at the first L-BFGS guess against the
    ground-truth rollout (wind 0.015 * normalize(1, 0.1, 1), frequency 10, phase 0.5, k_stretch 550:
    OptimizationTaskSetup.cpp:163-173) — both rollouts by the oracle. Loss = sum_frames |x - x_truth|^2 / (251 N)
    (Simulation.cpp:3260-3274). ~1 minute of CPU: the one long test of the CPU suite."""
    g, P, F, att, o = tshirt
    cfg = scenes.TSHIRT
    xf = P[att].reshape(-1)

    def rollout(k_stretch, fw):
        o.set(k_stretch=float(k_stretch)); o.set_wind(True, 2, fw[0:3] / np.linalg.norm(fw[0:3]), float(np.linalg.norm(fw[0:3])), float(fw[3]), float(fw[4]))
        o.build(); o.clear_records()
        x = P.reshape(-1).copy(); v = np.zeros_like(x)
        traj = [x.copy()]
        for k in range(1, 251):
            out = o.step(x, v, xf, t_prev=(k - 1) * cfg["h"])
            x, v = out["x"], out["v"]
            traj.append(x.copy())
            if k % 50 == 0:
                o.clear_records()
        return np.asarray(traj)

    truth = rollout(550.0, np.array([*(0.015 * np.array([1, 0.1, 1]) / np.sqrt(2.01)), 10.0, 0.5]))
    guess = rollout(g["log_k"][0], g["log_wind"][0])
    L = ((guess - truth) ** 2).sum() / (251 * len(P))
    print(f"\n[golden tshirt] loss of logged evaluation 0: oracle {L:.5f}, reference log {float(g['losses'][0]):.5f}")
    assert abs(L - float(g["losses"][0])) <= 0.01 * float(g["losses"][0])
    # restore the fixture's parameters for other tests of this module
    fw = g["f_wind"]
    o.set(k_stretch=float(g["k_stretch"])); o.set_wind(True, 2, fw[0:3] / np.linalg.norm(fw[0:3]), float(np.linalg.norm(fw[0:3])), float(fw[3]), float(fw[4])); o.build()
