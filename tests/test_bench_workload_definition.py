"""CPU: where bench.py's workload comes from (VERDICT r05 "weak" 8): the package, not tests/. Importing bench.py must not pull in the test helpers or the
oracle (only its cpu_baseline leg may, when it runs), and the workload it builds is the one the parity tests feed to the oracle."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_importing_bench_does_not_import_tests_or_the_oracle():
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "bad = [m for m in ('orc', 'meshes', 'scenes', 'records', 'ledger') if m in sys.modules]; "
            "assert not bad, bad; assert not any(p.rstrip('/').endswith('/tests') for p in sys.path), sys.path; print('ok')" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-2000:]


def test_headline_workload_is_defined_in_the_package_and_is_what_the_tests_use():
    sys.path.insert(0, ROOT)
    import meshes                      # tests' names: re-exports of the package's definitions
    import scenes
    from diffcloth_amd import workloads
    assert meshes.grid_cloth is workloads.grid_cloth and meshes.fold_flap is workloads.fold_flap
    assert scenes.HAT is workloads.HAT and scenes.normalise_model is workloads.normalise_model
    V, F, V0, flap, center = workloads.c4_scene(100, 5, 0.02)
    assert V.shape == (10000, 3) and F.shape == (19602, 3) and int(flap.sum()) == 500
    assert np.array_equal(V, V.astype(np.float32).astype(np.float64)) and np.array_equal(center, center.astype(np.float32).astype(np.float64))
    X, MU = workloads.c4_rollout_inputs(V0, np.arange(3))
    X2, MU2 = workloads.c4_rollout_inputs(V0, np.array([2]))
    assert np.array_equal(X[2], X2[0]) and MU[2, 0] == MU2[0, 0] and 0.1 <= MU.min() and MU.max() <= 0.9      # seeded by the global rollout id
    for key, fn in workloads.SECONDARY_WORKLOADS.items():
        w = fn()
        X0, V0s, lead, timed, mus = w["start"](2, np.random.default_rng(0))
        assert X0.shape == (2, 3 * w["P"].shape[0]) and np.isfinite(X0).all(), key
        if lead is not None:
            assert lead.shape[1:] == (2, 3 * len(w["att"])) and timed(3).shape == (3, 2, 3 * len(w["att"])), key
