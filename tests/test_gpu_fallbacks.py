"""The fallback kernels (taken for meshes the packet-ELL / element-window paths cannot hold: N > 10 240, bandwidth
> 511 even after renumbering, windows that do not fit LDS) must stay parity-green. They are selected per process by
development switches read once by the launchers, so each variant runs the core parity tests in a subprocess."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORE = "test_forward_step_matches_oracle_with_contact or test_backward_step_matches_oracle or test_forward_and_backward_with_self_contacts " \
       "or test_parameter_gradients_match_oracle or test_free_running_tshirt or test_fused_rollout_with_self_contacts " \
       "or test_two_contexts_agree or test_random_scene_step_matches_oracle"


@pytest.mark.parametrize("env", [
    {"DC_FWD_VARIANT": "global", "DC_WINDOWS": "0"},      # everything in global memory (any N)
    {"DC_FWD_VARIANT": "0", "DC_WINDOWS": "0"},           # ELL resident PCG, global-memory element passes
    {"DC_FWD_VARIANT": "1"},                              # ELL resident PCG (second thread shape), windowed adjoint
    {"DC_RENUMBER": "1"},                                 # default kernels on renumbered grids
    {"DC_SELF_LDS": "0"},                                 # layered self-contact passes through global memory (working sets beyond LDS)
    {"DC_DENSE_MAX_N": "0"},                              # resident PCG instead of the explicit inverse small meshes get by default
])
def test_core_parity_on_fallback_kernels(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"),
                        os.path.join(ROOT, "tests", "test_gpu_selfcontact.py"), os.path.join(ROOT, "tests", "test_gpu_edge_cases.py"), os.path.join(ROOT, "tests", "test_gpu_random_scenes.py"), "-q", "-x", "-k", CORE, "-p", "no:cacheprovider"],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, f"{env}:\n{tail}\n{r.stderr[-2000:]}"
    assert " passed" in tail and "failed" not in tail, tail
