"""GPU: randomised scenes — grid size, fabric, time step, friction, attachments, obstacle kind and pose, self-collision switch and
adjoint mode drawn from a seed — one settled state each, forward and backward step against the fp64 oracle. Catches what the
hand-picked scenes of the other files do not exercise (odd sizes against the 64-row chunks and the window boundaries, stiff /
soft extremes, obstacles off-centre)."""
import numpy as np
import pytest

import meshes
import orc
import records
from diffcloth_amd import capi

pytestmark = pytest.mark.gpu


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.mark.parametrize("seed", range(30))
def test_random_scene_step_matches_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    nx, ny = int(rng.integers(5, 38)), int(rng.integers(5, 38))
    dim = float(rng.uniform(2.0, 5.0))
    V, F = meshes.grid_cloth(nx, ny, dim, dim * ny / nx, "DOWN")
    V = f32(V + 0.15 * (dim / nx) * rng.standard_normal(V.shape) * np.array([1, 0, 1]))      # irregular triangles
    h = 1.0 / float(rng.choice([90, 120, 180, 240]))
    fabric = dict(density=float(rng.uniform(0.1, 0.5)), k_stretch=float(10 ** rng.uniform(1.5, 3.2)), k_bend=float(10 ** rng.uniform(-4, 0.5)))
    mu = float(rng.uniform(0.05, 0.95))
    natt = int(rng.integers(0, 4))
    att = sorted(int(a) for a in rng.choice(len(V), size=natt, replace=False)) if natt else []
    selfc = bool(rng.integers(0, 2))
    mode = int(rng.integers(0, 2))
    kind = int(rng.integers(0, 3))          # sphere / capsule / plane
    centre = f32(V.mean(axis=0) + np.array([rng.uniform(-0.3, 0.3) * dim, 0.0, rng.uniform(-0.3, 0.3) * dim]))
    o = orc.Oracle(V, F, h=h, fwd_tol=1e-9, bwd_tol=1e-10, attachments=att, selfcollision=selfc, gradient_clipping=False, **fabric)
    if kind == 0:
        R = float(rng.uniform(0.6, 2.0)); c = f32(centre - np.array([0, R + 0.05, 0]))
        o.add_sphere(c, R, mu); prim = dict(kind=capi.DC_PRIM_SPHERE, center=c, radius=R)
    elif kind == 1:
        R = float(rng.uniform(0.4, 0.9)); L = float(rng.uniform(1.0, 3.0))
        axis = rng.standard_normal(3) * np.array([1, 0.1, 1]); axis /= np.linalg.norm(axis)
        top = f32(axis * L); c = f32(centre - np.array([0, R + 0.15, 0]) - 0.5 * top)
        o.add_capsule(c, top, R, L, mu); prim = dict(kind=capi.DC_PRIM_CAPSULE, center=c, top_offset=top, radius=R, length=L)
    else:
        w = float(rng.uniform(0.8, 1.6)); tilt = float(rng.uniform(-0.3, 0.3))
        ul, ur = f32([-w, tilt, -w]), f32([w, tilt, -w]); c = f32(centre - np.array([0, 0.25, 0]))
        o.add_plane(c, ul, ur, mu); prim = dict(kind=capi.DC_PRIM_PLANE, center=c, top_offset=ul, corner2=ur, radius=0.0)
    o.build()
    e = capi.Engine(0)
    e.set_mesh(V, F); e.set_attachments(att)
    e.set_params(time_step=h, forward_tol=1e-9, backward_tol=1e-10, cg_rel_tol=1e-6, cg_max_iter=4000, gradient_clipping=0,
                 selfcollision_enabled=int(selfc), adjoint_mode=mode, adjoint_rel_tol=1e-8, adjoint_iter_cap=400, **fabric)
    e.set_primitives([dict(group=0, mu=mu, **prim)])
    e.build()
    # settle a few steps with the oracle (loose tolerance), clips drifting
    x, v = f32(V.reshape(-1)), np.zeros(V.size)
    xf = V[att].reshape(-1).copy() if att else None
    o.set(fwd_tol=1e-6); o.build()
    for s in range(int(rng.integers(3, 14))):
        if att:
            xf = xf + np.tile([0.01, -0.01, 0.005], len(att))
        out = o.step(x, v, None if xf is None else f32(xf)); x, v = f32(out["x"]), f32(out["v"])
    o.set(fwd_tol=1e-9); o.build()
    e.alloc_batch(2, 1)
    e.set_state(0, np.stack([x, x]), np.stack([v, v]))
    XF = None if xf is None else np.stack([f32(xf), f32(xf)])
    st = e.step_forward(0, fixed_pts=XF)
    ref = o.step(x, v, None if xf is None else f32(xf))
    x1, v1 = e.get_state(1)
    if not ref["converged"]:
        # a draw too stiff for the iteration cap ((-log10 tol) * 150, Simulation.cpp:1182): both sides return their best iterate
        # (that path has its own test, test_gpu_parity.py); here only that the device agrees it did not converge
        assert st["converged"][0] != 1 and st["pd_iters"][0] == ref["iters"]
        assert np.abs(x1[0] - ref["x"]).max() <= 1e-3
        return
    assert np.all(np.isin(st["converged"], (1, 2)))
    assert st["prim_contacts"][0] == ref["nprim"] and st["self_contacts"][0] == ref["nself"]
    dx = np.abs(x1[0] - ref["x"]).max()
    gx = f32(rng.standard_normal(V.size)); gv = f32(0.01 * rng.standard_normal(V.size))
    gb = e.step_backward(1, np.stack([gx, gx]), np.stack([gv, gv]))
    rb = o.step_backward(ref["id"], gx, gv, is_start=False, direct=True)
    ex, ev = rel(gb["dL_dx"][0], rb["dL_dx"]), rel(gb["dL_dv"][0], rb["dL_dv"])
    print(f"\n[random scene {seed}] {nx}x{ny} h=1/{round(1 / h)} k=({fabric['k_stretch']:.0f}, {fabric['k_bend']:.1e}) att {natt} prim {kind} self {int(selfc)} "
          f"mode {mode}: contacts {ref['nprim']}/{ref['nself']}, PD {st['pd_iters'][0]}/{ref['iters']}, max|dx| {dx:.1e}, grad err {ex:.1e} {ev:.1e}, "
          f"adjoint its {gb['adjoint_iters'][0]} status {gb['converged'][0]}")
    assert dx <= 5e-5 * max(dim / 4.5, 1.0)
    assert np.array_equal(x1[0], x1[1]) and np.array_equal(gb["dL_dx"][0], gb["dL_dx"][1])       # the two slots got the same input
    # END TO END the flat 1e-4 gate holds wherever the two PD loops stopped after the same number of iterations and no contact sits on a
    # switching surface of the friction law. Slowly converging draws (stiff, ~1000 PD iterations at a contraction of 0.99) can stop a few
    # iterations apart: positions then differ by ~2e-6 and, through the moved linearisation point, the gradients by up to 1e-3 (the hat
    # scene of test_gpu_configs.py shows the same). Contacts within 1e-4 of a switching surface (take-off / stick / slide) can land on
    # the other side of it in fp32: the step is not differentiable there and the two gradients belong to different branches (seen: one of
    # 334 contacts classified differently, 5 % gradient difference). Those draws REPORT the end-to-end difference; what is gated on EVERY
    # draw is the adjoint on one and the same record (the oracle adopts the engine's, tests/records.py) — flat 1e-4.
    con = o.prim_contacts(ref["id"])
    fo, _ = o.record_fr(ref["id"])
    near = 0
    for k, i in enumerate(con["particle"]):
        n = con["normal"][k]; d = fo[3 * i:3 * i + 3]; sd = d @ n; dT = d - sd * n; nd = max(np.linalg.norm(d), 1e-30)
        near += abs(np.linalg.norm(dT) - mu * abs(sd)) / nd < 1e-4 or abs(sd) / nd < 1e-4
    same_stop = int(st["pd_iters"][0]) == int(ref["iters"])
    matched = records.oracle_adopts_gpu_record(o, ref["id"], e, 1, 0, x, x1[0], v1[0], e.get_record(1)[0][0], h)
    assert matched == ref["nself"]
    rb3 = o.step_backward(ref["id"], gx, gv, is_start=False, direct=True)
    ea = max(rel(gb["dL_dx"][0], rb3["dL_dx"]), rel(gb["dL_dv"][0], rb3["dL_dv"]))
    print(f"[random scene {seed}] same record (oracle adopts the engine's): {ea:.1e}; end to end {'GATED 1e-4' if same_stop and not near else 'reported'} "
          f"(same PD stop {same_stop}, contacts near a switching surface {near})")
    assert ea <= 1e-4
    if same_stop and not near:
        assert ex <= 1e-4 and ev <= 1e-4
    else:
        assert ex <= 0.2 and ev <= 0.2          # sanity only: a different branch / stopping point, not a different algorithm
