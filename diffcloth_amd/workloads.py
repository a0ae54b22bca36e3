"""Workload definitions: the synthetic cloth of the headline metric and the reference's demo scenes as plain arrays.

What is restated here is scene GEOMETRY the callers of the hot path feed it (no stepping, no solver): the reference's grid builder
(Simulation::createClothMeshFromConfig, /root/reference/src/code/simulation/Simulation.cpp:2611-2757; getInitParticlePos :1783-1791;
orientation handling Simulation.h:641-671), its model-mesh normalisation (createClothMeshFromModel, :2170-2226), the attachment
search (createAttachments, :2258-2310), primitive placement (initScene, :1894-1944) and the demo tables
(OptimizationTaskConfigurations.cpp:65-163). `bench.py` builds its headline and secondary workloads from this module; the tests
feed the SAME arrays to the fp64 oracle and to the engine (tests/meshes.py and tests/scenes.py re-export these names).

Raw mesh data of the reference's assets (OBJ vertex / face arrays, no code) is a fixture: tests/golden/meshes.npz, written by
tests/golden/make_fixtures.py in the build container; `load_mesh` reads it (override the path with DC_MESH_FIXTURE).
"""
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(_ROOT, "tests", "golden")
GRAVITY = 9.8


def f32(a):
    """values as the device sees them: rounded to float32, carried as float64 (the C-ABI's host edge is fp64)"""
    return np.asarray(a, dtype=np.float32).astype(np.float64)


# ---------------------------------------------------------------- grid cloth ----
def _rot_axis_angle(axis, angle):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def axis_to_rotation(final_dir, initial_dir):
    """engine/UtilityFunctions.h:77-88."""
    f = np.asarray(final_dir, float) / np.linalg.norm(final_dir)
    i = np.asarray(initial_dir, float) / np.linalg.norm(initial_dir)
    if np.linalg.norm(f - i) > 1e-5:
        perp = np.cross(i, f)
        return _rot_axis_angle(perp, np.arccos(np.dot(f, i)))
    return np.eye(3)


def orient(points, orientation, up_vector=(0, 1, 0)):
    """rotatePointsAccordingToConfig + rotatePointsAroundCenter (rotates p - minDim)."""
    if orientation == "FRONT":
        return points
    if orientation == "DOWN":
        R = axis_to_rotation((0, 1, 0), (0, 0, 1))
    elif orientation == "BACK":
        R = axis_to_rotation((0, 0, 1), (1, 0, 0)) @ axis_to_rotation((1, 0, 0), (0, 0, -1))
    elif orientation == "CUSTOM":
        R = axis_to_rotation(up_vector, (0, 1, 0))
    else:
        raise ValueError(orientation)
    return (points - points.min(axis=0)) @ R.T


def grid_cloth(nx, ny=None, dim_x=4.5, dim_y=None, orientation="DOWN"):
    """Returns (verts [N,3] float64, tris [T,3] int32) exactly as the reference numbers/winds them."""
    ny = nx if ny is None else ny
    dim_y = dim_x if dim_y is None else dim_y
    gsx = dim_x / (nx - 1)
    gsy = dim_y / (ny - 1)
    origin = np.array([-(ny - 1) / 4.0 * gsy, 15.0, 0.0])
    ii, jj = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    pts = np.stack([jj * gsy, -ii * gsx, np.zeros_like(ii, dtype=float)], axis=-1).reshape(-1, 3) + origin
    pts = orient(pts, orientation)
    mn, mx = pts.min(axis=0), pts.max(axis=0)
    pts = pts - mn - (mx - mn) / 2

    def pid(a, b):
        if a < 0 or b < 0 or a >= ny or b >= nx:
            return -1
        return a * nx + b

    tris = []
    for i in range(ny):
        for j in range(nx):
            this, left, up, upr = pid(i, j), pid(i, j - 1), pid(i - 1, j), pid(i - 1, j + 1)
            if min(this, up, upr) >= 0:
                tris.append((upr, up, this))   # createTriangle(a,b,c) stores (c,b,a)
            if min(up, this, left) >= 0:
                tris.append((left, this, up))
    return pts, np.asarray(tris, dtype=np.int32)


def sphere_scene_center(verts, radius=2.0):
    """Sphere placement of PLANE_AND_SPHERE (Simulation.cpp:1894-1903) for a grid cloth."""
    mn, mx = verts.min(axis=0), verts.max(axis=0)
    center_low = 0.5 * (mn + mx)
    center_low[1] = mn[1]
    plane_center = center_low - np.array([0, radius * 2 + 0.1, 0])
    return plane_center + np.array([radius * 0.3, radius, radius * 0.1])


def fold_flap(verts, nx, ny, rows, gap):
    """Folds the last `rows` grid rows of a grid_cloth() mesh back over the cloth: row i_f + d (i_f = ny - 1 - rows)
    is laid exactly above row i_f - d, `gap` higher — a flap resting on the cloth, every flap vertex within contact
    distance of the vertex below it when gap < r_a + r_b (Simulation.cpp:194-220, radii :2407-2431). Returns the folded
    positions and the boolean flap mask."""
    V = np.array(verts, dtype=np.float64).reshape(ny, nx, 3).copy()
    i_f = ny - 1 - rows
    assert rows >= 1 and i_f - rows >= 0
    flap = np.zeros((ny, nx), dtype=bool)
    for d in range(1, rows + 1):
        V[i_f + d] = V[i_f - d] + np.array([0.0, gap, 0.0])
        flap[i_f + d] = True
    return V.reshape(-1, 3), flap.reshape(-1)


# The headline workload's material and scene (sphereFabric, OptimizationTaskConfigurations.cpp:81-96; rotatingSphereScene :228-244)
C4_CLOTH = dict(dim=4.5, density=0.3, k_stretch=150.0, k_bend=1e-5, h=1.0 / 180, sphere_radius=2.0, sphere_mu=0.9)


def c4_scene(grid=100, fold_rows=5, fold_gap=0.02):
    """Rest mesh, folded start shape, flap mask, sphere centre of the C4 workload (fp32-representable, as the device sees them)."""
    V, F = grid_cloth(grid, grid, C4_CLOTH["dim"], C4_CLOTH["dim"], "DOWN")
    V = f32(V)
    center = f32(sphere_scene_center(V, C4_CLOTH["sphere_radius"]))
    if fold_rows > 0:
        V0, flap = fold_flap(V, grid, grid, fold_rows, fold_gap)
        V0 = f32(V0)
    else:
        V0, flap = V.copy(), np.zeros(V.shape[0], dtype=bool)
    return V, F, V0, flap, center


def c4_flap_force(mass, flap, multiple):
    """Constant per-vertex force (3N): the flap pressed onto the cloth with `multiple` times its own weight."""
    f = np.zeros((mass.size, 3))
    f[flap, 1] = -multiple * GRAVITY * mass[flap]
    return f32(f).reshape(-1)


def c4_rollout_inputs(V0, ids):
    """Per-rollout start state and friction coefficient, seeded by the global rollout id (translation and mu)."""
    X = np.empty((len(ids), V0.size)); MU = np.empty((len(ids), 1))
    for k, gid in enumerate(ids):
        rng = np.random.default_rng(1000 + int(gid))
        shift = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.09, -0.02), rng.uniform(-0.5, 0.5)])
        X[k] = (V0 + shift).astype(np.float32).reshape(-1)
        MU[k, 0] = rng.uniform(0.1, 0.9)
    return X, MU


# ---------------------------------------------------------------- model meshes and demo tables ----
def load_mesh(name):
    path = os.environ.get("DC_MESH_FIXTURE", os.path.join(GOLDEN, "meshes.npz"))
    z = np.load(path)
    return z[name + "_v"].copy(), z[name + "_f"].copy()


def normalise_model(V, orientation, cloth_dim, up_vector=(0, 1, 0)):
    """Returns (rest positions, restShapeMinDim, restShapeMaxDim)."""
    P = orient(V.copy(), orientation, up_vector)
    mn, mx = P.min(axis=0), P.max(axis=0)
    dim = mx - mn
    scale = dim.max() / cloth_dim
    rest_max = dim / scale
    rest_min = np.zeros(3)
    tr = rest_max / 2.0
    rest_min = rest_min - tr
    rest_max = rest_max - tr
    P = (P - mn) / scale - rest_max
    return P, rest_min, rest_max


def corner_attachments(P, rest_min, rest_max):
    """LEFT_RIGHT_CORNERS_2 on a model mesh: the vertices closest to the upper-left / upper-right goal points."""
    zmid = (rest_min[2] + rest_max[2]) / 2.0
    goals = [np.array([rest_min[0], rest_max[1], zmid]), np.array([rest_max[0], rest_max[1], zmid])]
    out = []
    for g in goals:
        best = 0
        for i in range(len(P)):
            if np.linalg.norm(P[i] - g) < np.linalg.norm(P[best] - g):
                best = i
        out.append(best)
    return out


TSHIRT = dict(mesh="tshirt", orientation="BACK", cloth_dim=6.0, k_stretch=550.0, k_bend=0.01, density=0.124,
              h=1.0 / 90, steps=250, fwd_tol=1e-8, bwd_tol=5e-4)
HAT = dict(mesh="hat", orientation="FRONT", cloth_dim=6.0, k_stretch=1200.0, k_bend=120.0, density=0.224,
           h=1.0 / 100, steps=400, fwd_tol=1e-8, bwd_tol=5e-4, attachments=[394, 32], sphere_radius=2.1, sphere_mu=0.1)
SOCK = dict(mesh="sock", orientation="CUSTOM", cloth_dim=5.0, k_stretch=600.0, k_bend=1.0, density=0.224,
            h=1.0 / 160, steps=400, fwd_tol=1e-9, bwd_tol=5e-4, attachments=[14, 30, 3, 81])
# the garment used for the self-contact parity cases (dress_twirl's mesh in the squashed pose of tests/test_gpu_configs.py)
DRESS = dict(mesh="dress", orientation="FRONT", cloth_dim=8.0, k_stretch=800.0, k_bend=0.05, density=0.2, h=1.0 / 120)
# the reference's slope fabric (Slope/perfFabric4-96x96-onPlane.obj) in the mesh file's own coordinates
PERF_FABRIC = dict(mesh="perf96", raw=True, k_stretch=50.0, k_bend=1e-5, density=0.2, h=1.0 / 100, plane_mu=0.2)


def hat_head_center(rest_min, rest_max, radius=2.1):
    """sphere_head placement of PLANE_BUST_WEARHAT (Simulation.cpp:1932-1944)."""
    low = 0.5 * (rest_min + rest_max)
    low[1] = rest_min[1]
    plane = low - np.array([0, 0.5, 0]) - np.array([0, 0, 4.0])
    return plane + np.array([0, radius + 0.5, -4.0])


def sock_leg(rest_min, rest_max):
    """LowerLeg of the FOOT scene (Simulation.cpp:1916-1925, Primitive.h:350-374): centre and (kind, centerInit,
    topOffset, radius, length) of joint sphere, foot capsule, leg capsule."""
    high = 0.5 * (rest_min + rest_max)
    high[1] = rest_max[1]
    center = high + np.array([0, 3.0, -4.0])
    radius, foot_len, leg_len = 0.8, 4.0, 5.0
    axis = np.array([0.0, 1.0, 0.0])
    foot_rot = axis_to_rotation(axis, (0, 1, 0))
    foot_global = axis_to_rotation(foot_rot @ np.array([0, 1.0, 0]), (0, 1, 0))
    leg_center = foot_rot @ np.array([0, foot_len, 0])
    leg_rot = axis_to_rotation((0, 0.7, 0.3), (0, 1, 0))
    leg_global = axis_to_rotation(leg_rot @ axis, (0, 1, 0))
    children = [
        (0, leg_center, np.zeros(3), radius + 0.05, 0.0),
        (1, np.zeros(3), foot_global @ np.array([0, foot_len, 0]), radius, foot_len),
        (1, leg_center, leg_global @ np.array([0, leg_len, 0]), radius, leg_len),
    ]
    return center, children


def slope_plane(P):
    """the plane the slope fabric lies on: least-squares plane through the mesh, 0.02 below it (centre, two edge offsets)"""
    c0 = P.mean(axis=0)
    n = np.linalg.svd(P - c0)[2][2]
    n = -n if n[1] < 0 else n
    ex = np.array([1.0, 0.0, 0.0]); ex = ex - n * (ex @ n); ex /= np.linalg.norm(ex)
    es = np.cross(n, ex)
    return f32(c0 - 0.02 * n), f32(-3.6 * ex + 3.6 * es), f32(3.6 * ex + 3.6 * es)


# ---------------------------------------------------------------- secondary contact workloads of bench.py ----
# Each returns a dict: P, F, params (dc_params fields), prims (engine primitive dicts; kinds as integers of capi.DC_PRIM_*), att,
# and `start(B, rng)` -> (X0, V0, lead_xf [S0][B][3Af] or None, timed_xf_fn(K) -> [K][B][3Af] or None, mus or None): the state the
# lead-in steps start from, the clip targets of the lead-in steps (run by the ENGINE itself: the workload needs no oracle) and
# of the timed steps. These are the loaded states of tests/test_gpu_configs.py (hat pressed onto the head, sock pulled along the
# foot, squashed dress) and tools/bench_configs.py (slope fabric), not free fall.
PRIM_SPHERE, PRIM_CAPSULE, PRIM_PLANE = 0, 1, 2


def hat_workload():
    """C3: wear_hat, 579 vertices, two clips lowered onto the head sphere (mu 0.1) until the hat is pressed on — clip motion of
    tests/test_gpu_configs.py::test_c3_hat_batch_64 (0.05 down, 0.3 back per step), per-rollout clip offsets and friction."""
    cfg = HAT
    V, F = load_mesh(cfg["mesh"])
    P, rmin, rmax = normalise_model(V, cfg["orientation"], cfg["cloth_dim"])
    P = f32(P)
    center = f32(hat_head_center(rmin, rmax, cfg["sphere_radius"]))
    att = cfg["attachments"]
    step = np.tile([0.0, -0.05, -0.3], len(att))

    def start(B, rng, lead=18):
        x0 = f32(P.reshape(-1))
        X0 = np.stack([f32(x0 + 0.002 * rng.standard_normal(x0.size)) for _ in range(B)])
        off = 0.02 * rng.standard_normal((B, 3 * len(att)))
        base = P[att].reshape(-1)
        lead_xf = np.stack([f32(base + (s + 1) * step + off) for s in range(lead)])

        def timed(K):
            return np.stack([f32(base + (lead + s + 1) * step + off) for s in range(K)])
        return X0, np.zeros_like(X0), lead_xf, timed, f32(rng.uniform(0.05, 0.6, (B, 1)))
    return dict(name="C3 hat x64 pressed onto the head", P=P, F=F, att=att,
                params=dict(time_step=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], selfcollision_enabled=0),
                prims=[dict(kind=PRIM_SPHERE, group=0, center=center, radius=cfg["sphere_radius"], mu=cfg["sphere_mu"])],
                fwd_tol=1e-6, start=start, B=64)


def sock_workload():
    """C5: wear_sock, 1055 vertices, four clips, LowerLeg (joint sphere + two capsules, one friction group); the opening slipped
    over the tip of the foot capsule and pulled along it (tests/test_gpu_configs.py::test_c5_sock_batch_512)."""
    cfg = SOCK
    V, F = load_mesh(cfg["mesh"])
    P, rmin, rmax = normalise_model(V, cfg["orientation"], cfg["cloth_dim"], up_vector=(0, 1, 0))
    P = f32(P)
    center, children = sock_leg(rmin, rmax)
    center = f32(center)
    att = cfg["attachments"]
    prims = [dict(kind=PRIM_SPHERE if k == 0 else PRIM_CAPSULE, group=0, center=center + f32(c0), radius=float(np.float32(r)), mu=0.4,
                  top_offset=f32(t), length=float(np.float32(l))) for k, c0, t, r, l in children]
    rim = P[att[:2] + att[3:]].mean(axis=0)
    Xs = P + (np.array([0.0, 6.3, -4.0]) - rim)
    pull = np.tile(np.array([0.0, 1.0, 0.0]) * 0.04, len(att))

    def start(B, rng, lead=6):
        x0 = f32(Xs.reshape(-1))
        X0 = np.stack([f32(x0 + 0.001 * rng.standard_normal(x0.size)) for _ in range(B)])
        off = 0.01 * rng.standard_normal((B, 3 * len(att)))
        base = Xs[att].reshape(-1)
        lead_xf = np.stack([f32(base + (s + 1) * pull + off) for s in range(lead)])

        def timed(K):
            return np.stack([f32(base + (lead + s + 1) * pull + off) for s in range(K)])
        return X0, np.zeros_like(X0), lead_xf, timed, f32(rng.uniform(0.2, 0.9, (B, 1)))
    return dict(name="C5 sock x512 pulled along the foot", P=P, F=F, att=att,
                params=dict(time_step=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], selfcollision_enabled=0),
                prims=prims, fwd_tol=1e-9, start=start, B=512)


def dress_workload():
    """C4 (shipped garment): the 3 634-vertex dress hanging from its top rim, flattened along z with a closing speed so that the
    sheets touch — self-collision detection, layering, layered friction (tests/test_gpu_configs.py::test_c4_dress_self_contact_batch)."""
    cfg = DRESS
    V, F = load_mesh(cfg["mesh"])
    P, rmin, rmax = normalise_model(V, cfg["orientation"], cfg["cloth_dim"])
    P = f32(P)
    top = np.argsort(-P[:, 1])[:6].tolist()
    X = P.copy(); X[:, 2] *= 0.9
    vel = np.zeros_like(X); vel[:, 2] = -0.1 * np.sign(P[:, 2])

    def start(B, rng, lead=1):
        X0 = np.stack([f32((X + 0.0005 * rng.standard_normal(X.shape)).reshape(-1)) for _ in range(B)])
        V0 = np.stack([f32((vel + 0.005 * rng.standard_normal(X.shape)).reshape(-1)) for _ in range(B)])
        xf = np.stack([f32(X[top].reshape(-1)) for _ in range(B)])
        lead_xf = np.stack([xf for _ in range(lead)])
        return X0, V0, lead_xf, (lambda K: np.stack([xf for _ in range(K)])), None
    return dict(name="dress (3634 vertices) x256 squashed, sheets in self contact", P=P, F=F, att=top,
                params=dict(time_step=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], selfcollision_enabled=1),
                prims=[], fwd_tol=1e-8, start=start, B=256)


def perf_fabric_workload():
    """The reference's 96 x 96 slope fabric (9 216 vertices) sliding on its plane: every vertex in sliding contact."""
    cfg = PERF_FABRIC
    V, F = load_mesh(cfg["mesh"])
    P = f32(V)
    c, t, c2 = slope_plane(P)

    def start(B, rng, lead=2):
        x0 = P.reshape(-1)
        X0 = np.stack([f32(x0 + 0.001 * rng.standard_normal(x0.size)) for _ in range(B)])
        return X0, np.zeros_like(X0), None, (lambda K: None), None
    return dict(name="perfFabric 96x96 (9216 vertices) x256 sliding on the slope plane", P=P, F=F, att=[],
                params=dict(time_step=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], selfcollision_enabled=0),
                prims=[dict(kind=PRIM_PLANE, group=0, center=c, top_offset=t, corner2=c2, radius=0.0, mu=cfg["plane_mu"])],
                fwd_tol=1e-8, start=start, B=256, lead=2)


SECONDARY_WORKLOADS = dict(hat=hat_workload, sock=sock_workload, dress=dress_workload, perf_fabric=perf_fabric_workload)
