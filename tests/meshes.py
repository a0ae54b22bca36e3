"""Synthetic cloth meshes for the tests — test infrastructure.

grid_cloth() restates the reference's grid builder (Simulation::createClothMeshFromConfig,
/root/reference/src/code/simulation/Simulation.cpp:2611-2757; getInitParticlePos :1783-1791;
orientation handling Simulation.h:641-671) so that oracle and product can be fed the same raw mesh.
"""
import numpy as np


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
    pts = np.zeros((ny * nx, 3))
    for i in range(ny):
        for j in range(nx):
            pts[i * nx + j] = np.array([j * gsy, -i * gsx, 0.0]) + origin
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
