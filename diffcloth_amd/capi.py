"""ctypes binding of libdiffcloth_hip.so (include/diffcloth_hip.h).

This is the thinnest possible Python view of the C-ABI: numpy float64 arrays in the reference's layout
(xyz-interleaved, length 3N per rollout) go straight to the `dc_*` entry points. There is no CPU fallback:
if the library is missing, or no HIP device is present, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdiffcloth_hip.so")

DC_PRIM_SPHERE, DC_PRIM_CAPSULE, DC_PRIM_PLANE, DC_PRIM_BOWL, DC_PRIM_SPHERE_DISCRETIZED = 0, 1, 2, 3, 4


class DcError(RuntimeError):
    pass


class dc_primitive(C.Structure):
    _fields_ = [("kind", C.c_int), ("group", C.c_int), ("center", C.c_double * 3), ("top_offset", C.c_double * 3),
                ("corner2", C.c_double * 3), ("radius", C.c_double), ("length", C.c_double), ("mu", C.c_double), ("rotates", C.c_int)]


class dc_params(C.Structure):
    _fields_ = [("time_step", C.c_double), ("density", C.c_double), ("k_stretch", C.c_double), ("k_bend", C.c_double),
                ("k_att", C.c_double), ("gravity", C.c_double * 3), ("forward_tol", C.c_double),
                ("backward_tol", C.c_double), ("gravity_enabled", C.c_int), ("contact_enabled", C.c_int),
                ("selfcollision_enabled", C.c_int), ("gradient_clipping", C.c_int),
                ("gradient_clipping_threshold", C.c_double), ("pd_iter_cap", C.c_int), ("adjoint_iter_cap", C.c_int),
                ("cg_rel_tol", C.c_double), ("cg_max_iter", C.c_int), ("stall_window", C.c_int),
                ("adjoint_mode", C.c_int), ("adjoint_rel_tol", C.c_double), ("adjoint_block_precond", C.c_int), ("adjoint_fp32_only", C.c_int),
                ("max_self_contacts", C.c_int), ("forward_deflation", C.c_int)]


class dc_step_stats(C.Structure):
    _fields_ = [("converged", C.c_int), ("pd_iters", C.c_int), ("cg_iters", C.c_int), ("prim_contacts", C.c_int),
                ("self_contacts", C.c_int), ("last_xdiff", C.c_float), ("self_overflow", C.c_int)]


class dc_bwd_stats(C.Structure):
    _fields_ = [("converged", C.c_int), ("adjoint_iters", C.c_int), ("cg_iters", C.c_int), ("clipped", C.c_int), ("used_direct", C.c_int),
                ("last_udiff", C.c_float), ("refine_cycles", C.c_int), ("fp64_iters", C.c_int), ("residual_verified", C.c_int), ("workgroups", C.c_int)]


class dc_record(C.Structure):
    _fields_ = [("x", C.POINTER(C.c_double)), ("v", C.POINTER(C.c_double)), ("f", C.POINTER(C.c_double)), ("r", C.POINTER(C.c_double)),
                ("prim", C.POINTER(C.c_int)), ("normal", C.POINTER(C.c_double)), ("x_fixed", C.POINTER(C.c_double)),
                ("self_count", C.POINTER(C.c_int)), ("self_pairs", C.POINTER(C.c_int)), ("self_layer", C.POINTER(C.c_int)),
                ("self_normal", C.POINTER(C.c_double)), ("self_d", C.POINTER(C.c_double))]


EXPORTED_SYMBOLS = [
    "dc_create", "dc_destroy", "dc_last_error", "dc_version", "dc_set_mesh", "dc_set_attachments", "dc_set_params",
    "dc_set_primitives", "dc_build", "dc_default_params", "dc_set_solver", "dc_set_flags", "dc_get_counts", "dc_get_system_matrix",
    "dc_get_vertex_data", "dc_alloc_batch", "dc_set_state", "dc_get_state", "dc_set_mu", "dc_set_uniform_force", "dc_set_vertex_forces", "dc_set_vertex_force_field", "dc_get_force_gradient",
    "dc_step_forward", "dc_get_record", "dc_get_contacts", "dc_get_self_contacts", "dc_step_backward", "dc_rollout_forward",
    "dc_seed_gradient", "dc_rollout_backward", "dc_get_gradient", "dc_get_param_gradients", "dc_get_stats", "dc_sync", "dc_timer_start",
    "dc_timer_stop", "dc_kernel_times", "dc_get_cluster", "dc_set_gradient", "dc_set_fixed_point_schedule", "dc_set_force_schedule",
    "dc_set_seed_schedule", "dc_clear_schedules", "dc_get_states", "dc_get_dxfixed", "dc_get_layout", "dc_comm_unique_id", "dc_comm_init", "dc_allreduce_sum", "dc_comm_destroy",
    "dc_get_deflation", "dc_set_record", "dc_set_trajectory_start", "dc_keep_force_gradients", "dc_get_force_gradients", "dc_use_stream", "dc_set_state_dev", "dc_get_state_dev", "dc_step_forward_dev", "dc_step_backward_dev",
]

_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def load_library():
    """Loads libdiffcloth_hip.so; raises if it has not been built (see diffcloth_amd/build.py)."""
    global _lib
    if _lib is None:
        path = os.environ.get("DC_LIB", LIB_PATH)        # development switch: A/B runs of two builds of the library
        if not os.path.exists(path):
            raise DcError(f"{path} not found: run `python -m diffcloth_amd.build` (hipcc, gfx950)")
        lib = C.CDLL(path)
        lib.dc_last_error.restype = C.c_char_p
        lib.dc_version.restype = C.c_char_p
        _lib = lib
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _i(a):
    return a.ctypes.data_as(_ip) if a is not None else None


def _f64(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _stats_to_dict(arr, fields):
    return {f: np.array([getattr(s, f) for s in arr]) for f in fields}


class Engine:
    """One dc_ctx. Mirrors the call sequence of include/diffcloth_hip.h one to one."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.dc_create(C.c_int(device), C.byref(h))
        if rc != 0:
            raise DcError(f"dc_create failed with code {rc}: no HIP device available (there is no CPU fallback)")
        self.h = h
        self.device = device
        self.params = dc_params()
        self.lib.dc_default_params(C.byref(self.params))
        self.N = self.T = self.E = self.Af = 0
        self.B = 0
        self.ngroups = 1

    def close(self):
        if getattr(self, "h", None):
            self.lib.dc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise DcError(f"code {rc}: {self.lib.dc_last_error(self.h).decode()}")

    # ---- system ----
    def set_mesh(self, verts, tris):
        v = _f64(verts).reshape(-1)
        t = _i32(tris).reshape(-1)
        self._chk(self.lib.dc_set_mesh(self.h, C.c_int(v.size // 3), _d(v), C.c_int(t.size // 3), _i(t)))

    def set_attachments(self, vertices):
        a = _i32(vertices).reshape(-1)
        self._chk(self.lib.dc_set_attachments(self.h, C.c_int(a.size), _i(a)))

    def set_params(self, **kw):
        for k, v in kw.items():
            if k == "gravity":
                for d in range(3):
                    self.params.gravity[d] = float(v[d])
            elif hasattr(self.params, k):
                setattr(self.params, k, v)
            else:
                raise KeyError(k)
        self._chk(self.lib.dc_set_params(self.h, C.byref(self.params)))

    def set_solver(self, forward_tol=None, backward_tol=None, gradient_clipping=None, clip_threshold=None, force_direct_adjoint=None):
        """dc_set_solver: the solver knobs the reference reads from its process-global statics at every step; no rebuild, the batch and
        its tape stay valid. Arguments left None keep their current value."""
        p = self.params
        if forward_tol is not None: p.forward_tol = forward_tol
        if backward_tol is not None: p.backward_tol = backward_tol
        if gradient_clipping is not None: p.gradient_clipping = int(gradient_clipping)
        if clip_threshold is not None: p.gradient_clipping_threshold = clip_threshold
        if force_direct_adjoint is not None: p.adjoint_mode = 1 if force_direct_adjoint else 0
        self._chk(self.lib.dc_set_solver(self.h, C.c_double(p.forward_tol), C.c_double(p.backward_tol), C.c_int(p.gradient_clipping),
                                         C.c_double(p.gradient_clipping_threshold), C.c_int(int(p.adjoint_mode == 1))))

    def set_primitives(self, prims):
        """prims: list of dicts(kind, group, center, top_offset, radius, length, mu, rotates)."""
        arr = (dc_primitive * max(len(prims), 1))()
        groups = []
        for k, p in enumerate(prims):
            arr[k].kind = p.get("kind", DC_PRIM_SPHERE)
            arr[k].group = p.get("group", k)
            for d in range(3):
                arr[k].center[d] = float(p["center"][d])
                arr[k].top_offset[d] = float(p.get("top_offset", (0, 0, 0))[d])
                arr[k].corner2[d] = float(p.get("corner2", (0, 0, 0))[d])
            arr[k].radius = float(p["radius"])
            arr[k].length = float(p.get("length", 0.0))
            arr[k].mu = float(p.get("mu", 0.0))
            arr[k].rotates = int(p.get("rotates", 0))
            if arr[k].group not in groups:
                groups.append(arr[k].group)
        self.ngroups = max(len(groups), 1)
        self._chk(self.lib.dc_set_primitives(self.h, C.c_int(len(prims)), arr))

    def build(self):
        self._chk(self.lib.dc_build(self.h))
        c = _i32(np.zeros(6))
        self._chk(self.lib.dc_get_counts(self.h, _i(c)))
        self.N, self.T, self.E, self.Af, self.nnz, self.rows = [int(x) for x in c]
        return self

    def system_matrix(self):
        ptr = _i32(np.zeros(self.N + 1)); col = _i32(np.zeros(self.nnz)); val = _f64(np.zeros(self.nnz))
        self._chk(self.lib.dc_get_system_matrix(self.h, _i(ptr), _i(col), _d(val)))
        return ptr, col, val

    def vertex_data(self):
        m = _f64(np.zeros(self.N)); a = _f64(np.zeros(self.N)); r = _f64(np.zeros(self.N))
        self._chk(self.lib.dc_get_vertex_data(self.h, _d(m), _d(a), _d(r)))
        return m, a, r

    # ---- batch ----
    def alloc_batch(self, batch, tape_steps):
        self._chk(self.lib.dc_alloc_batch(self.h, C.c_int(batch), C.c_int(tape_steps)))
        self.B = batch
        self.tape = tape_steps

    def _vec(self, a, per):
        a = _f64(a).reshape(-1)
        if a.size != self.B * per:
            raise ValueError(f"expected {self.B}x{per} values, got {a.size}")
        return a

    def set_state(self, slot, x, v):
        x = self._vec(x, 3 * self.N); v = self._vec(v, 3 * self.N)
        self._chk(self.lib.dc_set_state(self.h, C.c_int(slot), _d(x), _d(v)))

    def get_state(self, slot):
        x = np.zeros((self.B, 3 * self.N)); v = np.zeros((self.B, 3 * self.N))
        self._chk(self.lib.dc_get_state(self.h, C.c_int(slot), _d(x), _d(v)))
        return x, v

    def set_mu(self, mu):
        m = None if mu is None else self._vec(mu, self.ngroups)
        self._chk(self.lib.dc_set_mu(self.h, _d(m)))

    def set_uniform_force(self, f):
        a = None if f is None else self._vec(f, 3)
        self._chk(self.lib.dc_set_uniform_force(self.h, _d(a)))

    def set_vertex_forces(self, f):
        a = None if f is None else self._vec(f, 3 * self.N)
        self._chk(self.lib.dc_set_vertex_forces(self.h, _d(a)))

    def set_vertex_force_field(self, f):
        """second per-vertex force term with factor 1 (the constant force field next to a scheduled wind with fall-off)"""
        a = None if f is None else self._vec(f, 3 * self.N)
        self._chk(self.lib.dc_set_vertex_force_field(self.h, _d(a)))

    def get_force_gradient(self):
        """h^2 (I + dr_df)^T u* per vertex of the last backward step (dL_dfext_vec of the reference)."""
        out = np.zeros((self.B, 3 * self.N))
        self._chk(self.lib.dc_get_force_gradient(self.h, _d(out)))
        return out

    def keep_force_gradients(self, keep=True):
        self._chk(self.lib.dc_keep_force_gradients(self.h, C.c_int(int(keep))))

    def get_force_gradients(self, slot0, nslots):
        """h^2 (I + dr_df)^T u* per vertex of the backward steps through records slot0 .. slot0 + nslots - 1 (after keep_force_gradients)."""
        out = np.zeros((nslots, self.B, 3 * self.N))
        self._chk(self.lib.dc_get_force_gradients(self.h, C.c_int(slot0), C.c_int(nslots), _d(out)))
        return out

    # ---- hot path ----
    def step_forward(self, slot, fixed_pts=None, want_stats=True):
        fp = None if fixed_pts is None else self._vec(fixed_pts, 3 * self.Af)
        st = (dc_step_stats * self.B)() if want_stats else None
        self._chk(self.lib.dc_step_forward(self.h, C.c_int(slot), _d(fp), st))
        if want_stats:
            return _stats_to_dict(st, ["converged", "pd_iters", "cg_iters", "prim_contacts", "self_contacts", "last_xdiff", "self_overflow"])
        return None

    def get_record(self, slot):
        f = np.zeros((self.B, 3 * self.N)); r = np.zeros((self.B, 3 * self.N))
        self._chk(self.lib.dc_get_record(self.h, C.c_int(slot), _d(f), _d(r)))
        return f, r

    def set_record(self, slot, x, v, f, prim, normal, r=None, x_fixed=None, self_contacts=None):
        """dc_set_record: the forward record of `slot` handed in from outside (fp64 values kept for the adjoint's fp64 operator).
        prim: [B][N] index into the primitive list or -1; self_contacts: per rollout a dict(pairs [C][2], layer [C], normal [C][3], d [C][3])
        in layer order, or None."""
        rec = dc_record()
        keep = []

        def dptr(a, per):
            if a is None:
                return None
            a = self._vec(a, per); keep.append(a)
            return _d(a)
        n3 = 3 * self.N
        rec.x = dptr(x, n3); rec.v = dptr(v, n3); rec.f = dptr(f, n3); rec.r = dptr(r, n3); rec.normal = dptr(normal, n3)
        rec.x_fixed = dptr(x_fixed, 3 * self.Af)
        pr = _i32(prim).reshape(-1)
        if pr.size != self.B * self.N:
            raise ValueError("prim: expected B x N entries")
        keep.append(pr); rec.prim = _i(pr)
        if self_contacts is not None:
            cnt = _i32([0 if sc is None else len(sc["layer"]) for sc in self_contacts])
            live = [sc for sc in self_contacts if sc is not None and len(sc["layer"])]
            pairs = _i32(np.concatenate([np.asarray(sc["pairs"]).reshape(-1, 2) for sc in live]) if live else np.zeros((0, 2)))
            layer = _i32(np.concatenate([np.asarray(sc["layer"]).reshape(-1) for sc in live]) if live else np.zeros(0))
            nrm = _f64(np.concatenate([np.asarray(sc["normal"]).reshape(-1, 3) for sc in live]) if live else np.zeros((0, 3)))
            dd = _f64(np.concatenate([np.asarray(sc["d"]).reshape(-1, 3) for sc in live]) if live else np.zeros((0, 3)))
            keep += [cnt, pairs, layer, nrm, dd]
            rec.self_count = _i(cnt); rec.self_pairs = _i(pairs); rec.self_layer = _i(layer); rec.self_normal = _d(nrm); rec.self_d = _d(dd)
        self._chk(self.lib.dc_set_record(self.h, C.c_int(slot), C.byref(rec)))

    def get_contacts(self, slot):
        g = np.zeros((self.B, self.N), dtype=np.int32); n = np.zeros((self.B, 3 * self.N))
        self._chk(self.lib.dc_get_contacts(self.h, C.c_int(slot), _i(g), _d(n)))
        return g, n

    def get_self_contacts(self, slot, rollout=0, cap=8192):
        cnt = C.c_int(); nl = C.c_int()
        pairs = np.zeros((cap, 2), dtype=np.int32); layer = np.zeros(cap, dtype=np.int32); nrm = np.zeros((cap, 3))
        self._chk(self.lib.dc_get_self_contacts(self.h, C.c_int(slot), C.c_int(rollout), C.c_int(cap), C.byref(cnt), C.byref(nl),
                                                _i(pairs), _i(layer), _d(nrm)))
        n = min(cnt.value, cap)
        return dict(count=cnt.value, layers=nl.value, pairs=pairs[:n], layer=layer[:n], normal=nrm[:n])

    def step_backward(self, slot, dL_dxnew, dL_dvnew, dL_dxinit=None, dL_dvinit=None, is_start=False):
        n3 = 3 * self.N
        gx = self._vec(dL_dxnew, n3); gv = self._vec(dL_dvnew, n3)
        ix = None if dL_dxinit is None else self._vec(dL_dxinit, n3)
        iv = None if dL_dvinit is None else self._vec(dL_dvinit, n3)
        dx = np.zeros((self.B, n3)); dv = np.zeros((self.B, n3))
        dxf = np.zeros((self.B, max(3 * self.Af, 1))); dmu = np.zeros((self.B, self.ngroups))
        st = (dc_bwd_stats * self.B)()
        self._chk(self.lib.dc_step_backward(self.h, C.c_int(slot), _d(gx), _d(gv), _d(ix), _d(iv), C.c_int(int(is_start)),
                                            _d(dx), _d(dv), _d(dxf), _d(dmu), st))
        out = dict(dL_dx=dx, dL_dv=dv, dL_dxfixed=dxf[:, :3 * self.Af], dL_dmu=dmu)
        out.update(_stats_to_dict(st, ["converged", "adjoint_iters", "cg_iters", "clipped", "used_direct", "last_udiff", "refine_cycles", "fp64_iters", "residual_verified", "workgroups"]))
        return out

    # ---- device-pointer boundary (torch tensors on this GPU: no host copies, no synchronisation) ----
    def _tp(self, t, per):
        """(device pointer, is_f32) of a contiguous CUDA tensor of B * per elements, fp32 or fp64, on the engine's GPU"""
        dev = getattr(self, "device", None)
        import torch
        if t is None:
            return None, 1
        if t.is_cuda and dev is not None and t.device.index != dev:
            raise ValueError(f"tensor on cuda:{t.device.index}, the engine runs on device {dev}")
        if not t.is_cuda or not t.is_contiguous() or t.dtype not in (torch.float32, torch.float64) or t.numel() != per:
            raise ValueError(f"expected a contiguous CUDA float32 / float64 tensor of {per} elements, got {t.dtype} {tuple(t.shape)} on {t.device}")
        return C.c_void_p(t.data_ptr()), int(t.dtype == torch.float32)

    def use_stream(self, stream=None):
        """run the context on a torch.cuda.Stream (None: back to its own stream)"""
        self._chk(self.lib.dc_use_stream(self.h, C.c_void_p(stream.cuda_stream) if stream is not None else None))

    def set_state_dev(self, slot, x, v):
        px, f = self._tp(x, self.B * 3 * self.N); pv, f2 = self._tp(v, self.B * 3 * self.N)
        assert f == f2, "x and v must have one dtype"
        self._chk(self.lib.dc_set_state_dev(self.h, C.c_int(slot), px, pv, C.c_int(f)))

    def get_state_dev(self, slot, x, v):
        px, f = self._tp(x, self.B * 3 * self.N); pv, f2 = self._tp(v, self.B * 3 * self.N)
        assert f == f2, "x and v must have one dtype"
        self._chk(self.lib.dc_get_state_dev(self.h, C.c_int(slot), px, pv, C.c_int(f)))

    def step_forward_dev(self, slot, fixed_pts=None):
        p, f = self._tp(fixed_pts, self.B * 3 * self.Af)
        self._chk(self.lib.dc_step_forward_dev(self.h, C.c_int(slot), p, C.c_int(f)))

    def step_backward_dev(self, slot, gx, gv, dx, dv, dxfixed=None, dmu=None, ix=None, iv=None, is_start=False):
        n = self.B * 3 * self.N
        pgx, f = self._tp(gx, n); pgv, _ = self._tp(gv, n); pdx, f3_ = self._tp(dx, n); pdv, _ = self._tp(dv, n)
        pix, _ = self._tp(ix, n); piv, _ = self._tp(iv, n)
        pxf, _ = self._tp(dxfixed, self.B * 3 * self.Af); pmu, _ = self._tp(dmu, self.B * self.ngroups)
        for t in (gv, dx, dv, ix, iv, dxfixed, dmu):
            assert t is None or t.dtype == gx.dtype, "all tensors of a call must have one dtype"
        self._chk(self.lib.dc_step_backward_dev(self.h, C.c_int(slot), pgx, pgv, pix, piv, C.c_int(int(is_start)), pdx, pdv, pxf, pmu, C.c_int(f)))

    # ---- device-resident rollouts ----
    def rollout_forward(self, slot, nsteps):
        self._chk(self.lib.dc_rollout_forward(self.h, C.c_int(slot), C.c_int(nsteps)))

    def seed_gradient(self, slot, target=None, scale=1.0):
        t = None if target is None else _f64(target).reshape(-1)
        self._chk(self.lib.dc_seed_gradient(self.h, C.c_int(slot), _d(t), C.c_double(scale)))

    def set_trajectory_start(self, start_slot):
        """tape slot of the trajectory's initial state (default 0; -1: this tape holds a later segment — no backward step of it is the isStart step)"""
        self._chk(self.lib.dc_set_trajectory_start(self.h, C.c_int(start_slot)))

    def rollout_backward(self, slot, nsteps):
        self._chk(self.lib.dc_rollout_backward(self.h, C.c_int(slot), C.c_int(nsteps)))

    # ---- device-resident schedules (dc_set_*_schedule) ----
    def set_gradient(self, dL_dx, dL_dv):
        n3 = 3 * self.N
        self._chk(self.lib.dc_set_gradient(self.h, _d(self._vec(dL_dx, n3)), _d(self._vec(dL_dv, n3))))

    def set_fixed_point_schedule(self, slot0, xf):
        """xf: [nsteps][B][3 Af] targets of the steps slot0+k -> slot0+k+1"""
        xf = np.ascontiguousarray(np.asarray(xf, dtype=np.float64).reshape(-1, self.B, 3 * self.Af))
        self._chk(self.lib.dc_set_fixed_point_schedule(self.h, C.c_int(slot0), C.c_int(xf.shape[0]), _d(xf)))

    def set_force_schedule(self, slot0, nsteps, fu=None, fv_scale=None):
        fu = None if fu is None else np.ascontiguousarray(np.asarray(fu, dtype=np.float64).reshape(nsteps, self.B, 3))
        fs = None if fv_scale is None else np.ascontiguousarray(np.asarray(fv_scale, dtype=np.float64).reshape(nsteps, self.B))
        self._chk(self.lib.dc_set_force_schedule(self.h, C.c_int(slot0), C.c_int(nsteps), _d(fu), _d(fs)))

    def set_seed_schedule(self, slot0, dL_dx, dL_dv=None):
        """dL_dx: [nslots][B][3 N] loss gradient w.r.t. the state at slot slot0+k"""
        gx = np.ascontiguousarray(np.asarray(dL_dx, dtype=np.float64).reshape(-1, self.B, 3 * self.N))
        gv = None if dL_dv is None else np.ascontiguousarray(np.asarray(dL_dv, dtype=np.float64).reshape(gx.shape))
        self._chk(self.lib.dc_set_seed_schedule(self.h, C.c_int(slot0), C.c_int(gx.shape[0]), _d(gx), _d(gv)))

    def clear_schedules(self):
        self._chk(self.lib.dc_clear_schedules(self.h))

    def get_states(self, slot0, nslots):
        x = np.zeros((nslots, self.B, 3 * self.N)); v = np.zeros((nslots, self.B, 3 * self.N))
        self._chk(self.lib.dc_get_states(self.h, C.c_int(slot0), C.c_int(nslots), _d(x), _d(v)))
        return x, v

    def get_dxfixed(self, slot0, nslots):
        out = np.zeros((nslots, self.B, max(3 * self.Af, 1)))
        if self.Af > 0:
            out = np.zeros((nslots, self.B, 3 * self.Af))
            self._chk(self.lib.dc_get_dxfixed(self.h, C.c_int(slot0), C.c_int(nslots), _d(out)))
        return out[:, :, :3 * self.Af]

    def get_gradient(self):
        dx = np.zeros((self.B, 3 * self.N)); dv = np.zeros((self.B, 3 * self.N)); dmu = np.zeros((self.B, self.ngroups))
        self._chk(self.lib.dc_get_gradient(self.h, _d(dx), _d(dv), _d(dmu)))
        return dx, dv, dmu

    def get_mu_gradient(self):
        """dL/dmu accumulated by the backward sweep, per rollout and friction group (small copy; waits for the stream)."""
        dmu = np.zeros((self.B, self.ngroups))
        self._chk(self.lib.dc_get_gradient(self.h, None, None, _d(dmu)))
        return dmu

    def get_param_gradients(self, slot):
        out = np.zeros((self.B, 8))
        self._chk(self.lib.dc_get_param_gradients(self.h, C.c_int(slot), _d(out)))
        return dict(dL_dk=out[:, 0:3], dL_ddensity=out[:, 3], sum_dfext=out[:, 4:7])

    def get_stats(self, slot):
        f = (dc_step_stats * self.B)(); b = (dc_bwd_stats * self.B)()
        self._chk(self.lib.dc_get_stats(self.h, C.c_int(slot), f, b))
        return (_stats_to_dict(f, ["converged", "pd_iters", "cg_iters", "prim_contacts", "self_contacts", "last_xdiff", "self_overflow"]),
                _stats_to_dict(b, ["converged", "adjoint_iters", "cg_iters", "clipped", "used_direct", "last_udiff", "refine_cycles", "fp64_iters", "residual_verified", "workgroups"]))

    def sync(self):
        self._chk(self.lib.dc_sync(self.h))

    # ---- RCCL collective of the C-ABI (C++ callers; Python callers normally use torch.distributed) ----
    @staticmethod
    def comm_unique_id():
        lib = load_library()
        buf = C.create_string_buffer(128)
        rc = lib.dc_comm_unique_id(buf)
        if rc != 0:
            raise DcError(f"code {rc}: dc_comm_unique_id failed (RCCL not available?)")
        return buf.raw

    def comm_init(self, nranks, rank, unique_id):
        self._chk(self.lib.dc_comm_init(self.h, C.c_int(nranks), C.c_int(rank), C.c_char_p(unique_id)))

    def allreduce_sum(self, values):
        a = np.ascontiguousarray(np.asarray(values, dtype=np.float64).reshape(-1)).copy()
        self._chk(self.lib.dc_allreduce_sum(self.h, _d(a), C.c_int(a.size)))
        return a

    def comm_destroy(self):
        self._chk(self.lib.dc_comm_destroy(self.h))

    def layout(self):
        """which kernel set dc_build chose (dc_get_layout)"""
        c = _i32(np.zeros(6))
        self._chk(self.lib.dc_get_layout(self.h, _i(c)))
        return dict(renumbered=bool(c[0]), bandwidth=int(c[1]), packet_kernel=bool(c[2]), element_windows=bool(c[3]), windows=int(c[4]), dense_inverse=bool(c[5]))

    def deflation(self):
        """(vectors, probe iterations): the deflation space dc_build chose for the forward solve (dc_get_deflation)"""
        k = C.c_int(); pi = C.c_int()
        self._chk(self.lib.dc_get_deflation(self.h, C.byref(k), C.byref(pi)))
        return k.value, pi.value

    def cluster(self):
        """workgroups per rollout the engine chose for this batch (dc_get_cluster); 1 = one workgroup per rollout"""
        k = C.c_int(); nb = C.c_int()
        self._chk(self.lib.dc_get_cluster(self.h, C.byref(k), C.byref(nb)))
        return k.value

    def timer_start(self):
        self._chk(self.lib.dc_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        self._chk(self.lib.dc_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def kernel_times(self, reset=False):
        a = C.c_float(); b = C.c_float(); na = C.c_int(); nb = C.c_int()
        self._chk(self.lib.dc_kernel_times(self.h, C.byref(a), C.byref(na), C.byref(b), C.byref(nb), C.c_int(int(reset))))
        return dict(fwd_ms=a.value, fwd_launches=na.value, bwd_ms=b.value, bwd_launches=nb.value)
