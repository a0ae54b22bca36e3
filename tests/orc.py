"""ctypes binding of the fp64 CPU oracle (oracle/liboracle.so) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB_PATH = os.path.join(_ROOT, "oracle", "liboracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _i(a):
    return a.ctypes.data_as(_ip) if a is not None else None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle"), "-s"])
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_create.restype = C.c_void_p
        _lib.orc_step.restype = C.c_int
        _lib.orc_get_prim_contacts.restype = C.c_int
        _lib.orc_get_self_contacts.restype = C.c_int
    return _lib


def f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


class Oracle:
    """One fp64 simulation instance. Vectors are xyz-interleaved float64 of length 3N, as in the reference."""

    def __init__(self, verts, tris, h=1 / 90, density=0.1, k_stretch=100.0, k_bend=0.01, k_att=10000.0,
                 gravity=(0, -9.8, 0), fwd_tol=1e-7, bwd_tol=5e-5, attachments=(), gravity_enabled=True,
                 contact=True, selfcollision=True, gradient_clipping=False, calc_atp=False, pd_iter_cap=-1, threads=1):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_create())
        verts = f64(verts).reshape(-1)
        tris = i32(tris).reshape(-1)
        self.N = verts.size // 3
        self.L.orc_set_mesh(self.h, C.c_int(self.N), _d(verts), C.c_int(tris.size // 3), _i(tris))
        att = i32(attachments)
        self.Af = att.size
        self.L.orc_set_attachments(self.h, C.c_int(att.size), _i(att))
        self.params = dict(h=h, density=density, k_stretch=k_stretch, k_bend=k_bend, k_att=k_att, gravity=gravity,
                           fwd_tol=fwd_tol, bwd_tol=bwd_tol)
        self.flags = dict(gravity=int(gravity_enabled), contact=int(contact), selfcollision=int(selfcollision),
                          clip=int(gradient_clipping), atp=int(calc_atp), cap=pd_iter_cap, threads=threads)
        self._built = False
        self.nprim = 0

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    def _push_params(self):
        p = self.params
        arr = f64([p["h"], p["density"], p["k_stretch"], p["k_bend"], p["k_att"], *p["gravity"], p["fwd_tol"], p["bwd_tol"]])
        f = self.flags
        fl = i32([f["gravity"], f["contact"], f["selfcollision"], f["clip"], f["atp"], f["cap"], f["threads"]])
        self.L.orc_set_params(self.h, _d(arr), _i(fl))

    def set(self, **kw):
        for k, v in kw.items():
            if k in self.params:
                self.params[k] = v
            elif k in self.flags:
                self.flags[k] = v
            else:
                raise KeyError(k)
        self._built = False

    def set_wind(self, enabled, config, wind, norm, freq, phase):
        self.L.orc_set_wind(self.h, C.c_int(int(enabled)), C.c_int(config), _d(f64([*wind, norm, freq, phase])))

    def set_force_extras(self, falloff=None, field=None, per_step_factor=1.0):
        """windFallOff (3N), constant force field (3N), perstepWindFactor of the next step (Simulation.cpp:55-116)"""
        fo = None if falloff is None else f64(falloff).reshape(-1)
        fi = None if field is None else f64(field).reshape(-1)
        self.L.orc_set_force_extras(self.h, None if fo is None else _d(fo), None if fi is None else _d(fi), C.c_double(per_step_factor))

    def add_sphere(self, center, radius, mu, rotates=False):
        self.L.orc_add_sphere(self.h, _d(f64(center)), C.c_double(radius), C.c_double(mu), C.c_int(int(rotates)))
        self.nprim += 1

    def add_discretized_sphere(self, center, radius, mu, resolution=40):
        """Sphere with discretized = true (the BIG_SPHERE scene, Simulation.cpp:1905-1911)"""
        self.L.orc_add_discretized_sphere(self.h, _d(f64(center)), C.c_double(radius), C.c_double(mu), C.c_int(int(resolution)))
        self.nprim += 1

    def add_capsule(self, center, top_offset, radius, length, mu):
        self.L.orc_add_capsule(self.h, _d(f64(center)), _d(f64(top_offset)), C.c_double(radius), C.c_double(length), C.c_double(mu))
        self.nprim += 1

    def add_plane(self, center, upper_left, upper_right, mu):
        """corners relative to the centre (Plane ctor, Primitive.cpp:13-21)"""
        self.L.orc_add_plane(self.h, _d(f64(center)), _d(f64(upper_left)), _d(f64(upper_right)), C.c_double(mu))
        self.nprim += 1

    def add_bowl(self, center, radius, mu):
        self.L.orc_add_bowl(self.h, _d(f64(center)), C.c_double(radius), C.c_double(mu))
        self.nprim += 1

    def add_lower_leg(self, center, mu, children):
        ch = f64(children).reshape(-1)
        self.L.orc_add_lower_leg(self.h, _d(f64(center)), C.c_double(mu), C.c_int(ch.size // 9), _d(ch))
        self.nprim += 1

    def set_mu(self, prim, mu):
        self.L.orc_set_mu(self.h, C.c_int(prim), C.c_double(mu))

    def build(self):
        self._push_params()
        self.L.orc_build(self.h)
        self._built = True
        c = i32(np.zeros(6))
        self.L.orc_counts(self.h, _i(c))
        self.N, self.T, self.E, self.Af, self.nnz, self.nrows = [int(v) for v in c]
        return self

    def P_csr(self):
        ptr = i32(np.zeros(self.N + 1)); col = i32(np.zeros(self.nnz)); val = f64(np.zeros(self.nnz))
        self.L.orc_get_P(self.h, _i(ptr), _i(col), _d(val))
        return ptr, col, val

    def vertex_data(self):
        m = f64(np.zeros(self.N)); a = f64(np.zeros(self.N)); r = f64(np.zeros(self.N))
        self.L.orc_get_vertex_data(self.h, _d(m), _d(a), _d(r))
        return m, a, r

    def bends(self):
        idx = i32(np.zeros(4 * self.E)); wv = f64(np.zeros(4 * self.E)); n = f64(np.zeros(self.E))
        self.L.orc_get_bends(self.h, _i(idx), _d(wv), _d(n))
        return idx.reshape(-1, 4), wv.reshape(-1, 4), n

    def tri_project(self, t, x):
        o = f64(np.zeros(6)); self.L.orc_tri_project(self.h, C.c_int(t), _d(f64(x)), _d(o)); return o

    def tri_project_backward(self, t, x):
        o = f64(np.zeros(54)); self.L.orc_tri_project_backward(self.h, C.c_int(t), _d(f64(x)), _d(o)); return o.reshape(6, 9)

    def bend_project(self, e, x):
        o = f64(np.zeros(3)); self.L.orc_bend_project(self.h, C.c_int(e), _d(f64(x)), _d(o)); return o

    def bend_backward(self, e, x):
        o = f64(np.zeros(36)); self.L.orc_bend_backward(self.h, C.c_int(e), _d(f64(x)), _d(o)); return o.reshape(3, 12)

    def solveP(self, rhs):
        o = f64(np.zeros(3 * self.N)); self.L.orc_solveP(self.h, _d(f64(rhs)), _d(o)); return o

    def clear_records(self):
        self.L.orc_clear_records(self.h)

    def step(self, x, v, x_fixed=None, t_prev=0.0, frozen=-1):
        if not self._built:
            self.build()
        else:
            self._push_params()
        x = f64(x).reshape(-1); v = f64(v).reshape(-1)
        xf = f64(x_fixed).reshape(-1) if x_fixed is not None else f64(np.zeros(max(3 * self.Af, 1)))
        xn = f64(np.zeros(3 * self.N)); vn = f64(np.zeros(3 * self.N)); info = i32(np.zeros(5))
        rid = self.L.orc_step(self.h, _d(x), _d(v), _d(xf), C.c_double(t_prev), _d(xn), _d(vn), _i(info), C.c_int(frozen))
        return dict(id=rid, x=xn, v=vn, converged=bool(info[0]), iters=int(info[1]), nprim=int(info[2]),
                    nself=int(info[3]), nlayers=int(info[4]))

    def record_fr(self, rid):
        f = f64(np.zeros(3 * self.N)); r = f64(np.zeros(3 * self.N))
        self.L.orc_get_record(self.h, C.c_int(rid), _d(f), _d(r))
        return f, r

    def prim_contacts(self, rid):
        cap = self.N
        ints = i32(np.zeros(3 * cap)); d = f64(np.zeros(9 * cap))
        n = self.L.orc_get_prim_contacts(self.h, C.c_int(rid), _i(ints), _d(d), C.c_int(cap))
        ints = ints.reshape(-1, 3)[:n]; d = d.reshape(-1, 9)[:n]
        return dict(particle=ints[:, 0], prim=ints[:, 1], type=ints[:, 2], normal=d[:, 0:3], d=d[:, 3:6], r=d[:, 6:9])

    def self_contacts(self, rid):
        cap = 8 * self.N
        ints = i32(np.zeros(4 * cap)); d = f64(np.zeros(6 * cap))
        n = self.L.orc_get_self_contacts(self.h, C.c_int(rid), _i(ints), _d(d), C.c_int(cap))
        n = min(n, cap)
        ints = ints.reshape(-1, 4)[:n]; d = d.reshape(-1, 6)[:n]
        return dict(p1=ints[:, 0], p2=ints[:, 1], layer=ints[:, 2], type=ints[:, 3], normal=d[:, 0:3], d=d[:, 3:6])

    def step_backward(self, rid, dL_dxnew, dL_dvnew, dL_dxinit=None, dL_dvinit=None, is_start=True, direct=True, num_mu=None):
        n3 = 3 * self.N
        z = np.zeros(n3)
        gx = f64(dL_dxnew).reshape(-1); gv = f64(dL_dvnew).reshape(-1)
        ix = f64(dL_dxinit if dL_dxinit is not None else z).reshape(-1)
        iv = f64(dL_dvinit if dL_dvinit is not None else z).reshape(-1)
        num_mu = self.nprim if num_mu is None else num_mu
        dx = f64(np.zeros(n3)); dv = f64(np.zeros(n3)); dxf = f64(np.zeros(max(3 * self.Af, 1)))
        dmu = f64(np.zeros(max(num_mu, 1))); scal = f64(np.zeros(11)); info = i32(np.zeros(3)); fvec = f64(np.zeros(n3))
        self.L.orc_step_backward(self.h, C.c_int(rid), _d(gx), _d(gv), _d(ix), _d(iv), C.c_int(int(is_start)),
                                 C.c_int(int(direct)), _d(dx), _d(dv), _d(dxf), C.c_int(num_mu), _d(dmu), _d(scal), _i(info), _d(fvec))
        return dict(dL_dx=dx, dL_dv=dv, dL_dxfixed=dxf[:3 * self.Af], dL_dmu=dmu[:num_mu], dL_dk=scal[0:3],
                    dL_ddensity=scal[3], dL_dwind=scal[4:9], dL_dwindtimestep=float(scal[9]), dL_dfext_vec=fvec,
                    converged=bool(info[0]), iters=int(info[1]),
                    used_direct=bool(info[2]), direct_residual=float(scal[10]))

    def override_record(self, rid, x=None, f=None):
        """diagnostic: replace x_new and / or f of record `rid` (contact vectors d, r re-derived from f)"""
        xx = None if x is None else f64(x).reshape(-1)
        ff = None if f is None else f64(f).reshape(-1)
        self.L.orc_override_record(self.h, C.c_int(rid), None if xx is None else _d(xx), None if ff is None else _d(ff))

    def adopt_record(self, rid, x, f, prim_normal=None, self_pairs=None, self_normal=None):
        """Make record `rid` the record ANOTHER engine holds of the same step (same contact sets): x_new, f, and — when given — the
        contact normals (prim_normal: 3N, read at the vertices in primitive contact; self contacts matched by their (id1, id2) pair);
        d and r of every contact are re-derived from f with those normals. Returns the number of self contacts matched."""
        matched = 0
        pn = None if prim_normal is None else f64(prim_normal).reshape(-1)
        if pn is not None or self_pairs is not None:
            sp = i32(self_pairs if self_pairs is not None else np.zeros((0, 2))).reshape(-1)
            sn = f64(self_normal if self_normal is not None else np.zeros((0, 3))).reshape(-1)
            self.L.orc_override_contacts.restype = C.c_int
            matched = self.L.orc_override_contacts(self.h, C.c_int(rid), None if pn is None else _d(pn), C.c_int(sp.size // 2), _i(sp), _d(sn))
        self.override_record(rid, x=x, f=f)
        return matched

    def step_backward_lu(self, rid, dL_dxnew, dL_dvnew, **kw):
        """step_backward(direct=True) with the adjoint system K u = g solved by a sparse LU of the explicit K (scipy, SuperLU) — what the
        reference's solveDirect does (SparseLU, Simulation.cpp:1431-1440) and the robust choice for near-singular K, where the oracle's
        restarted GMRES needs thousands of products. Gradient clipping must be off (g is taken as passed). Returns the usual dict plus
        lu_residual = |g - K u| / |g|."""
        import scipy.sparse.linalg as spla
        assert not self.flags["clip"], "step_backward_lu: gradient clipping rescales g inside the oracle"
        K = self.adjoint_matrix(rid)
        g = f64(dL_dxnew).reshape(-1)
        lu = spla.splu(K.tocsc())
        u = lu.solve(g)
        u = u + lu.solve(g - K @ u)                  # one step of iterative refinement
        res = float(np.linalg.norm(g - K @ u) / max(np.linalg.norm(g), 1e-300))
        uu = f64(u)
        self.L.orc_set_given_u(self.h, _d(uu))
        try:
            out = self.step_backward(rid, dL_dxnew, dL_dvnew, direct=True, **kw)
        finally:
            self.L.orc_set_given_u(self.h, None)
        out["lu_residual"] = res
        return out

    def diagnostics(self, bits):
        """process-wide diagnostic switches of the oracle library (orc_emulate_fp32_F): 1 = deformation gradient rounded to fp32, 2 = velocity
        iterate rounded to fp32, 4 = a PD loop that hits pd_iter_cap keeps its last iterate (the loop stopped after exactly `cap` iterations)"""
        self.L.orc_emulate_fp32_F(C.c_int(int(bits)))

    def adjoint_matrix(self, rid):
        """K = P - dP^T of the direct adjoint solve of record `rid` (3N x 3N, xyz-interleaved) as a scipy CSC matrix — diagnostic
        for the solver prototypes (tests/proto_adjoint.py)."""
        import scipy.sparse as sp
        n3 = 3 * self.N
        self._push_params()
        colptr = i32(np.zeros(n3 + 1))
        self.L.orc_adjoint_matrix.restype = C.c_int
        nnz = self.L.orc_adjoint_matrix(self.h, C.c_int(rid), _i(colptr), None, None)
        rows = i32(np.zeros(nnz)); vals = f64(np.zeros(nnz))
        self.L.orc_adjoint_matrix(self.h, C.c_int(rid), _i(colptr), _i(rows), _d(vals))
        return sp.csc_matrix((vals, rows, colptr), shape=(n3, n3))

    def detect(self, x, v):
        a = C.c_int(); b = C.c_int(); c = C.c_int()
        self.L.orc_detect(self.h, _d(f64(x)), _d(f64(v)), C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value


def friction(n, f, mu):
    L = lib()
    r = f64(np.zeros(3)); J = f64(np.zeros(9)); dmu = f64(np.zeros(3)); ty = C.c_int()
    L.orc_friction(_d(f64(n)), _d(f64(f)), C.c_double(mu), _d(r), C.byref(ty), _d(J), _d(dmu))
    return r, ty.value, J.reshape(3, 3), dmu
