// TEST INFRASTRUCTURE ONLY — C entry points of the fp64 oracle, bound from tests/ via ctypes.
// Nothing under diffcloth_amd/ may include, link or call this code.
#include "orc_sim.h"
#include <cstring>
#include <chrono>
#include <omp.h>

using namespace orc;

extern "C" {

void *orc_create() { return new Sim(); }
void orc_destroy(void *s) { delete (Sim *) s; }

void orc_set_mesh(void *s, int N, const double *pos, int T, const int *tris) { ((Sim *) s)->setMesh(N, pos, T, tris); }
void orc_set_attachments(void *s, int Af, const int *vtx) { ((Sim *) s)->att.assign(vtx, vtx + Af); }

// params: h, density, k_stretch, k_bend, k_att, gx, gy, gz, fwd_tol, bwd_tol   (10 doubles)
// flags:  gravity, contact, selfcollision, gradientClipping, calcSeparateAtp, pd_iter_cap, threads   (7 ints)
void orc_set_params(void *s, const double *p, const int *flags) {
  Sim *S = (Sim *) s;
  S->P.h = p[0]; S->P.density = p[1]; S->P.k_stretch = p[2]; S->P.k_bend = p[3]; S->P.k_att = p[4];
  S->P.gravity = V3(p[5], p[6], p[7]); S->P.fwd_tol = p[8]; S->P.bwd_tol = p[9];
  S->P.gravityEnabled = flags[0]; S->P.contactEnabled = flags[1]; S->P.selfcollisionEnabled = flags[2];
  S->P.gradientClipping = flags[3]; S->P.calcSeparateAtp = flags[4]; S->P.pd_iter_cap = flags[5];
  S->P.threads = flags[6] > 0 ? flags[6] : 1;
}
// windConfig (0 none, 1 constant, 2 sin), wind[3], windNorm, frequency, phase
void orc_set_wind(void *s, int enabled, int config, const double *w) {
  Sim *S = (Sim *) s;
  S->P.windEnabled = enabled; S->P.windConfig = config;
  S->P.wind = V3(w[0], w[1], w[2]); S->P.windNorm = w[3]; S->P.windFrequency = w[4]; S->P.windPhase = w[5];
}
// per-vertex wind fall-off (3N or NULL = ones), constant force field (3N or NULL = off), per-step wind factor of the next step
void orc_set_force_extras(void *s, const double *falloff, const double *field, double perStepFactor) {
  Sim *S = (Sim *) s;
  const size_t n3 = 3 * (size_t) S->N;
  S->windFallOff.clear(); S->external_force_field.clear();
  if (falloff) S->windFallOff.assign(falloff, falloff + n3);
  if (field) S->external_force_field.assign(field, field + n3);
  S->P.enableConstantForcefield = field != nullptr;
  S->P.perStepWindFactor = perStepFactor;
}
void orc_clear_primitives(void *s) { ((Sim *) s)->prims.clear(); }
void orc_add_sphere(void *s, const double *center, double radius, double mu, int rotates) {
  Primitive p; p.kind = PRIM_SPHERE; p.center = V3(center[0], center[1], center[2]); p.radius = radius; p.mu = mu; p.rotates = rotates;
  ((Sim *) s)->prims.push_back(p);
}
// Sphere with discretized = true (Simulation.cpp:1905-1911, BIG_SPHERE): face normals of its resolution x resolution mesh (40 in the reference)
void orc_add_discretized_sphere(void *s, const double *center, double radius, double mu, int resolution) {
  Primitive p; p.kind = PRIM_SPHERE; p.center = V3(center[0], center[1], center[2]); p.radius = radius; p.mu = mu; p.rotates = false;
  p.discretized = true; p.mesh = buildSphereMesh(radius, resolution > 0 ? resolution : 40);
  ((Sim *) s)->prims.push_back(p);
}
// diagnostics: the mesh of buildSphereMesh as 12 doubles per triangle (p0, p1, p2, normal); returns the triangle count
int orc_sphere_mesh(double radius, int resolution, double *out, int cap) {
  const std::vector<Primitive::Tri> m = buildSphereMesh(radius, resolution);
  for (int t = 0; t < (int) m.size() && t < cap; t++) {
    const V3 *v[4] = {&m[t].p0, &m[t].p1, &m[t].p2, &m[t].normal};
    for (int k = 0; k < 4; k++) for (int d = 0; d < 3; d++) out[12 * t + 3 * k + d] = (*v[k])[d];
  }
  return (int) m.size();
}
void orc_add_capsule(void *s, const double *center, const double *topOffset, double radius, double length, double mu) {
  Primitive p; p.kind = PRIM_CAPSULE; p.center = V3(center[0], center[1], center[2]);
  p.topOffset = V3(topOffset[0], topOffset[1], topOffset[2]); p.radius = radius; p.length = length; p.mu = mu;
  ((Sim *) s)->prims.push_back(p);
}
// Plane (Primitive.cpp:13-47): centre + the corners upperLeft, upperRight given relative to the centre
void orc_add_plane(void *s, const double *center, const double *upperLeft, const double *upperRight, double mu) {
  Primitive p; p.kind = PRIM_PLANE; p.center = V3(center[0], center[1], center[2]); p.mu = mu;
  p.upperLeft = V3(upperLeft[0], upperLeft[1], upperLeft[2]); p.upperRight = V3(upperRight[0], upperRight[1], upperRight[2]);
  ((Sim *) s)->prims.push_back(p);
}
void orc_add_bowl(void *s, const double *center, double radius, double mu) {
  Primitive p; p.kind = PRIM_BOWL; p.center = V3(center[0], center[1], center[2]); p.radius = radius; p.mu = mu;
  ((Sim *) s)->prims.push_back(p);
}
// LowerLeg = joint sphere + foot capsule + leg capsule (Primitive.cpp:383-418); children given by their
// centerInit offsets: child k: kind, centerInit[3], topOffset[3], radius, length  (9 doubles each)
void orc_add_lower_leg(void *s, const double *center, double mu, int nchild, const double *child) {
  Primitive p; p.kind = PRIM_LOWER_LEG; p.center = V3(center[0], center[1], center[2]); p.mu = mu;
  for (int k = 0; k < nchild; k++) {
    const double *c = child + 9 * k;
    Primitive q; q.kind = (int) c[0]; q.centerInit = V3(c[1], c[2], c[3]); q.topOffset = V3(c[4], c[5], c[6]);
    q.radius = c[7]; q.length = c[8];
    p.children.push_back(q);
  }
  ((Sim *) s)->prims.push_back(p);
}
void orc_set_mu(void *s, int prim, double mu) { ((Sim *) s)->prims[prim].mu = mu; }
void orc_build(void *s) { ((Sim *) s)->build(); }

void orc_counts(void *s, int *out) {   // N, T, E, Af, nnz(P), rows
  Sim *S = (Sim *) s;
  out[0] = S->N; out[1] = (int) S->tris.size(); out[2] = (int) S->bends.size(); out[3] = (int) S->att.size();
  out[4] = (int) S->Pcol.size(); out[5] = (int) S->rows.size();
}
void orc_get_P(void *s, int *ptr, int *col, double *val) {
  Sim *S = (Sim *) s;
  std::memcpy(ptr, S->Pptr.data(), sizeof(int) * S->Pptr.size());
  std::memcpy(col, S->Pcol.data(), sizeof(int) * S->Pcol.size());
  std::memcpy(val, S->Pval.data(), sizeof(double) * S->Pval.size());
}
void orc_get_vertex_data(void *s, double *mass, double *area, double *radii) {
  Sim *S = (Sim *) s;
  std::memcpy(mass, S->mass.data(), sizeof(double) * S->N);
  std::memcpy(area, S->area.data(), sizeof(double) * S->N);
  std::memcpy(radii, S->radii.data(), sizeof(double) * S->N);
}
void orc_get_bends(void *s, int *idx, double *wv, double *n) {
  Sim *S = (Sim *) s;
  for (size_t e = 0; e < S->bends.size(); e++) {
    for (int k = 0; k < 4; k++) { idx[4 * e + k] = S->bends[e].v[k]; wv[4 * e + k] = S->bends[e].wv[k]; }
    n[e] = S->bends[e].n;
  }
}
void orc_tri_project(void *s, int t, const double *x, double *out6) { Sim *S = (Sim *) s; S->triProject(S->tris[t], x, out6); for (int k = 0; k < 6; k++) out6[k] *= S->tris[t].w; }
void orc_tri_project_backward(void *s, int t, const double *x, double *out54) {
  Sim *S = (Sim *) s; Mat J = S->triProjectBackward(S->tris[t], x) * S->tris[t].w;
  std::memcpy(out54, J.a.data(), sizeof(double) * 54);
}
void orc_bend_project(void *s, int e, const double *x, double *out3) { Sim *S = (Sim *) s; S->bendProject(S->bends[e], x, out3); for (int k = 0; k < 3; k++) out3[k] *= S->bends[e].w; }
void orc_bend_backward(void *s, int e, const double *x, double *out36) {
  Sim *S = (Sim *) s; Mat J = S->bendBackward(S->bends[e], x);
  std::memcpy(out36, J.a.data(), sizeof(double) * 36);
}
void orc_friction(const double *n, const double *f, double mu, double *r, int *type, double *J9, double *dmu) {
  V3 nn(n[0], n[1], n[2]), ff(f[0], f[1], f[2]);
  int ty; V3 rr = Sim::dryFrictionForce(nn, ff, mu, ty);
  r[0] = rr.x; r[1] = rr.y; r[2] = rr.z; *type = ty;
  M3 J = Sim::dri_dfi(nn, ff, mu); std::memcpy(J9, J.data(), sizeof(double) * 9);
  V3 dm = Sim::dri_dmu(nn, ff, mu); dmu[0] = dm.x; dmu[1] = dm.y; dmu[2] = dm.z;
}
void orc_solveP(void *s, const double *rhs, double *out) {
  Sim *S = (Sim *) s;
  std::vector<double> r(rhs, rhs + 3 * S->N), o;
  S->solveP(r, o);
  std::memcpy(out, o.data(), sizeof(double) * o.size());
}

void orc_clear_records(void *s) { ((Sim *) s)->records.clear(); }
// One forward step from (x_n, v_n) with fixed-point targets x_fixed (3*Af). Returns the record index.
// info: [converged, convergeIter, nPrimContacts, nSelfContacts, nLayers]
int orc_step(void *s, const double *x_n, const double *v_n, const double *x_fixed, double t_prev, double *x_new,
             double *v_new, int *info, int frozenContactsFrom) {
  Sim *S = (Sim *) s;
  int id = S->step(x_n, v_n, x_fixed, t_prev, frozenContactsFrom);
  const Record &r = S->records[id];
  std::memcpy(x_new, r.x.data(), sizeof(double) * r.x.size());
  std::memcpy(v_new, r.v.data(), sizeof(double) * r.v.size());
  if (info) {
    info[0] = r.converged; info[1] = r.convergeIter; info[2] = (int) r.prim.size();
    int ns = 0; for (auto &l : r.layers) ns += (int) l.size();
    info[3] = ns; info[4] = (int) r.layers.size();
  }
  return id;
}
void orc_get_record(void *s, int id, double *f, double *r) {
  Sim *S = (Sim *) s; const Record &rec = S->records[id];
  if (f) std::memcpy(f, rec.f.data(), sizeof(double) * rec.f.size());
  if (r) std::memcpy(r, rec.r.data(), sizeof(double) * rec.r.size());
}
// prim contacts of a record: per contact [particle, prim, type] ints and [normal(3), d(3), r(3)] doubles
int orc_get_prim_contacts(void *s, int id, int *ints, double *dbls, int cap) {
  Sim *S = (Sim *) s; const Record &rec = S->records[id];
  int n = std::min((int) rec.prim.size(), cap);
  for (int k = 0; k < n; k++) {
    const PrimContact &c = rec.prim[k];
    ints[3 * k] = c.particleId; ints[3 * k + 1] = c.primitiveId; ints[3 * k + 2] = c.type;
    for (int d = 0; d < 3; d++) { dbls[9 * k + d] = c.normal[d]; dbls[9 * k + 3 + d] = c.d[d]; dbls[9 * k + 6 + d] = c.r[d]; }
  }
  return (int) rec.prim.size();
}
// self contacts of a record, in layer order: [p1, p2, layer, type] ints and [normal(3), d(3)] doubles
int orc_get_self_contacts(void *s, int id, int *ints, double *dbls, int cap) {
  Sim *S = (Sim *) s; const Record &rec = S->records[id];
  int k = 0, total = 0;
  for (size_t l = 0; l < rec.layers.size(); l++)
    for (const SelfContact &c : rec.layers[l]) {
      total++;
      if (k >= cap) continue;
      ints[4 * k] = c.particleId1; ints[4 * k + 1] = c.particleId2; ints[4 * k + 2] = (int) l; ints[4 * k + 3] = c.type;
      for (int d = 0; d < 3; d++) { dbls[6 * k + d] = c.normal[d]; dbls[6 * k + 3 + d] = c.d[d]; }
      k++;
    }
  return total;
}
// Backward through record `id`. scal: [dL_dk_stretch, dL_dk_bend, dL_dk_att, dL_ddensity, dL_dwind(5), dL_dwindtimestep, relative
// residual of the direct solve or -1] (11 doubles)
// info: [converged, backwardIters, usedDirect]
void orc_step_backward(void *s, int id, const double *dL_dxnew, const double *dL_dvnew, const double *dL_dxinit,
                       const double *dL_dvinit, int isStart, int forceDirect, double *dL_dx, double *dL_dv,
                       double *dL_dxfixed, int numMu, double *dL_dmu, double *scal, int *info, double *dfext_vec /*3N or NULL*/) {
  Sim *S = (Sim *) s;
  BackwardOut o = S->stepBackward(S->records[id], dL_dxnew, dL_dvnew, dL_dxinit, dL_dvinit, isStart, forceDirect, numMu);
  std::memcpy(dL_dx, o.dL_dx.data(), sizeof(double) * o.dL_dx.size());
  std::memcpy(dL_dv, o.dL_dv.data(), sizeof(double) * o.dL_dv.size());
  if (dL_dxfixed) std::memcpy(dL_dxfixed, o.dL_dxfixed.data(), sizeof(double) * o.dL_dxfixed.size());
  if (dL_dmu) std::memcpy(dL_dmu, o.dL_dmu.data(), sizeof(double) * o.dL_dmu.size());
  if (scal) { for (int k = 0; k < 3; k++) scal[k] = o.dL_dk[k]; scal[3] = o.dL_ddensity; for (int k = 0; k < 5; k++) scal[4 + k] = o.dL_dwind[k]; }
  if (info) { info[0] = o.converged; info[1] = o.backwardIters; info[2] = o.usedDirect; }
  if (scal) { scal[9] = o.dL_dwindtimestep; scal[10] = o.directResidual; }
  if (dfext_vec) std::memcpy(dfext_vec, o.dL_dfext_vec.data(), sizeof(double) * o.dL_dfext_vec.size());
}
// Collision detection + layering only (for tests of Sim.cpp:225-624).
int orc_detect(void *s, const double *x_n, const double *v, int *nprim, int *nself, int *nlayers) {
  Sim *S = (Sim *) s;
  std::vector<double> x(x_n, x_n + 3 * S->N), vv(v, v + 3 * S->N);
  std::vector<PrimContact> prim; std::vector<std::vector<SelfContact>> layers;
  S->collisionDetection(x, vv, seg3(x, 0), prim, layers);
  *nprim = (int) prim.size(); *nlayers = (int) layers.size();
  int ns = 0; for (auto &l : layers) ns += (int) l.size();
  *nself = ns;
  return 0;
}

}  // extern "C"

// Diagnostic for the parity analysis: round fields of a stored record to fp32, to measure what the precision of
// the tape alone does to the gradient. which: bit 0 = x (x_new), bit 1 = f, r and the contact d vectors.
extern "C" void orc_round_record(void *s, int id, int which) {
  Sim *S = (Sim *) s; Record &rec = S->records[id];
  auto r32 = [](std::vector<double> &a) { for (double &q : a) q = (double) (float) q; };
  if (which & 1) r32(rec.x);
  if (which & 2) {
    r32(rec.f); r32(rec.r);
    for (PrimContact &c : rec.prim) { c.d = V3((float) c.d.x, (float) c.d.y, (float) c.d.z); c.r = V3((float) c.r.x, (float) c.r.y, (float) c.r.z); }
    for (auto &L : rec.layers) for (SelfContact &c : L) c.d = V3((float) c.d.x, (float) c.d.y, (float) c.d.z);
  }
}

// Diagnostic for the solver prototypes (tests/proto_adjoint.py): K = P - dP^T of record `id` as CSC. Two calls: with rowidx == NULL
// it returns the number of entries (and fills colptr), then the caller allocates and calls again.
extern "C" int orc_adjoint_matrix(void *s, int id, int *colptr, int *rowidx, double *val) {
  Sim *S = (Sim *) s;
  static thread_local std::vector<int> cp, ri;
  static thread_local std::vector<double> va;
  if (!rowidx) {
    omp_set_num_threads(S->P.threads);
    S->adjointMatrix(S->records[id], cp, ri, va, 0.0);
    std::memcpy(colptr, cp.data(), sizeof(int) * cp.size());
    return (int) ri.size();
  }
  std::memcpy(rowidx, ri.data(), sizeof(int) * ri.size());
  std::memcpy(val, va.data(), sizeof(double) * va.size());
  return (int) ri.size();
}

// Diagnostic (tests/analyze_dump.py): overwrite x (x_new) and / or f of a stored record with values recorded elsewhere (the HIP
// path's tape) and re-derive the contact vectors d and r from f exactly as Simulation::calculateDryFrictionVector does — to tell
// which recorded quantity a gradient difference between two implementations comes from.
extern "C" void orc_override_record(void *s, int id, const double *x, const double *f) {
  Sim *S = (Sim *) s; Record &rec = S->records[id];
  if (x) rec.x.assign(x, x + rec.x.size());
  if (f) {
    rec.f.assign(f, f + rec.f.size());
    S->dryFrictionVector(rec.f, rec.prim, rec.layers, rec.r);
  }
}

// Diagnostic: the solution a forced direct solve of orc_step_backward is to take (3N doubles; NULL: solve with GMRES again)
extern "C" void orc_set_given_u(void *s, const double *u) {
  Sim *S = (Sim *) s;
  if (u) S->given_u.assign(u, u + 3 * (size_t) S->N); else S->given_u.clear();
}

// Diagnostic, used with orc_override_record to differentiate a record that another engine produced (the "same record on both sides"
// parity tests): contact normals of record `id` replaced — prim_normal (3N, read at the vertices in contact with a primitive; a rotating
// sphere's v_out follows its normal, Primitive.cpp:254-257) and the normals of the self contacts named by pairs (particleId1, particleId2).
// Call it BEFORE orc_override_record(f): that call re-derives d and r of every contact with the normals in place.
extern "C" int orc_override_contacts(void *s, int id, const double *prim_normal, int nself, const int *pairs, const double *self_normal) {
  Sim *S = (Sim *) s; Record &rec = S->records[id];
  if (prim_normal)
    for (PrimContact &c : rec.prim) {
      if (c.primitiveId < 0) continue;
      const V3 old_n = c.normal;
      c.normal = V3(prim_normal[3 * c.particleId], prim_normal[3 * c.particleId + 1], prim_normal[3 * c.particleId + 2]);
      const Primitive &p = S->prims[c.primitiveId];
      if (p.rotates) c.v_out += (V3(0, 1, 0).cross(c.normal) - V3(0, 1, 0).cross(old_n)) * 8;
    }
  int matched = 0;
  if (nself > 0 && pairs && self_normal) {
    std::map<std::pair<int, int>, int> where;
    for (int k = 0; k < nself; k++) where[{pairs[2 * k], pairs[2 * k + 1]}] = k;
    for (auto &layer : rec.layers)
      for (SelfContact &c : layer) {
        auto it = where.find({c.particleId1, c.particleId2});
        if (it == where.end()) continue;
        c.normal = V3(self_normal[3 * it->second], self_normal[3 * it->second + 1], self_normal[3 * it->second + 2]);
        matched++;
      }
  }
  return matched;
}

// Diagnostic: let the local projections see the deformation gradient (the weighted bending vector) rounded to fp32, as an fp32
// evaluation of F = [x1 - x0, x2 - x0] inv_deltaUV delivers it — measures what that rounding alone does to a step and its gradient.
namespace orc { extern bool g_emulate_fp32_F, g_emulate_fp32_v, g_cap_keeps_last; }
extern "C" void orc_emulate_fp32_F(int on) { orc::g_emulate_fp32_F = (on & 1) != 0; orc::g_emulate_fp32_v = (on & 2) != 0; orc::g_cap_keeps_last = (on & 4) != 0; }
