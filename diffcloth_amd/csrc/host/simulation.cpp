// Host-side `Simulation` (see simulation.h). Scene construction follows the reference's
// Simulation::createSystem / createClothMeshFromConfig / createClothMeshFromModel / createAttachments / initScene
// (reference Simulation.cpp:1804-2067, 2170-2405, 2611-2757); the per-step work is delegated to the dc_* C-ABI.
#include <cstring>
#include "simulation.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace dchost {

double Simulation::forwardConvergenceThreshold = 1e-7;          // Simulation.cpp:17
double Simulation::backwardConvergenceThreshold = 1e-4 * 0.5;   // Simulation.cpp:19
std::string Simulation::assetRoot = "";

namespace {

typedef std::array<double, 9> Mat3;
Mat3 axisAngle(Vec3d axis, double angle) {
  double n = std::sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
  double x = axis[0] / n, y = axis[1] / n, z = axis[2] / n, c = std::cos(angle), s = std::sin(angle), t = 1 - c;
  return {t * x * x + c, t * x * y - s * z, t * x * z + s * y, t * x * y + s * z, t * y * y + c, t * y * z - s * x,
          t * x * z - s * y, t * y * z + s * x, t * z * z + c};
}
Mat3 identity3() { return {1, 0, 0, 0, 1, 0, 0, 0, 1}; }
Mat3 mul(const Mat3 &a, const Mat3 &b) {
  Mat3 c{};
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) c[3 * i + j] += a[3 * i + k] * b[3 * k + j];
  return c;
}
Vec3d rotv(const Mat3 &m, const Vec3d &v) {
  return {m[0] * v[0] + m[1] * v[1] + m[2] * v[2], m[3] * v[0] + m[4] * v[1] + m[5] * v[2], m[6] * v[0] + m[7] * v[1] + m[8] * v[2]};
}
Vec3d normalized(Vec3d v) { double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); return {v[0] / n, v[1] / n, v[2] / n}; }
// engine/UtilityFunctions.h:77-88
Mat3 axisToRotation(Vec3d finalDir, Vec3d initialDir) {
  finalDir = normalized(finalDir); initialDir = normalized(initialDir);
  Vec3d d = {finalDir[0] - initialDir[0], finalDir[1] - initialDir[1], finalDir[2] - initialDir[2]};
  if (std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) > 1e-5) {
    Vec3d perp = {initialDir[1] * finalDir[2] - initialDir[2] * finalDir[1], initialDir[2] * finalDir[0] - initialDir[0] * finalDir[2],
                  initialDir[0] * finalDir[1] - initialDir[1] * finalDir[0]};
    double angle = std::acos(finalDir[0] * initialDir[0] + finalDir[1] * initialDir[1] + finalDir[2] * initialDir[2]);
    return axisAngle(perp, angle);
  }
  return identity3();
}
void bbox(const VecXd &p, Vec3d &mn, Vec3d &mx) {
  mn = mx = {p[0], p[1], p[2]};
  for (size_t i = 0; i < p.size() / 3; i++)
    for (int d = 0; d < 3; d++) { mn[d] = std::min(mn[d], p[3 * i + d]); mx[d] = std::max(mx[d], p[3 * i + d]); }
}
void check(dc_ctx *c, int rc, const char *what) {
  if (rc != DC_OK) throw std::runtime_error(std::string(what) + ": " + (c ? dc_last_error(c) : "no context"));
}
// engine/MeshFileHandler.h:137-267 (OBJ subset: v / f with optional /vt/vn)
void loadObj(const std::string &path, VecXd &verts, std::vector<int> &tris) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("cannot open mesh file " + path);
  std::string line;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    std::string tag;
    ss >> tag;
    if (tag == "v") { double x, y, z; ss >> x >> y >> z; verts.push_back(x); verts.push_back(y); verts.push_back(z); }
    else if (tag == "f") {
      std::vector<int> idx;
      std::string tok;
      while (ss >> tok) idx.push_back(std::stoi(tok.substr(0, tok.find('/'))) - 1);
      for (size_t k = 1; k + 1 < idx.size(); k++) { tris.push_back(idx[0]); tris.push_back(idx[k]); tris.push_back(idx[k + 1]); }
    }
  }
}

}  // namespace

Simulation::~Simulation() {
  forEachContext([](dc_ctx *c) { if (c) dc_destroy(c); });
}

// The active set's data live in the plain members (everything written for one set keeps working); switching stores them back and loads
// the other set's. The contexts are NOT touched here: the caller hands the state over (selectSetForStep).
void Simulation::activateSet(int i) {
  if (i == currentSysmatId || i < 0 || i >= (int) attachmentSets.size()) return;
  AttachmentSet &o = attachmentSets[currentSysmatId];
  o.vertices = attachmentVertices; o.fixedRest = fixedPointRest; o.fixedCur = fixedPointCur; o.splines = controlPointSplines; o.ctx = ctx;
  const AttachmentSet &n = attachmentSets[i];
  attachmentVertices = n.vertices; fixedPointRest = n.fixedRest; fixedPointCur = n.fixedCur; controlPointSplines = n.splines; ctx = n.ctx;
  currentSysmatId = i;
  paramsFwdTol = paramsBwdTol = -1;        // the solver knobs of this context may be stale: pushParams sends them again
}

// Simulation::step, Simulation.cpp:1053-1068: the last set whose start frame has been reached (counted in records, the initial one
// included) is the one this step runs with. The state the step starts from is handed to that set's context at the same tape slot.
void Simulation::selectSetForStep() {
  if (attachmentSets.size() < 2) return;
  for (int i = (int) attachmentSets.size() - 1; i >= 0; i--)
    if ((int) forwardRecords.size() >= attachmentSets[i].startFrameNum) {
      if (i != currentSysmatId) {
        activateSet(i);
        const ForwardInformation &prev = forwardRecords.back();
        check(ctx, dc_set_state(ctx, prev.deviceSlot, prev.x.data(), prev.v.data()), "dc_set_state (attachment set hand-over)");
      }
      break;
    }
}

// createAttachments, Simulation.cpp:2389-2393: one spline per fixed point, start = end = its rest position
std::vector<Spline> Simulation::restSplines(const std::vector<int> &vertices) const {
  std::vector<Spline> r;
  for (size_t a = 0; a < vertices.size(); a++) {
    const Vec3d q = {rest[3 * (size_t) vertices[a]], rest[3 * (size_t) vertices[a] + 1], rest[3 * (size_t) vertices[a] + 2]};
    r.emplace_back(q, q, 10, (int) a);
  }
  return r;
}

// rotatePointsAccordingToConfig + rotatePointsAroundCenter (Simulation.h:641-671, Simulation.cpp:2151-2168)
static void orientPoints(VecXd &p, const SceneConfiguration &cfg) {
  Mat3 R = identity3();
  switch (cfg.orientation) {
    case FRONT: return;
    case DOWN: R = axisToRotation({0, 1, 0}, {0, 0, 1}); break;
    case BACK: R = mul(axisToRotation({0, 0, 1}, {1, 0, 0}), axisToRotation({1, 0, 0}, {0, 0, -1})); break;
    case CUSTOM_ORIENTATION: R = axisToRotation(cfg.upVector, {0, 1, 0}); break;
  }
  Vec3d mn, mx;
  bbox(p, mn, mx);
  for (size_t i = 0; i < p.size() / 3; i++) {
    Vec3d q = rotv(R, {p[3 * i] - mn[0], p[3 * i + 1] - mn[1], p[3 * i + 2] - mn[2]});
    p[3 * i] = q[0]; p[3 * i + 1] = q[1]; p[3 * i + 2] = q[2];
  }
}

Simulation *Simulation::createSystem(SceneConfiguration cfg, Vec3d /*center*/, bool runBackward_) {
  VecXd pts;
  std::vector<int> tr;
  if (cfg.fabric.isModel) {
    std::string root = assetRoot;
    if (root.empty()) { const char *e = std::getenv("DIFFCLOTH_ASSETS"); root = e ? e : "/root/reference/src/assets/meshes"; }
    loadObj(root + "/" + cfg.fabric.name, pts, tr);
  } else {
    // getInitParticlePos (Simulation.cpp:1783-1791) + triangle pattern of createClothMeshFromConfig (:2716-2735)
    const int nx = cfg.fabric.gridNumX, ny = cfg.fabric.gridNumY;
    const double gsx = cfg.fabric.clothDimX / (nx - 1), gsy = cfg.fabric.clothDimY / (ny - 1);
    for (int i = 0; i < ny; i++)
      for (int j = 0; j < nx; j++) { pts.push_back(j * gsy - (ny - 1) / 4.0 * gsy); pts.push_back(15 - i * gsx); pts.push_back(0); }
    auto pid = [&](int a, int b) { return (a < 0 || b < 0 || a >= ny || b >= nx) ? -1 : a * nx + b; };
    for (int i = 0; i < ny; i++)
      for (int j = 0; j < nx; j++) {
        int self = pid(i, j), left = pid(i, j - 1), up = pid(i - 1, j), upRight = pid(i - 1, j + 1);
        if (self >= 0 && up >= 0 && upRight >= 0) { tr.push_back(upRight); tr.push_back(up); tr.push_back(self); }   // createTriangle stores (c,b,a)
        if (up >= 0 && self >= 0 && left >= 0) { tr.push_back(left); tr.push_back(self); tr.push_back(up); }
      }
  }
  Simulation *s = new Simulation();
  s->sceneConfig = cfg;
  s->runBackward = runBackward_;
  try {
    s->buildFromMesh(pts, tr, cfg.fabric.isModel);
  } catch (...) { delete s; throw; }
  return s;
}

Simulation *Simulation::createSystemFromMesh(SceneConfiguration cfg, const VecXd &verts, const std::vector<int> &tr, bool runBackward_) {
  Simulation *s = new Simulation();
  s->sceneConfig = cfg;
  s->runBackward = runBackward_;
  try {
    s->buildFromMesh(verts, tr, true);
  } catch (...) { delete s; throw; }
  return s;
}

void Simulation::buildFromMesh(VecXd pts, const std::vector<int> &tr, bool isModel) {
  orientPoints(pts, sceneConfig);
  Vec3d mn, mx;
  bbox(pts, mn, mx);
  Vec3d dim = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
  N = (int) pts.size() / 3;
  if (isModel) {   // createClothMeshFromModel (Simulation.cpp:2170-2226)
    const bool keep = sceneConfig.fabric.keepOriginalScalePoint;
    double scale = keep ? 1.0 : std::max(std::max(dim[0], dim[1]), dim[2]) / sceneConfig.fabric.clothDimX;
    if (keep) { restShapeMaxDim = mx; restShapeMinDim = mn; }
    else {
      for (int d = 0; d < 3; d++) { restShapeMaxDim[d] = dim[d] / scale; restShapeMinDim[d] = 0; }
      Vec3d tr2 = {restShapeMaxDim[0] / 2, restShapeMaxDim[1] / 2, restShapeMaxDim[2] / 2};
      for (int d = 0; d < 3; d++) { restShapeMinDim[d] -= tr2[d]; restShapeMaxDim[d] -= tr2[d]; }
      for (int i = 0; i < N; i++)
        for (int d = 0; d < 3; d++) pts[3 * i + d] = (pts[3 * i + d] - mn[d]) / scale - restShapeMaxDim[d];
    }
  } else {         // createClothMeshFromConfig (Simulation.cpp:2677-2703)
    if (!sceneConfig.fabric.keepOriginalScalePoint)
      for (int i = 0; i < N; i++)
        for (int d = 0; d < 3; d++) pts[3 * i + d] = pts[3 * i + d] - mn[d] - dim[d] / 2;
    for (int d = 0; d < 3; d++) { restShapeMaxDim[d] = dim[d] / 2; restShapeMinDim[d] = -dim[d] / 2; }
  }
  for (int d = 0; d < 3; d++) restShapeMidPoint[d] = 0.5 * (restShapeMinDim[d] + restShapeMaxDim[d]);
  rest = pts;
  tris = tr;

  // createAttachments (Simulation.cpp:2258-2405)
  attachmentVertices.clear();
  if (sceneConfig.attachmentPoints == CUSTOM_ARRAY) {
    if (!sceneConfig.customAttachmentVertexIdx.empty()) attachmentVertices = sceneConfig.customAttachmentVertexIdx[0].second;
  } else if (sceneConfig.attachmentPoints == LEFT_RIGHT_CORNERS_2) {
    if (isModel) {
      double zmid = (restShapeMinDim[2] + restShapeMaxDim[2]) / 2.0;
      Vec3d goals[2] = {{restShapeMinDim[0], restShapeMaxDim[1], zmid}, {restShapeMaxDim[0], restShapeMaxDim[1], zmid}};
      for (auto &g : goals) {
        int best = 0;
        auto d2 = [&](int i) { double s = 0; for (int d = 0; d < 3; d++) s += (rest[3 * i + d] - g[d]) * (rest[3 * i + d] - g[d]); return std::sqrt(s); };
        for (int i = 0; i < N; i++) if (d2(i) < d2(best)) best = i;
        attachmentVertices.push_back(best);
      }
    } else {
      attachmentVertices.push_back(0);
      attachmentVertices.push_back(sceneConfig.fabric.gridNumX - 1);
    }
  }
  fixedPointRest.clear();
  for (int a : attachmentVertices) for (int d = 0; d < 3; d++) fixedPointRest.push_back(rest[3 * a + d]);
  fixedPointCur = fixedPointRest;
  initScene();
  // the reference's sysMat vector: one entry per attachment set (one for every configuration but CUSTOM_ARRAY with several entries)
  attachmentSets.clear();
  currentSysmatId = 0;
  const size_t nsets = sceneConfig.attachmentPoints == CUSTOM_ARRAY ? std::max<size_t>(sceneConfig.customAttachmentVertexIdx.size(), 1) : 1;
  attachmentSets.resize(nsets);
  for (size_t si = 1; si < nsets; si++) {
    AttachmentSet &a = attachmentSets[si];
    const auto &cfgSet = sceneConfig.customAttachmentVertexIdx[si];
    a.startFrameNum = (int) (cfgSet.first * sceneConfig.stepNum);
    a.vertices = cfgSet.second;
    for (int v : a.vertices) {
      if (v < 0 || v >= N) throw std::runtime_error("customAttachmentVertexIdx: vertex index out of range");
      for (int d = 0; d < 3; d++) a.fixedRest.push_back(rest[3 * (size_t) v + d]);
    }
    a.fixedCur = a.fixedRest;
    a.splines = restSplines(a.vertices);
  }
  if (nsets > 1) attachmentSets[0].startFrameNum = (int) (sceneConfig.customAttachmentVertexIdx[0].first * sceneConfig.stepNum);
  configureDevice();
  resetSystem();
}

// Simulation::initScene (Simulation.cpp:1804-2067): analytic obstacles of the shipped scenes.
void Simulation::initScene() {
  primitives.clear();
  Vec3d low = {restShapeMidPoint[0], restShapeMinDim[1], restShapeMidPoint[2]};
  switch (sceneConfig.primitiveConfig) {
    case PLANE_AND_SPHERE: {        // :1894-1903
      Primitive s; s.type = SPHERE; s.radius = 2; s.mu = 0.9;
      Vec3d plane = {low[0], low[1] - (s.radius * 2 + 0.1), low[2]};
      s.center = s.centerInit = {plane[0] + s.radius * 0.3, plane[1] + s.radius, plane[2] + s.radius * 0.1};
      primitives.push_back(s);
      break;
    }
    case PLANE_BUST_WEARHAT: {      // :1932-1944
      Primitive s; s.type = SPHERE; s.radius = 2.1; s.mu = 0.1;
      Vec3d plane = {low[0], low[1] - 0.5, low[2] - 4};
      s.center = s.centerInit = {plane[0], plane[1] + s.radius + 0.5, plane[2] - 4};
      primitives.push_back(s);
      break;
    }
    case BIG_SPHERE: {              // :1905-1912
      Primitive s; s.type = SPHERE; s.radius = 15; s.mu = 0.0; s.center = s.centerInit = {-0.50, -16.00, 0.00};
      s.discretized = true;         // veryBigSphere.discretized = true (:1910)
      primitives.push_back(s);
      break;
    }
    case Y0PLANE: {                 // :1811-1819, :1887-1892: the bowl of Simulation.h:458 (the reference also gives every
      Primitive b; b.type = BOWL; b.radius = 0.5; b.mu = 0; b.center = b.centerInit = {0, 0.5, 0};   // particle v = (0, -10, 0))
      primitives.push_back(b);
      break;
    }
    case SLOPE: {                   // :1946-1962 with the slope plane of Simulation.h:474
      Primitive pl; pl.type = PLANE; pl.mu = 0.2;
      const Vec3d c0 = {0, -11, 10}, ul = {-8 - c0[0], -1 - c0[1], -1 - c0[2]}, ur = {8 - c0[0], -1 - c0[1], -1 - c0[2]};
      pl.upperLeft = ul; pl.upperRight = ur;
      const Vec3d lowerRight = {-ul[0], -ul[1], -ul[2]};
      const Vec3d shift = {(lowerRight[0] - ur[0]) * 0.5, (lowerRight[1] - ur[1]) * 0.5, (lowerRight[2] - ur[2]) * 0.5};
      const Vec3d ref = {(restShapeMaxDim[0] + restShapeMinDim[0]) * 0.5, restShapeMinDim[1], restShapeMinDim[2] - 1.0};
      pl.center = pl.centerInit = {ref[0] + shift[0], ref[1] + shift[1] - 2, ref[2] + shift[2]};
      primitives.push_back(pl);
      break;
    }
    case FOOT: {                    // :1916-1925 + LowerLeg::createNewMesh (Primitive.h:350-374)
      Primitive leg; leg.type = LOWER_LEG; leg.isPrimitiveCollection = true; leg.mu = 0;
      Vec3d high = {restShapeMidPoint[0], restShapeMaxDim[1], restShapeMidPoint[2]};
      leg.center = leg.centerInit = {high[0], high[1] + 3, high[2] - 4};
      const double radius = 0.8, footLength = 4, legLength = 5;
      Vec3d axis = normalized(sceneConfig.sockLegOrientation);
      Mat3 footRot = axisToRotation(axis, {0, 1, 0});                       // foot: parentAxis (0,1,0), axis
      Vec3d footGlobalAxis = rotv(footRot, {0, 1, 0});
      Mat3 footGlobalRot = axisToRotation(footGlobalAxis, {0, 1, 0});
      Vec3d legCenter = rotv(footRot, {0, footLength, 0});
      Mat3 legRot = axisToRotation({0, 0.7, 0.3}, {0, 1, 0});               // leg: parentAxis = axis, axis (0,0.7,0.3)
      Vec3d legGlobalAxis = rotv(legRot, axis);
      Mat3 legGlobalRot = axisToRotation(legGlobalAxis, {0, 1, 0});
      Primitive joint; joint.type = SPHERE; joint.radius = radius + 0.05; joint.centerInit = joint.center = legCenter;
      Primitive foot; foot.type = CAPSULE; foot.radius = radius; foot.length = footLength; foot.centerInit = foot.center = {0, 0, 0};
      foot.topOffset = rotv(footGlobalRot, {0, footLength, 0});
      Primitive lg; lg.type = CAPSULE; lg.radius = radius; lg.length = legLength; lg.centerInit = lg.center = legCenter;
      lg.topOffset = rotv(legGlobalRot, {0, legLength, 0});
      leg.primitives = {joint, foot, lg};
      primitives.push_back(leg);
      break;
    }
    default: break;
  }
  // control-point splines: one per fixed point, start = end = rest position (createAttachments, :2389-2393); the grid
  // scenes with CORNERS_2_UP lift the two corners to the opposite edge (:2337-2358)
  controlPointSplines.clear();
  const size_t Af = attachmentVertices.size();
  auto restOf = [&](size_t a) { return Vec3d{fixedPointRest[3 * a], fixedPointRest[3 * a + 1], fixedPointRest[3 * a + 2]}; };
  if (sceneConfig.attachmentPoints == CUSTOM_ARRAY)
    for (size_t a = 0; a < Af; a++) controlPointSplines.emplace_back(restOf(a), restOf(a), 10, (int) a);
  else if (!sceneConfig.fabric.isModel && sceneConfig.trajectory == CORNERS_2_UP && Af >= 2) {
    const int gx = sceneConfig.fabric.gridNumX, gy = sceneConfig.fabric.gridNumY;
    auto initPos = [&](int i, int j) {        // getInitParticlePos (:1783-1791)
      const double sx = sceneConfig.fabric.clothDimX / (gx - 1), sy = sceneConfig.fabric.clothDimY / (gy - 1);
      return Vec3d{j * sy - (gy - 1) / 4.0 * sy, 15 - i * sx, 0};
    };
    controlPointSplines.emplace_back(restOf(0), initPos(gy - 1, 0), 8, 0);
    controlPointSplines.emplace_back(restOf(1), initPos(gy - 1, gx - 1), 8, 1);
  }
  // scene-dependent end points (:1994-2052)
  switch (sceneConfig.trajectory) {
    case CORNERS_1_WEARHAT:
    case CORNERS_2_WEARHAT: {
      if (sceneConfig.primitiveConfig != PLANE_BUST_WEARHAT || primitives.empty() || controlPointSplines.empty()) break;
      const Primitive &head = primitives[0];
      Vec3d tr;
      for (int d = 0; d < 3; d++) tr[d] = head.center[d] - 0.5 * (restShapeMinDim[d] + restShapeMaxDim[d]);
      tr[1] += head.radius * 0.6;
      const size_t n = sceneConfig.trajectory == CORNERS_1_WEARHAT ? 1 : std::min<size_t>(2, controlPointSplines.size());
      for (size_t k = 0; k < n; k++) {
        Spline &sp = controlPointSplines[k];
        sp.segments[0].yUp = 15;
        Vec3d r = restOf(k);
        sp.moveEndPoint(0, {r[0] + tr[0], r[1] + tr[1], r[2] + tr[2]});
      }
      break;
    }
    case CORNERS_2_WEARSOCK: {
      if (sceneConfig.primitiveConfig != FOOT || primitives.empty() || controlPointSplines.size() < 2) break;
      const Primitive &leg = primitives[0];
      const Primitive &shin = leg.primitives.back();
      Vec3d footTop = leg.center;
      footTop[1] += shin.length + shin.radius * 2;
      Vec3d sockTop = {0.5 * (restShapeMinDim[0] + restShapeMaxDim[0]), restShapeMaxDim[1], restShapeMinDim[2] + shin.radius};
      for (Spline &sp : controlPointSplines) {
        sp.segments[0].yUp = -28;
        Vec3d r = restOf((size_t) sp.pFixed);
        sp.moveEndPoint(0, {r[0] + footTop[0] - sockTop[0], r[1] + footTop[1] - sockTop[1], r[2] + footTop[2] - sockTop[2]});
      }
      break;
    }
    default: break;
  }
}

void Simulation::resetSystem(const std::vector<Spline> &controlPoints) {   // Simulation.cpp:2858-2862
  controlPointSplines = controlPoints;
  resetSystem();
}

void Simulation::configureDevice() {
  std::vector<dc_primitive> flat;
  for (size_t g = 0; g < primitives.size(); g++) {
    const Primitive &p = primitives[g];
    auto add = [&](const Primitive &q, const Vec3d &c) {
      dc_primitive d{};
      d.kind = q.type == CAPSULE ? DC_PRIM_CAPSULE : (q.type == PLANE ? DC_PRIM_PLANE : (q.type == BOWL ? DC_PRIM_BOWL : (q.discretized ? DC_PRIM_SPHERE_DISCRETIZED : DC_PRIM_SPHERE)));
      d.group = (int) g;
      for (int k = 0; k < 3; k++) { d.center[k] = c[k]; d.top_offset[k] = q.type == PLANE ? q.upperLeft[k] : q.topOffset[k]; d.corner2[k] = q.upperRight[k]; }
      d.radius = q.radius; d.length = q.length; d.mu = p.mu; d.rotates = q.rotates;
      flat.push_back(d);
    };
    if (p.isPrimitiveCollection)
      for (const Primitive &q : p.primitives) add(q, {p.center[0] + q.centerInit[0], p.center[1] + q.centerInit[1], p.center[2] + q.centerInit[2]});
    else add(p, p.center);
  }
  dc_params prm;
  dc_default_params(&prm);
  prm.time_step = sceneConfig.timeStep;
  prm.density = sceneConfig.fabric.density;
  prm.k_stretch = sceneConfig.fabric.k_stiff_stretching;
  prm.k_bend = sceneConfig.fabric.k_stiff_bending;
  for (int d = 0; d < 3; d++) prm.gravity[d] = gravity[d];
  prm.gravity_enabled = gravityEnabled; prm.contact_enabled = contactEnabled; prm.selfcollision_enabled = selfcollisionEnabled;
  prm.forward_tol = forwardConvergenceThreshold; prm.backward_tol = backwardConvergenceThreshold;
  prm.gradient_clipping = gradientClipping; prm.gradient_clipping_threshold = gradientClippingThreshold;
  tapeSlots = std::max(sceneConfig.stepNum, 1) + 8;
  // one context per attachment set (set 0 = the active members)
  for (size_t si = 0; si < std::max<size_t>(attachmentSets.size(), 1); si++) {
    dc_ctx *c = nullptr;
    check(nullptr, dc_create(0, &c) == DC_OK ? DC_OK : DC_ERR_HIP, "dc_create (no HIP device: the stepper has no CPU path)");
    const std::vector<int> &att = si == 0 ? attachmentVertices : attachmentSets[si].vertices;
    if (si == 0) ctx = c; else attachmentSets[si].ctx = c;
    check(c, dc_set_mesh(c, N, rest.data(), (int) tris.size() / 3, tris.data()), "dc_set_mesh");
    check(c, dc_set_attachments(c, (int) att.size(), att.data()), "dc_set_attachments");
    check(c, dc_set_primitives(c, (int) flat.size(), flat.data()), "dc_set_primitives");
    check(c, dc_set_params(c, &prm), "dc_set_params");
    check(c, dc_build(c), "dc_build");
    check(c, dc_alloc_batch(c, 1, tapeSlots), "dc_alloc_batch");
  }
  fieldSignature.clear();
  paramsFwdTol = prm.forward_tol; paramsBwdTol = prm.backward_tol; paramsClip = gradientClipping;
  paramsClipThr = gradientClippingThreshold; paramsDirect = false;
}

// The constant force field does not change between steps: it goes to the device when its values (or its switch) changed, not at every step.
void Simulation::uploadForceField(dc_ctx *c, bool field) {
  unsigned long long sig = 0;
  if (field) {      // FNV-1a over the bit patterns (3N doubles: microseconds; an upload is a host-to-device copy + a stream synchronisation)
    sig = 1469598103934665603ull;
    for (double v : external_force_field) { unsigned long long b; std::memcpy(&b, &v, 8); sig = (sig ^ b) * 1099511628211ull; }
    if (sig == 0) sig = 1;
  }
  auto it = fieldSignature.find(c);
  if (it != fieldSignature.end() && it->second == sig) return;
  check(c, dc_set_vertex_force_field(c, field ? external_force_field.data() : nullptr), "dc_set_vertex_force_field");
  fieldSignature[c] = sig;
}

// New fabric parameters (stiffness per constraint type, density): constraint weights, lumped masses and the system matrix
// are rebuilt on the device (setConstraintWeight / updateMassMatrix / initializePrefactoredMatrices, Simulation.cpp:3500-3556).
void Simulation::rebuildSystem() {
  dc_params prm;
  dc_default_params(&prm);
  prm.time_step = sceneConfig.timeStep;
  prm.density = sceneConfig.fabric.density;
  prm.k_stretch = sceneConfig.fabric.k_stiff_stretching;
  prm.k_bend = sceneConfig.fabric.k_stiff_bending;
  prm.k_att = k_stiff_attachment;
  for (int d = 0; d < 3; d++) prm.gravity[d] = gravity[d];
  prm.gravity_enabled = gravityEnabled; prm.contact_enabled = contactEnabled; prm.selfcollision_enabled = selfcollisionEnabled;
  prm.forward_tol = forwardConvergenceThreshold; prm.backward_tol = backwardConvergenceThreshold;
  prm.gradient_clipping = gradientClipping; prm.gradient_clipping_threshold = gradientClippingThreshold;
  prm.adjoint_mode = backwardGradientForceDirectSolver ? 1 : 0;
  forEachContext([&](dc_ctx *c) {
    check(c, dc_set_params(c, &prm), "dc_set_params");
    check(c, dc_build(c), "dc_build");
  });
  fieldSignature.clear();
  paramsFwdTol = prm.forward_tol; paramsBwdTol = prm.backward_tol; paramsClip = gradientClipping;
  paramsClipThr = gradientClippingThreshold; paramsDirect = backwardGradientForceDirectSolver;
}

// The reference reads its mutable statics at every step; mirror that by refreshing the solver knobs when they changed.
void Simulation::pushParams() {
  if (paramsFwdTol == forwardConvergenceThreshold && paramsBwdTol == backwardConvergenceThreshold &&
      paramsClip == gradientClipping && paramsClipThr == gradientClippingThreshold && paramsDirect == backwardGradientForceDirectSolver)
    return;
  forEachContext([&](dc_ctx *c) {
    check(c, dc_set_solver(c, forwardConvergenceThreshold, backwardConvergenceThreshold, gradientClipping ? 1 : 0,
                           gradientClippingThreshold, backwardGradientForceDirectSolver ? 1 : 0), "dc_set_solver");
  });
  paramsFwdTol = forwardConvergenceThreshold; paramsBwdTol = backwardConvergenceThreshold; paramsClip = gradientClipping;
  paramsClipThr = gradientClippingThreshold; paramsDirect = backwardGradientForceDirectSolver;
}

void Simulation::setWindAncCollision(bool wind_, bool collision, bool selfCollision, bool) {
  windEnabled = wind_; contactEnabled = collision; selfcollisionEnabled = selfCollision;
  forEachContext([&](dc_ctx *c) { check(c, dc_set_flags(c, gravityEnabled ? 1 : 0, contactEnabled ? 1 : 0, selfcollisionEnabled ? 1 : 0), "dc_set_flags"); });
}

// Simulation::resetSystem (Simulation.cpp:3490-3584 without the parameter overloads): back to the rest pose.
void Simulation::resetSystem() {
  activateSet(0);                      // currentSysmatId = 0 (Simulation.cpp:2841); only set 0's fixed points go back to rest (:2818-2826)
  forwardRecords.clear();
  perStepGradient.clear();
  ForwardInformation r0;
  r0.x = rest; r0.v.assign(rest.size(), 0.0);
  r0.x_prev = r0.x; r0.v_prev = r0.v;
  r0.f.assign(rest.size(), 0.0); r0.r = r0.f; r0.s_n = r0.x;
  r0.x_fixedpoints = fixedPointRest;
  r0.stepIdx = 0; r0.deviceSlot = 0; r0.t = 0;
  forwardRecords.push_back(r0);
  fixedPointCur = fixedPointRest;
  forEachContext([&](dc_ctx *c) { check(c, dc_clear_schedules(c), "dc_clear_schedules"); });     // per-slot inputs of an earlier device-resident evaluation
  check(ctx, dc_set_state(ctx, 0, r0.x.data(), r0.v.data()), "dc_set_state");
}

// ---- Spline (reference Spline.h): cubic Hermite basis h00 p0 + h10 m0 + h01 p1 + h11 m1 on each segment ----
namespace {
inline double h00(double t, int o) { return o == 1 ? 6 * t * t - 6 * t : 2 * t * t * t - 3 * t * t + 1; }
inline double h01(double t, int o) { return o == 1 ? -6 * t * t + 6 * t : -2 * t * t * t + 3 * t * t; }
inline double h10(double t, int o) { return o == 1 ? 3 * t * t - 4 * t + 1 : t * t * t - 2 * t * t + t; }
inline double h11(double t, int o) { return o == 1 ? 3 * t * t - 2 * t : t * t * t - t * t; }
}  // namespace

void Spline::retangent(Segment &seg) {
  for (int d = 0; d < 3; d++) seg.m0[d] = seg.m1[d] = seg.p1[d] - seg.p0[d];
  seg.m0[1] += seg.yUp; seg.m1[1] -= seg.yUp;
}
Spline::Spline(Vec3d p0, Vec3d p1, double yUp, int pFixed_, double startFraction, double endFraction) : pFixed(pFixed_) {
  Segment seg;
  seg.p0 = p0; seg.p1 = p1; seg.yUp = yUp; seg.segId = 0; seg.startFraction = startFraction; seg.endFraction = endFraction;
  retangent(seg);
  segments.push_back(seg);
}
void Spline::addSegment(Vec3d p1, double yUp, double startFraction, double endFraction) {
  if (segments.empty()) { std::fprintf(stderr, "WARNING: calling addSegment but spline is empty at the moment, please init spline with a segment\n"); return; }
  Segment seg;
  seg.p0 = segments.back().p1; seg.p1 = p1; seg.yUp = yUp; seg.segId = (int) segments.size();
  seg.startFraction = startFraction; seg.endFraction = endFraction;
  retangent(seg);
  segments.push_back(seg);
}
void Spline::moveEndPoint(int segId, Vec3d newp1) {
  Segment &seg = segments.at(segId);
  seg.p1 = newp1;
  retangent(seg);
  if (segId + 1 < (int) segments.size()) { Segment &nx = segments[segId + 1]; nx.p0 = newp1; retangent(nx); }
}
const Spline::Segment &Spline::segmentAt(double t) const {
  for (const Segment &s : segments) if (s.endFraction >= t) return s;
  return segments.back();
}
double Spline::localTime(const Segment &seg, double t) {
  if (t > seg.endFraction) return 1;
  if (t < seg.startFraction) return 0;
  return (t - seg.startFraction) / (seg.endFraction - seg.startFraction);
}
Vec3d Spline::evalute(double t, int order) const {
  t = std::min(std::max(t, 0.0), 1.0);
  const Segment &seg = segmentAt(t);
  const double s = localTime(seg, t);
  Vec3d out;
  for (int d = 0; d < 3; d++) out[d] = h00(s, order) * seg.p0[d] + h10(s, order) * seg.m0[d] + h01(s, order) * seg.p1[d] + h11(s, order) * seg.m1[d];
  return out;
}
std::vector<double> Spline::dxfixed_dcontrolPoints(double t) const {
  const int per = parametersPerSegment(type), np = getParameterNumber();
  std::vector<double> J(3 * (size_t) np, 0.0);
  const Segment &seg = segmentAt(t);
  const double s = localTime(seg, t);
  const int off = per * seg.segId;
  for (int d = 0; d < 3; d++) {
    double *row = &J[(size_t) d * np + off];
    if (type == ENDPOINT_AND_TANGENTS) { row[d] = h01(s, 0); row[3 + d] = h10(s, 0); row[6 + d] = h11(s, 0); }
    else row[d] = h01(s, 0) + h10(s, 0) + h11(s, 0);          // p1 moves both tangents with it
  }
  if (type == ENDPOINT_AND_UP) J[(size_t) 1 * np + off + 3] = h10(s, 0) - h11(s, 0);
  return J;
}
VecXd Spline::paramToVector() const {
  VecXd out;
  for (const Segment &seg : segments) {
    for (int d = 0; d < 3; d++) out.push_back(seg.p1[d]);
    if (type == ENDPOINT_AND_UP) out.push_back(seg.yUp);
    if (type == ENDPOINT_AND_TANGENTS) { for (int d = 0; d < 3; d++) out.push_back(seg.m0[d]); for (int d = 0; d < 3; d++) out.push_back(seg.m1[d]); }
  }
  return out;
}
void Spline::updateControlPoints(const VecXd &step) {
  const int per = parametersPerSegment(type);
  if ((int) step.size() != getParameterNumber())
    std::fprintf(stderr, "WARNING: Updating control points for spline type %d but dimension is %zu instead of %d\n", (int) type, step.size(), getParameterNumber());
  for (size_t i = 0; i < segments.size() && (i + 1) * per <= step.size(); i++) {
    Segment &seg = segments[i];
    const double *q = &step[i * per];
    for (int d = 0; d < 3; d++) seg.p1[d] += q[d];
    if (type == ENDPOINT_AND_TANGENTS) { for (int d = 0; d < 3; d++) { seg.m0[d] += q[3 + d]; seg.m1[d] += q[6 + d]; } }
    else { if (type == ENDPOINT_AND_UP) seg.yUp += q[3]; retangent(seg); }
  }
}

double Simulation::windFactorAt(double t, int stepIdx) const {   // fillForces (Simulation.cpp:64-87)
  switch (sceneConfig.windConfig) {
    case WIND_SIN: case WIND_SIN_AND_FALLOFF: return (std::sin(windFrequency * t + windPhase) + 1.0) / 2.0;
    case NO_WIND: return 0.0;
    case WIND_FACTOR_PER_STEP: return stepIdx >= 0 && (size_t) stepIdx < perstepWindFactor.size() ? perstepWindFactor[stepIdx] : 1.0;
    default: return 1.0;
  }
}

void Simulation::setWindFallOffFromFocusPoint(const Vec3d &focus) {
  windFallOff.assign(3 * (size_t) N, 1.0);
  const VecXd &x = forwardRecords.empty() ? rest : forwardRecords.back().x;
  for (int i = 0; i < N; i++) {
    const double dx = focus[0] - x[3 * i], dy = focus[1] - x[3 * i + 1], dz = focus[2] - x[3 * i + 2];
    const double f = std::min(1.0 / std::sqrt(dx * dx + dy * dy + dz * dz), 1.0);     // (the reference's "distSquared" is the norm)
    for (int d = 0; d < 3; d++) windFallOff[3 * (size_t) i + d] = f;
  }
}

// stepFixPoints (Simulation.cpp:964-1018)
VecXd Simulation::fixedPointTargets(double t) {
  const size_t Af = attachmentVertices.size();
  switch (sceneConfig.trajectory) {
    case NO_TRAJECTORY: break;
    case PER_STEP_TRAJECTORY:
      if (rlFixedPointPos.size() == 3 * Af) fixedPointCur = rlFixedPointPos;
      break;
    case FIXED_POINT_TRAJECTORY: {
      const size_t k = forwardRecords.size() - 1;
      if (k < fixedPointTrajectory.size() && fixedPointTrajectory[k].size() == 3 * Af) fixedPointCur = fixedPointTrajectory[k];
      break;
    }
    case TRAJECTORY_DRESS_TWIRL: {
      Mat3 R = axisAngle({0, 1, 0}, 0.02);
      for (size_t a = 0; a < Af; a++) {
        Vec3d c = {restShapeMidPoint[0], fixedPointCur[3 * a + 1], restShapeMidPoint[2]};
        Vec3d rel = {fixedPointCur[3 * a] - c[0], 0.0, fixedPointCur[3 * a + 2] - c[2]};
        Vec3d q = rotv(R, rel);
        fixedPointCur[3 * a] = q[0] + c[0]; fixedPointCur[3 * a + 2] = q[2] + c[2];
      }
      break;
    }
    default: {     // spline-driven trajectories (CORNERS_2_UP, CORNERS_*_WEARHAT, CORNERS_2_WEARSOCK)
      const double frac = t / (sceneConfig.timeStep * sceneConfig.stepNum);
      for (const Spline &s : controlPointSplines)
        if (s.pFixed >= 0 && (size_t) s.pFixed < Af) {
          Vec3d q = s.evalute(frac);
          for (int d = 0; d < 3; d++) fixedPointCur[3 * (size_t) s.pFixed + d] = q[d];
        }
      break;
    }
  }
  return fixedPointCur;
}

void Simulation::step() {
  if ((int) forwardRecords.size() >= tapeSlots) throw std::runtime_error("Simulation::step: tape exhausted (stepNum + 8 records)");
  const auto tStart = std::chrono::steady_clock::now();
  selectSetForStep();
  pushParams();
  const ForwardInformation &prev = forwardRecords.back();
  ForwardInformation rec;
  rec.t = prev.t + sceneConfig.timeStep;
  rec.stepIdx = (int) forwardRecords.size();
  rec.sysMatId = currentSysmatId;
  rec.deviceSlot = prev.deviceSlot + 1;
  rec.x_prev = prev.x; rec.v_prev = prev.v;
  rec.windFactor = windFactorAt(rec.t, rec.stepIdx);       // (the reference indexes perstepWindFactor by forwardRecords.size())
  // fillForces (Simulation.cpp:55-116): uniform wind through dc_set_uniform_force; wind with per-vertex fall-off through
  // dc_set_vertex_forces, the constant force field through dc_set_vertex_force_field (the same two terms a fused rollout uses)
  const bool fallOff = windEnabled && windHasFallOff() && windFallOff.size() == 3 * (size_t) N;
  const bool field = enableConstantForcefield && external_force_field.size() == 3 * (size_t) N;
  if (windEnabled && !fallOff) {
    double f[3];
    for (int d = 0; d < 3; d++) f[d] = wind[d] * windNorm * rec.windFactor;
    check(ctx, dc_set_uniform_force(ctx, f), "dc_set_uniform_force");
  } else check(ctx, dc_set_uniform_force(ctx, nullptr), "dc_set_uniform_force");
  if (fallOff) {
    VecXd fv(3 * (size_t) N, 0.0);
    for (size_t k = 0; k < fv.size(); k++) fv[k] = wind[k % 3] * windNorm * rec.windFactor * windFallOff[k];
    check(ctx, dc_set_vertex_forces(ctx, fv.data()), "dc_set_vertex_forces");
  } else check(ctx, dc_set_vertex_forces(ctx, nullptr), "dc_set_vertex_forces");
  uploadForceField(ctx, field);
  rec.x_fixedpoints = fixedPointTargets(rec.t);
  rec.simDurartionFraction = rec.t / (sceneConfig.timeStep * sceneConfig.stepNum);
  rec.splines = controlPointSplines;
  dc_step_stats st;
  check(ctx, dc_step_forward(ctx, prev.deviceSlot, rec.x_fixedpoints.empty() ? nullptr : rec.x_fixedpoints.data(), &st), "dc_step_forward");
  rec.x.resize(3 * (size_t) N); rec.v.resize(3 * (size_t) N); rec.f.resize(3 * (size_t) N); rec.r.resize(3 * (size_t) N);
  check(ctx, dc_get_state(ctx, rec.deviceSlot, rec.x.data(), rec.v.data()), "dc_get_state");
  check(ctx, dc_get_record(ctx, rec.deviceSlot, rec.f.data(), rec.r.data()), "dc_get_record");
  std::vector<int> grp(N);
  VecXd nrm(3 * (size_t) N);
  check(ctx, dc_get_contacts(ctx, rec.deviceSlot, grp.data(), nrm.data()), "dc_get_contacts");
  for (int i = 0; i < N; i++)
    if (grp[i] >= 0) rec.primitiveCollisions.push_back({grp[i], i, {nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]}});
  if (st.self_contacts > 0) {   // layered self contacts of this step (collisionInfos.second)
    const int cap = st.self_contacts;
    std::vector<int> pairs(2 * (size_t) cap), layer(cap);
    VecXd sn(3 * (size_t) cap);
    int cnt = 0, nl = 0;
    check(ctx, dc_get_self_contacts(ctx, rec.deviceSlot, 0, cap, &cnt, &nl, pairs.data(), layer.data(), sn.data()), "dc_get_self_contacts");
    rec.selfCollisionLayers.assign(std::max(nl, 1), {});
    for (int k = 0; k < std::min(cnt, cap); k++)
      rec.selfCollisionLayers[layer[k]].push_back({pairs[2 * k], pairs[2 * k + 1], layer[k], {sn[3 * k], sn[3 * k + 1], sn[3 * k + 2]}});
  }
  rec.converged = st.converged != 0;
  rec.convergeIter = st.pd_iters;
  rec.totalConverged = prev.totalConverged + (rec.converged ? 1 : 0);
  rec.cumulateIter = prev.cumulateIter + st.pd_iters;
  rec.s_n.assign(3 * (size_t) N, 0.0);
  rec.totalRuntime = prev.totalRuntime + std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tStart).count();
  forwardRecords.push_back(std::move(rec));
}

// Simulation::stepNN (Simulation.cpp:1020-1042)
void Simulation::stepNN(int idx, const VecXd &x, const VecXd &v, const VecXd &fixedPointPos) {
  sceneConfig.trajectory = PER_STEP_TRAJECTORY;
  if (x.size() != 3 * (size_t) N || v.size() != 3 * (size_t) N) throw std::runtime_error("stepNN: x / v must have 3 * num_particles entries");
  if (fixedPointPos.size() != 3 * attachmentVertices.size())
    std::fprintf(stderr, "WARNING: require %zu fixed point dofs but input fixed point dof is %zu\n", 3 * attachmentVertices.size(), fixedPointPos.size());
  ForwardInformation &cur = forwardRecords.back();
  cur.x = x; cur.v = v;
  check(ctx, dc_set_state(ctx, cur.deviceSlot, x.data(), v.data()), "dc_set_state");
  rlFixedPointPos = fixedPointPos;
  step();
  forwardRecords.back().stepIdx = idx;
}

// Simulation::stepBackwardNN (Simulation.cpp:1443-1452)
BackwardInformation Simulation::stepBackwardNN(BackwardTaskInformation &taskInfo, VecXd &dL_dxnew, VecXd &dL_dvnew,
                                               const ForwardInformation &forwardInfo_new, bool isStart, const VecXd &dL_dxinit,
                                               const VecXd &dL_dvinit) {
  BackwardInformation g;
  g.dL_dx = dL_dxnew; g.dL_dv = dL_dvnew;
  return stepBackward(taskInfo, g, forwardInfo_new, isStart, dL_dxinit, dL_dvinit);
}

// Simulation::stepBackward (Simulation.cpp:1455-1780)
bool Simulation::needsForceVector(const BackwardTaskInformation &taskInfo) const {
  // dL_dfext_vec = h^2 (I + dr_df)^T u* per vertex (:1700-1760): its plain sum comes with the parameter gradients; the
  // per-vertex vector is needed only when a fall-off weighting, the force field or the per-step factors use it
  return taskInfo.dL_dconstantForceField || taskInfo.dL_dwindFactor ||
         ((taskInfo.dL_dfext || taskInfo.dL_dfwind) && sceneConfig.windConfig == WIND_SIN_AND_FALLOFF);
}

BackwardInformation Simulation::stepBackward(BackwardTaskInformation &taskInfo, BackwardInformation &gradient_new,
                                             const ForwardInformation &fwd, bool isStart, const VecXd &dL_dxinit, const VecXd &dL_dvinit) {
  // the record's own system matrix (SystemMatrix &currentSysMat = sysMat[forwardInfo_new.sysMatId], Simulation.cpp:1482): its context
  // holds the tape slot of this step; the set active for stepping is restored afterwards
  struct SetGuard {
    Simulation *s; int keep;
    SetGuard(Simulation *s_, int want) : s(s_), keep(s_->currentSysmatId) { s->activateSet(want); }
    ~SetGuard() { s->activateSet(keep); }
  } guard(this, fwd.sysMatId);
  const size_t n3 = 3 * (size_t) N, Af = attachmentVertices.size();
  if (gradient_new.dL_dx.size() != n3 || gradient_new.dL_dv.size() != n3) throw std::runtime_error("stepBackward: gradient size mismatch");
  if (fwd.deviceSlot < 1 || fwd.deviceSlot >= (int) forwardRecords.size() + 1) throw std::runtime_error("stepBackward: record has no device slot");
  const auto tStart = std::chrono::steady_clock::now();
  pushParams();
  VecXd dx(n3), dv(n3);
  DeviceBackwardStep d;
  d.dxf.assign(3 * std::max<size_t>(Af, 1), 0.0);
  d.dmu.assign(std::max<size_t>(primitives.size(), 1), 0.0);
  const bool haveInit = dL_dxinit.size() == n3 && dL_dvinit.size() == n3;
  dc_bwd_stats st;
  check(ctx, dc_step_backward(ctx, fwd.deviceSlot, gradient_new.dL_dx.data(), gradient_new.dL_dv.data(),
                              haveInit ? dL_dxinit.data() : nullptr, haveInit ? dL_dvinit.data() : nullptr, isStart ? 1 : 0,
                              dx.data(), dv.data(), d.dxf.data(), d.dmu.data(), &st), "dc_step_backward");
  d.converged = st.converged; d.iters = st.adjoint_iters;
  // parameter gradients of this step (Simulation.cpp:1672-1764): the device returns this step's contributions
  check(ctx, dc_get_param_gradients(ctx, fwd.deviceSlot, d.par), "dc_get_param_gradients");
  if (needsForceVector(taskInfo)) {
    d.fvec.resize(n3);
    check(ctx, dc_get_force_gradient(ctx, d.fvec.data()), "dc_get_force_gradient");
  }
  BackwardInformation ret = accumulateBackward(taskInfo, gradient_new, fwd, d);
  ret.dL_dx = std::move(dx); ret.dL_dv = std::move(dv);
  ret.totalRuntime = gradient_new.totalRuntime + std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tStart).count();
  return ret;
}

BackwardInformation Simulation::accumulateBackward(BackwardTaskInformation &taskInfo, const BackwardInformation &gradient_new,
                                                   const ForwardInformation &fwd, const DeviceBackwardStep &d) {
  const size_t n3 = 3 * (size_t) N, Af = attachmentVertices.size();
  const VecXd &dxf = d.dxf, &dmu = d.dmu, &fvec = d.fvec;
  const double *par = d.par;
  BackwardInformation ret;
  ret.converged = d.converged != 0;
  ret.backwardIters = d.iters;
  ret.backwardTotalIters = d.iters + gradient_new.backwardTotalIters;
  ret.convergedAccum = gradient_new.convergedAccum + (d.converged == 1 ? 1 : 0);
  ret.loss = gradient_new.loss;
  ret.dL_dxfixed.assign(3 * Af, 0.0);
  ret.dL_dxfixed_accum.assign(3 * Af, 0.0);
  if (taskInfo.dL_dcontrolPoints && Af > 0) {   // Simulation.cpp:1642-1660
    ret.dL_dxfixed.assign(dxf.begin(), dxf.begin() + 3 * Af);
    perStepGradient.push_back(ret.dL_dxfixed);
    if (fwd.stepIdx == 1) std::reverse(perStepGradient.begin(), perStepGradient.end());
    if (gradient_new.dL_dxfixed_accum.size() == 3 * Af)
      for (size_t k = 0; k < 3 * Af; k++) ret.dL_dxfixed_accum[k] = ret.dL_dxfixed[k] + gradient_new.dL_dxfixed_accum[k];
    // spline parameters (:1658-1669): dL_dspline += (dx_fixed/dparams)^T dL_dx_fixed of the driven point
    ret.dL_dsplines = gradient_new.dL_dsplines;
    const size_t sid = (size_t) std::max(fwd.sysMatId, 0);      // ret.dL_dsplines[sysMatId][splineIdx] (:1668)
    if (ret.dL_dsplines.size() <= sid) ret.dL_dsplines.resize(std::max<size_t>(sid + 1, attachmentSets.size()));
    if (ret.dL_dsplines[sid].size() != controlPointSplines.size()) {
      ret.dL_dsplines[sid].clear();
      for (const Spline &sp : controlPointSplines) ret.dL_dsplines[sid].push_back(VecXd(sp.getParameterNumber(), 0.0));
    }
    for (size_t k = 0; k < controlPointSplines.size(); k++) {
      const Spline &sp = controlPointSplines[k];
      if (sp.pFixed < 0 || (size_t) sp.pFixed >= Af) continue;
      const int np = sp.getParameterNumber();
      std::vector<double> J = sp.dxfixed_dcontrolPoints(fwd.simDurartionFraction);
      for (int q = 0; q < np; q++)
        for (int dd = 0; dd < 3; dd++) ret.dL_dsplines[sid][k][q] += J[(size_t) dd * np + q] * dxf[3 * (size_t) sp.pFixed + dd];
    }
  }
  ret.dL_ddensity = gradient_new.dL_ddensity;
  if (taskInfo.dL_density) ret.dL_ddensity += par[3];
  ret.dL_dk_pertype = gradient_new.dL_dk_pertype;      // Constraint::ConstraintType order: spring, attachment, triangle, bending
  const double perType[4] = {0.0, par[2], par[0], par[1]};
  for (int k = 0; k < 4; k++)
    if (taskInfo.dL_dk_pertype[k]) ret.dL_dk_pertype[k] = gradient_new.dL_dk_pertype[k] + perType[k];
  const bool needVec = needsForceVector(taskInfo) && fvec.size() == n3;
  const bool haveFall = windFallOff.size() == n3;
  double total[3] = {par[4], par[5], par[6]};       // sum_i dL_dfext_vec_i, fall-off weighted for WIND_SIN_AND_FALLOFF
  if (needVec && sceneConfig.windConfig == WIND_SIN_AND_FALLOFF && haveFall) {
    total[0] = total[1] = total[2] = 0;
    for (size_t k = 0; k < n3; k++) total[k % 3] += fvec[k] * windFallOff[k];
  }
  ret.dL_dfext = gradient_new.dL_dfext;
  if (taskInfo.dL_dfext)                        // :1700-1712
    for (int dd = 0; dd < 3; dd++) ret.dL_dfext[dd] += total[dd] * fwd.windFactor;
  ret.dL_dconstantForceField = gradient_new.dL_dconstantForceField;
  if (taskInfo.dL_dconstantForceField && needVec) {        // :1714-1718
    if (ret.dL_dconstantForceField.size() != n3) ret.dL_dconstantForceField.assign(n3, 0.0);
    for (size_t k = 0; k < n3; k++) ret.dL_dconstantForceField[k] += fvec[k];
  }
  ret.dL_dwindtimestep = gradient_new.dL_dwindtimestep;
  if (taskInfo.dL_dwindFactor && needVec) {                // :1720-1729
    if (ret.dL_dwindtimestep.size() <= (size_t) fwd.stepIdx) ret.dL_dwindtimestep.resize((size_t) fwd.stepIdx + 1, 0.0);
    double acc = 0;
    for (size_t k = 0; k < n3; k++) acc += fvec[k] * wind[k % 3] * windNorm * (haveFall ? windFallOff[k] : 1.0);
    ret.dL_dwindtimestep[fwd.stepIdx] = acc;
  }
  ret.dL_dwind = gradient_new.dL_dwind;
  if (taskInfo.dL_dfwind) {                     // :1731-1760 (sin wind model, with or without fall-off)
    const double c = std::cos(windFrequency * fwd.t + windPhase);
    double tf = 0;
    for (int dd = 0; dd < 3; dd++) tf += total[dd] * wind[dd] * windNorm;
    for (int dd = 0; dd < 3; dd++) ret.dL_dwind[dd] += total[dd] * fwd.windFactor;
    ret.dL_dwind[3] += tf * c * 0.5 * fwd.t;
    ret.dL_dwind[4] += tf * c * 0.5;
  }
  if (taskInfo.dL_dmu)                          // Simulation.cpp:1622-1632
    for (size_t k = 0; k < taskInfo.mu_primitives.size(); k++) {
      int prim = taskInfo.mu_primitives[k];
      double prev = k < gradient_new.dL_dmu.size() ? gradient_new.dL_dmu[k].second : 0.0;
      ret.dL_dmu.push_back({prim, prev + (prim >= 0 && prim < (int) dmu.size() ? dmu[prim] : 0.0)});
    }
  return ret;
}

// ---------------------------------------------------------------------------------------------------------------
// device-resident evaluation: the loops of runBackwardTask (Simulation.cpp:3853-3961) as two launches
// ---------------------------------------------------------------------------------------------------------------
bool Simulation::rolloutOnDevice(int nsteps) {
  static const bool envOff = std::getenv("DIFFCLOTH_DEVICE_ROLLOUTS") && std::getenv("DIFFCLOTH_DEVICE_ROLLOUTS")[0] == '0';
  if (!deviceResidentRollouts || envOff || nsteps < 1 || forwardRecords.empty()) return false;
  if (sceneConfig.trajectory == PER_STEP_TRAJECTORY) return false;          // the targets of a step arrive with the step (RL action)
  if (attachmentSets.size() > 1) return false;                               // the system matrix changes inside the rollout: per-step path
  if ((int) forwardRecords.size() + nsteps > tapeSlots) throw std::runtime_error("Simulation::rolloutOnDevice: tape exhausted (stepNum + 8 records)");
  const size_t n3 = 3 * (size_t) N, Af = attachmentVertices.size();
  const bool fallOff = windEnabled && windHasFallOff() && windFallOff.size() == n3;
  const bool field = enableConstantForcefield && external_force_field.size() == n3;
  const auto tStart = std::chrono::steady_clock::now();
  pushParams();
  const int slot0 = forwardRecords.back().deviceSlot;
  const size_t first = forwardRecords.size();
  // ---- the per-step inputs of step(): wind factor, fillForces terms, stepFixPoints targets (in order: the twirl is incremental) ----
  std::vector<double> fu((size_t) nsteps * 3, 0.0), fvs(nsteps, 1.0), xf((size_t) nsteps * 3 * std::max<size_t>(Af, 1), 0.0);
  for (int k = 0; k < nsteps; k++) {
    const ForwardInformation &prev = forwardRecords.back();
    ForwardInformation rec;
    rec.t = prev.t + sceneConfig.timeStep;
    rec.stepIdx = (int) forwardRecords.size();
    rec.deviceSlot = prev.deviceSlot + 1;
    rec.windFactor = windFactorAt(rec.t, rec.stepIdx);
    if (windEnabled && !fallOff) for (int d = 0; d < 3; d++) fu[3 * (size_t) k + d] = wind[d] * windNorm * rec.windFactor;
    if (fallOff) fvs[k] = rec.windFactor;
    rec.x_fixedpoints = fixedPointTargets(rec.t);
    for (size_t q = 0; q < 3 * Af; q++) xf[(size_t) k * 3 * Af + q] = rec.x_fixedpoints[q];
    rec.simDurartionFraction = rec.t / (sceneConfig.timeStep * sceneConfig.stepNum);
    rec.splines = controlPointSplines;
    forwardRecords.push_back(std::move(rec));
  }
  // from here on a failing device call (capacity overflow, exchange time-out ...) must not leave half-made records behind
  try {
  check(ctx, dc_clear_schedules(ctx), "dc_clear_schedules");
  // the two per-vertex terms of fillForces (Simulation.cpp:91-106): the wind with fall-off, whose factor changes from step to step (the
  // schedule's fv_scale), and the constant force field, factor 1 in every step (dc_set_vertex_force_field)
  if (fallOff) {
    VecXd fv(n3, 0.0);
    for (size_t k = 0; k < n3; k++) fv[k] = wind[k % 3] * windNorm * windFallOff[k];
    check(ctx, dc_set_vertex_forces(ctx, fv.data()), "dc_set_vertex_forces");
  } else check(ctx, dc_set_vertex_forces(ctx, nullptr), "dc_set_vertex_forces");
  uploadForceField(ctx, field);
  check(ctx, dc_set_uniform_force(ctx, nullptr), "dc_set_uniform_force");
  check(ctx, dc_set_force_schedule(ctx, slot0, nsteps, (windEnabled && !fallOff) ? fu.data() : nullptr, fallOff ? fvs.data() : nullptr), "dc_set_force_schedule");
  if (Af > 0) check(ctx, dc_set_fixed_point_schedule(ctx, slot0, nsteps, xf.data()), "dc_set_fixed_point_schedule");
  check(ctx, dc_rollout_forward(ctx, slot0, nsteps), "dc_rollout_forward");
  // ---- records: states in one download, solver statistics per slot ----
  VecXd X((size_t) nsteps * n3), V((size_t) nsteps * n3);
  check(ctx, dc_get_states(ctx, slot0 + 1, nsteps, X.data(), V.data()), "dc_get_states");
  const long long us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tStart).count();
  for (int k = 0; k < nsteps; k++) {
    ForwardInformation &rec = forwardRecords[first + k];
    const ForwardInformation &prev = forwardRecords[first + k - 1];
    rec.x.assign(X.begin() + (size_t) k * n3, X.begin() + (size_t) (k + 1) * n3);
    rec.v.assign(V.begin() + (size_t) k * n3, V.begin() + (size_t) (k + 1) * n3);
    rec.x_prev = prev.x; rec.v_prev = prev.v;
    dc_step_stats st;
    check(ctx, dc_get_stats(ctx, rec.deviceSlot, &st, nullptr), "dc_get_stats");
    rec.converged = st.converged != 0;
    rec.convergeIter = st.pd_iters;
    rec.totalConverged = prev.totalConverged + (rec.converged ? 1 : 0);
    rec.cumulateIter = prev.cumulateIter + st.pd_iters;
    rec.totalRuntime = prev.totalRuntime + us / nsteps;
  }
  } catch (...) {
    forwardRecords.resize(first);
    throw;
  }
  return true;
}

void Simulation::loadRecordDetails(int recordIdx) {
  if (recordIdx < 1 || recordIdx >= (int) forwardRecords.size()) throw std::runtime_error("loadRecordDetails: no such record");
  ForwardInformation &rec = forwardRecords[recordIdx];
  const size_t n3 = 3 * (size_t) N;
  rec.f.resize(n3); rec.r.resize(n3);
  check(ctx, dc_get_record(ctx, rec.deviceSlot, rec.f.data(), rec.r.data()), "dc_get_record");
  std::vector<int> grp(N);
  VecXd nrm(n3);
  check(ctx, dc_get_contacts(ctx, rec.deviceSlot, grp.data(), nrm.data()), "dc_get_contacts");
  rec.primitiveCollisions.clear();
  for (int i = 0; i < N; i++)
    if (grp[i] >= 0) rec.primitiveCollisions.push_back({grp[i], i, {nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]}});
  dc_step_stats st;
  check(ctx, dc_get_stats(ctx, rec.deviceSlot, &st, nullptr), "dc_get_stats");
  rec.selfCollisionLayers.clear();
  if (st.self_contacts > 0) {
    const int cap = st.self_contacts;
    std::vector<int> pairs(2 * (size_t) cap), layer(cap);
    VecXd sn(3 * (size_t) cap);
    int cnt = 0, nl = 0;
    check(ctx, dc_get_self_contacts(ctx, rec.deviceSlot, 0, cap, &cnt, &nl, pairs.data(), layer.data(), sn.data()), "dc_get_self_contacts");
    rec.selfCollisionLayers.assign(std::max(nl, 1), {});
    for (int k = 0; k < std::min(cnt, cap); k++)
      rec.selfCollisionLayers[layer[k]].push_back({pairs[2 * k], pairs[2 * k + 1], layer[k], {sn[3 * k], sn[3 * k + 1], sn[3 * k + 2]}});
  }
}

std::vector<BackwardInformation> Simulation::sweepBackwardOnDevice(BackwardTaskInformation &taskInfo,
                                                                   const std::vector<std::pair<VecXd, VecXd>> &seeds, double loss) {
  static const bool envOff = std::getenv("DIFFCLOTH_DEVICE_ROLLOUTS") && std::getenv("DIFFCLOTH_DEVICE_ROLLOUTS")[0] == '0';
  const int frames = (int) forwardRecords.size();
  if (!deviceResidentRollouts || envOff || frames < 2 || (int) seeds.size() != frames) return {};
  const bool needVec = needsForceVector(taskInfo);  // the per-vertex force gradient of every step: kept per tape slot by the sweep (dc_keep_force_gradients)
  if (forwardRecords[0].deviceSlot != 0) return {};      // isStart of the sweep is tied to tape slot 1
  if (attachmentSets.size() > 1) return {};              // several system matrices: the per-step loop picks the context of each record
  for (int i = 1; i < frames; i++) if (forwardRecords[i].deviceSlot != forwardRecords[i - 1].deviceSlot + 1) return {};
  const size_t n3 = 3 * (size_t) N, Af = attachmentVertices.size();
  const auto tStart = std::chrono::steady_clock::now();
  pushParams();
  const int slot0 = forwardRecords[0].deviceSlot, last = forwardRecords.back().deviceSlot, nsteps = frames - 1;
  VecXd SX((size_t) nsteps * n3), SV((size_t) nsteps * n3);
  for (int i = 0; i < nsteps; i++) {
    if (seeds[i].first.size() != n3 || seeds[i].second.size() != n3) return {};
    std::copy(seeds[i].first.begin(), seeds[i].first.end(), SX.begin() + (size_t) i * n3);
    std::copy(seeds[i].second.begin(), seeds[i].second.end(), SV.begin() + (size_t) i * n3);
  }
  check(ctx, dc_set_seed_schedule(ctx, slot0, nsteps, SX.data(), SV.data()), "dc_set_seed_schedule");
  check(ctx, dc_set_gradient(ctx, seeds[frames - 1].first.data(), seeds[frames - 1].second.data()), "dc_set_gradient");
  check(ctx, dc_keep_force_gradients(ctx, needVec ? 1 : 0), "dc_keep_force_gradients");
  check(ctx, dc_rollout_backward(ctx, last, nsteps), "dc_rollout_backward");
  VecXd FV;
  if (needVec) {
    FV.assign((size_t) nsteps * n3, 0.0);
    check(ctx, dc_get_force_gradients(ctx, slot0 + 1, nsteps, FV.data()), "dc_get_force_gradients");
    check(ctx, dc_keep_force_gradients(ctx, 0), "dc_keep_force_gradients");
  }
  VecXd dx(n3), dv(n3), dmuTotal(std::max<size_t>(primitives.size(), 1), 0.0);
  check(ctx, dc_get_gradient(ctx, dx.data(), dv.data(), dmuTotal.data()), "dc_get_gradient");
  VecXd DXF((size_t) nsteps * 3 * std::max<size_t>(Af, 1), 0.0);
  if (Af > 0) check(ctx, dc_get_dxfixed(ctx, slot0 + 1, nsteps, DXF.data()), "dc_get_dxfixed");
  const long long us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tStart).count();
  BackwardInformation derivative;
  derivative.dL_dx = seeds[frames - 1].first; derivative.dL_dv = seeds[frames - 1].second; derivative.loss = loss;
  std::vector<BackwardInformation> all = {derivative};
  for (int idx = frames - 1; idx >= 1; idx--) {
    const ForwardInformation &fwd = forwardRecords[idx];
    DeviceBackwardStep d;
    d.dxf.assign(3 * std::max<size_t>(Af, 1), 0.0);
    for (size_t q = 0; q < 3 * Af; q++) d.dxf[q] = DXF[(size_t) (idx - 1) * 3 * Af + q];
    d.dmu.assign(dmuTotal.size(), 0.0);
    if (idx == 1) d.dmu = dmuTotal;               // the device accumulates dL_dmu over the sweep: the total enters at the last step
    if (needVec) d.fvec.assign(FV.begin() + (size_t) (idx - 1) * n3, FV.begin() + (size_t) idx * n3);
    dc_bwd_stats st;
    check(ctx, dc_get_stats(ctx, fwd.deviceSlot, nullptr, &st), "dc_get_stats");
    d.converged = st.converged; d.iters = st.adjoint_iters;
    check(ctx, dc_get_param_gradients(ctx, fwd.deviceSlot, d.par), "dc_get_param_gradients");
    BackwardInformation next = accumulateBackward(taskInfo, derivative, fwd, d);
    next.totalRuntime = derivative.totalRuntime + us / nsteps;
    if (idx == 1) { next.dL_dx = dx; next.dL_dv = dv; }
    derivative = next;
    all.push_back(std::move(next));
  }
  return all;
}

}  // namespace dchost
