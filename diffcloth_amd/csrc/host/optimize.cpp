// Rollout-level driver of the reference's optimisation demos on top of the device stepper: loss functions and their
// seeds (Simulation::calculateLossAndGradient, reference Simulation.cpp:3237-3488), parameter injection
// (resetSystemWithParams, :3490-3584), the forward rollout + backward sweep (runBackwardTask, :3853-3961) and the
// OptimizeHelper that maps a flat parameter vector to those (optimization/OptimizeHelper.cpp,
// optimization/OptimizationTaskSetup.cpp). Host code only: every time step goes through the dc_* C-ABI — step by step
// (Simulation::step() / stepBackward()) or, where the scene allows it, as one fused launch per direction with the per-step inputs
// uploaded as schedules (Simulation::rolloutOnDevice / sweepBackwardOnDevice).
#include "optimize.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

namespace dchost {

namespace {
double sqnorm_diff(const VecXd &a, const VecXd &b) {
  double s = 0;
  for (size_t k = 0; k < a.size() && k < b.size(); k++) s += (a[k] - b[k]) * (a[k] - b[k]);
  return s;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// losses
// ---------------------------------------------------------------------------------------------------------------
double Simulation::calculateLossAndGradient(LossType lossType, LossInfo &lossInfo, VecXd &dL_dx, VecXd &dL_dv, int idx, bool calculateLoss) {
  const size_t n3 = 3 * (size_t) N;
  const int lastIdx = (int) forwardRecords.size() - 1, frames = (int) forwardRecords.size();
  dL_dx.assign(n3, 0.0); dL_dv.assign(n3, 0.0);
  double L = 0;
  switch (lossType) {
    case MATCH_TRAJECTORY:
    case MATCH_VELOCITY: {
      const bool vel = lossType == MATCH_VELOCITY;
      if ((int) lossInfo.targetSimulation.size() != frames)
        std::fprintf(stderr, "WARNING: calculate trajectory loss frame number mismatch, but records has size %d while target has size %zu\n",
                     frames, lossInfo.targetSimulation.size());
      const double k = 1.0 / ((double) frames * N);
      auto rec = [&](int i) -> const VecXd & { return vel ? forwardRecords[i].v : forwardRecords[i].x; };
      auto tgt = [&](int i) -> const VecXd & { const auto &t = lossInfo.targetSimulation.at(i); return vel ? t.second : t.first; };
      if (calculateLoss)
        for (int i = 0; i < frames && i < (int) lossInfo.targetSimulation.size(); i++) L += k * sqnorm_diff(rec(i), tgt(i));
      const int at = calculateLoss ? lastIdx : idx;
      VecXd &g = vel ? dL_dv : dL_dx;
      for (size_t q = 0; q < n3; q++) g[q] = k * 2 * (rec(at)[q] - tgt(at)[q]);
      break;
    }
    case MATCH_TRAJECTORY_MAX: {
      const double k = 1.0 / N;
      int maxFrame = 0;
      for (int i = 0; i < frames; i++) {
        const int lossFrame = (int) (k * sqnorm_diff(forwardRecords[i].x, lossInfo.targetSimulation.at(i).first));   // (sic) integer, as the reference
        if (lossFrame > L) { L = lossFrame; maxFrame = i; }
      }
      if (idx == maxFrame)
        for (size_t q = 0; q < n3; q++) dL_dx[q] = k * 2 * (forwardRecords[idx].x[q] - lossInfo.targetSimulation.at(idx).first[q]);
      break;
    }
    case DRESS_ANGLE: {
      const double targetHeight = restShapeMinDim[1] + (restShapeMaxDim[1] - restShapeMinDim[1]) * lossInfo.targetTwirlHeight;
      const double k = lossInfo.loopPoints.empty() ? 0.0 : 1.0 / lossInfo.loopPoints.size();
      for (int p : lossInfo.loopPoints) {
        const double y = forwardRecords[lastIdx].x[3 * (size_t) p + 1];
        L += (y - targetHeight) * (y - targetHeight) * k;
        if (idx == lastIdx) dL_dx[3 * (size_t) p + 1] += 2 * (y - targetHeight) * k;
      }
      break;
    }
    case MATCHSHAPE_TRANSLATION_INVARINT: {
      VecXd target = lossInfo.targetShape, cur = forwardRecords[lastIdx].x;
      if (target.size() != n3) throw std::runtime_error("MATCHSHAPE_TRANSLATION_INVARINT: targetShape has the wrong size");
      const Vec3d t0 = {target[0], target[1], target[2]}, c0 = {cur[0], cur[1], cur[2]};
      for (int i = 0; i < N; i++) for (int d = 0; d < 3; d++) { target[3 * i + d] -= t0[d]; cur[3 * i + d] -= c0[d]; }
      L += sqnorm_diff(cur, target) / N;
      if (idx == lastIdx) {
        for (size_t q = 0; q < n3; q++) dL_dx[q] = 2 * (cur[q] - target[q]);
        dL_dx[0] = dL_dx[1] = dL_dx[2] = 0;
        for (int i = 1; i < N; i++) for (int d = 0; d < 3; d++) dL_dx[d] += dL_dx[3 * i + d];   // (sic) the reference adds, see :3388-3391
        for (size_t q = 0; q < n3; q++) dL_dx[q] /= N;
      }
      break;
    }
    case MULTISTEP_MATCHSHAPE: {
      for (auto &fs : lossInfo.targetFrameShape) {
        const VecXd &cur = forwardRecords.at(fs.first).x;
        if (calculateLoss) L += sqnorm_diff(cur, fs.second) / N;
        if (idx == fs.first) for (size_t q = 0; q < n3; q++) dL_dx[q] = 2 * (cur[q] - fs.second[q]) / N;
      }
      break;
    }
    case MATCHSHAPE_WITH_TRANSLATION: {
      const VecXd &cur = forwardRecords[lastIdx].x;
      VecXd target(n3);
      for (int i = 0; i < N; i++) for (int d = 0; d < 3; d++) target[3 * i + d] = rest[3 * i + d] + lossInfo.targetTranslation[d];   // pos_init + translation
      if (calculateLoss) L += sqnorm_diff(cur, target) / N;
      if (idx == lastIdx) for (size_t q = 0; q < n3; q++) dL_dx[q] = 2 * (cur[q] - target[q]) / N;
      break;
    }
    case ASSISTED_DRESSING_KEYPOINTS: {
      const double total = (double) lossInfo.targetPosPairs.size();
      for (const CorresPondenceTargetInfo &pr : lossInfo.targetPosPairs) {
        const VecXd &cur = forwardRecords.at(pr.frameIdx).x;
        int far = pr.particleIndices.at(0);
        auto dist2 = [&](int p) { double s = 0; for (int d = 0; d < 3; d++) s += (cur[3 * (size_t) p + d] - pr.targetPos[d]) * (cur[3 * (size_t) p + d] - pr.targetPos[d]); return s; };
        double maxDist = dist2(far);
        for (int p : pr.particleIndices) { const double dd = dist2(p); if (dd > maxDist) { maxDist = dd; far = p; } }
        if (calculateLoss) L += maxDist / total;
        if (pr.frameIdx == idx) for (int d = 0; d < 3; d++) dL_dx[3 * (size_t) far + d] = 2 * (cur[3 * (size_t) far + d] - pr.targetPos[d]) / total;
      }
      break;
    }
  }
  if (calculateLoss && forwardRecords.size() >= 2)
    for (size_t q = 0; q < n3; q++) dL_dx[q] += dL_dv[q] / sceneConfig.timeStep;
  return L;
}

// ---------------------------------------------------------------------------------------------------------------
// parameters -> system
// ---------------------------------------------------------------------------------------------------------------
void Simulation::resetSystemWithParams(BackwardTaskInformation &task, ParamInfo &param) {
  bool systemChanged = false;
  // Constraint::ConstraintType order: spring, attachment, triangle, bending
  if (task.dL_dk_pertype[1]) { k_stiff_attachment = param.k_pertype[1]; systemChanged = true; }
  if (task.dL_dk_pertype[2]) { sceneConfig.fabric.k_stiff_stretching = param.k_pertype[2]; systemChanged = true; }
  if (task.dL_dk_pertype[3]) { sceneConfig.fabric.k_stiff_bending = param.k_pertype[3]; systemChanged = true; }
  auto norm3 = [](const double *v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
  if (task.dL_dfext && param.f_ext.size() == 3) {
    const double n = norm3(param.f_ext.data());
    windNorm = n;
    for (int d = 0; d < 3; d++) wind[d] = n > 0 ? param.f_ext[d] / n : 0.0;
  }
  if (task.dL_dfwind) {
    const double n = norm3(param.f_extwind.data());
    windNorm = n;
    for (int d = 0; d < 3; d++) wind[d] = n > 0 ? param.f_extwind[d] / n : 0.0;
    windFrequency = param.f_extwind[3]; windPhase = param.f_extwind[4];
  }
  if (task.dL_dconstantForceField) external_force_field = param.f_constantForceField;   // Simulation.cpp:3524-3526 (enableConstantForcefield stays the caller's switch)
  if (task.dL_dwindFactor) perstepWindFactor = param.f_ext_timestep;                    // :3528-3530
  if (task.dL_dcontrolPoints && !param.controlPointSplines.empty()) controlPointSplines = param.controlPointSplines[0];
  if (task.dL_dmu)
    for (const auto &pm : param.mu) primitives.at(pm.first).mu = pm.second;
  if (task.dL_density) { sceneConfig.fabric.density = param.density; systemChanged = true; }
  if (systemChanged) rebuildSystem();
  if (task.dL_dmu && !primitives.empty()) {
    VecXd mu(primitives.size());
    for (size_t k = 0; k < primitives.size(); k++) mu[k] = primitives[k].mu;
    forEachContext([&](dc_ctx *c) { if (dc_set_mu(c, mu.data()) != DC_OK) throw std::runtime_error(std::string("dc_set_mu: ") + dc_last_error(c)); });
  }
  resetSystem();
  if (task.dL_dx0 && param.x0.size() == 3 * (size_t) N) {
    forwardRecords[0].x = param.x0;
    if (dc_set_state(ctx, 0, forwardRecords[0].x.data(), forwardRecords[0].v.data()) != DC_OK)
      throw std::runtime_error(std::string("dc_set_state: ") + dc_last_error(ctx));
  }
  perStepGradient.clear();
}

std::vector<BackwardInformation> Simulation::runBackwardTask(BackwardTaskInformation task, LossType lossType, LossInfo &lossInfo,
                                                             TaskSolveStatistics &stats, int FORWARD_STEPS, ParamInfo guess, bool lossOnly,
                                                             bool skipForward) {
  if (!skipForward) resetSystemWithParams(task, guess);
  forwardConvergenceThreshold = task.forwardAccuracyLevel;
  backwardConvergenceThreshold = task.backwardAccuracyLevel;
  if (!skipForward && !rolloutOnDevice(FORWARD_STEPS))        // all steps in one launch where the scene allows it (simulation.h)
    for (int i = 0; i < FORWARD_STEPS; i++) step();
  BackwardInformation first;
  if (forwardRecords.empty()) { first.loss = 0; return {first}; }
  const int frames = (int) forwardRecords.size();
  VecXd dL_dlastx, dL_dlastv;
  const double L = calculateLossAndGradient(lossType, lossInfo, dL_dlastx, dL_dlastv, frames - 1, true);
  forwardRecords.back().loss = L;
  stats.totalForwardSim++;
  stats.completeForwardLog.emplace_back(guess, forwardRecords.back());
  if (lossOnly) { first.loss = L; return {first}; }
  BackwardInformation derivative;
  derivative.dL_dx = dL_dlastx; derivative.dL_dv = dL_dlastv; derivative.loss = L;
  std::vector<BackwardInformation> all = {derivative};
  if (FORWARD_STEPS + 1 != frames)
    std::fprintf(stderr, "WARNING: invariant violated: FORWARD_STEPS:%d fowardRecords: %d\n", FORWARD_STEPS, frames);
  VecXd dL_dxinit, dL_dvinit;
  // the whole sweep in one launch with the per-frame loss gradients as a device schedule, or step by step
  std::vector<BackwardInformation> fused;
  if (deviceResidentRollouts && !needsForceVector(task)) {
    std::vector<std::pair<VecXd, VecXd>> seeds(frames);
    for (int i = 0; i + 1 < frames; i++) calculateLossAndGradient(lossType, lossInfo, seeds[i].first, seeds[i].second, i, false);
    seeds[frames - 1] = {dL_dlastx, dL_dlastv};
    fused = sweepBackwardOnDevice(task, seeds, L);
  }
  if (!fused.empty()) all = std::move(fused);
  else
    for (int idx = frames - 1; idx >= 1; idx--) {
      calculateLossAndGradient(lossType, lossInfo, dL_dxinit, dL_dvinit, idx - 1, false);
      derivative = stepBackward(task, derivative, forwardRecords[idx], (idx - 1) == 0, dL_dxinit, dL_dvinit);
      all.push_back(derivative);
    }
  std::reverse(all.begin(), all.end());
  all[0].correspondingForwardIdxInStats = stats.totalForwardSim - 1;
  stats.totalBackprop++;
  stats.completeBackwardLog.emplace_back(guess, all[0]);
  return all;
}

// ---------------------------------------------------------------------------------------------------------------
// OptimizeHelper
// ---------------------------------------------------------------------------------------------------------------
OptimizeHelper::OptimizeHelper(Simulation *system_, const LossInfo &lossInfo_, const BackwardTaskInformation &taskInfo_, LossType lossType_,
                               int steps, const ParamInfo &paramActual)
    : lossInfo(lossInfo_), taskInfo(taskInfo_), system(system_), FORWARD_STEPS(steps), lossType(lossType_), param_actual(paramActual) {
  // initial spline guess: every end point moved to start + (-1, 1, 0) for the first two curves, (1, 1, 0) for the others
  param_guess.controlPointSplines.push_back(system->controlPointSplines);
  for (size_t k = 0; k < param_guess.controlPointSplines[0].size(); k++) {
    Spline &s = param_guess.controlPointSplines[0][k];
    for (size_t seg = 0; seg < s.segments.size(); seg++) {
      const Vec3d p0 = s.segments[seg].p0;
      s.moveEndPoint((int) seg, {p0[0] + (k < 2 ? -1.0 : 1.0), p0[1] + 1.0, p0[2]});
    }
  }
  param_guess.mu.resize(taskInfo.mu_primitives.size());
  setParameterBounds();
}

void OptimizeHelper::setParameterBounds() {
  std::vector<std::pair<double, double>> bounds;
  auto add = [&](double lo, double hi, bool logScale, const char *name) { bounds.push_back({lo, hi}); paramLogScaleTransformOn.push_back(logScale); paramName.push_back(name); };
  totalParamNumber = 0; totalSplineParamNumber = 0;
  if (taskInfo.dL_dfwind) {
    offset.dL_dfwind = totalParamNumber; totalParamNumber += 5;
    for (int i = 0; i < 3; i++) add(-0.1, 0.1, false, "windForce");
    add(0.01, 15, false, "windFreq");
    add(-5, 5, false, "windPhase");
  }
  if (taskInfo.dL_dfext) {
    offset.dL_dfext = totalParamNumber; totalParamNumber += 3;
    for (int i = 0; i < 3; i++) add(-3, 3, false, "windDir");
  }
  static const std::pair<double, double> stiffnessBounds[4] = {{0, 200}, {63, 10000}, {80, 1500}, {1e-7, 5}};
  static const char *typeNames[4] = {"CONSTRAINT_SPRING", "CONSTRAINT_ATTACHMENT", "CONSTRAINT_TRIANGLE", "CONSTRAINT_TRIANGLE_BENDING"};
  if (taskInfo.dL_dx0) {
    offset.dL_dx0 = totalParamNumber; totalParamNumber += 3 * system->getNumParticles();
    for (int i = 0; i < 3 * system->getNumParticles(); i++) add(-1e30, 1e30, false, "x0");   // the reference bounds x0 by the scene box
  }
  if (taskInfo.dL_dconstantForceField) {     // OptimizeHelper.cpp:98-107
    offset.dL_dconstantForceField = totalParamNumber; totalParamNumber += 3 * system->getNumParticles();
    for (int i = 0; i < 3 * system->getNumParticles(); i++) add(-10, 10, false, "constantForceField");
  }
  for (int i = 0; i < 4; i++)
    if (taskInfo.dL_dk_pertype[i]) { offset.dL_k[i] = totalParamNumber; totalParamNumber += 1; add(stiffnessBounds[i].first, stiffnessBounds[i].second, false, typeNames[i]); }
  if (taskInfo.dL_density) { offset.dL_density = totalParamNumber; totalParamNumber += 1; add(0.01, 1.0, false, "density"); }
  if (taskInfo.dL_dcontrolPoints) {
    offset.dL_dspline = totalParamNumber;
    for (const Spline &s : system->controlPointSplines) {
      const int per = Spline::parametersPerSegment(s.type), np = s.getParameterNumber();
      totalParamNumber += np; totalSplineParamNumber += np;
      for (int q = 0; q < np; q++) {
        const int r = q % per;            // end point inside the scene box (not modelled here: wide box), tangents in [-50, 50]
        if (s.type == Spline::ENDPOINT_AND_TANGENTS && r >= 3) add(-50, 50, false, "spline");
        else add(-1e3, 1e3, false, "spline");
      }
    }
  }
  if (taskInfo.dL_dmu)
    for (size_t k = 0; k < taskInfo.mu_primitives.size(); k++) { offset.dL_dmu.push_back(totalParamNumber); totalParamNumber++; add(0.01, 0.95, false, "mu"); }
  paramLowerBound.assign(totalParamNumber, 0.0); paramUpperBound.assign(totalParamNumber, 0.0);
  for (int i = 0; i < totalParamNumber; i++) { paramLowerBound[i] = bounds[i].first; paramUpperBound[i] = bounds[i].second; }
}

VecXd OptimizeHelper::paramInfoToVecXd(const ParamInfo &param) const {
  VecXd x(totalParamNumber, 0.0);
  if (taskInfo.dL_dfwind) for (int i = 0; i < 5; i++) x[offset.dL_dfwind + i] = param.f_extwind[i];
  if (taskInfo.dL_dfext) for (int i = 0; i < 3 && i < (int) param.f_ext.size(); i++) x[offset.dL_dfext + i] = param.f_ext[i];
  if (taskInfo.dL_dx0) for (size_t i = 0; i < param.x0.size(); i++) x[offset.dL_dx0 + i] = param.x0[i];
  if (taskInfo.dL_dconstantForceField) for (size_t i = 0; i < param.f_constantForceField.size(); i++) x[offset.dL_dconstantForceField + i] = param.f_constantForceField[i];
  for (int i = 0; i < 4; i++) if (taskInfo.dL_dk_pertype[i]) x[offset.dL_k[i]] = param.k_pertype[i];
  if (taskInfo.dL_density) x[offset.dL_density] = param.density;
  if (taskInfo.dL_dmu) for (size_t i = 0; i < taskInfo.mu_primitives.size() && i < param.mu.size(); i++) x[offset.dL_dmu[i]] = param.mu[i].second;
  if (taskInfo.dL_dcontrolPoints && !param.controlPointSplines.empty()) {
    int at = offset.dL_dspline;
    for (const Spline &s : param.controlPointSplines[0]) { VecXd v = s.paramToVector(); for (double q : v) x[at++] = q; }
  }
  return x;
}

ParamInfo OptimizeHelper::vecXdToParamInfo(const VecXd &x) const {
  if ((int) x.size() != totalParamNumber) throw std::runtime_error("vecXdToParamInfo: expected " + std::to_string(totalParamNumber) + " parameters");
  ParamInfo param;
  if (taskInfo.dL_dfwind) for (int i = 0; i < 5; i++) param.f_extwind[i] = x[offset.dL_dfwind + i];
  if (taskInfo.dL_dfext) param.f_ext.assign(x.begin() + offset.dL_dfext, x.begin() + offset.dL_dfext + 3);
  if (taskInfo.dL_dx0) param.x0.assign(x.begin() + offset.dL_dx0, x.begin() + offset.dL_dx0 + 3 * system->getNumParticles());
  if (taskInfo.dL_dconstantForceField)
    param.f_constantForceField.assign(x.begin() + offset.dL_dconstantForceField, x.begin() + offset.dL_dconstantForceField + 3 * system->getNumParticles());
  for (int i = 0; i < 4; i++) if (taskInfo.dL_dk_pertype[i]) param.k_pertype[i] = x[offset.dL_k[i]];
  if (taskInfo.dL_density) param.density = x[offset.dL_density];
  if (taskInfo.dL_dmu) for (size_t i = 0; i < taskInfo.mu_primitives.size(); i++) param.mu.push_back({taskInfo.mu_primitives[i], x[offset.dL_dmu[i]]});
  if (taskInfo.dL_dcontrolPoints) {
    param.controlPointSplines.emplace_back();
    int at = offset.dL_dspline;
    for (const Spline &guess : param_guess.controlPointSplines[0]) {
      // Spline::splineFromParam: absolute parameters replace the guess's (tangents re-derived for the end-point types)
      Spline s = guess;
      const VecXd cur = s.paramToVector();
      VecXd step(cur.size());
      for (size_t q = 0; q < cur.size(); q++) step[q] = x[at + q] - cur[q];
      s.updateControlPoints(step);
      at += (int) cur.size();
      param.controlPointSplines[0].push_back(s);
    }
  }
  return param;
}

VecXd OptimizeHelper::gradientInfoToVecXd(const BackwardInformation &b) const {
  VecXd g(totalParamNumber, 0.0);
  if (taskInfo.dL_dfwind) for (int i = 0; i < 5; i++) g[offset.dL_dfwind + i] = b.dL_dwind[i];
  if (taskInfo.dL_dx0) for (size_t i = 0; i < b.dL_dx.size(); i++) g[offset.dL_dx0 + i] = b.dL_dx[i];
  if (taskInfo.dL_dfext) for (int i = 0; i < 3; i++) g[offset.dL_dfext + i] = b.dL_dfext[i];
  if (taskInfo.dL_dconstantForceField) for (size_t i = 0; i < b.dL_dconstantForceField.size(); i++) g[offset.dL_dconstantForceField + i] = b.dL_dconstantForceField[i];
  for (int i = 0; i < 4; i++) if (taskInfo.dL_dk_pertype[i]) g[offset.dL_k[i]] = b.dL_dk_pertype[i];
  if (taskInfo.dL_density) g[offset.dL_density] = b.dL_ddensity;
  if (taskInfo.dL_dmu) for (size_t i = 0; i < taskInfo.mu_primitives.size() && i < b.dL_dmu.size(); i++) g[offset.dL_dmu[i]] = b.dL_dmu[i].second;
  if (taskInfo.dL_dcontrolPoints && !b.dL_dsplines.empty()) {
    int at = offset.dL_dspline;
    for (const VecXd &gs : b.dL_dsplines[0]) for (double q : gs) g[at++] = q;
  }
  return g;
}

bool OptimizeHelper::paramIsWithinBound(const VecXd &x) const {
  for (int i = 0; i < totalParamNumber; i++) if (x[i] > paramUpperBound[i] || x[i] < paramLowerBound[i]) return false;
  return true;
}

VecXd OptimizeHelper::getRandomParam(int randSeed) {
  std::srand(randSeed);
  for (;;) {
    const int seed = std::rand();
    taskInfo.randSeed = seed; taskInfo.srandSeed = randSeed;
    std::srand(seed);
    VecXd x(totalParamNumber);
    for (int i = 0; i < totalParamNumber; i++) {
      const double u = (double) std::rand() / RAND_MAX;
      // uniform in the parameter's bounds (OptimizeHelper.cpp:316-322); the boxes this host class does not model (spline end points,
      // x0: bounded by the scene box in the reference, +-1e3 / +-1e30 here) are cut to +-50 — only where that leaves an interval
      double lo = paramLowerBound[i], hi = paramUpperBound[i];
      if (std::max(lo, -50.0) < std::min(hi, 50.0)) { lo = std::max(lo, -50.0); hi = std::min(hi, 50.0); }
      x[i] = lo + u * (hi - lo);
    }
    ParamInfo param = vecXdToParamInfo(x);
    if (taskInfo.dL_dcontrolPoints && !param.controlPointSplines.empty()) {
      // all curves share the first curve's end-point translation (reduces the initial search space, OptimizeHelper.cpp:326-337)
      std::vector<Spline> init = param_guess.controlPointSplines[0];
      const Spline &first = param.controlPointSplines[0].at(0);
      Vec3d tr;
      for (int d = 0; d < 3; d++) tr[d] = first.segments[0].p1[d] - first.segments[0].p0[d];
      for (Spline &s : init) s.moveEndPoint(0, {s.segments[0].p0[0] + tr[0], s.segments[0].p0[1] + tr[1], s.segments[0].p0[2] + tr[2]});
      param.controlPointSplines[0] = init;
    }
    auto clampNorm = [](double *v, double cap) { const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); if (n > cap && n > 0) for (int d = 0; d < 3; d++) v[d] *= cap / n; };
    if (taskInfo.dL_dfext && param.f_ext.size() == 3) clampNorm(param.f_ext.data(), 1.0);
    if (taskInfo.dL_dfwind) clampNorm(param.f_extwind.data(), 2.0);
    x = paramInfoToVecXd(param);
    if (paramIsWithinBound(x)) return x;
  }
}

double OptimizeHelper::runSimulationAndGetLoss(const VecXd &x) {
  ParamInfo param = vecXdToParamInfo(x);
  return system->runBackwardTask(taskInfo, lossType, lossInfo, statistics, FORWARD_STEPS, param, true).at(0).loss;
}

std::vector<BackwardInformation> OptimizeHelper::runSimulationAndGetLossAndGradients(const VecXd &x) {
  ParamInfo param = vecXdToParamInfo(x);
  return system->runBackwardTask(taskInfo, lossType, lossInfo, statistics, FORWARD_STEPS, param, false);
}

void OptimizeHelper::saveLastIter() {
  system->backwardOptimizationRecords.emplace_back(lastBackwardOptRecord);
  system->backwardOptimizationGuesses.push_back(lastGuess);
  system->exportStatistics(demoNum, statistics, taskInfo);
}

double OptimizeHelper::operator()(const VecXd &x, VecXd &grad) {
  iter++;
  ParamInfo param = vecXdToParamInfo(x);
  std::vector<BackwardInformation> records = system->runBackwardTask(taskInfo, lossType, lossInfo, statistics, FORWARD_STEPS, param, false);
  lastBackwardOptRecord = {system->forwardRecords, records};
  lastGuess = {param, records[0].loss};
  if (exportEveryEvaluation) saveLastIter();
  grad = gradientInfoToVecXd(records[0]);
  return records[0].loss;
}

// ---------------------------------------------------------------------------------------------------------------
// demo set-up (optimization/OptimizationTaskSetup.cpp:154-225 and :48-152, BackwardTaskSolver.cpp:77-137)
// ---------------------------------------------------------------------------------------------------------------
OptimizeHelper *makeOptimizeHelperForDemo(const std::string &name, Simulation *sim) {
  BackwardTaskInformation task;
  ParamInfo truth;
  LossInfo loss;
  LossType lossType = MATCH_TRAJECTORY;
  bool generateGroundtruth = false;
  task.forwardAccuracyLevel = sim->sceneConfig.forwardConvergenceThresh;
  task.backwardAccuracyLevel = sim->sceneConfig.backwardConvergenceThresh;
  if (name == "wind_tshirt") {
    sim->setWindAncCollision(true, true, true, false);
    task.dL_dk_pertype[2] = true; truth.k_pertype[2] = sim->sceneConfig.fabric.k_stiff_stretching;
    task.dL_dfwind = true;
    const double n = std::sqrt(1 + 0.01 + 1);
    truth.f_extwind = {1 / n * 0.015, 0.1 / n * 0.015, 1 / n * 0.015, 10, 0.5};
    lossType = MATCH_TRAJECTORY; generateGroundtruth = true;
  } else if (name == "sphere") {
    sim->setWindAncCollision(false, true, true, false);
    task.dL_dmu = true; task.mu_primitives.push_back(0); truth.mu.push_back({0, 0.3});
    lossType = MATCH_TRAJECTORY; generateGroundtruth = true;
  } else if (name == "wear_hat" || name == "wear_sock") {
    sim->setWindAncCollision(false, true, true, false);
    task.dL_dcontrolPoints = true;
    for (Spline &s : sim->controlPointSplines) s.type = Spline::ENDPOINT_AND_TANGENTS;     // resetSplineConfigsForControlTasks
    if (name == "wear_hat") {
      lossType = MATCHSHAPE_WITH_TRANSLATION;
      const Primitive &head = sim->primitives.at(0);
      for (int d = 0; d < 3; d++) loss.targetTranslation[d] = head.center[d] - 0.5 * (sim->restShapeMinDim[d] + sim->restShapeMaxDim[d]);
      loss.targetTranslation[1] += head.radius * 0.6;
    } else {
      lossType = ASSISTED_DRESSING_KEYPOINTS;
      loss.targetPosPairs = sockKeypointTargets(*sim);
    }
  } else if (name == "dress_twirl") {
    sim->setWindAncCollision(false, true, true, false);
    task.dL_density = true; task.dL_dk_pertype[3] = true;
    truth.density = 0.01; truth.k_pertype[2] = 2.0;
    lossType = DRESS_ANGLE;
    loss.targetTwirlHeight = 0.3;
    const VecXd &rest = sim->restPositions();
    for (int i = 0; i < sim->getNumParticles(); i++)
      if (std::fabs(rest[3 * (size_t) i + 1] - sim->restShapeMinDim[1]) < 1.2) loss.loopPoints.push_back(i);
  } else {
    throw std::runtime_error("Undefined example name (" + name + ").");
  }
  if (generateGroundtruth) {
    sim->resetSystemWithParams(task, truth);
    const double saveF = Simulation::forwardConvergenceThreshold;
    Simulation::forwardConvergenceThreshold = task.forwardAccuracyLevel;
    for (int i = 0; i < sim->sceneConfig.stepNum; i++) sim->step();
    Simulation::forwardConvergenceThreshold = saveF;
    sim->groundTruthForwardRecords.clear();
    for (const ForwardInformation &r : sim->forwardRecords) sim->groundTruthForwardRecords.push_back({r.x, r.v});
    loss.targetSimulation = sim->groundTruthForwardRecords;
  }
  return new OptimizeHelper(sim, loss, task, lossType, sim->sceneConfig.stepNum, truth);
}

// Key points of the lower leg the sock's opening / toe / heel vertices are pulled to (OptimizationTaskSetup.cpp:75-126).
// Joint-bind positions are mapped as centre + R (p) with the child's global rotation, as Capsule::getTransformedPosFromJointBindPos.
std::vector<CorresPondenceTargetInfo> sockKeypointTargets(const Simulation &sim) {
  const Primitive &leg = sim.primitives.at(0);
  if (!leg.isPrimitiveCollection || leg.primitives.size() < 3) throw std::runtime_error("sockKeypointTargets: the scene has no lower leg");
  const Primitive &foot = leg.primitives[1], &shin = leg.primitives[2];
  auto frame = [&](const Primitive &c, Vec3d p) {
    // the capsule's axis is topOffset / length; build the rotation that takes +y to it (axisToRotation of the reference)
    Vec3d ax = {c.topOffset[0] / c.length, c.topOffset[1] / c.length, c.topOffset[2] / c.length};
    Vec3d y = {0, 1, 0};
    Vec3d v = {y[1] * ax[2] - y[2] * ax[1], y[2] * ax[0] - y[0] * ax[2], y[0] * ax[1] - y[1] * ax[0]};
    const double s = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), co = ax[1];
    Vec3d out = p;
    if (s > 1e-9) {                      // Rodrigues
      Vec3d k = {v[0] / s, v[1] / s, v[2] / s};
      const double ang = std::atan2(s, co), cs = std::cos(ang), sn = std::sin(ang);
      Vec3d kxp = {k[1] * p[2] - k[2] * p[1], k[2] * p[0] - k[0] * p[2], k[0] * p[1] - k[1] * p[0]};
      const double kp = k[0] * p[0] + k[1] * p[1] + k[2] * p[2];
      for (int d = 0; d < 3; d++) out[d] = p[d] * cs + kxp[d] * sn + k[d] * kp * (1 - cs);
    }
    for (int d = 0; d < 3; d++) out[d] += leg.center[d] + c.centerInit[d];
    return out;
  };
  const Vec3d centerTopLeft = frame(shin, {-shin.radius, shin.length, 0}), centerTopRight = frame(shin, {shin.radius, shin.length, 0});
  const Vec3d centerTopFront = frame(shin, {0, shin.length, shin.radius}), centerTopBack = frame(shin, {0, shin.length, -shin.radius});
  const Vec3d calf = frame(shin, {0, shin.length * 0.4, -shin.radius});
  const Vec3d heel = frame(foot, {0, foot.length, -foot.radius}), arch = frame(foot, {0, foot.length * 0.5, foot.radius});
  const Vec3d toe = frame(foot, {0, -foot.radius, 0}), tipBack = frame(foot, {0, 0, -foot.radius});
  const Vec3d tipLeft = frame(foot, {-foot.radius, 0, 0}), tipRight = frame(foot, {foot.radius, 0, 0});
  const std::vector<int> topFront = {104, 27, 43, 475, 392, 903, 416, 413, 895}, topLeft = {11, 30, 164, 755, 30}, topRight = {563, 43, 474, 14},
                         toes = {865, 420, 946, 250, 80}, openingBack = {102, 81, 842, 318, 12};
  const int last = sim.sceneConfig.stepNum;
  std::vector<CorresPondenceTargetInfo> m;
  m.push_back({last, heel, {2, 20, 336, 792, 995}});
  m.push_back({last, toe, toes});
  m.push_back({last, arch, {282, 343, 249}});
  m.push_back({last, centerTopFront, topFront});
  m.push_back({last, centerTopLeft, topLeft});
  m.push_back({last, centerTopRight, topRight});
  m.push_back({last, centerTopBack, openingBack});
  m.push_back({last, calf, {37, 241, 349}});
  for (int i = 0; i < 3; i++) {          // the opening gets extra weight while it passes the toes
    const int f = (int) (last * 0.62 + i);
    m.push_back({f, toe, topFront});
    m.push_back({f, tipBack, openingBack});
    m.push_back({f, tipLeft, topLeft});
    m.push_back({f, tipRight, topRight});
  }
  return m;
}

}  // namespace dchost
