// pybind11 module `diffcloth_py`: same module name, functions, classes and attribute names as the reference's
// src/code/python_interface.cpp:164-378, backed by the MI355X stepper. Vectors cross the boundary as numpy float64
// arrays (the reference uses pybind11/eigen.h, which gives Python exactly that). Additions (makeSimFromMesh, Spline,
// writable OptimizeHelper.forward_steps, ...) do not change the reference surface; batched rollouts are driven through
// the C-ABI (diffcloth_amd/capi.py).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>
#include <cstdio>
#include <fstream>
#include <memory>
#include "simulation.h"
#include "optimize.h"

namespace py = pybind11;
using namespace dchost;

typedef py::array_t<double, py::array::c_style | py::array::forcecast> NpArr;

static VecXd toVec(const NpArr &a) {
  auto r = a.unchecked();
  VecXd v((size_t) a.size());
  const double *p = a.data();
  for (size_t k = 0; k < v.size(); k++) v[k] = p[k];
  (void) r;
  return v;
}
static py::array_t<double> toNp(const VecXd &v) { return py::array_t<double>((py::ssize_t) v.size(), v.data()); }
template <size_t K>
static py::array_t<double> toNp(const std::array<double, K> &v) { return py::array_t<double>((py::ssize_t) K, v.data()); }

static Simulation *makeSim(const std::string &exampleName, bool runBackward) {
  Simulation::forwardConvergenceThreshold = 1e-5;      // python_interface.cpp:13
  SceneConfiguration cfg = sceneByName(exampleName);   // throws "Undefined example name (...)" like throwError
  Simulation *sim = Simulation::createSystem(cfg, {0, 0, 0}, runBackward);
  if (exampleName == "wear_hat") {                     // python_interface.cpp:21-25
    const Primitive &head = sim->primitives.at(0);
    Vec3d bust = {head.center[0], head.center[1] + head.radius * 0.6, head.center[2]};
    for (int d = 0; d < 3; d++) sim->taskLossInfo.targetTranslation[d] = bust[d] - 0.5 * (sim->restShapeMinDim[d] + sim->restShapeMaxDim[d]);
  }
  return sim;
}

static OptimizeHelper *makeOptimizeHelperWithSim(const std::string &exampleName, Simulation *sim) {
  Simulation::forwardConvergenceThreshold = 1e-5;      // python_interface.cpp:99
  sim->setPrintVerbose(false);
  OptimizeHelper *h = makeOptimizeHelperForDemo(exampleName, sim);      // BackwardTaskSolver::getOptimizeHelperPointer
  if (exampleName == "wear_hat") {                     // debug target shape of the hat demo (assets: remeshed/Hat/hat_target.txt)
    std::string root = Simulation::assetRoot.empty() ? (std::getenv("DIFFCLOTH_ASSETS") ? std::getenv("DIFFCLOTH_ASSETS") : "/root/reference/src/assets/meshes") : Simulation::assetRoot;
    std::ifstream in(root + "/remeshed/Hat/hat_target.txt");
    VecXd shape;
    double v;
    while (in >> v) shape.push_back(v);
    if (shape.size() == 3 * (size_t) sim->getNumParticles()) h->lossInfo.targetFrameShape.push_back({sim->sceneConfig.stepNum, shape});
  }
  return h;
}

static OptimizeHelper *makeOptimizeHelper(const std::string &exampleName) {
  Simulation *sim = Simulation::createSystem(sceneByName("wear_hat"), {0, 0, 0}, false);   // python_interface.cpp:139-144
  return makeOptimizeHelperWithSim(exampleName, sim);
}

PYBIND11_MODULE(diffcloth_py, m) {
  m.doc() = "MI355X-native DiffCloth stepper with the reference's diffcloth_py surface";

  py::enum_<WindConfig>(m, "WindConfig")
      .value("NO_WIND", NO_WIND).value("WIND_CONSTANT", WIND_CONSTANT).value("WIND_SIN", WIND_SIN)
      .value("WIND_SIN_AND_FALLOFF", WIND_SIN_AND_FALLOFF).value("WIND_FACTOR_PER_STEP", WIND_FACTOR_PER_STEP).export_values();

  py::class_<SceneConfiguration>(m, "SceneConfiguration")
      .def_readwrite("timeStep", &SceneConfiguration::timeStep)
      .def_readwrite("windConfig", &SceneConfiguration::windConfig)
      .def_readonly("stepNum", &SceneConfiguration::stepNum)
      .def_readwrite("customAttachmentVertexIdx", &SceneConfiguration::customAttachmentVertexIdx);

  py::class_<PrimitiveCollisionInformation>(m, "PrimitiveCollisionInformation")
      .def_readonly("primitiveId", &PrimitiveCollisionInformation::primitiveId)
      .def_readonly("particleId", &PrimitiveCollisionInformation::particleId);
  py::class_<SelfCollisionInformation>(m, "SelfCollisionInformation")
      .def_readonly("particleId1", &SelfCollisionInformation::particleId1)
      .def_readonly("particleId2", &SelfCollisionInformation::particleId2);

  py::class_<ForwardInformation>(m, "ForwardInformation")
      .def_readonly("simDurartionFraction", &ForwardInformation::simDurartionFraction)
      .def_readonly("splines", &ForwardInformation::splines)
      .def_property_readonly("x", [](const ForwardInformation &r) { return toNp(r.x); })
      .def_property_readonly("v", [](const ForwardInformation &r) { return toNp(r.v); })
      .def_property_readonly("x_prev", [](const ForwardInformation &r) { return toNp(r.x_prev); })
      .def_property_readonly("v_prev", [](const ForwardInformation &r) { return toNp(r.v_prev); })
      .def_property_readonly("f", [](const ForwardInformation &r) { return toNp(r.f); })
      .def_property_readonly("r", [](const ForwardInformation &r) { return toNp(r.r); })
      .def_property_readonly("x_fixedpoints", [](const ForwardInformation &r) { return toNp(r.x_fixedpoints); })
      .def_readonly("stepIdx", &ForwardInformation::stepIdx)
      .def_readonly("sysMatId", &ForwardInformation::sysMatId)
      .def_readonly("t", &ForwardInformation::t)
      .def_readonly("avgDeformation", &ForwardInformation::avgDeformation)
      .def_readonly("maxDeformation", &ForwardInformation::maxDeformation)
      .def_readonly("converged", &ForwardInformation::converged)
      .def_readonly("convergeIter", &ForwardInformation::convergeIter)
      .def_readonly("cumulateIter", &ForwardInformation::cumulateIter)         // "Total PD Iters" of forwardLog.txt
      .def_readonly("totalConverged", &ForwardInformation::totalConverged)     // "Total Frames Converged"
      .def_readonly("totalRuntime", &ForwardInformation::totalRuntime)
      .def_property_readonly("collisionInfos", [](const ForwardInformation &r) {
        // completeCollisionInfo = ((primitive contacts, self contacts), layers)
        std::vector<SelfCollisionInformation> flat;
        for (auto &l : r.selfCollisionLayers) flat.insert(flat.end(), l.begin(), l.end());
        return py::make_tuple(py::make_tuple(r.primitiveCollisions, flat), r.selfCollisionLayers);
      });

  py::enum_<Spline::SplineType>(m, "SplineType")
      .value("ENDPOINT", Spline::ENDPOINT).value("ENDPOINT_AND_UP", Spline::ENDPOINT_AND_UP)
      .value("ENDPOINT_AND_TANGENTS", Spline::ENDPOINT_AND_TANGENTS);
  py::class_<Spline>(m, "Spline")
      .def(py::init([](const VecXd &p0, const VecXd &p1, double yUp, int pFixed, double f0, double f1) {
             return Spline({p0.at(0), p0.at(1), p0.at(2)}, {p1.at(0), p1.at(1), p1.at(2)}, yUp, pFixed, f0, f1);
           }), py::arg("p0"), py::arg("p1"), py::arg("yUp"), py::arg("pFixed"), py::arg("startFraction") = 0.0, py::arg("endFraction") = 1.0)
      .def("addSegment", [](Spline &s, const VecXd &p1, double yUp, double f0, double f1) { s.addSegment({p1.at(0), p1.at(1), p1.at(2)}, yUp, f0, f1); })
      .def_readwrite("type", &Spline::type)
      .def_readonly("pFixed", &Spline::pFixed)
      .def("getParameterNumber", &Spline::getParameterNumber)
      .def("evalute", [](const Spline &s, double t, int order) { Vec3d q = s.evalute(t, order); return toNp(q); }, py::arg("t"), py::arg("order") = 0)
      .def("dxfixed_dcontrolPoints", [](const Spline &s, double t) {
        std::vector<double> J = s.dxfixed_dcontrolPoints(t);
        const int np = s.getParameterNumber();
        py::array_t<double> a({3, np});
        std::memcpy(a.mutable_data(), J.data(), J.size() * sizeof(double));
        return a;
      })
      .def("paramToVector", [](const Spline &s) { return toNp(s.paramToVector()); })
      .def("updateControlPoints", [](Spline &s, const VecXd &step) { s.updateControlPoints(step); })
      .def("moveEndPoint", [](Spline &s, int seg, const VecXd &p) { s.moveEndPoint(seg, {p.at(0), p.at(1), p.at(2)}); });

  py::class_<BackwardInformation>(m, "BackwardInformation")
      .def_property_readonly("dL_dx", [](const BackwardInformation &b) { return toNp(b.dL_dx); })
      .def_property_readonly("dL_dv", [](const BackwardInformation &b) { return toNp(b.dL_dv); })
      .def_property_readonly("dL_dxfixed", [](const BackwardInformation &b) { return toNp(b.dL_dxfixed); })
      .def_property_readonly("dL_dsplines", [](const BackwardInformation &b) {          // python_interface.cpp:219
        py::list sets;
        for (const auto &set : b.dL_dsplines) { py::list l; for (const VecXd &g : set) l.append(toNp(g)); sets.append(l); }
        return sets;
      })
      .def_property_readonly("dL_dfext", [](const BackwardInformation &b) { return toNp(b.dL_dfext); })
      .def_property_readonly("dL_dwind", [](const BackwardInformation &b) { return toNp(b.dL_dwind); })
      .def_readonly("dL_ddensity", &BackwardInformation::dL_ddensity)
      .def_readonly("dL_dk_pertype", &BackwardInformation::dL_dk_pertype)
      .def_readonly("dL_dmu", &BackwardInformation::dL_dmu)
      .def_readonly("loss", &BackwardInformation::loss)
      .def_readonly("totalRuntime", &BackwardInformation::totalRuntime)
      .def_property_readonly("dL_dconstantForceField", [](const BackwardInformation &b) { return toNp(b.dL_dconstantForceField); })
      .def_property_readonly("dL_dwindtimestep", [](const BackwardInformation &b) { return toNp(b.dL_dwindtimestep); })
      .def_readonly("converged", &BackwardInformation::converged)
      .def_readonly("convergedAccum", &BackwardInformation::convergedAccum)
      .def_readonly("backwardIters", &BackwardInformation::backwardIters)
      .def_readonly("backwardTotalIters", &BackwardInformation::backwardTotalIters);

  py::class_<BackwardTaskInformation>(m, "BackwardTaskInformation")
      .def(py::init<>())
      .def_readwrite("dL_dk_pertype", &BackwardTaskInformation::dL_dk_pertype)
      .def_readwrite("dL_density", &BackwardTaskInformation::dL_density)
      .def_readwrite("dL_dfext", &BackwardTaskInformation::dL_dfext)
      .def_readwrite("dL_dfwind", &BackwardTaskInformation::dL_dfwind)
      .def_readwrite("adddr_dd", &BackwardTaskInformation::adddr_dd)
      .def_readwrite("dL_dcontrolPoints", &BackwardTaskInformation::dL_dcontrolPoints)
      .def_readwrite("dL_dmu", &BackwardTaskInformation::dL_dmu)
      .def_readwrite("mu_primitives", &BackwardTaskInformation::mu_primitives)
      .def_readwrite("dL_dx0", &BackwardTaskInformation::dL_dx0)
      .def_readwrite("dL_dwindFactor", &BackwardTaskInformation::dL_dwindFactor)
      .def_readwrite("dL_dconstantForceField", &BackwardTaskInformation::dL_dconstantForceField)
      .def_readonly("forwardAccuracyLevel", &BackwardTaskInformation::forwardAccuracyLevel)
      .def_readonly("backwardAccuracyLevel", &BackwardTaskInformation::backwardAccuracyLevel)
      .def_readonly("randSeed", &BackwardTaskInformation::randSeed)
      .def_readonly("srandSeed", &BackwardTaskInformation::srandSeed);

  // python_interface.cpp:245-249
  py::class_<CorresPondenceTargetInfo>(m, "CorresPondenceTargetInfo")
      .def(py::init<>())
      .def_readwrite("frameIdx", &CorresPondenceTargetInfo::frameIdx)
      .def_property("targetPos", [](const CorresPondenceTargetInfo &c) { return toNp(c.targetPos); },
                    [](CorresPondenceTargetInfo &c, const NpArr &a) { VecXd v = toVec(a); for (int d = 0; d < 3; d++) c.targetPos[d] = v.at(d); })
      .def_readwrite("particleIndices", &CorresPondenceTargetInfo::particleIndices);

  py::class_<LossInfo>(m, "LossInfo")
      .def_property("targetLoc", [](const LossInfo &l) { return toNp(l.targetLoc); },
                    [](LossInfo &l, const NpArr &a) { VecXd v = toVec(a); for (int d = 0; d < 3; d++) l.targetLoc[d] = v.at(d); })
      .def_property("targetTranslation", [](const LossInfo &l) { return toNp(l.targetTranslation); },
                    [](LossInfo &l, const NpArr &a) { VecXd v = toVec(a); for (int d = 0; d < 3; d++) l.targetTranslation[d] = v.at(d); })
      .def_property("targetFrameShape",
                    [](const LossInfo &l) { py::list out; for (auto &p : l.targetFrameShape) out.append(py::make_tuple(p.first, toNp(p.second))); return out; },
                    [](LossInfo &l, const std::vector<std::pair<int, NpArr>> &v) { l.targetFrameShape.clear(); for (auto &p : v) l.targetFrameShape.push_back({p.first, toVec(p.second)}); })
      .def_readwrite("targetPosPairs", &LossInfo::targetPosPairs);      // python_interface.cpp:256 (ASSISTED_DRESSING_KEYPOINTS targets)

  py::class_<Primitive> primitive(m, "Primitive");
  py::enum_<PrimitiveType>(primitive, "PrimitiveType")
      .value("PLANE", PLANE).value("CUBE", CUBE).value("SPHERE", SPHERE).value("CAPSULE", CAPSULE)
      .value("FOOT", FOOT_PRIM).value("LOWER_LEG", LOWER_LEG).value("BOWL", BOWL).export_values();
  primitive.def_readwrite("primitives", &Primitive::primitives)
      .def_readwrite("isPrimitiveCollection", &Primitive::isPrimitiveCollection)
      .def_readwrite("type", &Primitive::type)
      .def_property("center", [](const Primitive &p) { return toNp(p.center); },
                    [](Primitive &p, const NpArr &a) { VecXd v = toVec(a); for (int d = 0; d < 3; d++) p.center[d] = v.at(d); })
      .def_property_readonly("centerInit", [](const Primitive &p) { return toNp(p.centerInit); })
      .def_readonly("radius", &Primitive::radius)
      .def_readwrite("mu", &Primitive::mu)
      .def("getPointVec", [](const Primitive &p) { return toNp(p.getPointVec()); }, "getPointVec");

  py::class_<Simulation>(m, "Simulation")
      .def_readonly("taskLossInfo", &Simulation::taskLossInfo)
      .def_readonly("primitives", &Simulation::primitives)
      .def_readonly("sceneConfig", &Simulation::sceneConfig)
      .def_readwrite("forwardRecords", &Simulation::forwardRecords)
      .def_readwrite("useCustomRLFixedPoint", &Simulation::useCustomRLFixedPoint)
      .def_property("perStepGradient",
                    [](const Simulation &s) { py::list out; for (auto &g : s.perStepGradient) out.append(toNp(g)); return out; },
                    [](Simulation &s, const std::vector<NpArr> &v) { s.perStepGradient.clear(); for (auto &a : v) s.perStepGradient.push_back(toVec(a)); })
      .def_readwrite("gradientClipping", &Simulation::gradientClipping)
      // additive: runBackwardTask / OptimizeHelper evaluations as one fused launch per direction (simulation.h); records of such an
      // evaluation carry x, v, fixed points and statistics, loadRecordDetails(i) fetches f, r and the contact lists of record i
      .def_readwrite("deviceResidentRollouts", &Simulation::deviceResidentRollouts)
      .def("loadRecordDetails", &Simulation::loadRecordDetails)
      .def_readwrite("controlPointSplines", &Simulation::controlPointSplines)    // sysMat[0].controlPointSplines of the reference (the ACTIVE set's here)
      .def_property_readonly("attachmentSetCount", &Simulation::attachmentSetCount)
      .def_property_readonly("currentAttachmentSet", &Simulation::currentAttachmentSet)
      .def_property_readonly("attachmentSetStartFrames", &Simulation::attachmentSetStartFrames)
      .def("resetSystemWithSplines", [](Simulation &s, const std::vector<Spline> &c) { s.resetSystem(c); })
      .def_readwrite("gradientClippingThreshold", &Simulation::gradientClippingThreshold)
      .def_readwrite("backwardGradientForceDirectSolver", &Simulation::backwardGradientForceDirectSolver)
      .def_property_readonly("ndof_u", &Simulation::getActionDim)
      .def_property_readonly("num_particles", &Simulation::getNumParticles)
      .def_readwrite_static("forwardConvergenceThreshold", &Simulation::forwardConvergenceThreshold)
      .def_readwrite_static("backwardConvergenceThreshold", &Simulation::backwardConvergenceThreshold)
      .def_readwrite_static("assetRoot", &Simulation::assetRoot)
      .def("resetSystem", [](Simulation &s) { s.resetSystem(); }, "reset the simulation")
      .def("step", &Simulation::step, "forward one step")
      .def("getCurrentPosVelocityVec", [](const Simulation &s) { auto p = s.getCurrentPosVelocityVec(); return py::make_tuple(toNp(p.first), toNp(p.second)); }, "get posvel vecs")
      .def("appendPerStepGradient", [](Simulation &s, const NpArr &x) { s.appendPerStepGradient(toVec(x)); }, "append grad", py::arg("x"))
      .def("stepNN", [](Simulation &s, int idx, const NpArr &x, const NpArr &v, const NpArr &fp) { s.stepNN(idx, toVec(x), toVec(v), toVec(fp)); },
           "forward one step with arg", py::arg("idx"), py::arg("x"), py::arg("v"), py::arg("fixedPointPos"))
      .def("setWindAndCollision", &Simulation::setWindAncCollision, "setWindAndCollision", py::arg("windEnable"), py::arg("collisionEnable"),
           py::arg("selfCollisionEnable"), py::arg("enableConstantForcefield"))
      .def("getStateInfo", &Simulation::getStateInfo, "get the forward info of the current step")
      .def("setAction", [](Simulation &s, const NpArr &a) { s.setAction(toVec(a)); }, "set the target position for clips")
      .def("exportCurrentMeshPos", &Simulation::exportCurrentMeshPos, "export the mesh at certain step", py::arg("step"), py::arg("filename"))
      .def("setPrintVerbose", &Simulation::setPrintVerbose, "set whether to print verbose info", py::arg("flag"))
      .def("getPastStateInfo", &Simulation::getPastStateInfo, "get the forward info of a past time step", py::arg("stepIdx"))
      .def("exportCurrentSimulation", &Simulation::exportCurrentSimulation, "export the simulation to files", py::arg("fileName"))
      // on-disk formats beyond the reference's Python surface (its C++ side: Simulation.cpp:4003-4238, Simulation.h:574-620)
      .def_readwrite_static("outputRoot", &Simulation::outputRoot)
      // wind fall-off / per-step wind factors / constant force field of fillForces (Simulation.cpp:55-116; C++-only members of the reference)
      .def_property("windFallOff", [](Simulation &s) { return toNp(s.windFallOff); }, [](Simulation &s, const NpArr &a) { s.windFallOff = toVec(a); })
      .def_property("perstepWindFactor", [](Simulation &s) { return toNp(s.perstepWindFactor); }, [](Simulation &s, const NpArr &a) { s.perstepWindFactor = toVec(a); })
      .def_property("external_force_field", [](Simulation &s) { return toNp(s.external_force_field); }, [](Simulation &s, const NpArr &a) { s.external_force_field = toVec(a); })
      .def_readwrite("enableConstantForcefield", &Simulation::enableConstantForcefield)
      .def_readwrite("windEnabled", &Simulation::windEnabled)
      .def("setWind", [](Simulation &s, const NpArr &dir, double norm, double freq, double phase) { VecXd d = toVec(dir); for (int k = 0; k < 3; k++) s.wind[k] = d.at(k); s.windNorm = norm; s.windFrequency = freq; s.windPhase = phase; },
           "wind direction (unit vector), norm, sin frequency and phase (Simulation.h:357-360)", py::arg("direction"), py::arg("norm"), py::arg("frequency") = 14.0, py::arg("phase") = 0.0)
      .def("setWindFallOffFromFocusPoint", [](Simulation &s, const NpArr &p) { VecXd v = toVec(p); s.setWindFallOffFromFocusPoint({v.at(0), v.at(1), v.at(2)}); }, py::arg("focus"))
      .def("exportSimulation", [](Simulation &s, const std::string &name) { s.exportSimulation(name, s.forwardRecords); },
           "write <name>/<i>.obj + info.txt, the layout the reference's viewer replays", py::arg("fileName"))
      .def("resetForwardRecordsFromFolder", &Simulation::resetForwardRecordsFromFolder, "append one record per <i>.obj of the folder", py::arg("subFolder"))
      .def_static("parameterToString", &Simulation::parameterToString, py::arg("taskInfo"), py::arg("param"))
      .def("stepBackward",
           [](Simulation &s, BackwardTaskInformation &ti, BackwardInformation &g, const ForwardInformation &f, bool isStart, const NpArr &ix, const NpArr &iv) {
             return s.stepBackward(ti, g, f, isStart, toVec(ix), toVec(iv));
           }, "stepbackward one step", py::arg("taskInfo"), py::arg("dL_dxvfnew"), py::arg("forwardInfo_new"), py::arg("isStart"),
           py::arg("dL_dxinit"), py::arg("dL_dvinit"))
      .def("stepBackwardNN",
           [](Simulation &s, BackwardTaskInformation &ti, const NpArr &gx, const NpArr &gv, const ForwardInformation &f, bool isStart, const NpArr &ix, const NpArr &iv) {
             VecXd a = toVec(gx), b = toVec(gv);
             return s.stepBackwardNN(ti, a, b, f, isStart, toVec(ix), toVec(iv));
           }, "stepbackward one step", py::arg("taskInfo"), py::arg("dL_dxnew"), py::arg("dL_dvnew"), py::arg("forwardInfo_new"), py::arg("isStart"),
           py::arg("dL_dxinit"), py::arg("dL_dvinit"))
      // additive: rest mesh access for callers that build their own inputs
      .def("getRestPositions", [](const Simulation &s) { return toNp(s.restPositions()); })
      .def("getTriangles", [](const Simulation &s) { return s.triangles(); })
      .def("getAttachmentVertices", [](const Simulation &s) { return s.attachments(); });

  py::enum_<LossType>(m, "LossType")
      .value("MATCHSHAPE_WITH_TRANSLATION", MATCHSHAPE_WITH_TRANSLATION).value("MULTISTEP_MATCHSHAPE", MULTISTEP_MATCHSHAPE)
      .value("MATCHSHAPE_TRANSLATION_INVARINT", MATCHSHAPE_TRANSLATION_INVARINT).value("ASSISTED_DRESSING_KEYPOINTS", ASSISTED_DRESSING_KEYPOINTS)
      .value("MATCH_TRAJECTORY", MATCH_TRAJECTORY).value("MATCH_TRAJECTORY_MAX", MATCH_TRAJECTORY_MAX).value("MATCH_VELOCITY", MATCH_VELOCITY)
      .value("DRESS_ANGLE", DRESS_ANGLE);

  py::class_<ParamInfo>(m, "ParamInfo")                 // python_interface.cpp:260-266
      .def(py::init<>())
      .def_property("x0", [](const ParamInfo &p) { return toNp(p.x0); }, [](ParamInfo &p, const NpArr &a) { p.x0 = toVec(a); })
      .def_property("v0", [](const ParamInfo &p) { return toNp(p.v0); }, [](ParamInfo &p, const NpArr &a) { p.v0 = toVec(a); })
      .def_property("f_ext", [](const ParamInfo &p) { return toNp(p.f_ext); }, [](ParamInfo &p, const NpArr &a) { p.f_ext = toVec(a); })
      .def_property("f_extwind", [](const ParamInfo &p) { return toNp(p.f_extwind); },
                    [](ParamInfo &p, const NpArr &a) { VecXd v = toVec(a); for (int i = 0; i < 5; i++) p.f_extwind[i] = v.at(i); })
      .def_readwrite("density", &ParamInfo::density)
      .def_readonly("k_pertype", &ParamInfo::k_pertype)
      .def_readonly("controlPointSplines", &ParamInfo::controlPointSplines)
      .def_readonly("mu", &ParamInfo::mu);

  py::class_<OptimizeHelper>(m, "OptimizeHelper")      // python_interface.cpp:337-365
      .def_property_readonly("paramLowerBound", [](const OptimizeHelper &h) { return toNp(h.paramLowerBound); })
      .def_property_readonly("paramUpperBound", [](const OptimizeHelper &h) { return toNp(h.paramUpperBound); })
      .def_readwrite("forward_steps", &OptimizeHelper::FORWARD_STEPS)    // writable here (read-only in the reference): shorter rollouts for checks
      .def_readonly("sim", &OptimizeHelper::system, py::return_value_policy::reference)
      .def_readonly("paramLogScaleTransformOn", &OptimizeHelper::paramLogScaleTransformOn)
      .def_readonly("paramName", &OptimizeHelper::paramName)
      .def_readonly("taskInfo", &OptimizeHelper::taskInfo)
      .def_readonly("lossType", &OptimizeHelper::lossType)
      .def_readonly("lossInfo", &OptimizeHelper::lossInfo)
      .def_readonly("paramActual", &OptimizeHelper::param_actual)
      .def("getActualParam", [](const OptimizeHelper &h) { return toNp(h.getActualParam()); }, "getactualparam")
      .def("getRandomParam", [](OptimizeHelper &h, int seed) { return toNp(h.getRandomParam(seed)); }, "generate random initial parameters",
           py::arg("randSeed") = 0)
      .def("runSimulationAndGetLoss", [](OptimizeHelper &h, const NpArr &x) { return h.runSimulationAndGetLoss(toVec(x)); },
           "compute loss from parameter vector", py::arg("x"))
      .def("vecXdToParamInfo", [](const OptimizeHelper &h, const NpArr &x) { return h.vecXdToParamInfo(toVec(x)); }, py::arg("x"))
      .def("paramInfoToVecXd", [](const OptimizeHelper &h, const ParamInfo &p) { return toNp(h.paramInfoToVecXd(p)); }, py::arg("param"))
      .def("gradientInfoToVecXd", [](const OptimizeHelper &h, const BackwardInformation &g) { return toNp(h.gradientInfoToVecXd(g)); },
           "convert grad struct to grad vector", py::arg("grad"))
      .def("runSimulationAndGetLossGradient", [](OptimizeHelper &h, const NpArr &x) { return h.runSimulationAndGetLossAndGradients(toVec(x)); },
           "compute loss and grads from parameter vector", py::arg("x"))
      .def_readwrite("exportEveryEvaluation", &OptimizeHelper::exportEveryEvaluation)
      .def_property_readonly("experimentName", [](OptimizeHelper &h) { return h.statistics.experimentName; })
      .def("evaluate", [](OptimizeHelper &h, const NpArr &x) { VecXd g; const double L = h(toVec(x), g); return py::make_tuple(L, toNp(g)); },
           "the reference's L-BFGS callback (OptimizeHelper::operator()), headless: loss, flat gradient; logs the iteration to disk", py::arg("x"));

  m.def("loadObjFile", [](const std::string &file) {
    VecXd pts; std::vector<int> tris;
    if (!Simulation::loadObjFile(file, pts, tris)) throw std::runtime_error("cannot read " + file);
    return py::make_tuple(toNp(pts), py::array_t<int>((py::ssize_t) tris.size(), tris.data()));
  }, "read the v / f records of an OBJ file (flat xyz array, flat 0-based triangle array)", py::arg("file"));
  m.def("makeSim", &makeSim, "initialize a simulation instance", py::arg("exampleName"), py::arg("runBackward") = true);
  m.def("makeSimFromMesh",
        [](const std::string &sceneName, const NpArr &verts, const std::vector<int> &tris, bool runBackward,
           const std::vector<std::pair<double, std::vector<int>>> &attachmentSets, int stepNum) {
          SceneConfiguration cfg = sceneByName(sceneName);
          if (!attachmentSets.empty()) { cfg.attachmentPoints = CUSTOM_ARRAY; cfg.customAttachmentVertexIdx = attachmentSets; }
          if (stepNum > 0) cfg.stepNum = stepNum;
          return Simulation::createSystemFromMesh(cfg, toVec(verts), tris, runBackward);
        }, "additive: build a scene from an in-memory mesh (raw file coordinates) instead of an asset path; attachmentSets replaces the scene's "
           "customAttachmentVertexIdx ((start fraction of the rollout, vertex indices) per set — the C++-only SceneConfiguration field of the reference)",
        py::arg("sceneName"), py::arg("verts"), py::arg("tris"), py::arg("runBackward") = true,
        py::arg("attachmentSets") = std::vector<std::pair<double, std::vector<int>>>(), py::arg("stepNum") = 0);
  m.def("makeOptimizeHelper", &makeOptimizeHelper, "initialize an optimize helper", py::arg("exampleName"));
  m.def("makeOptimizeHelperWithSim", &makeOptimizeHelperWithSim, "initialize an optimize helper", py::arg("exampleName"), py::arg("sim"));
  m.def("enableOpenMP", [](int n_threads) { (void) n_threads; std::printf("diffcloth_py (MI355X): host threads are not used by the GPU stepper\n"); },
        "set up Open MP", py::arg("n_threads") = 5);
  m.def("render", [](Simulation *, bool, bool) { throw std::runtime_error("render: the OpenGL viewer is not part of the MI355X stepper"); },
        "rendering the previous trajectry", py::arg("sim"), py::arg("renderPosPairs") = false, py::arg("autoExit") = true);
}
