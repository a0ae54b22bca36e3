// Host-side `Simulation` of the MI355X DiffCloth stepper: keeps the entry points and record types of the
// reference class (reference: /root/reference/src/code/simulation/Simulation.h) and forwards the hot path to
// the C-ABI of libdiffcloth_hip.so (include/diffcloth_hip.h). No Eigen: vectors are std::vector<double> in the
// reference's layout (xyz-interleaved, length 3N); the pybind layer exposes them as numpy arrays exactly as
// pybind11/eigen.h does for the reference.
//
//   reference member                         here
//   Simulation::step()            :703       Simulation::step()
//   stepNN(idx,x,v,fixedPointPos) :705       Simulation::stepNN(...)
//   stepBackward(...)             :569-572   Simulation::stepBackward(...)
//   stepBackwardNN(...)           :564-567   Simulation::stepBackwardNN(...)
//   createSystem(scene,center,rb) :499-500   Simulation::createSystem(...)
//   resetSystem()                 :923-941   Simulation::resetSystem()
//   getStateInfo/getPastStateInfo :966-989   same names
#pragma once
#include <array>
#include <map>
#include <string>
#include <utility>
#include <vector>
#include "../../../include/diffcloth_hip.h"

namespace dchost {

typedef std::vector<double> VecXd;
typedef std::array<double, 3> Vec3d;

enum Orientation { FRONT, DOWN, BACK, CUSTOM_ORIENTATION };                       // engine/Constants.h:35-37
enum AttachmentConfigs { NO_ATTACHMENTS, LEFT_RIGHT_CORNERS_2, CUSTOM_ARRAY };    // :38-42
enum TrajectoryConfigs { NO_TRAJECTORY, CORNERS_2_UP, CORNERS_2_WEARHAT, CORNERS_1_WEARHAT, CORNERS_2_WEARSOCK,
                         TRAJECTORY_DRESS_TWIRL, FIXED_POINT_TRAJECTORY, PER_STEP_TRAJECTORY };
enum PrimitiveConfiguration { PRIM_NONE, PLANE_AND_SPHERE, PLANE_BUST_WEARHAT, FOOT, BIG_SPHERE, Y0PLANE, SLOPE };
enum WindConfig { NO_WIND, WIND_CONSTANT, WIND_SIN, WIND_SIN_AND_FALLOFF, WIND_FACTOR_PER_STEP };   // :55-61
enum PrimitiveType { PLANE, CUBE, SPHERE, CAPSULE, FOOT_PRIM, LOWER_LEG, BOWL };

struct FabricConfiguration {      // Simulation.h:103-118
  double clothDimX = 6, clothDimY = 6;
  double k_stiff_stretching = 100, k_stiff_bending = 0.01;
  int gridNumX = 25, gridNumY = 25;
  double density = 0.1;
  bool keepOriginalScalePoint = false, isModel = false;
  std::string name;             // mesh path (relative to the asset root) when isModel
};

struct SceneConfiguration {       // Simulation.h:276-296
  FabricConfiguration fabric;
  Orientation orientation = FRONT;
  Vec3d upVector = {0, 1, 0};
  AttachmentConfigs attachmentPoints = NO_ATTACHMENTS;
  std::vector<std::pair<double, std::vector<int>>> customAttachmentVertexIdx;
  TrajectoryConfigs trajectory = NO_TRAJECTORY;
  PrimitiveConfiguration primitiveConfig = PRIM_NONE;
  WindConfig windConfig = NO_WIND;
  Vec3d sockLegOrientation = {0, 1, 0};
  double timeStep = 1.0 / 90;
  int stepNum = 100;
  double forwardConvergenceThresh = 1e-7, backwardConvergenceThresh = 5e-4;
  std::string name;
};

struct Primitive {                // the fields of Primitive.h the Python side reads
  PrimitiveType type = SPHERE;
  Vec3d center = {0, 0, 0}, centerInit = {0, 0, 0};
  double radius = 1, length = 0, mu = 0;
  bool rotates = false, isPrimitiveCollection = false;
  bool discretized = false;       // Sphere::discretized (Primitive.h:222): face normals of the sphere's mesh as contact normals
  Vec3d topOffset = {0, 0, 0};
  Vec3d upperLeft = {0, 0, 0}, upperRight = {0, 0, 0};   // PLANE: two corners relative to the centre (Primitive.cpp:13-21)
  std::vector<Primitive> primitives;            // children of a LowerLeg
  VecXd getPointVec() const { return VecXd(center.begin(), center.end()); }
};

// Trajectory of one fixed point: piecewise cubic Hermite curve over the simulated time span (reference Spline.h).
// Parameters per segment: ENDPOINT p1 (tangents follow as m0 = p1 - p0 + yUp e_y, m1 = p1 - p0 - yUp e_y),
// ENDPOINT_AND_UP (p1, yUp), ENDPOINT_AND_TANGENTS (p1, m0, m1).
struct Spline {
  enum SplineType { ENDPOINT, ENDPOINT_AND_UP, ENDPOINT_AND_TANGENTS };
  struct Segment {
    Vec3d p0 = {0, 0, 0}, m0 = {0, 0, 0}, p1 = {0, 0, 0}, m1 = {0, 0, 0};
    double yUp = 8, startFraction = 0, endFraction = 1;
    int segId = 0;
  };
  std::vector<Segment> segments;
  int pFixed = 0;                 // index of the fixed point this curve drives
  SplineType type = ENDPOINT;

  Spline() {}
  Spline(Vec3d p0, Vec3d p1, double yUp, int pFixed, double startFraction = 0.0, double endFraction = 1.0);
  static int parametersPerSegment(SplineType t) { return t == ENDPOINT ? 3 : (t == ENDPOINT_AND_UP ? 4 : 9); }
  int getParameterNumber() const { return parametersPerSegment(type) * (int) segments.size(); }
  void addSegment(Vec3d p1, double yUp, double startFraction, double endFraction);
  void moveEndPoint(int segId, Vec3d newp1);
  Vec3d evalute(double t, int order = 0) const;                 // (sic) position, or d/ds for order 1, at simulation fraction t
  std::vector<double> dxfixed_dcontrolPoints(double t) const;   // 3 x getParameterNumber(), row-major
  VecXd paramToVector() const;
  void updateControlPoints(const VecXd &step);
 private:
  const Segment &segmentAt(double t) const;
  static double localTime(const Segment &seg, double t);
  static void retangent(Segment &seg);
};

struct PrimitiveCollisionInformation { int primitiveId = -1, particleId = -1; Vec3d normal = {0, 0, 0}; };   // Simulation.h:39-51
struct SelfCollisionInformation { int particleId1 = -1, particleId2 = -1, layerId = 0; Vec3d normal = {0, 0, 0}; };   // Simulation.h:53-63

struct ForwardInformation {       // Simulation.h:68-100 (hot-path fields)
  VecXd x, v, x_prev, v_prev, f, r, s_n, x_fixedpoints;
  std::vector<PrimitiveCollisionInformation> primitiveCollisions;
  std::vector<std::vector<SelfCollisionInformation>> selfCollisionLayers;   // collisionInfos.second (contactSorting output)
  int sysMatId = 0;
  double t = 0, windFactor = 0, avgDeformation = 0, maxDeformation = 0;
  bool converged = false;
  int convergeIter = 0, totalConverged = 0, cumulateIter = 0, stepIdx = 0;
  double loss = 0;
  long long totalRuntime = 0;     // microseconds spent in step() up to and including this record (Simulation.cpp:1396-1397)
  double simDurartionFraction = 0;   // (sic) t / (timeStep * stepNum), the spline parameter of this step
  std::vector<Spline> splines;
  int deviceSlot = 0;             // tape slot of libdiffcloth_hip holding this record
};

struct BackwardInformation {      // Simulation.h:136-162 (hot-path fields)
  VecXd dL_dx, dL_dv, dL_dxfixed, dL_dxfixed_accum;
  Vec3d dL_dfext = {0, 0, 0};
  std::array<double, 5> dL_dwind = {0, 0, 0, 0, 0};
  double dL_ddensity = 0;
  std::array<double, 4> dL_dk_pertype = {0, 0, 0, 0};
  std::vector<std::pair<int, double>> dL_dmu;
  std::vector<std::vector<VecXd>> dL_dsplines;   // [attachment set][spline] -> gradient w.r.t. the spline parameters
  double loss = 0;
  long long totalRuntime = 0;
  bool converged = false;
  int convergedAccum = 0, backwardIters = 0, backwardTotalIters = 0;
  int correspondingForwardIdxInStats = 0;
  VecXd dL_dconstantForceField;   // 3N, accumulated over the steps (Simulation.cpp:1714-1718)
  VecXd dL_dwindtimestep;         // one entry per step (Simulation.cpp:1720-1729)
};

struct BackwardTaskInformation {  // Simulation.h:188-209
  std::array<bool, 4> dL_dk_pertype = {false, false, false, false};
  bool dL_density = false, dL_dfext = false, dL_dconstantForceField = false, dL_dfwind = false, adddr_dd = false;
  bool dL_dcontrolPoints = false, dL_dxfixed = false, dL_dmu = false, dL_dx0 = false, dL_dwindFactor = false;
  double forwardAccuracyLevel = 1e-7, backwardAccuracyLevel = 5e-4;
  std::vector<int> mu_primitives;
  int randSeed = 0, srandSeed = 0;
};

enum LossType {                   // engine/Constants.h:12-22 (same order)
  MATCHSHAPE_WITH_TRANSLATION, MULTISTEP_MATCHSHAPE, MATCHSHAPE_TRANSLATION_INVARINT, ASSISTED_DRESSING_KEYPOINTS,
  MATCH_TRAJECTORY, MATCH_TRAJECTORY_MAX, MATCH_VELOCITY, DRESS_ANGLE
};

struct CorresPondenceTargetInfo {  // (sic) Simulation.h:211-219: the farthest of `particleIndices` is pulled to targetPos at frameIdx
  int frameIdx = 0;
  Vec3d targetPos = {0, 0, 0};
  std::vector<int> particleIndices;
};

struct ForwardInformation;
struct LossInfo {                 // Simulation.h:252-261
  Vec3d targetLoc = {0, 0, 0}, targetTranslation = {0, 0, 0};
  std::vector<std::pair<int, VecXd>> targetFrameShape;
  std::vector<std::pair<VecXd, VecXd>> targetSimulation;   // (x, v) per frame of the ground-truth run
  VecXd targetShape;
  std::vector<int> loopPoints;     // sorted particle ids (std::set in the reference)
  double targetTwirlHeight = 0;
  std::vector<CorresPondenceTargetInfo> targetPosPairs;
};

struct ParamInfo {                // Simulation.h:120-133
  VecXd x0, v0, f_ext, f_ext_timestep, f_constantForceField;
  std::array<double, 5> f_extwind = {0, 0, 0, 0, 0};
  double density = 0;
  std::array<double, 4> k_pertype = {0, 0, 0, 0};
  std::vector<std::vector<Spline>> controlPointSplines;
  std::vector<std::pair<int, double>> mu;
};

struct TaskSolveStatistics {      // Simulation.h:221-250
  int totalForwardSim = 0, totalBackprop = 0;
  int optimizationRecordsSaved = 0, forwardWritten = 0, backwardWritten = 0;
  bool configWritten = false;
  std::string experimentName;
  std::vector<std::pair<ParamInfo, ForwardInformation>> completeForwardLog;     // last frame of every forward run
  std::vector<std::pair<ParamInfo, BackwardInformation>> completeBackwardLog;   // first frame of every backward sweep
};

class Simulation {
 public:
  // process-global thresholds, as in the reference (Simulation.h:333-334, Simulation.cpp:17-19)
  static double forwardConvergenceThreshold, backwardConvergenceThreshold;
  static std::string assetRoot;   // directory holding "remeshed/..." meshes (DIFFCLOTH_ASSETS or the reference's src/assets/meshes)

  SceneConfiguration sceneConfig;
  std::vector<Primitive> primitives;
  std::vector<ForwardInformation> forwardRecords;
  std::vector<VecXd> perStepGradient;
  LossInfo taskLossInfo;
  bool gradientClipping = true;              // Simulation.h:330
  double gradientClippingThreshold = 16.0;   // :331
  bool useCustomRLFixedPoint = false;
  bool backwardGradientForceDirectSolver = false;   // :324
  bool printVerbose = false;
  bool windEnabled = false, contactEnabled = true, selfcollisionEnabled = true, gravityEnabled = true;   // Simulation.cpp:9-16
  Vec3d gravity = {0, -9.8, 0};              // :356
  Vec3d wind = {0.01, 0, 1};                 // :357
  double windNorm = 0.15, windFrequency = 14, windPhase = 0;
  VecXd windFallOff;                          // 3N per-vertex factors of the wind (Simulation.h:349; ones unless set, :2592-2593)
  VecXd perstepWindFactor;                    // WIND_FACTOR_PER_STEP: factor of step k (Simulation.h:350, :3213-3214)
  VecXd external_force_field;                 // 3N constant force field, added when enableConstantForcefield (Simulation.cpp:87-89)
  bool enableConstantForcefield = false;
  void setWindFallOffFromFocusPoint(const Vec3d &focus);   // min(1 / |focus - x_i|, 1) per vertex (Simulation.cpp:3125-3130)
  double k_stiff_attachment = 10000;           // AttachmentSpring::k_stiff (AttachmentSpring.cpp:10)
  Vec3d restShapeMinDim = {0, 0, 0}, restShapeMaxDim = {0, 0, 0}, restShapeMidPoint = {0, 0, 0};
  VecXd rlFixedPointPos;
  std::vector<Spline> controlPointSplines;   // sysMat[0].controlPointSplines
  std::vector<VecXd> fixedPointTrajectory;   // FIXED_POINT_TRAJECTORY: targets per step

  ~Simulation();
  static Simulation *createSystem(SceneConfiguration sceneConfig, Vec3d center, bool runBackward = true);
  // same, from an in-memory mesh (already in the reference's raw file coordinates): used for the GPU-box tests
  static Simulation *createSystemFromMesh(SceneConfiguration sceneConfig, const VecXd &verts, const std::vector<int> &tris,
                                          bool runBackward = true);

  void resetSystem();
  void resetSystem(const std::vector<Spline> &controlPoints);   // Simulation.cpp:2858-2862
  // rollout-level driver of the optimisation demos (csrc/host/optimize.cpp)
  double calculateLossAndGradient(LossType lossType, LossInfo &lossInfo, VecXd &dL_dx, VecXd &dL_dv, int idx, bool calculateLoss);   // Simulation.cpp:3237-3488
  void resetSystemWithParams(BackwardTaskInformation &taskConfiguration, ParamInfo &param);                                          // :3490-3584
  std::vector<BackwardInformation> runBackwardTask(BackwardTaskInformation taskConfiguration, LossType lossType, LossInfo &lossInfo,
                                                   TaskSolveStatistics &taskStatistics, int FORWARD_STEPS, ParamInfo guess, bool lossOnly,
                                                   bool skipForward = false);                                                       // :3853-3961
  void rebuildSystem();          // stiffness / density changed: new constraint weights, masses and system matrix on the device
  std::vector<std::pair<VecXd, VecXd>> groundTruthForwardRecords;
  void step();
  // Device-resident evaluation (include/diffcloth_hip.h "device-resident schedules"): the per-step inputs of `nsteps` steps (wind
  // factors, fixed-point targets) are uploaded once and the steps run in ONE launch; the records get x, v, fixed points and solver
  // statistics (f, r and the contact lists stay on the device: loadRecordDetails). Returns false without doing anything when the
  // scene needs a per-step host decision (PER_STEP trajectory, fall-off wind together with a force field) or
  // `deviceResidentRollouts` is off — the caller then loops over step().
  bool rolloutOnDevice(int nsteps);
  void loadRecordDetails(int recordIdx);          // f, r, primitive and self contacts of forwardRecords[recordIdx] from the device
  // backward sweep over all records in ONE launch with the per-frame loss gradients as a device schedule; `seeds[i]` = (dL_dx, dL_dv)
  // of the loss w.r.t. the state of record i. Returns the BackwardInformation per record like the stepBackward loop of
  // runBackwardTask (entry 0 complete; dL_dx / dL_dv of the intermediate entries are not downloaded). Empty = not applicable.
  std::vector<BackwardInformation> sweepBackwardOnDevice(BackwardTaskInformation &taskInfo, const std::vector<std::pair<VecXd, VecXd>> &seeds, double loss);
  bool deviceResidentRollouts = true;             // environment DIFFCLOTH_DEVICE_ROLLOUTS=0 switches the fast path off
  void stepNN(int idx, const VecXd &x, const VecXd &v, const VecXd &fixedPointPos);
  BackwardInformation stepBackward(BackwardTaskInformation &taskInfo, BackwardInformation &gradient_new,
                                   const ForwardInformation &forwardInfo_new, bool isStart, const VecXd &dL_dxinit,
                                   const VecXd &dL_dvinit);
  BackwardInformation stepBackwardNN(BackwardTaskInformation &taskInfo, VecXd &dL_dxnew, VecXd &dL_dvnew,
                                     const ForwardInformation &forwardInfo_new, bool isStart, const VecXd &dL_dxinit,
                                     const VecXd &dL_dvinit);
  ForwardInformation getStateInfo() const { return forwardRecords.back(); }
  ForwardInformation getPastStateInfo(int stepIdx) const { return forwardRecords.at(stepIdx); }
  std::pair<VecXd, VecXd> getCurrentPosVelocityVec() const { return {forwardRecords.back().x, forwardRecords.back().v}; }
  void setAction(const VecXd &a) { rlFixedPointPos = a; }
  int getActionDim() const { return 3 * (int) attachmentVertices.size(); }
  int getNumParticles() const { return N; }
  void setPrintVerbose(bool v) { printVerbose = v; }
  void setWindAncCollision(bool wind_, bool collision, bool selfCollision, bool constantForceField);
  void appendPerStepGradient(const VecXd &x) { perStepGradient.push_back(x); }
  // ---- on-disk formats (export.cpp); every path is relative to outputFolder() ----
  static std::string outputRoot;   // "" -> $DIFFCLOTH_OUTPUT, else "output" (the reference writes to SOURCE_PATH/output/)
  static std::string outputFolder();
  static bool loadObjFile(const std::string &file, VecXd &points, std::vector<int> &triangles);
  void exportCurrentMeshPos(int step, const std::string &fileName) const;
  void exportCurrentSimulation(const std::string &fileName) const;
  void exportSimulation(const std::string &fileName, const std::vector<ForwardInformation> &records) const;
  void exportFrameInfo(const ForwardInformation &record, const std::string &file) const;
  int resetForwardRecordsFromFolder(const std::string &subFolder);    // appends one record per "<i>.obj"; returns the frame count
  void exportStatistics(int demoIdx, TaskSolveStatistics &statistics, const BackwardTaskInformation &taskInfo, bool writePerf = true);
  static std::string taskInfoToString(const BackwardTaskInformation &taskInfo);
  static std::string parameterToString(const BackwardTaskInformation &taskInfo, const ParamInfo &param);
  static std::string forwardInfoToString(const BackwardTaskInformation &taskInfo, const ForwardInformation &record);
  static std::string backwrdInfoAndGradToString(const BackwardTaskInformation &taskInfo, const BackwardInformation &grad);   // (sic)
  double meshArea(const VecXd &x) const;
  // records of the optimisation iterations kept for exportStatistics (Simulation.h:441-443)
  std::vector<std::pair<std::vector<ForwardInformation>, std::vector<BackwardInformation>>> backwardOptimizationRecords;
  std::vector<std::pair<ParamInfo, double>> backwardOptimizationGuesses;
  const VecXd &restPositions() const { return rest; }
  const std::vector<int> &triangles() const { return tris; }
  const std::vector<int> &attachments() const { return attachmentVertices; }
  dc_ctx *context() const { return ctx; }
  // Several attachment sets (SceneConfiguration::customAttachmentVertexIdx with more than one entry; the reference's sysMat vector,
  // Simulation.cpp:2371-2393): set i takes over at record startFrameNum_i = (int) (fraction_i * stepNum) (Simulation::step, :1053-1068).
  int attachmentSetCount() const { return (int) attachmentSets.size(); }
  int currentAttachmentSet() const { return currentSysmatId; }
  std::vector<int> attachmentSetStartFrames() const { std::vector<int> r; for (const auto &a : attachmentSets) r.push_back(a.startFrameNum); return r; }

 private:
  Simulation() {}
  void buildFromMesh(VecXd pts, const std::vector<int> &tris, bool isModel);
  void initScene();
  void configureDevice();
  void pushParams();
  VecXd fixedPointTargets(double t);
  struct DeviceBackwardStep { VecXd dxf, dmu, fvec; double par[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int converged = 0, iters = 0; };
  // host side of stepBackward (Simulation.cpp:1608-1764): accumulate this step's device outputs onto the carried gradient record
  BackwardInformation accumulateBackward(BackwardTaskInformation &taskInfo, const BackwardInformation &gradient_new, const ForwardInformation &fwd,
                                         const DeviceBackwardStep &d);
  bool needsForceVector(const BackwardTaskInformation &taskInfo) const;
  double windFactorAt(double t, int stepIdx) const;
  bool windHasFallOff() const { return sceneConfig.windConfig == WIND_SIN_AND_FALLOFF || sceneConfig.windConfig == WIND_FACTOR_PER_STEP; }

  // One attachment set = one system matrix of the reference (SystemMatrix: its fixed points, attachment springs, splines, P). Here: one
  // engine context per set (same mesh, primitives and parameters; its own attachment rows in P, its own tape). The members
  // attachmentVertices / fixedPointRest / fixedPointCur / controlPointSplines / ctx are the ACTIVE set's; activateSet swaps them.
  struct AttachmentSet {
    int startFrameNum = 0;
    std::vector<int> vertices;
    VecXd fixedRest, fixedCur;
    std::vector<Spline> splines;
    dc_ctx *ctx = nullptr;
  };
  std::vector<AttachmentSet> attachmentSets;
  int currentSysmatId = 0;
  void activateSet(int i);
  void selectSetForStep();              // Simulation::step, Simulation.cpp:1053-1068
  // constant force field as last uploaded per context (signature of its values; 0 = none on the device): Simulation::step re-uploads it only
  // when enableConstantForcefield / external_force_field changed (it was uploaded, with a stream synchronisation, at every step)
  std::map<dc_ctx *, unsigned long long> fieldSignature;
  void uploadForceField(dc_ctx *c, bool field);
  template <class F> void forEachContext(F f) {
    for (size_t i = 0; i < attachmentSets.size(); i++) f((int) i == currentSysmatId ? ctx : attachmentSets[i].ctx);
    if (attachmentSets.empty() && ctx) f(ctx);
  }
  std::vector<Spline> restSplines(const std::vector<int> &vertices) const;

  dc_ctx *ctx = nullptr;
  int N = 0, tapeSlots = 0;
  VecXd rest;
  std::vector<int> tris;
  std::vector<int> attachmentVertices;
  VecXd fixedPointRest, fixedPointCur;
  bool runBackward = true;
  double paramsFwdTol = -1, paramsBwdTol = -1;
  bool paramsClip = true, paramsDirect = false;
  double paramsClipThr = 16.0;
};

// scene tables (optimization/OptimizationTaskConfigurations.cpp:10-349)
SceneConfiguration sceneByName(const std::string &name);

}  // namespace dchost
