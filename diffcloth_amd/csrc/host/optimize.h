// OptimizeHelper of the reference (optimization/OptimizeHelper.h): maps a flat parameter vector to ParamInfo, runs the
// rollout + backward sweep through Simulation::runBackwardTask and flattens the gradients again — the object the
// reference's L-BFGS driver and `diffcloth_py.OptimizeHelper` users call.
#pragma once
#include <string>
#include <vector>
#include "simulation.h"

namespace dchost {

class OptimizeHelper {
 public:
  struct Offsets {
    int dL_dfwind = 0, dL_dfext = 0, dL_density = 0, dL_dspline = 0, dL_dx0 = 0, dL_dconstantForceField = 0;
    int dL_k[4] = {0, 0, 0, 0};
    std::vector<int> dL_dmu;
  };
  LossInfo lossInfo;
  BackwardTaskInformation taskInfo;
  Simulation *system = nullptr;
  TaskSolveStatistics statistics;
  VecXd paramLowerBound, paramUpperBound;
  std::vector<std::string> paramName;
  std::vector<bool> paramLogScaleTransformOn;
  int FORWARD_STEPS = 0;
  LossType lossType = MATCH_TRAJECTORY;
  ParamInfo param_guess, param_actual;
  int totalParamNumber = 0, totalSplineParamNumber = 0;
  Offsets offset;

  OptimizeHelper(Simulation *system, const LossInfo &lossInfo, const BackwardTaskInformation &taskInfo, LossType lossType, int FORWARD_STEPS,
                 const ParamInfo &paramActual);
  ParamInfo vecXdToParamInfo(const VecXd &x) const;
  VecXd paramInfoToVecXd(const ParamInfo &param) const;
  VecXd gradientInfoToVecXd(const BackwardInformation &backwardInfo) const;
  VecXd getActualParam() const { return paramInfoToVecXd(param_actual); }
  bool paramIsWithinBound(const VecXd &x) const;
  VecXd getRandomParam(int randSeed = 0);
  double runSimulationAndGetLoss(const VecXd &x);
  std::vector<BackwardInformation> runSimulationAndGetLossAndGradients(const VecXd &x);
  // The L-BFGS callback of the reference (OptimizeHelper::operator(), OptimizeHelper.cpp:533-575) without its render
  // window: one rollout + backward sweep, the iteration is recorded and written to disk (saveLastIter), returns the loss.
  double operator()(const VecXd &x, VecXd &grad);
  void saveLastIter();            // OptimizeHelper.cpp:528-532
  int demoNum = 0, iter = 0;
  bool exportEveryEvaluation = true;
  std::pair<std::vector<ForwardInformation>, std::vector<BackwardInformation>> lastBackwardOptRecord;
  std::pair<ParamInfo, double> lastGuess;

 private:
  void setParameterBounds();
};

// BackwardTaskSolver::getOptimizeHelperPointer for the demos this build carries (wind_tshirt, sphere, wear_hat,
// wear_sock, dress_twirl): gradient switches, ground-truth parameters / rollout, loss type and loss targets.
OptimizeHelper *makeOptimizeHelperForDemo(const std::string &exampleName, Simulation *sim);
std::vector<CorresPondenceTargetInfo> sockKeypointTargets(const Simulation &sim);

}  // namespace dchost
