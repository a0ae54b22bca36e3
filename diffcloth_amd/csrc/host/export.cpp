// On-disk formats of the host `Simulation`: OBJ frame dumps, per-frame clip files, parameter / log text files of an
// optimisation run, and the reader that turns a folder of frames back into forward records.
// The layouts are the reference's, so that batched GPU runs can be replayed by its viewer (`-mode visualize`) and
// compared with the runs it ships under output/ (reference: /root/reference/src/code/simulation/):
//   <root>/<name>/<i>/0-CLOTH.obj + info.txt, <root>/area.txt      exportCurrentSimulation     Simulation.cpp:3788-3842
//   <root>/<name>.txt (3 decimals) + <name>.obj                     exportCurrentMeshPos        :3844-3851
//   <root>/<name>/<i>.obj + info.txt                                exportSimulation            :4195-4238
//   info.txt  "CLIP_<k>:x,y,z" (5 decimals)                         exportFrameInfo             Simulation.h:846-861
//   param.txt / forwardLog.txt / backwardLog.txt / perf.txt ...     exportStatistics            Simulation.cpp:4003-4130
//   reader (numeric .obj names, sorted)                             resetForwardRecordsFromFolder  Simulation.h:574-620
//   "v x y z" / "f a b c" (1-based) with default stream precision   MeshFileHandler::saveOBJFile   engine/MeshFileHandler.h:137-160
#include <sys/stat.h>
#include <dirent.h>
#include <algorithm>
#include <cmath>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <stdexcept>
#include "simulation.h"

namespace dchost {

std::string Simulation::outputRoot = "";

namespace {
const char *kConstraintNames[4] = {"CONSTRAINT_SPRING_STRETCH", "CONSTRAINT_ATTACHMENT", "CONSTRAINT_TRIANGLE", "CONSTRAINT_TRIANGLE_BENDING"};
const char *kWindNames[5] = {"NO_WIND", "WIND_CONSTANT", "WIND_SIN", "WIND_SIN_AND_FALLOFF", "WIND_FACTOR_PER_STEP"};

std::string fixedStr(double v, int precision = 2) {
  std::ostringstream s;
  s << std::fixed << std::setprecision(precision) << v;
  return s.str();
}
template <class V> std::string tupleStr(const V &v, int precision = 2) {
  std::ostringstream s;
  s << std::fixed << std::setprecision(precision) << "(";
  for (size_t i = 0; i < v.size(); i++) s << (i ? "," : "") << v[i];
  s << ")";
  return s.str();
}
void makeDirs(const std::string &path) {      // mkdir -p
  for (size_t i = 1; i <= path.size(); i++)
    if (i == path.size() || path[i] == '/') {
      const std::string sub = path.substr(0, i);
      if (!sub.empty() && sub != "." && sub != "/") ::mkdir(sub.c_str(), 0755);
    }
}
void writeText(const std::string &file, const std::string &text, bool append = false) {
  std::ofstream os(file, append ? std::ios::app : std::ios::trunc);
  if (!os) throw std::runtime_error("cannot write " + file);
  os << text;
}
void writeObj(const std::string &file, const VecXd &x, const std::vector<int> &tris) {
  std::ofstream os(file);
  if (!os) throw std::runtime_error("cannot write " + file);
  for (size_t i = 0; i + 2 < x.size(); i += 3) os << "v " << x[i] << " " << x[i + 1] << " " << x[i + 2] << "\n";
  for (size_t t = 0; t + 2 < tris.size(); t += 3) os << "f " << tris[t] + 1 << " " << tris[t + 1] + 1 << " " << tris[t + 2] + 1 << "\n";
}
}  // namespace

std::string Simulation::outputFolder() {
  std::string root = outputRoot;
  if (root.empty()) root = std::getenv("DIFFCLOTH_OUTPUT") ? std::getenv("DIFFCLOTH_OUTPUT") : "output";
  if (root.back() != '/') root += "/";
  return root;
}

bool Simulation::loadObjFile(const std::string &file, VecXd &points, std::vector<int> &triangles) {
  std::ifstream is(file);
  if (!is) return false;
  points.clear(); triangles.clear();
  std::string line;
  while (std::getline(is, line)) {
    std::istringstream ls(line);
    std::string tag;
    ls >> tag;
    if (tag == "v") {
      double a, b, c;
      if (ls >> a >> b >> c) { points.push_back(a); points.push_back(b); points.push_back(c); }
    } else if (tag == "f") {
      std::string tok;
      int n = 0, idx[3];
      while (n < 3 && ls >> tok) idx[n++] = std::stoi(tok.substr(0, tok.find('/'))) - 1;    // "a", "a/b", "a//c", "a/b/c"
      if (n == 3) triangles.insert(triangles.end(), idx, idx + 3);
    }
  }
  return true;
}

void Simulation::exportFrameInfo(const ForwardInformation &record, const std::string &file) const {
  std::string out;
  for (size_t k = 0; k + 2 < record.x_fixedpoints.size(); k += 3)
    out += "CLIP_" + std::to_string(k / 3) + ":" + fixedStr(record.x_fixedpoints[k], 5) + "," + fixedStr(record.x_fixedpoints[k + 1], 5) + "," +
           fixedStr(record.x_fixedpoints[k + 2], 5) + "\n";
  writeText(file, out);
}

double Simulation::meshArea(const VecXd &x) const {
  double total = 0;
  for (size_t t = 0; t + 2 < tris.size(); t += 3) {
    const double *a = &x[3 * tris[t]], *b = &x[3 * tris[t + 1]], *c = &x[3 * tris[t + 2]];
    const double u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, w[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    const double n[3] = {u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0]};
    total += 0.5 * std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  }
  return total;
}

void Simulation::exportCurrentMeshPos(int step, const std::string &fileName) const {
  const ForwardInformation &r = forwardRecords.at(step);
  const std::string root = outputFolder();
  makeDirs(root + fileName.substr(0, fileName.find_last_of('/') == std::string::npos ? 0 : fileName.find_last_of('/')));
  std::ostringstream txt;
  txt << std::fixed << std::setprecision(3);
  for (int i = 0; i < N; i++) txt << r.x[3 * i] << " " << r.x[3 * i + 1] << " " << r.x[3 * i + 2] << "\n";
  writeText(root + fileName + ".txt", txt.str());
  writeObj(root + fileName + ".obj", r.x, tris);
}

void Simulation::exportCurrentSimulation(const std::string &fileName) const {
  const std::string root = outputFolder();
  std::string areas;
  for (size_t i = 0; i < forwardRecords.size(); i++) {
    const std::string folder = root + fileName + "/" + std::to_string(i) + "/";
    makeDirs(folder);
    writeObj(folder + "0-CLOTH.obj", forwardRecords[i].x, tris);
    areas += "Frame " + std::to_string(i) + ":" + fixedStr(meshArea(forwardRecords[i].x), 7) + "\n";
    exportFrameInfo(forwardRecords[i], folder + "info.txt");
  }
  writeText(root + "area.txt", areas);
}

void Simulation::exportSimulation(const std::string &fileName, const std::vector<ForwardInformation> &records) const {
  const std::string folder = outputFolder() + fileName + "/";
  makeDirs(folder);
  for (size_t i = 0; i < records.size(); i++) {
    writeObj(folder + std::to_string(i) + ".obj", records[i].x, tris);
    exportFrameInfo(records[i], folder + "info.txt");      // one file per folder: the last frame's clips remain (as in the reference)
  }
}

int Simulation::resetForwardRecordsFromFolder(const std::string &subFolder) {
  const std::string folder = outputFolder() + subFolder + "/";
  std::vector<std::pair<int, std::string>> frames;
  DIR *d = ::opendir(folder.c_str());
  if (!d) throw std::runtime_error("cannot open folder " + folder);
  while (dirent *ent = ::readdir(d)) {
    const std::string name = ent->d_name;
    if (name.size() <= 4 || name.substr(name.size() - 4) != ".obj") continue;
    const std::string stem = name.substr(0, name.size() - 4);
    if (stem.find_first_not_of("0123456789") != std::string::npos) continue;      // primitives ("1-SPHERE.obj") are not frames
    frames.push_back({std::stoi(stem), name});
  }
  ::closedir(d);
  std::sort(frames.begin(), frames.end());
  if (forwardRecords.empty()) resetSystem();
  const ForwardInformation init = forwardRecords[0];
  VecXd pts;
  std::vector<int> ftris;
  for (size_t i = 0; i < frames.size(); i++) {
    if (!loadObjFile(folder + frames[i].second, pts, ftris) || pts.size() < 3 * (size_t) N)
      throw std::runtime_error("frame " + folder + frames[i].second + " does not hold the " + std::to_string(N) + " cloth vertices");
    ForwardInformation rec = init;
    rec.t = sceneConfig.timeStep * (double) i;
    rec.x.assign(pts.begin(), pts.begin() + 3 * N);
    rec.stepIdx = (int) i;
    rec.x_fixedpoints.resize(3 * attachmentVertices.size());
    for (size_t k = 0; k < attachmentVertices.size(); k++)
      for (int c = 0; c < 3; c++) rec.x_fixedpoints[3 * k + c] = rec.x[3 * attachmentVertices[k] + c];
    forwardRecords.push_back(rec);
  }
  return (int) frames.size();
}

// ---------------------------------------------------------------------------------------------------------------
// text records of an optimisation run
// ---------------------------------------------------------------------------------------------------------------
std::string Simulation::taskInfoToString(const BackwardTaskInformation &task) {   // Simulation.cpp:4240-4281
  std::string out = "============Task Configuration:======================\n";
  out += "Forward Accuracy:" + fixedStr(task.forwardAccuracyLevel, 11) + "\n";
  out += "Backward Accuracy:" + fixedStr(task.backwardAccuracyLevel, 11) + "\n";
  out += "Rand seed:" + std::to_string(task.randSeed) + "\n";
  out += "Srand seed:" + std::to_string(task.srandSeed) + "\n";
  out += "Optimizer:LBFGS\n";
  auto on = [&](const std::string &what, bool flag) { if (flag) out += what + ": ON\n"; };
  for (int i = 0; i < 4; i++) on(kConstraintNames[i], task.dL_dk_pertype[i]);
  on("density", task.dL_density); on("f_ext", task.dL_dfext); on("f_wind", task.dL_dfwind); on("mu", task.dL_dmu);
  on("spline", task.dL_dcontrolPoints);
  if (task.dL_dmu) {
    out += "primitive mu id:\n";
    for (int id : task.mu_primitives) out += std::to_string(id) + ",";
    out += "\n";
  }
  return out;
}

std::string Simulation::parameterToString(const BackwardTaskInformation &task, const ParamInfo &p) {   // Simulation.cpp:4283-4352
  std::string out = "============Parameter Info:======================\n";
  for (int i = 0; i < 4; i++)
    if (task.dL_dk_pertype[i]) out += std::string("k_") + kConstraintNames[i] + ":" + fixedStr(p.k_pertype[i], 6) + "\n";
  if (task.dL_density) out += "density:" + fixedStr(p.density, 6) + "\n";
  if (task.dL_dmu)
    for (const auto &pm : p.mu) out += "mu from prim" + std::to_string(pm.first) + ":" + fixedStr(pm.second, 6) + "\n";
  if (task.dL_dx0 && !p.x0.empty()) {
    double n2 = 0;
    for (double v : p.x0) n2 += v * v;
    out += "x0: norm:" + fixedStr(std::sqrt(n2), 5) + "\n";
    for (size_t i = 0; i < p.x0.size() / 3; i += 200)
      out += "pId-" + std::to_string(i) + tupleStr(std::array<double, 3>{p.x0[3 * i], p.x0[3 * i + 1], p.x0[3 * i + 2]}, 5) + "\n";
  }
  if (task.dL_dfext && p.f_ext.size() == 3) out += "f_ext:" + tupleStr(p.f_ext, 6) + "\n";
  if (task.dL_dfwind) out += "f_wind:" + tupleStr(p.f_extwind, 6) + "\n";
  if (task.dL_dcontrolPoints)
    for (size_t set = 0; set < p.controlPointSplines.size(); set++) {
      out += "====set" + std::to_string(set) + "=====\n";
      for (const Spline &sp : p.controlPointSplines[set])
        for (const Spline::Segment &s : sp.segments) {
          out += "t[" + fixedStr(s.startFraction) + "," + fixedStr(s.endFraction) + "]:p1:" + tupleStr(s.p1, 6);
          if (sp.type == Spline::ENDPOINT_AND_UP) out += ",up:" + fixedStr(s.yUp, 4);
          if (sp.type == Spline::ENDPOINT_AND_TANGENTS) out += ",m0:" + tupleStr(s.m0, 4) + ",m1:" + tupleStr(s.m1, 4);
          out += "\n";
        }
    }
  return out;
}

std::string Simulation::forwardInfoToString(const BackwardTaskInformation &, const ForwardInformation &f) {   // Simulation.cpp:4452-4475
  std::string out = "============Forward Info:======================\n";
  out += "Total PD Iters:" + std::to_string(f.cumulateIter) + "\n";
  out += "Total Frames Converged:" + std::to_string(f.totalConverged) + "\n";
  out += "Forward Total Runtime[ms]:" + fixedStr(f.totalRuntime / 1000.0, 5) + "\n";
  out += "Loss:" + fixedStr(f.loss, 5) + "\n";
  return out;
}

std::string Simulation::backwrdInfoAndGradToString(const BackwardTaskInformation &task, const BackwardInformation &g) {   // (sic) :4354-4450
  std::string out = "============Backward Iter Info:======================\n";
  out += "Loss:" + fixedStr(g.loss, 3) + "\n";
  out += "Number of Converged Iter:" + std::to_string(g.convergedAccum) + "\n";
  out += "Total Backward Iter:" + std::to_string(g.backwardTotalIters) + "\n";
  out += "Backward Total Runtime[ms]:" + fixedStr(g.totalRuntime / 1000.0, 7) + "\n";
  out += "\n============Gradient Info:======================\n";
  for (int i = 0; i < 4; i++)
    if (task.dL_dk_pertype[i]) out += std::string("dL/dk_") + kConstraintNames[i] + ":" + fixedStr(g.dL_dk_pertype[i], 5) + "\n";
  if (task.dL_density) out += "dL/ddensity:" + fixedStr(g.dL_ddensity, 5) + "\n";
  if (task.dL_dmu)
    for (const auto &pm : g.dL_dmu) out += "dL_dmu_" + std::to_string(pm.first) + ":" + fixedStr(pm.second, 5) + "\n";
  if (task.dL_dfext) out += "dL/df_ext:" + tupleStr(g.dL_dfext, 4) + "\n";
  if (task.dL_dfwind) out += "dL/df_wind:" + tupleStr(g.dL_dwind, 4) + "\n";
  if (task.dL_dcontrolPoints)
    for (size_t set = 0; set < g.dL_dsplines.size(); set++)
      for (size_t k = 0; k < g.dL_dsplines[set].size(); k++)
        out += "dL/dspline_" + std::to_string(set) + "_" + std::to_string(k) + ":" + tupleStr(g.dL_dsplines[set][k], 4) + "\n";
  return out;
}

void Simulation::exportStatistics(int demoIdx, TaskSolveStatistics &st, const BackwardTaskInformation &task, bool writePerf) {
  if (!st.configWritten && st.experimentName.empty()) {
    char stamp[32];
    std::time_t now = std::time(nullptr);
    std::strftime(stamp, sizeof stamp, "%Y%m%d-%H%M%S", std::localtime(&now));
    st.experimentName = sceneConfig.name + "-randseed-" + std::to_string(task.srandSeed) + "-" + stamp + "-forwardThresh-" +
                        fixedStr(std::log10(task.forwardAccuracyLevel), 1);
  }
  const std::string rel = st.experimentName + "-LBFGS/";
  const std::string parent = outputFolder() + rel;
  makeDirs(parent);
  writeText(parent + "iters.txt", "Total forward:" + std::to_string(st.totalForwardSim) + "\nTotal backprop:" + std::to_string(st.totalBackprop));
  if (!st.configWritten) {
    writeText(parent + "task_info.txt", taskInfoToString(task));
    std::string cfg = "demoName:" + sceneConfig.name + " \n" + "demoIdx:" + std::to_string(demoIdx) + " \n";
    cfg += "FPS:" + std::to_string((int) std::lround(1.0 / sceneConfig.timeStep)) + "\n";
    cfg += "Frame Number:" + std::to_string(sceneConfig.stepNum + 1) + "\n";
    cfg += std::string("Collision:") + (contactEnabled ? "ON" : "OFF") + "\n";
    cfg += std::string("Self-Collision:") + (selfcollisionEnabled ? "ON" : "OFF") + "\n";
    cfg += std::string("Wind:") + (windEnabled ? "ON" : "OFF") + "\n";
    cfg += std::string("WindMode:") + kWindNames[sceneConfig.windConfig] + "\n";
    cfg += "Object Number:" + std::to_string(primitives.size() + 1 + attachmentVertices.size()) + "\n";
    writeText(parent + "scene-config.txt", cfg);
    st.configWritten = true;
  }
  const int prevForward = st.forwardWritten;
  for (size_t i = st.forwardWritten; i < st.completeForwardLog.size(); i++)
    writeText(parent + "forwardLog.txt", "Record " + std::to_string(i) + "\n" + forwardInfoToString(task, st.completeForwardLog[i].second) +
                                             parameterToString(task, st.completeForwardLog[i].first), true);
  st.forwardWritten = (int) st.completeForwardLog.size();
  for (size_t i = st.backwardWritten; i < st.completeBackwardLog.size(); i++)
    writeText(parent + "backwardLog.txt", "Record " + std::to_string(i) + "\n" + backwrdInfoAndGradToString(task, st.completeBackwardLog[i].second) +
                                              parameterToString(task, st.completeBackwardLog[i].first) + "Corresponding forward Idx: " +
                                              std::to_string(st.completeBackwardLog[i].second.correspondingForwardIdxInStats) + "\n", true);
  st.backwardWritten = (int) st.completeBackwardLog.size();
  for (size_t i = st.optimizationRecordsSaved; i < backwardOptimizationRecords.size(); i++) {
    const std::string iter = rel + "iter" + std::to_string(i);
    exportSimulation(iter, backwardOptimizationRecords[i].first);
    writeText(outputFolder() + iter + "/param.txt", parameterToString(task, backwardOptimizationGuesses[i].first));
  }
  st.optimizationRecordsSaved = (int) backwardOptimizationRecords.size();
  makeDirs(parent + "last_frame_meshes/");
  for (size_t i = prevForward; i < st.completeForwardLog.size(); i++) {
    writeObj(parent + "last_frame_meshes/0-CLOTH_forwardlastframe_iter_" + std::to_string(i) + ".obj", st.completeForwardLog[i].second.x, tris);
    exportFrameInfo(st.completeForwardLog[i].second, parent + "last_frame_meshes/" + std::to_string(i) + "_info.txt");
  }
  if (writePerf) {
    long long fwd = 0, bwd = 0;
    std::string out = "Demo Name:" + sceneConfig.name + "\n";
    out += std::string("Collision:") + (contactEnabled ? "ON" : "OFF") + "\n";
    out += std::string("Self-Collision:") + (selfcollisionEnabled ? "ON" : "OFF") + "\n";
    out += std::string("Wind:") + (windEnabled ? "ON" : "OFF") + "\n";
    out += std::string("BackwardSolver:") + (backwardGradientForceDirectSolver ? "Jacobi" : "Direct") + "\n";   // (labels as the reference prints them)
    out += "Total Particles:" + std::to_string(N) + "\n";
    out += "Fabric Name:" + sceneConfig.fabric.name + "\n";
    out += "======Backward iters =====\n";
    for (const auto &b : st.completeBackwardLog) {
      out += "Total Jacobi iters:" + std::to_string(b.second.backwardTotalIters) + "\n";
      out += "Jacobi total converged:" + std::to_string(b.second.convergedAccum) + "\n";
      out += "Total backward props:" + std::to_string(sceneConfig.stepNum) + "\n";
    }
    out += "======Forward Runtime (unit, [s])=====\n";
    for (size_t i = 0; i < st.completeForwardLog.size(); i++) {
      fwd += st.completeForwardLog[i].second.totalRuntime;
      out += "iter" + std::to_string(i) + ":" + fixedStr(st.completeForwardLog[i].second.totalRuntime / 1e6, 8) + "\n";
    }
    out += "======Backward Runtime (unit, [s])=====\n";
    for (size_t i = 0; i < st.completeBackwardLog.size(); i++) {
      bwd += st.completeBackwardLog[i].second.totalRuntime;
      out += "iter" + std::to_string(i) + ":" + fixedStr(st.completeBackwardLog[i].second.totalRuntime / 1e6, 8) + "\n";
    }
    out += "Total Forward Time:" + fixedStr(fwd / 1e6, 6) + "\n";
    out += "Total Backward Time:" + fixedStr(bwd / 1e6, 6) + "\n";
    out += "Total Time:" + fixedStr((fwd + bwd) / 1e6, 6) + "\n";
    writeText(parent + "perf.txt", out);
  }
}

}  // namespace dchost
