// Fabric / scene tables of the shipped demos. Values restated from the reference's
// optimization/OptimizationTaskConfigurations.cpp:10-349 (fabric and scene structs); only the fields that reach
// the hot path are kept.
#include "simulation.h"
#include <stdexcept>

namespace dchost {

namespace {
FabricConfiguration fabric(double dim, double ks, double kb, int gx, int gy, double rho, bool model, const char *name) {
  FabricConfiguration f;
  f.clothDimX = f.clothDimY = dim; f.k_stiff_stretching = ks; f.k_stiff_bending = kb;
  f.gridNumX = gx; f.gridNumY = gy; f.density = rho; f.isModel = model; f.name = name;
  return f;
}
}  // namespace

SceneConfiguration sceneByName(const std::string &name) {
  SceneConfiguration s;
  s.name = name;
  if (name == "wear_hat" || name == "demo_wearhat") {            // hatScene :166-183, agenthat579 :132-146
    s.fabric = fabric(6, 1200, 120, 40, 80, 0.224, true, "remeshed/agenthat2-579-rotated.obj");
    s.orientation = FRONT; s.attachmentPoints = CUSTOM_ARRAY; s.customAttachmentVertexIdx = {{0.0, {394, 32}}};
    s.trajectory = CORNERS_2_WEARHAT; s.primitiveConfig = PLANE_BUST_WEARHAT; s.windConfig = NO_WIND;
    s.timeStep = 1.0 / 100.0; s.stepNum = 400; s.forwardConvergenceThresh = 1e-8; s.backwardConvergenceThresh = 5e-4;
  } else if (name == "wear_sock" || name == "wear_sock1") {      // sockScene :331-349, sock482 :148-163
    s.fabric = fabric(5, 600, 1, 40, 80, 0.224, true, "remeshed/sock1055-2081.obj");
    s.orientation = CUSTOM_ORIENTATION; s.upVector = {0, 1, 0}; s.attachmentPoints = CUSTOM_ARRAY;
    s.customAttachmentVertexIdx = {{0.0, {14, 30, 3, 81}}};
    s.trajectory = CORNERS_2_WEARSOCK; s.primitiveConfig = FOOT; s.windConfig = NO_WIND; s.sockLegOrientation = {0, 1, 0};
    s.timeStep = 1.0 / 160.0; s.stepNum = 400; s.forwardConvergenceThresh = 1e-9; s.backwardConvergenceThresh = 5e-4;
  } else if (name == "sphere" || name == "rotating_sphere") {    // rotatingSphereScene :228-244, sphereFabric :81-96
    s.fabric = fabric(4.5, 150, 0.00001, 25, 25, 0.3, false, "sphereFabric");
    s.orientation = DOWN; s.attachmentPoints = NO_ATTACHMENTS; s.trajectory = NO_TRAJECTORY;
    s.primitiveConfig = PLANE_AND_SPHERE; s.windConfig = NO_WIND;
    s.timeStep = 1.0 / 180.0; s.stepNum = 350; s.forwardConvergenceThresh = 1e-9; s.backwardConvergenceThresh = 5e-4;
  } else if (name == "wind_tshirt" || name == "tshirt") {        // tshirtScene :265-283, tshirt1000 :65-79
    s.fabric = fabric(6, 550, 0.01, 40, 80, 0.124, true, "remeshed/T-shirt/tshirt1000-tri.obj");
    s.orientation = BACK; s.attachmentPoints = LEFT_RIGHT_CORNERS_2; s.trajectory = NO_TRAJECTORY;
    s.primitiveConfig = PRIM_NONE; s.windConfig = WIND_SIN;
    s.timeStep = 1.0 / 90.0; s.stepNum = 250; s.forwardConvergenceThresh = 1e-8; s.backwardConvergenceThresh = 5e-4;
  } else if (name == "dress_twirl" || name == "inverse_design") { // dressScene :285-309, dressv7khandsUpDrape :115-129
    s.fabric = fabric(13, 3000, 0.3, 40, 80, 0.3, true, "remeshed/dress-handsup-drape.obj");
    s.orientation = FRONT; s.attachmentPoints = CUSTOM_ARRAY;
    s.customAttachmentVertexIdx = {{0.0, {1335, 1336, 1334, 1360, 1339, 1347, 1345, 1342, 1349, 1351, 1352, 3604, 1145, 1150, 1137, 1142,
                                          1143, 1285, 3496, 3497, 3501, 1152, 1153, 3499, 3498, 3500, 3559, 1146, 1333, 1355, 1350}}};
    s.trajectory = TRAJECTORY_DRESS_TWIRL; s.primitiveConfig = PRIM_NONE; s.windConfig = NO_WIND;
    s.timeStep = 1.0 / 120.0; s.stepNum = 125; s.forwardConvergenceThresh = 1e-10; s.backwardConvergenceThresh = 5e-4;
  } else {
    throw std::runtime_error("Undefined example name (" + name + ").");   // python_interface.cpp:86
  }
  return s;
}

}  // namespace dchost
