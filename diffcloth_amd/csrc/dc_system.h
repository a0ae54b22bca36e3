// Host-side constraint-system builder of the MI355X DiffCloth stepper.
//
// Turns a raw triangle mesh + fabric parameters into the flat SoA tables the HIP kernels consume.
// Reference behaviour reproduced (paths relative to /root/reference/src/code/simulation/):
//   triangle rest data          Triangle.cpp:587-645, weight Triangle.h:173-175
//   bending flaps + cotan w.    Simulation.cpp:2096-2131, TriangleBending.cpp:186-239, weight TriangleBending.h:40-42
//   lumped areas / masses       Simulation.cpp:2894-2966
//   collision radii             Simulation.cpp:2407-2431
//   A rows, P = M + h^2 A^T A   Simulation.cpp:2969-3059 (+ Triangle.cpp:296-304, TriangleBending.cpp:20-24,
//                               AttachmentSpring.cpp:61-63)
// Every constraint row has the same coefficient on x, y and z, so A = A_s (x) I3 and only the scalar N x N
// matrix P_s is assembled.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace dc {

struct HostSystem {
  // ---- input ----
  int N = 0, T = 0;
  std::vector<double> rest;      // 3N
  std::vector<int> tri;          // 3T
  std::vector<int> att_vertex;   // Af

  // ---- derived topology (independent of stiffness / density / h) ----
  int E = 0;
  std::vector<double> tri_D;     // 4T  inv_deltaUV row-major
  std::vector<double> tri_area;  // T
  std::vector<int> bend_v;       // 4E
  std::vector<double> bend_w;    // 4E  cotan weights
  std::vector<double> bend_n;    // E   rest norm of the weighted sum
  std::vector<double> bend_A;    // E   A0 + A1
  std::vector<double> area;      // N   lumped vertex area
  std::vector<double> radii;     // N
  // vertex -> incident constraint corners, CSR; entries index the corner-output array of the kernels:
  //   triangle t corner k -> k*T + t            (k in 0..2)
  //   bend e corner k     -> 3T + k*E + e       (k in 0..3)
  std::vector<int> inc_ptr, inc_idx;
  // pairs of vertices sharing a triangle (pointpointConnectionTable), CSR, sorted — self-collision exclusion
  std::vector<int> conn_ptr, conn_idx;

  // ---- derived numerics (depend on parameters) ----
  std::vector<double> mass;      // N
  std::vector<double> tri_w2;    // T   area * k_stretch          (= constrainWeightSqrt^2)
  std::vector<double> bend_w2;   // E   k_bend * 3 / (A0 + A1)
  std::vector<int> P_ptr, P_col; // scalar CSR, columns sorted
  std::vector<double> P_val;

  std::string error;

  bool set_mesh(int n, const double *pos, int t, const int *tris);
  bool build_numerics(double h, double density, double k_stretch, double k_bend, double k_att);
  int rows() const { return 6 * T + 3 * E + 3 * (int) att_vertex.size(); }
};

// Reverse Cuthill-McKee ordering of the vertex graph of a triangle mesh: order[new] = old. Used to renumber the
// vertices on the device when the caller's numbering has a large bandwidth (the packet-ELL matrix format and the
// element windows need |i - j| of coupled vertices to be small).
std::vector<int> rcm_order(int n, int t, const int *tris);
// max |i - j| over the edges of the mesh
int mesh_bandwidth(int t, const int *tris);

}  // namespace dc
