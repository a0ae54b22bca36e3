// TEST INFRASTRUCTURE ONLY — fp64 CPU oracle for the DiffCloth hot path (see oracle/README.md).
// Nothing under diffcloth_amd/ may include, link or call this code.
//
// Restates, in dependency-free C++17 (no Eigen), the algorithm of
//   Simulation::step()          /root/reference/src/code/simulation/Simulation.cpp:1043-1428
//   Simulation::stepBackward()  Simulation.cpp:1455-1780
// and their callees. Every function cites the reference lines it follows.
// Parity status: pinned against output/tshirt-exampleopt — frames 0..40 of iter0 and the logged loss of evaluation 0
// (tests/test_golden_tshirt.py) — and validated by
// finite differences + an independent NumPy/SciPy single-step cross-check (tests/test_oracle_*.py).
#pragma once
#include "orc_math.h"
#include <array>
#include <map>
#include <set>
#include <string>

namespace orc {

typedef std::array<double, 9> M3;  // row-major 3x3

enum CollisionType { TAKE_OFF = 0, STICK = 1, SLIDE = 2 };
enum PrimKind { PRIM_SPHERE = 0, PRIM_CAPSULE = 1, PRIM_LOWER_LEG = 2, PRIM_PLANE = 3, PRIM_BOWL = 4 };

struct TriRest {          // Triangle.cpp:587-645 (ctor), Triangle.h:173-175 (weight)
  int v[3];
  double D[4];            // inv_deltaUV, row-major 2x2
  double area;
  double w;               // constrainWeightSqrt = sqrt(area_rest * k_stretch)
};
struct BendRest {         // TriangleBending.cpp:186-239, TriangleBending.h:40-42
  int v[4];
  double wv[4];           // weightVert (cotan weights)
  double n, A0, A1;
  double w;               // sqrt(k_bend * 3 / (A0 + A1))
};
struct Primitive {        // Primitive.cpp (isInContact family)
  int kind = PRIM_SPHERE;
  V3 center;              // Primitive::center (for a capsule: centre of its bottom cap)
  V3 centerInit;          // child offset inside a LowerLeg (Primitive.cpp:410-418)
  double radius = 1, mu = 0, length = 0;
  bool rotates = false;   // Sphere::rotates (Primitive.cpp:255-257)
  V3 topOffset;           // capsule: globalRotation * (0, length, 0)   (Primitive.cpp:582)
  V3 upperLeft, upperRight;   // plane: two corners relative to the centre; the others are their negatives (Primitive.cpp:13-21)
  V3 velocity;
  std::vector<Primitive> children;  // LowerLeg: joint sphere, foot capsule, leg capsule
  // Sphere::discretized (Primitive.h:222; set for the BIG_SPHERE scene, Simulation.cpp:1910): the contact normal is a face normal of the
  // sphere's own render mesh (Primitive.cpp:230-253). `mesh`: its triangles in creation order (buildSphereMesh).
  bool discretized = false;
  struct Tri { V3 p0, p1, p2, normal; };
  std::vector<Tri> mesh;
};
// Sphere::Sphere (Primitive.cpp:133-216): the latitude / longitude mesh of a sphere of `radius` around the origin, `resolution` x `resolution`.
std::vector<Primitive::Tri> buildSphereMesh(double radius, int resolution);
struct PrimContact {      // Simulation.h:39-51
  int primitiveId = -1, particleId = -1;
  V3 normal, v_out, d, r;
  double dist = 0;
  int type = TAKE_OFF;
};
struct SelfContact {      // Simulation.h:53-63
  int particleId1 = -1, particleId2 = -1;
  V3 normal, d, r;
  int layerId = -1;
  int type = TAKE_OFF;
};
struct Record {           // ForwardInformation, Simulation.h:68-100 (hot-path fields)
  std::vector<double> x, v, x_prev, v_prev, f, r, s_n, x_fixed;
  std::vector<PrimContact> prim;
  std::vector<std::vector<SelfContact>> layers;
  std::vector<double> Atp_weightless[3];  // At_p_weightless_pertype for {tri, bend, att}
  bool converged = false;
  int convergeIter = 0;
  double windFactor = 0, t = 0;
};
struct BackwardOut {      // BackwardInformation, Simulation.h:136-162 (hot-path fields)
  std::vector<double> dL_dx, dL_dv, dL_dxfixed, dL_dmu;
  double dL_dk[3] = {0, 0, 0};   // stretch, bend, attachment
  double dL_ddensity = 0;
  double dL_dwind[5] = {0, 0, 0, 0, 0};
  std::vector<double> dL_dfext_vec;   // h^2 (I + dr_df)^T u* per vertex (Sim.cpp:1700-1760)
  double dL_dwindtimestep = 0;       // this step's entry (Sim.cpp:1720-1729)
  bool converged = false;
  int backwardIters = 0;
  bool usedDirect = false;
  double directResidual = -1;        // relative residual |g - K u| / |g| the direct solve ended with (-1: it did not run)
};

struct Params {
  double h = 1.0 / 90;
  double density = 0.1, k_stretch = 100, k_bend = 0.01, k_att = 10000;   // AttachmentSpring.cpp:10
  V3 gravity = V3(0, -9.8, 0);                                           // Simulation.h:356
  double fwd_tol = 1e-7, bwd_tol = 5e-5;                                 // Simulation.cpp:17-19
  bool gravityEnabled = true, contactEnabled = true, selfcollisionEnabled = true, windEnabled = false;
  bool gradientClipping = true;                                          // Simulation.h:330-331
  double gradientClippingThreshold = 16.0;
  int windConfig = 0;                   // 0 NO_WIND, 1 WIND_CONSTANT, 2 WIND_SIN, 3 WIND_SIN_AND_FALLOFF, 4 WIND_FACTOR_PER_STEP (engine/Constants.h:55-61)
  double perStepWindFactor = 1.0;       // perstepWindFactor[step] of the step being taken (Sim.cpp:80-82)
  bool enableConstantForcefield = false;
  V3 wind = V3(0.01, 0, 1);             // Simulation.h:357
  double windNorm = 0.15, windFrequency = 14, windPhase = 0;   // Simulation.cpp:20-22
  bool calcSeparateAtp = false;         // calcualteSeperateAt_p
  int pd_iter_cap = -1;                 // <0: reference formula (-log10(tol))*150 (Simulation.cpp:1182)
  int threads = 1;
};

struct Sim {
  Params P;
  std::vector<double> given_u;   // diagnostic (orc_set_given_u): when it has 3N entries, a forced direct solve of stepBackward takes THIS solution instead of
                                 // running GMRES — tests solve near-singular adjoint systems with a sparse LU (scipy) as the reference's SparseLU does
  int N = 0;
  std::vector<double> windFallOff;          // 3N (Simulation.h:349), empty = ones
  std::vector<double> external_force_field; // 3N (Simulation.h:418)
  std::vector<double> rest;                 // 3N rest positions (pos_rest)
  std::vector<std::array<int, 3>> tris_in;
  std::vector<TriRest> tris;
  std::vector<BendRest> bends;
  std::vector<int> att;                     // attachment vertex per fixed point
  std::vector<double> area, mass, radii;    // per vertex
  std::vector<std::set<int>> connected;     // pointpointConnectionTable (share a triangle)
  std::vector<Primitive> prims;

  // scalar constraint rows: A = A_s (x) I3.  Row order: triangles (2 rows each), bends, attachments
  struct Row { int nv; int v[4]; double c[4]; int type; };   // coefficients include the constraint weight
  std::vector<Row> rows;
  // scalar P = M + h^2 A_s^T A_s  in CSR, and C = h^2 A_s^T A_s (same pattern)
  std::vector<int> Pptr, Pcol;
  std::vector<double> Pval, Cval;
  // per-type weightless A^T A (A_t_times_A_pertype, Simulation.cpp:3011) in the same pattern
  std::vector<double> Lval[3];
  // RCM + skyline Cholesky of P (stands in for Eigen::SimplicialLLT, Simulation.h:379)
  std::vector<int> perm, iperm, skyFirst;
  std::vector<size_t> skyPtr;
  std::vector<double> skyL;

  std::vector<Record> records;

  void setMesh(int n, const double *pos, int T, const int *tri);
  void build();
  void solveP(const std::vector<double> &rhs, std::vector<double> &out) const;   // 3N interleaved
  void mulS(const std::vector<double> &val, const std::vector<double> &x, std::vector<double> &y) const;

  // local physics
  void triProject(const TriRest &t, const double *x, double out[6]) const;
  Mat triProjectBackward(const TriRest &t, const double *x) const;   // 6x9, unweighted
  void bendProject(const BendRest &b, const double *x, double out[3]) const;
  Mat bendBackward(const BendRest &b, const double *x) const;        // 3x12, weighted

  // contact
  bool primInContact(const Primitive &p, const V3 &center_prim, const V3 &pos, const V3 &vel, V3 &normal,
                     double &dist, V3 &v_out) const;
  PrimContact isInContactWithObstacle(const V3 &pos, const V3 &v_in) const;
  bool isSelfCollision(int a, int b, const V3 &xa, const V3 &xb, const V3 &va, const V3 &vb, SelfContact &out) const;
  void collisionDetection(const std::vector<double> &x_n, const std::vector<double> &v, const V3 &particle0_pos,
                          std::vector<PrimContact> &prim, std::vector<std::vector<SelfContact>> &layers) const;
  static std::vector<std::vector<SelfContact>> contactSorting(const std::vector<PrimContact> &prim,
                                                              std::vector<SelfContact> &self);
  static V3 dryFrictionForce(const V3 &n, const V3 &f_i, double mu, int &type);
  static V3 dri_dmu(const V3 &n, const V3 &f_i, double mu);
  static M3 dri_dfi(const V3 &n, const V3 &f_i, double mu);
  void dryFrictionVector(const std::vector<double> &f, std::vector<PrimContact> &prim,
                         std::vector<std::vector<SelfContact>> &layers, std::vector<double> &r) const;

  // the hot path
  double fillForces(std::vector<double> &f_ext, double t_now) const;
  int step(const double *x_n, const double *v_n, const double *x_fixed, double t_prev, int frozenContactsFrom = -1);
  BackwardOut stepBackward(const Record &rec, const double *dL_dxnew, const double *dL_dvnew,
                           const double *dL_dxinit, const double *dL_dvinit, bool isStart, bool forceDirect,
                           int numMu) const;
  // diagnostic: K = P - dP^T of the direct adjoint solve of a record as CSC (tests/proto_adjoint.py)
  void adjointMatrix(const Record &rec, std::vector<int> &colptr, std::vector<int> &rowidx, std::vector<double> &val, double drop) const;
};

}  // namespace orc
