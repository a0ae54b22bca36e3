// Prints the face table of diffcloth_amd/csrc/dc_spheremesh.cpp (binary, 12 doubles per face) for tests/test_host_native.py, which
// compares it with the fp64 oracle's literal restatement of Sphere::Sphere (oracle/orc_sim.cpp buildSphereMesh).
#include "dc_spheremesh.h"
#include <cstdio>
#include <cstdlib>
int main(int argc, char **argv) {
  const double radius = argc > 1 ? std::atof(argv[1]) : 15.0;
  const int res = argc > 2 ? std::atoi(argv[2]) : 40;
  const std::vector<double> t = dc::sphere_mesh_table(radius, res);
  std::fwrite(t.data(), sizeof(double), t.size(), stdout);
  return 0;
}
