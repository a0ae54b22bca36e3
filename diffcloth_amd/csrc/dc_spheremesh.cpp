// Face table of the reference's sphere mesh (see dc_spheremesh.h). Written from the structure the constructor's loops produce, not as a
// transcription of them: vertices are the north pole N, rows 1 .. res - 1 of res vertices (row y at polar angle 180 y / res degrees, column x
// at azimuth 360 x / res degrees; Primitive.h:144-149), the south pole S. With V(y, x) the vertex of row y, column x, the faces come in
// this order (a face made from (a, b, c) is stored with corners p0 = c, p1 = b, p2 = a, Primitive.cpp:148-149):
//   row 1:   for x = 1 .. res - 1: (V(1,x), V(1,x-1), N);  after the last of them the closing cap face (V(1,0), V(1,res-1), N)
//   row y>1: for x = 1 .. res - 1: (V(y,x), V(y,x-1), V(y-1,x)), (V(y,x-1), V(y-1,x-1), V(y-1,x));
//            then the seam quad of the row: (V(y-1,0), V(y,0), V(y,res-1)), (V(y,res-1), V(y-1,res-1), V(y-1,0))
//   south:   (S, V(res-1,res-1), V(res-1,0)), then for x = 1 .. res - 1: (S, V(res-1,x-1), V(res-1,x))
#include "dc_spheremesh.h"
#include <cmath>

namespace dc {

std::vector<double> sphere_mesh_table(double radius, int res) {
  std::vector<double> out;
  if (res < 3) return out;
  const double deg = 0.01745329251994329576923690768489;       // glm::radians
  struct P3 { double x, y, z; };
  auto at = [&](double phi, double theta) {
    return P3{radius * std::cos(phi * deg) * std::sin(theta * deg), radius * std::sin(phi * deg) * std::sin(theta * deg), radius * std::cos(theta * deg)};
  };
  const double dphi = 360.0 / res, dtheta = 180.0 / res;
  const P3 north = at(0, 0), south = at(0, 180);
  auto V = [&](int y, int x) { return at(dphi * x, dtheta * y); };
  auto face = [&](const P3 &a, const P3 &b, const P3 &c) {
    const P3 p0 = c, p1 = b, p2 = a;
    const double ux = p1.x - p0.x, uy = p1.y - p0.y, uz = p1.z - p0.z, vx = p2.x - p0.x, vy = p2.y - p0.y, vz = p2.z - p0.z;
    double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
    const double len = std::sqrt(nx * nx + ny * ny + nz * nz);
    if (len > 0) { nx /= len; ny /= len; nz /= len; }
    const double row[12] = {p0.x, p0.y, p0.z, p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, nx, ny, nz};
    out.insert(out.end(), row, row + 12);
  };
  for (int x = 1; x < res; x++) face(V(1, x), V(1, x - 1), north);
  face(V(1, 0), V(1, res - 1), north);
  for (int y = 2; y < res; y++) {
    for (int x = 1; x < res; x++) {
      face(V(y, x), V(y, x - 1), V(y - 1, x));
      face(V(y, x - 1), V(y - 1, x - 1), V(y - 1, x));
    }
    face(V(y - 1, 0), V(y, 0), V(y, res - 1));
    face(V(y, res - 1), V(y - 1, res - 1), V(y - 1, 0));
  }
  face(south, V(res - 1, res - 1), V(res - 1, 0));
  for (int x = 1; x < res; x++) face(south, V(res - 1, x - 1), V(res - 1, x));
  return out;
}

}  // namespace dc
