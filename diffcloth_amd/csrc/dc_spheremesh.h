// Host side of DC_PRIM_SPHERE_DISCRETIZED: the latitude / longitude mesh the reference's Sphere carries for rendering (Sphere::Sphere,
// Primitive.cpp:133-216) and, when `discretized` is set, for its contact normals (Sphere::isInContact, Primitive.cpp:230-253).
#pragma once
#include <vector>

namespace dc {

// [ntri][12] doubles: corners p0, p1, p2 and the unit face normal (p1 - p0) x (p2 - p0), triangles in the reference's creation order
// (the contact code keeps the LAST face that qualifies). Sphere of `radius` around the origin, res x res (the reference: 40).
std::vector<double> sphere_mesh_table(double radius, int res);

}  // namespace dc
