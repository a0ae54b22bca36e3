// Host-side builder of the "element window" tables: the per-constraint work of a PD step (local projections,
// reference Triangle.cpp:310-351 / TriangleBending.cpp:138-151, and their derivatives in the adjoint,
// Triangle.cpp:354-451 / TriangleBending.cpp:154-172) is executed on the device window by window, entirely in LDS:
//
//   window w owns the vertices [v0, v1) (a multiple of 64, one or more 64-row chunks) and holds every triangle /
//   bending flap that touches an owned vertex; [lo, lo + vs) is the vertex span of those elements.
//     stage    the two input vectors restricted to the span -> LDS
//     phase A  one thread per element: gather its 3 / 4 vertices from LDS, compute, write the element's result
//              vectors (two per triangle: the columns of the 3x2 residual; one per flap) to LDS
//     phase B  one thread per owned vertex: sum coef * result-vector over its incident corners (a fixed order:
//              deterministic, no atomics) and hand the sum to the caller's per-vertex code
//   Elements that straddle two windows are evaluated once per window (a few % of redundant work) so that no
//   partial sums ever cross a window boundary.
//
// All tables are shared by the rollouts of a batch (they depend on topology and parameters only).
#pragma once
#include <cstddef>
#include <vector>
#include "dc_system.h"

namespace dc {

constexpr int kWinDumpSlots = 64;   // result slots behind a window's zero vector, one per lane: where masked elements store (dc_winlib.h)

struct HostWindows {
  bool ok = false;
  int own = 0;                    // owned vertices per window (multiple of 64)
  int nwin = 0;
  int vcap = 0;                   // max vertex span of a window
  int nrcap = 0;                  // max result vectors of a window (2 * triangles + flaps + 1 zero vector + kWinDumpSlots)
  size_t lds_bytes = 0;           // 4 * (6 * vcap + 3 * nrcap)
  std::vector<int> win;           // 8 ints per window: v0, v1, lo, vs, tri_off, ntri, bend_off, nbend
  std::vector<int> tri_rec;       // 4 ints per window-triangle: j0 | j1 << 16, j2, bits(area * k_stretch), global triangle id
  std::vector<float> tri_D;       // 4 floats per window-triangle: inv_deltaUV
  std::vector<int> bend_rec;      // 4 ints per window-flap: j0 | j1 << 16, j2 | j3 << 16, bits(rest norm), bits(weight^2)
  std::vector<float> bend_w;      // 4 floats per window-flap: cotan weights
  // low-order parts (value - fl32(value)) of the fp64 rest-shape data, for the precise record pass (dc_winlib.h: PreciseTriOp /
  // PreciseBendOp): tri_Dlo = those of inv_deltaUV; bend_lo = those of the cotan weights 1..3 and of the rest norm
  std::vector<float> tri_Dlo, bend_lo;
  // vertex -> (result vector, coefficient) pairs, wave-sliced by 64-vertex chunk: pair-packet (s, lane) at
  // chunk c of 64 owned vertices: inc_n[c] = nb4 | nt4 << 16; first nt4 packets of 8 triangle entries (16 bits: position << 1 | minus),
  // inc[inc_ptr[c] + 64 s + lane], s < nt4; then nb4 packets {q0, bits(w0), q1, bits(w1)} of flap pairs, s - nt4 < nb4
  std::vector<int> inc;
  std::vector<int> inc_ptr, inc_n;

  bool build(const HostSystem &H, size_t lds_budget);
  // the same tables for a given window size (`own_size` owned vertices per window, a multiple of 64); no LDS check
  bool build_own(const HostSystem &H, int own_size);
};

}  // namespace dc
