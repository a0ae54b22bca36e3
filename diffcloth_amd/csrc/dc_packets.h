// Host-side builder of the packet-ELL copy of the system matrix used by the resident PCG (dc_forward_pk.hip):
// P symmetrically scaled to unit diagonal (D^-1/2 P D^-1/2, so that plain CG on it is Jacobi-preconditioned CG on P),
// off-diagonals packed three to a 16-byte packet {v0, v1, v2, d0 | d1 << 10 | d2 << 20} with d = column - row + 512,
// wave-sliced: chunk c = rows 64c .. 64c+63, packet (s, lane) at pk[4 * (pk_ptr[c] + 64 s + lane)], pk_n[c] packets per
// row (a multiple of 4). Padding packets are {0, 0, 0, 512 | 512 << 10 | 512 << 20} (value 0 on the row itself).
#pragma once
#include <vector>
#include "dc_system.h"

namespace dc {

struct HostPackets {
  bool ok = false;
  int vpt = 0;                    // rows per thread of the kernel the tables are padded for (threads * vpt rows)
  int threads = 512;              // threads of that kernel: 512, or 768 for the largest meshes (3 waves per SIMD, dc_forward_pk.hip)
  int bandwidth = 0;              // max |column - row| of P
  std::vector<int> pk;            // 4 ints per packet
  std::vector<int> pk_ptr, pk_n;  // per 64-row chunk
  std::vector<float> sq_dinv;     // [512 * vpt] sqrt(1 / P_ii), 0 for padding rows

  // false when the tables cannot be used: N > 10 240 rows or bandwidth > 511
  bool build(const HostSystem &H);
  // the same tables padded to `rows_padded` rows (a multiple of 64, >= N), any N; false when the bandwidth exceeds 511
  bool build_rows(const HostSystem &H, int rows_padded);
};

}  // namespace dc
