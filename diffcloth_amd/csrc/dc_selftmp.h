// Size of the per-rollout scratch of the self-contact layering (DevWork::sd_tmp; layout: SelfTmp in dc_selflib.h).
#pragma once
namespace dc {
// in ints; U = 2 cap + 2 entries per per-vertex array (a contact list of C pairs has at most 2 C distinct vertices)
__host__ __device__ constexpr int self_tmp_ints(int cap) { return 10 * (2 * cap + 2) + 4 * cap + 16; }
constexpr int kSelfCells = 4096;               // bins of the 2-D broad-phase grid of the detection
constexpr int kSelfDetectLdsInts = 16 + (kSelfCells + 1) + kSelfCells + 1 + 2048;   // LDS ints self_detect_rollout needs
}  // namespace dc
