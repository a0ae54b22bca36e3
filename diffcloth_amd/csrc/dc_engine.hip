// C-ABI layer of libdiffcloth_hip.so: context, device memory, tape, and the calls that enqueue the
// persistent step kernels. See include/diffcloth_hip.h for the contract of every entry point.
#include <hip/hip_runtime.h>
// RCCL is bound with dlopen in dc_comm_* (no link-time and no header dependency): the handful of types and constants of its C API
// (rccl.h / nccl.h) the four entry points used here need, restated
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat64 = 8 } ncclDataType_t;      // ncclDouble
typedef enum { ncclSum = 0 } ncclRedOp_t;
#define NCCL_UNIQUE_ID_BYTES 128
}
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "dc_device.h"
#include "dc_system.h"
#include "dc_windows.h"
#include "dc_packets.h"
#include "dc_dense.h"
#include "dc_deflate.h"
#include "dc_spheremesh.h"
#include "dc_selftmp.h"
#include "dc_cluster.h"

using namespace dc;

namespace dc {
hipError_t launch_pd_step_cluster(const DevSystem &S, const DevCluster &CL, const DevWork &W, const FwdArgs &A, int b0, int nb, hipStream_t st);
hipError_t launch_adjoint_step_cluster(const DevSystem &S, const DevCluster &CL, const DevWork &W, const BwdArgs &A, int b0, int nb, hipStream_t st);
}

// Tables of the split kernels (dc_cluster.h) for one K, built when the batch size is known (dc_alloc_batch).
struct ClusterSet {
  bool ok = false;
  int K = 1, nb = 0;              // workgroups per rollout; rollouts per launch (K nb <= CUs)
  DevCluster D;
  std::vector<void *> allocs;
  size_t xch_bytes = 0;
};

struct dc_ctx {
  int device = 0;
  bool host_only = false;   // dc_create(-1): table building / inspection only, every compute call fails
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr; // the stream dc_create made (dc_use_stream may point `stream` at the caller's)
  std::string err;
  HostSystem host;
  dc_params params;
  std::vector<dc_primitive> prims;
  std::vector<int> group_of_prim;
  int ngroups = 0;
  bool mesh_set = false, built = false;
  // vertex renumbering on the device (empty = identity): user_of[device index] = caller's index, dev_of = inverse
  std::vector<int> user_of, dev_of;
  std::vector<int> att_user;        // attachment vertices in the caller's numbering
  const int *d_user_of = nullptr;   // device copy of user_of (null = identity)

  DevSystem S;
  std::vector<void *> table_allocs;

  int B = 0, tape = 0;
  DevWork W;
  std::vector<void *> batch_allocs;
  float *X = nullptr, *V = nullptr, *F = nullptr, *R = nullptr, *NRM = nullptr;   // [(tape+1)][B][3][N]
  int *PRIM = nullptr;                                                            // [(tape+1)][B][N]
  int2 *SC_pair = nullptr;          // [(tape+1)][B][cap] self contacts per record
  float4 *SC_nrm = nullptr, *SC_d = nullptr;
  int *SC_meta = nullptr;           // [(tape+1)][B][kMetaStride]
  int *SC_verts = nullptr;          // [(tape+1)][B][2 * cap] working-set vertex lists of the self contacts
  int self_cap = 0;
  float *xf_cur = nullptr;          // [B][3][Af]
  float *XF = nullptr;              // [(tape+1)][B][3][Af] fixed-point targets per record
  float *DPAR = nullptr;            // [(tape+1)][B][8] per-step parameter gradients
  float *mu = nullptr, *fu = nullptr;
  bool fu_set = false;
  float *fv = nullptr;              // [B][3][N] per-vertex extra force
  bool fv_set = false;
  float *fv2 = nullptr;             // [B][3][N] second per-vertex force term, factor 1 (dc_set_vertex_force_field)
  bool fv2_set = false;
  int start_slot = 0;              // dc_set_trajectory_start: the tape slot of the trajectory's initial state (-1: none in this tape)
  float *GX = nullptr, *GV = nullptr, *IX = nullptr, *IV = nullptr, *DMU = nullptr, *target = nullptr;
  float *DXF = nullptr;             // [(tape+1)][B][3][Af] dL_dxfixed of the step that produced the slot
  // device-resident schedules of the fused rollouts (dc_set_*_schedule); flags per tape slot
  float *FU_S = nullptr;            // [(tape+1)][B][3] uniform force of the step that produces the slot
  float *FVS_S = nullptr;           // [(tape+1)][B] factor on fv of that step
  float *SEEDX = nullptr, *SEEDV = nullptr;   // [(tape+1)][B][3][N] loss gradient w.r.t. the state at the slot (allocated on first use)
  std::vector<char> sched_xf, sched_fu, sched_fvs, sched_seed;
  void *comm = nullptr;             // RCCL communicator of dc_comm_init (ncclComm_t), one rank per context
  int comm_ranks = 0;
  std::vector<void *> sched_pool;
  // record handed in from outside (dc_set_record): fp64 values of x_new, f, primitive-contact normals [B][3][N], self-contact normals / d
  // [B][cap][3]; allocated on first use, valid for tape slot inj_slot only (-1 = none)
  float *YS = nullptr;              // [(tape+1)][B][3][N] y of every backward step (dc_keep_force_gradients), allocated on first use
  bool keep_y = false;
  double *INJ_X = nullptr, *INJ_F = nullptr, *INJ_N = nullptr, *INJ_SN = nullptr, *INJ_SD = nullptr;
  int inj_slot = -1;
  dc_step_stats *fstats = nullptr;  // [(tape+1)][B]
  dc_bwd_stats *bstats = nullptr;   // [(tape+1)][B], indexed by the slot whose record was differentiated
  double *stage[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t stage_elems = 0;
  hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
  float fwd_ms = 0, bwd_ms = 0;
  int fwd_launches = 0, bwd_launches = 0;
  ClusterSet cl;
  int cus = 0;                      // compute units of the device
  int bandwidth = 0;                // of the scalar system matrix in device numbering
  int defl_k = 0, defl_probe = 0;   // deflation space of the forward solve (dc_deflate.h)
  // the last deflation build of this context and what it was built from: a rebuild that leaves P unchanged (another tolerance, contact flags,
  // primitives ...) skips the probe solve and the eigen-solve (0.9 s on the 7 742-vertex dress)
  HostDeflation defl_cache;
  uint64_t defl_key = 0;
  bool defl_cache_valid = false, defl_cache_built = false;
};

namespace {

int fail(dc_ctx *c, int code, const std::string &msg) {
  if (c) c->err = msg;
  return code;
}
#define HIPCHK(c, call)                                                                         \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return fail(c, DC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));            \
  } while (0)

template <typename T>
int dev_alloc(dc_ctx *c, std::vector<void *> &pool, T **out, size_t count) {
  void *p = nullptr;
  size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  HIPCHK(c, hipMalloc(&p, bytes));
  HIPCHK(c, hipMemsetAsync(p, 0, bytes, c->stream));
  pool.push_back(p);
  *out = (T *) p;
  return DC_OK;
}
template <typename T, typename U>
int upload(dc_ctx *c, const T **out, const std::vector<U> &src) {
  std::vector<T> tmp(src.begin(), src.end());
  T *p = nullptr;
  int rc = dev_alloc(c, c->table_allocs, &p, tmp.size());
  if (rc) return rc;
  if (!tmp.empty()) HIPCHK(c, hipMemcpy(p, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = p;
  return DC_OK;
}
void free_pool(std::vector<void *> &pool) {
  for (void *p : pool) (void) hipFree(p);
  pool.clear();
}
size_t slot_elems(const dc_ctx *c) { return (size_t) c->B * 3 * c->host.N; }

int pd_cap(const dc_ctx *c) {
  if (c->params.pd_iter_cap >= 0) return c->params.pd_iter_cap;
  return (int) ((-std::log10(c->params.forward_tol)) * 150);   // Simulation.cpp:1182
}

// per_vertex: the array is indexed by vertex (the device renumbering applies); false: by attachment / other index
int h2d_planar(dc_ctx *c, const double *src, float *dst, int n_per_rollout, int which_stage, bool per_vertex) {
  size_t elems = (size_t) c->B * 3 * n_per_rollout;
  HIPCHK(c, hipMemcpyAsync(c->stage[which_stage], src, elems * sizeof(double), hipMemcpyHostToDevice, c->stream));
  launch_f64i_to_f32p(c->stage[which_stage], dst, c->B, n_per_rollout, per_vertex ? c->d_user_of : nullptr, c->stream);
  return DC_OK;
}
int d2h_planar(dc_ctx *c, const float *src, double *dst, int n_per_rollout, int which_stage, bool per_vertex) {
  size_t elems = (size_t) c->B * 3 * n_per_rollout;
  launch_f32p_to_f64i(src, c->stage[which_stage], c->B, n_per_rollout, per_vertex ? c->d_user_of : nullptr, c->stream);
  HIPCHK(c, hipMemcpyAsync(dst, c->stage[which_stage], elems * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  return DC_OK;
}

int check_batch(dc_ctx *c, int slot_lo, int slot_hi) {
  if (!c->built) return fail(c, DC_ERR_STATE, "dc_build has not been called");
  if (c->B <= 0) return fail(c, DC_ERR_STATE, "dc_alloc_batch has not been called");
  if (slot_lo < 0 || slot_hi > c->tape) return fail(c, DC_ERR_INVALID, "tape slot out of range");
  return DC_OK;
}

// The reference's self-contact list has no limit (Simulation.cpp:281-352, 422-624); ours has, and says so: a step whose list
// was cut is not the reference's step.
int check_self_overflow(dc_ctx *c, const dc_step_stats *st, int slot) {
  for (int b = 0; b < c->B; b++)
    if (st[b].self_overflow)
      return fail(c, DC_ERR_CAPACITY, "self-contact list overflow in the step that produced slot " + std::to_string(slot) + ", rollout " +
                  std::to_string(b) + ((st[b].self_overflow & 1) ? ": more pairs than max_self_contacts = " + std::to_string(c->self_cap) : (st[b].self_overflow & 2) ? std::string(": more than 4088 contact layers") : std::string(": inconsistent contact tables (internal error)")) +
                  " (raise dc_params::max_self_contacts and repeat the step)");
  return DC_OK;
}

FwdArgs fwd_args(dc_ctx *c, int slot) {
  const size_t se = slot_elems(c), sp = (size_t) c->B * c->host.N;
  FwdArgs A;
  A.x_in = c->X + se * slot; A.v_in = c->V + se * slot;
  A.x_out = c->X + se * (slot + 1); A.v_out = c->V + se * (slot + 1);
  A.rec_f = c->F + se * (slot + 1); A.rec_r = c->R + se * (slot + 1); A.rec_n = c->NRM + se * (slot + 1);
  A.rec_prim = c->PRIM + sp * (slot + 1);
  A.x_fixed = c->xf_cur; A.mu = c->mu; A.fu = c->fu_set ? c->fu : nullptr; A.fv = c->fv_set ? c->fv : nullptr;
  A.fv2 = c->fv2_set ? c->fv2 : nullptr;
  A.fv_scale = nullptr; A.slot_xfix = 0; A.slot_fu = 0; A.slot_fvs = 0;
  // scheduled values of this step (dc_set_*_schedule) take precedence over the current ones
  if (c->S.Af > 0 && c->sched_xf[slot + 1]) A.x_fixed = c->XF + (size_t) c->B * 3 * c->S.Af * (slot + 1);
  if (c->sched_fu[slot + 1]) A.fu = c->FU_S + (size_t) c->B * 3 * (slot + 1);
  if (c->sched_fvs[slot + 1] && A.fv) A.fv_scale = c->FVS_S + (size_t) c->B * (slot + 1);
  A.stats = c->fstats + (size_t) c->B * (slot + 1);
  {
    const size_t sc = (size_t) c->B * c->self_cap * (slot + 1), sm = (size_t) c->B * kMetaStride * (slot + 1);
    A.self.pair = c->SC_pair + sc; A.self.nrm = c->SC_nrm + sc; A.self.dvec = c->SC_d + sc; A.self.meta = c->SC_meta + sm;
    A.self.verts = c->SC_verts + 2 * sc;
  }
  A.fwd_tol = (float) c->params.forward_tol;
  A.cg_tol = (float) (c->params.cg_rel_tol > 0 ? c->params.cg_rel_tol : 1e-4);
  A.pd_cap = pd_cap(c);
  A.cg_max = c->params.cg_max_iter > 0 ? c->params.cg_max_iter : 500;
  A.stall_window = c->params.stall_window > 0 ? c->params.stall_window : 0x7fffffff;   // off by default: reference semantics
  static const bool seed_on = !(getenv("DC_CG_SEED") && getenv("DC_CG_SEED")[0] == '0');     // development switch
  A.cg_seed = seed_on ? 1 : 0;
  { const char *envp = getenv("DC_FWD_SELFFULL"); A.self_full = envp ? (envp[0] == '1') : 0; }
  A.nsteps = 1; A.inline_detect = 0; A.slot_state = se; A.slot_prim = sp; A.slot_stats = (size_t) c->B;
  A.slot_self = (size_t) c->B * c->self_cap; A.slot_meta = (size_t) c->B * kMetaStride;
  return A;
}
BwdArgs bwd_args(dc_ctx *c, int slot, bool is_start, bool with_init) {
  const size_t se = slot_elems(c), sp = (size_t) c->B * c->host.N;
  BwdArgs A;
  A.x_new = c->X + se * slot; A.rec_f = c->F + se * slot; A.rec_n = c->NRM + se * slot;
  A.rec_prim = c->PRIM + sp * slot; A.mu = c->mu;
  {
    const size_t sc = (size_t) c->B * c->self_cap * slot, sm = (size_t) c->B * kMetaStride * slot;
    A.self.pair = c->SC_pair + sc; A.self.nrm = c->SC_nrm + sc; A.self.dvec = c->SC_d + sc; A.self.meta = c->SC_meta + sm;
    A.self.verts = c->SC_verts + 2 * sc;
  }
  A.gx = c->GX; A.gv = c->GV;
  A.ix = with_init ? c->IX : nullptr; A.iv = with_init ? c->IV : nullptr; A.slot_ix = 0;
  if (!with_init && c->SEEDX && c->sched_seed[slot - 1]) {      // seed schedule: the loss gradient w.r.t. the state this step started from
    A.ix = c->SEEDX + se * (slot - 1); A.iv = c->SEEDV + se * (slot - 1); A.slot_ix = se;
  }
  A.d_xfixed = c->DXF + (size_t) c->B * 3 * c->S.Af * slot; A.d_mu = c->DMU;
  A.d_param = c->DPAR + (size_t) c->B * 8 * slot;
  A.x_fixed = c->XF + (size_t) c->B * 3 * c->S.Af * slot;
  A.x_prev = c->X + se * (slot - 1); A.v_prev = c->V + se * (slot - 1); A.v_new = c->V + se * slot; A.stats = c->bstats + (size_t) c->B * slot;
  A.bwd_tol = (float) c->params.backward_tol;
  A.cg_tol = (float) (c->params.cg_rel_tol > 0 ? c->params.cg_rel_tol : 1e-4);
  A.clip_thr = (float) c->params.gradient_clipping_threshold;
  A.it_cap = c->params.adjoint_iter_cap > 0 ? c->params.adjoint_iter_cap : 400;   // Simulation.cpp:1562
  A.cg_max = c->params.cg_max_iter > 0 ? c->params.cg_max_iter : 500;
  A.is_start = is_start; A.clip = c->params.gradient_clipping;
  A.start_at = c->start_slot + 1;
  A.mode = c->params.adjoint_mode;
  A.rel_tol = (float) (c->params.adjoint_rel_tol > 0 ? c->params.adjoint_rel_tol : 1e-6);
  A.stall_window = c->params.stall_window > 0 ? c->params.stall_window : 0x7fffffff;   // off by default: reference semantics
  { const char *envp = getenv("DC_BLOCK_PRE"); A.block_pre = envp ? (envp[0] != '0') : (c->params.adjoint_block_precond != 0); }   // (development switch)
  { const char *envp = getenv("DC_ADJ_FP32"); A.fp32_only = envp ? (envp[0] == '1') : (c->params.adjoint_fp32_only != 0); }     // (development switch)
  { const char *envp = getenv("DC_ADJ_DENSEY"); A.dense_y = envp ? (envp[0] == '1') : 0; }
  { const char *envp = getenv("DC_ADJ_VERIFY"); A.verify_all = envp ? (envp[0] == '1') : 0; }     // (development switch)
  { const char *envp = getenv("DC_ADJ_WARM"); A.warm = envp ? (envp[0] == '1') : 0; }            // (development switch)
  { const char *envp = getenv("DC_ADJ_CG"); A.cg_first = envp ? (envp[0] == '1') : 1; }          // (A/B switch: 0 = BiCGSTAB correction solves only)
  A.ycap = 0; A.ybase = 0;                                                                                    // (set by the launch, dc_adjoint.hip)
  A.nsteps = 1; A.slot = slot;
  const bool inj = c->inj_slot == slot && c->INJ_X;
  A.inj_x = inj ? c->INJ_X : nullptr; A.inj_f = inj ? c->INJ_F : nullptr; A.inj_n = inj ? c->INJ_N : nullptr;
  A.inj_sn = inj ? c->INJ_SN : nullptr; A.inj_sd = inj ? c->INJ_SD : nullptr;
  A.ys = (c->keep_y && c->YS) ? c->YS + se * slot : nullptr;
  A.slot_state = se; A.slot_prim = sp; A.slot_self = (size_t) c->B * c->self_cap; A.slot_meta = (size_t) c->B * kMetaStride;
  A.slot_param = (size_t) c->B * 8; A.slot_xf = (size_t) c->B * 3 * c->S.Af; A.slot_stats = (size_t) c->B;
  return A;
}


// ---- split execution (dc_cluster.h): choice of K, tables, launches ------------------------------------------------------------
static int round64(int v) { return (v + 63) / 64 * 64; }

void free_cluster(dc_ctx *c) {
  for (void *p : c->cl.allocs) (void) hipFree(p);
  c->cl = ClusterSet();
}

template <typename T, typename U>
int upload_cl(dc_ctx *c, const T **out, const std::vector<U> &src) {
  std::vector<T> tmp(src.begin(), src.end());
  T *p = nullptr;
  int rc = dev_alloc(c, c->cl.allocs, &p, tmp.size());
  if (rc) return rc;
  if (!tmp.empty()) HIPCHK(c, hipMemcpy(p, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = p;
  return DC_OK;
}

// Rollouts one launch of the split kernels can hold with EVERY workgroup resident (the exchange spins on its peers: a part that is not
// scheduled until another rollout has finished its whole sweep would let them run into the spin limit). A launch is padded to a
// multiple of 8 rollouts and the parts of rollout j all run on XCD j mod 8 (cluster_map, dc_cluster.h), so what bounds it is one XCD:
// ceil(nb / 8) * K workgroups on cus / 8 CUs, one workgroup (160 KB of LDS, up to 1024 threads) per CU.
static int cluster_capacity(int cus, int K) { return std::max(1, 8 * ((cus / 8) / std::max(K, 1))); }

// Tables for K parts per rollout; returns DC_OK with c->cl.ok = false when K does not fit this mesh (the caller tries K - 1).
int build_cluster(dc_ctx *c, int K, bool forced) {
  free_cluster(c);
  const HostSystem &H = c->host;
  const int N = H.N;
  if (K < 2 || c->bandwidth <= 0 || c->bandwidth > 511) return DC_OK;
  const int HB = std::max(64, round64(c->bandwidth));
  static const int allowed[] = {1, 2, 3, 4, 6, 8, 12};
  const int lds_cap = (160 * 1024 - 256) / 4 - kXchLdsFloats;      // floats
  HostWindows HW;
  int R = 0, wpp = 0, vpt = 0;
  for (int w = 1; w <= 8 && R == 0; w++) {
    const int own_w = round64((N + K * w - 1) / (K * w));
    const int Rc = own_w * w;
    if ((long long) (K - 1) * Rc >= N) break;           // a part would be empty
    if (Rc < HB) break;                                  // halo rows must come from the direct neighbours only
    if (Rc < 256 && !forced) break;                      // parts of fewer rows than half a workgroup: nothing left to save (the 1426-vertex
                                                         // T-shirt, one rollout: 21.9 / 19.0 / 18.4 / 18.2 ms per fwd+bwd step at K = 1 / 4 / 6 / 8
                                                         // once its parts share an XCD, tools/bench_tshirt_k.py)
    int v = 0;
    for (int a : allowed) if (a * 512 >= Rc) { v = a; break; }
    if (v == 0) continue;                                // more rows per part than the kernel holds in registers: more windows do not help
    if (!HW.build_own(H, own_w)) continue;
    const int win_floats = (int) (HW.lds_bytes / 4);
    const int fwd = std::max(std::max((v <= 6 ? 6 : 3) * (Rc + 2 * HB), win_floats), kSelfDetectLdsInts);      // (<= 6 rows per thread: pipelined CG, two gather arrays)
    const int bwd = (win_floats + 3) / 4 * 4 + 6 * HB;
    if (fwd + 4 > lds_cap || bwd + 4 > lds_cap) continue;
    // the element reach of every window must stay inside the boundary rows its part receives
    bool reach_ok = true;
    for (int q = 0; q < HW.nwin; q++) {
      const int part = q / w, p0 = part * Rc;
      const int lo = HW.win[8 * q + 2], vs = HW.win[8 * q + 3];
      if (lo < p0 - HB || lo + vs > p0 + Rc + HB) reach_ok = false;
    }
    if (!reach_ok) continue;
    R = Rc; wpp = w; vpt = v;
  }
  if (R == 0) return DC_OK;
  HostPackets HP;
  if (!HP.build_rows(H, K * R)) return DC_OK;
  ClusterSet &cl = c->cl;
  DevCluster &D = cl.D;
  std::memset(&D, 0, sizeof(D));
  D.K = K; D.R = R; D.HB = HB; D.wpp = wpp; D.xch_stride = kXchWaves + 2 * HB; D.pk_vpt = vpt;
  D.spin_limit = kSpinLimit; D.test_drop = 0;
  { const char *ev = getenv("DC_SELF_REDUNDANT"); D.redundant_self = ev ? (ev[0] == '1') : 1; }
  if (const char *ev = getenv("DC_TEST_SPIN_MS")) { const long long ms = atoll(ev); if (ms > 0) D.spin_limit = ms * 100000ll; }      // test hooks
  if (const char *ev = getenv("DC_TEST_DROP_PART")) D.test_drop = ev[0] == '1';
  int rc;
  const int *ip; const float *fp;
  if ((rc = upload_cl<int>(c, &ip, HW.win))) return rc;
  D.win = (const int4 *) ip;
  if ((rc = upload_cl<int>(c, &ip, HW.tri_rec))) return rc;
  D.wtri_rec = (const int4 *) ip;
  if ((rc = upload_cl<float>(c, &fp, HW.tri_D))) return rc;
  D.wtri_D = (const float4 *) fp;
  if ((rc = upload_cl<int>(c, &ip, HW.bend_rec))) return rc;
  D.wbend_rec = (const int4 *) ip;
  if ((rc = upload_cl<float>(c, &fp, HW.bend_w))) return rc;
  D.wbend_w = (const float4 *) fp;
  if ((rc = upload_cl<float>(c, &fp, HW.tri_Dlo))) return rc;
  D.wtri_Dlo = (const float4 *) fp;
  if ((rc = upload_cl<float>(c, &fp, HW.bend_lo))) return rc;
  D.wbend_lo = (const float4 *) fp;
  if ((rc = upload_cl<int>(c, &ip, HW.inc))) return rc;
  D.winc = (const int4 *) ip;
  if ((rc = upload_cl<int>(c, &D.winc_ptr, HW.inc_ptr))) return rc;
  if ((rc = upload_cl<int>(c, &D.winc_n, HW.inc_n))) return rc;
  D.nwin = HW.nwin; D.win_vcap = HW.vcap; D.win_nrcap = HW.nrcap; D.win_lds_bytes = (int) HW.lds_bytes;
  if ((rc = upload_cl<int>(c, &ip, HP.pk))) return rc;
  D.pk = (const int4 *) ip;
  if ((rc = upload_cl<int>(c, &D.pk_ptr, HP.pk_ptr))) return rc;
  if ((rc = upload_cl<int>(c, &D.pk_n, HP.pk_n))) return rc;
  if ((rc = upload_cl<float>(c, &D.sq_dinv, HP.sq_dinv))) return rc;
  cl.K = K;
  {   // rollouts per launch: all of them when they fit, else equal chunks (never a last launch with a handful of rollouts)
    const int nbmax = cluster_capacity(c->cus, K), nchunks = (c->B + nbmax - 1) / nbmax;
    cl.nb = std::max(1, (c->B + nchunks - 1) / nchunks);
  }
  D.nb = cl.nb;
  cl.xch_bytes = (size_t) cl.nb * K * 2 * D.xch_stride * sizeof(v4i);
  if ((rc = dev_alloc(c, cl.allocs, &D.xch, cl.xch_bytes / sizeof(v4i)))) return rc;
  if ((rc = dev_alloc(c, cl.allocs, &D.err, 4))) return rc;
  DevCluster *dD = nullptr;
  if ((rc = dev_alloc(c, cl.allocs, &dD, 1))) return rc;
  D.self_dev = dD;
  HIPCHK(c, hipMemcpy(dD, &D, sizeof(DevCluster), hipMemcpyHostToDevice));
  cl.ok = true;
  return DC_OK;
}

// Which deflation space the forward solve wants (dc_params::forward_deflation: -1 decide by the probe solve, 0 never, > 0 always; the
// development switch DC_DEFLATION overrides it: 0 = off, 1 = always) — one rule for device and host-only contexts, read at every build.
static int deflation_want(const dc_params &p) {
  const char *envd = getenv("DC_DEFLATION");
  if (envd) return atoi(envd) > 0 ? 16 : 0;
  return p.forward_deflation > 0 ? 16 : p.forward_deflation;
}
// HostDeflation::build through the context's cache (key: the scalar system matrix, the request, the padded row count)
static bool build_deflation_cached(dc_ctx *c, const HostSystem &H, int want, int rows_padded, HostDeflation **out) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void *q, size_t n) { const unsigned char *b = (const unsigned char *) q; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
  mix(&H.N, sizeof(int)); mix(&want, sizeof(int)); mix(&rows_padded, sizeof(int));
  mix(H.P_ptr.data(), H.P_ptr.size() * sizeof(H.P_ptr[0])); mix(H.P_col.data(), H.P_col.size() * sizeof(H.P_col[0]));
  mix(H.P_val.data(), H.P_val.size() * sizeof(H.P_val[0])); mix(H.mass.data(), H.mass.size() * sizeof(H.mass[0]));
  if (!c->defl_cache_valid || c->defl_key != h) {
    c->defl_cache_built = c->defl_cache.build(H, want, rows_padded);
    c->defl_key = h; c->defl_cache_valid = true;
  }
  *out = &c->defl_cache;
  return c->defl_cache_built;
}

// K for this batch: enough parts to give every CU a workgroup (B rollouts x K <= CUs, K <= 8), at least as many as a mesh too
// large for the one-workgroup kernel needs; DC_CLUSTER=k forces k (development switch; 0 / 1 = off).
int choose_cluster(dc_ctx *c) {
  free_cluster(c);
  if (c->host_only || c->B <= 0) return DC_OK;
  const char *env = getenv("DC_CLUSTER");
  const int forced = env ? atoi(env) : -1;
  if (forced == 0 || forced == 1) return DC_OK;
  if (c->S.dense_inv) { if (forced < 2) return DC_OK; }     // small meshes: the explicit-inverse kernels are the faster ones
  // Fewer rollouts than CUs: as many parts as fit (B K <= CUs), up to 8 — a rollout's speed-up grows with K (measured on C4: 1.3 /
  // 1.9 / 3.0 x at K = 2 / 4 / 8). A mesh too large for one workgroup needs kmin parts; when that oversubscribes the device the
  // batch runs in several launches and K is the one that wastes the least: score = fraction of the CUs busy x per-CU efficiency.
  // per-CU efficiency of K parts against one workgroup per rollout, re-measured in round 6 on the C4 workload (bench.py --total-batch 128 / 64 / 32
  // against 256: 6 032 / 5 031 / 4 003 against 9 880 rollout-steps/s -> 0.61 / 0.51 / 0.405 at K = 2 / 4 / 8; K = 3, 5, 6, 7 interpolated)
  static const double eff[9] = {0, 1.0, 0.61, 0.56, 0.51, 0.48, 0.45, 0.43, 0.405};
  const int kmin = (!c->S.pk_ok || !c->S.win_ok) ? std::max(2, std::min(8, (c->host.N + 6143) / 6144)) : 1;
  int K = 1;
  if (forced >= 2) K = std::min(forced, 8);
  else if (c->B <= cluster_capacity(c->cus, std::max(kmin, 2))) { K = std::max(kmin, 2); while (K + 1 <= 8 && c->B <= cluster_capacity(c->cus, K + 1)) K++; }
  else if (kmin > 1) {
    double best = -1;
    for (int k = kmin; k <= 8; k++) {
      const int nbmax = cluster_capacity(c->cus, k), nchunks = (c->B + nbmax - 1) / nbmax, nb = (c->B + nchunks - 1) / nchunks;
      const double score = (double) nb * k / c->cus * eff[k];
      if (score > best + 1e-9) { best = score; K = k; }
    }
  }
  for (; K >= 2; K--) {
    int rc = build_cluster(c, K, forced >= 2);
    if (rc) return rc;
    if (c->cl.ok) break;
    if (forced < 2 && K <= kmin) break;
  }
  return DC_OK;
}

// The error word of the split kernels is sticky on the device: it is zero until an exchange times out and is cleared only AFTER that
// has been reported — an asynchronous call (dc_step_forward without statistics, dc_step_backward ...) can therefore not lose a
// time-out to the next call: whichever synchronising call comes first (statistics, dc_get_state / gradient / record, dc_sync) reports it.
int cluster_begin(dc_ctx *) { return DC_OK; }
int cluster_check(dc_ctx *c) {       // after a synchronisation
  if (!c->cl.ok) return DC_OK;
  unsigned e[4] = {0, 0, 0, 0};
  HIPCHK(c, hipMemcpy(e, c->cl.D.err, sizeof(e), hipMemcpyDeviceToHost));
  if (e[0]) HIPCHK(c, hipMemset(c->cl.D.err, 0, 16));
  if (e[0]) return fail(c, DC_ERR_HIP, "split kernels: an inter-workgroup exchange timed out (waiting for sequence " + std::to_string(e[1]) + ", saw tag " +
                        std::to_string(e[2]) + ", site " + std::to_string(e[3] & 255u) + ", part " + std::to_string((e[3] >> 8) & 15u) + ", granule " +
                        std::to_string(e[3] >> 12) + "; set DC_CLUSTER=1 to run one workgroup per rollout)");
  return DC_OK;
}

bool use_cluster_fwd(const dc_ctx *c) { return c->cl.ok; }
bool use_cluster_bwd(const dc_ctx *c) { return c->cl.ok && c->params.adjoint_mode == 1; }

int enqueue_pd_step(dc_ctx *c, const FwdArgs &A) {
  {   // a forward step overwrites records: a record handed in from outside for one of them (dc_set_record) is gone
    const long first = (long) ((A.x_out - c->X) / (long) slot_elems(c));
    if (c->inj_slot >= first && c->inj_slot < first + A.nsteps) c->inj_slot = -1;
  }
  if (!use_cluster_fwd(c)) { launch_pd_step(c->S, c->W, A, c->B, c->stream); HIPCHK(c, hipGetLastError()); return DC_OK; }
  for (int b0 = 0; b0 < c->B; b0 += c->cl.nb) {
    HIPCHK(c, hipMemsetAsync(c->cl.D.xch, 0, c->cl.xch_bytes, c->stream));
    HIPCHK(c, launch_pd_step_cluster(c->S, c->cl.D, c->W, A, b0, std::min(c->cl.nb, c->B - b0), c->stream));
  }
  return DC_OK;
}
int enqueue_adjoint_step(dc_ctx *c, const BwdArgs &A) {
  if (!use_cluster_bwd(c)) { launch_adjoint_step(c->S, c->W, A, c->B, c->stream); HIPCHK(c, hipGetLastError()); return DC_OK; }
  for (int b0 = 0; b0 < c->B; b0 += c->cl.nb) {
    HIPCHK(c, hipMemsetAsync(c->cl.D.xch, 0, c->cl.xch_bytes, c->stream));
    HIPCHK(c, launch_adjoint_step_cluster(c->S, c->cl.D, c->W, A, b0, std::min(c->cl.nb, c->B - b0), c->stream));
  }
  return DC_OK;
}

}  // namespace

extern "C" {

const char *dc_version(void) { return "diffcloth_hip 0.1 (gfx950)"; }

void dc_default_params(dc_params *p) {
  std::memset(p, 0, sizeof(*p));
  p->time_step = 1.0 / 90; p->density = 0.1; p->k_stretch = 100; p->k_bend = 0.01; p->k_att = 10000;   // AttachmentSpring.cpp:10
  p->gravity[0] = 0; p->gravity[1] = -9.8; p->gravity[2] = 0;                                          // Simulation.h:356
  p->forward_tol = 1e-7; p->backward_tol = 5e-5;                                                        // Simulation.cpp:17-19
  p->gravity_enabled = 1; p->contact_enabled = 1; p->selfcollision_enabled = 0;
  p->gradient_clipping = 1; p->gradient_clipping_threshold = 16.0;                                      // Simulation.h:330-331
  p->pd_iter_cap = -1; p->adjoint_iter_cap = 400; p->cg_rel_tol = 1e-4; p->cg_max_iter = 500; p->stall_window = 0;
  p->adjoint_mode = 0; p->adjoint_rel_tol = 1e-6; p->adjoint_block_precond = 1; p->adjoint_fp32_only = 0;
  p->max_self_contacts = 0;      /* sized from the mesh in dc_build */
  p->forward_deflation = -1;     /* decided in dc_build by a probe solve */
}

int dc_create(int device_id, dc_ctx **out) {
  if (!out) return DC_ERR_INVALID;
  *out = nullptr;
  if (device_id == -1) {   // host-only context: no device memory, no kernels, no CPU compute path either
    dc_ctx *c = new dc_ctx();
    c->device = -1; c->host_only = true;
    dc_default_params(&c->params);
    std::memset(&c->S, 0, sizeof(c->S));
    std::memset(&c->W, 0, sizeof(c->W));
    *out = c;
    return DC_OK;
  }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return DC_ERR_HIP;   // no CPU fallback: fail loudly
  if (device_id < 0 || device_id >= count) return DC_ERR_INVALID;
  dc_ctx *c = new dc_ctx();
  c->device = device_id;
  dc_default_params(&c->params);
  std::memset(&c->S, 0, sizeof(c->S));
  std::memset(&c->W, 0, sizeof(c->W));
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess) { delete c; return DC_ERR_HIP; }
  c->own_stream = c->stream;
  if (hipDeviceGetAttribute(&c->cus, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess || c->cus <= 0) c->cus = 256;
  if (hipEventCreate(&c->ev_a) != hipSuccess || hipEventCreate(&c->ev_b) != hipSuccess || hipEventCreate(&c->ev_t0) != hipSuccess ||
      hipEventCreate(&c->ev_t1) != hipSuccess) { delete c; return DC_ERR_HIP; }
  *out = c;
  return DC_OK;
}

int dc_destroy(dc_ctx *c) {
  if (!c) return DC_OK;
  if (c->host_only) { delete c; return DC_OK; }
  (void) hipSetDevice(c->device);
  (void) hipStreamSynchronize(c->stream);
  (void) dc_comm_destroy(c);
  free_cluster(c);
  free_pool(c->table_allocs);
  free_pool(c->batch_allocs);
  (void) hipEventDestroy(c->ev_a); (void) hipEventDestroy(c->ev_b);
  (void) hipEventDestroy(c->ev_t0); (void) hipEventDestroy(c->ev_t1);
  (void) hipStreamDestroy(c->own_stream);
  delete c;
  return DC_OK;
}

const char *dc_last_error(const dc_ctx *c) { return c ? c->err.c_str() : "null context"; }

int dc_set_mesh(dc_ctx *c, int n, const double *pos, int t, const int *tris) {
  if (!c) return DC_ERR_INVALID;
  c->built = false;
  c->user_of.clear(); c->dev_of.clear();
  if (!c->host.set_mesh(n, pos, t, tris)) { c->mesh_set = false; return fail(c, DC_ERR_TOPOLOGY, c->host.error); }
  // Renumber the vertices on the device when the caller's numbering couples far-apart indices (typical of
  // modelling-tool exports): the packet-ELL matrix (|i - j| <= 511 over two rings) and the element windows need
  // locality. DC_RENUMBER=0 / 1 forces it off / on (development switch).
  const char *env = getenv("DC_RENUMBER");
  const bool want = env ? env[0] == '1' : 2 * mesh_bandwidth(t, tris) > 511;
  if (want) {
    std::vector<int> order = rcm_order(n, t, tris);
    std::vector<int> inv(n);
    for (int k = 0; k < n; k++) inv[order[k]] = k;
    std::vector<int> tri2(3 * (size_t) t);
    for (size_t k = 0; k < tri2.size(); k++) tri2[k] = inv[tris[k]];
    if (2 * mesh_bandwidth(t, tri2.data()) < 2 * mesh_bandwidth(t, tris) || env) {
      std::vector<double> pos2(3 * (size_t) n);
      for (int k = 0; k < n; k++) for (int d = 0; d < 3; d++) pos2[3 * (size_t) k + d] = pos[3 * (size_t) order[k] + d];
      if (!c->host.set_mesh(n, pos2.data(), t, tri2.data())) { c->mesh_set = false; return fail(c, DC_ERR_TOPOLOGY, c->host.error); }
      c->user_of = order; c->dev_of = inv;
    }
  }
  c->mesh_set = true;
  return DC_OK;
}

int dc_set_attachments(dc_ctx *c, int count, const int *vertex) {
  if (!c || count < 0 || (count > 0 && !vertex)) return fail(c, DC_ERR_INVALID, "dc_set_attachments: bad arguments");
  c->att_user.assign(vertex, vertex + count);
  c->built = false;
  return DC_OK;
}

int dc_set_params(dc_ctx *c, const dc_params *p) {
  if (!c || !p) return DC_ERR_INVALID;
  if (!(p->time_step > 0) || !(p->density > 0)) return fail(c, DC_ERR_INVALID, "dc_set_params: time_step and density must be > 0");
  c->params = *p;
  c->built = false;
  return DC_OK;
}

int dc_set_primitives(dc_ctx *c, int count, const dc_primitive *prims) {
  if (!c || count < 0 || count > kMaxPrims) return fail(c, DC_ERR_INVALID, "dc_set_primitives: at most 8 flattened primitives");
  if (count > 0 && !prims) return fail(c, DC_ERR_INVALID, "dc_set_primitives: null primitive array");
  int discretised = 0;
  for (int k = 0; k < count; k++) {
    if (prims[k].kind < DC_PRIM_SPHERE || prims[k].kind > DC_PRIM_SPHERE_DISCRETIZED) return fail(c, DC_ERR_INVALID, "dc_set_primitives: unknown primitive kind");
    if (prims[k].kind == DC_PRIM_SPHERE_DISCRETIZED) {
      discretised++;
      if (!(prims[k].radius > 0) || prims[k].length < 0 || prims[k].length > 512) return fail(c, DC_ERR_INVALID, "dc_set_primitives: discretised sphere needs radius > 0 and a resolution (length) of 0 or 3 ... 512");
    }
  }
  if (discretised > 1) return fail(c, DC_ERR_INVALID, "dc_set_primitives: at most one discretised sphere per context");
  c->prims.assign(prims, prims + count);
  // compact the caller's group ids to 0..ngroups-1 in order of first appearance
  std::vector<int> seen;
  c->group_of_prim.assign(count, 0);
  for (int k = 0; k < count; k++) {
    int g = -1;
    for (size_t s = 0; s < seen.size(); s++) if (seen[s] == prims[k].group) g = (int) s;
    if (g < 0) { g = (int) seen.size(); seen.push_back(prims[k].group); }
    c->group_of_prim[k] = g;
  }
  c->ngroups = (int) seen.size();
  c->built = false;
  return DC_OK;
}

int dc_build(dc_ctx *c) {
  if (!c) return DC_ERR_INVALID;
  if (!c->mesh_set) return fail(c, DC_ERR_STATE, "dc_build: dc_set_mesh has not been called");
  c->inj_slot = -1;      // a rebuilt system no longer matches a record handed in before (dc_set_record)
  const dc_params &p = c->params;
  HostSystem &H = c->host;
  H.att_vertex = c->att_user;
  for (int &a : H.att_vertex) {
    if (a < 0 || a >= H.N) return fail(c, DC_ERR_INVALID, "dc_build: attachment vertex out of range");
    if (!c->dev_of.empty()) a = c->dev_of[a];
  }
  if (!H.build_numerics(p.time_step, p.density, p.k_stretch, p.k_bend, p.k_att)) return fail(c, DC_ERR_TOPOLOGY, H.error);
  c->bandwidth = 0;
  for (int r = 0; r < H.N; r++)
    for (int k = H.P_ptr[r]; k < H.P_ptr[r + 1]; k++) c->bandwidth = std::max(c->bandwidth, std::abs(H.P_col[k] - r));
  if (c->host_only) {       // which kernel set this system would get (dc_get_layout): the same host-side table builders, nothing uploaded
    HostWindows HW; HostPackets HP;
    c->S.win_ok = HW.build(H, (size_t) 150 * 1024) ? 1 : 0;
    c->S.pk_ok = HP.build(H) ? 1 : 0;
    c->S.nwin = HW.nwin; c->S.pk_vpt = HP.vpt; c->S.pk_threads = HP.threads;
    {
      HostDeflation *HD = nullptr;
      c->defl_k = 0; c->defl_probe = 0;
      if (c->S.win_ok) { build_deflation_cached(c, H, deflation_want(p), c->S.pk_ok ? HP.threads * HP.vpt : round64(H.N), &HD); c->defl_k = HD->k; c->defl_probe = HD->probe_iterations; }
    }
    c->built = true;
    return DC_OK;
  }
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  free_cluster(c);
  free_pool(c->table_allocs);
  DevSystem &S = c->S;
  std::memset(&S, 0, sizeof(S));
  const int N = H.N, T = H.T, E = H.E, Af = (int) H.att_vertex.size();
  S.N = N; S.T = T; S.E = E; S.Af = Af; S.NC = 3 * T + 4 * E;
  S.user_of = nullptr; S.dev_of = nullptr; c->d_user_of = nullptr;
  if (!c->user_of.empty()) {
    int rcp;
    if ((rcp = upload<int>(c, &S.user_of, c->user_of))) return rcp;
    if ((rcp = upload<int>(c, &S.dev_of, c->dev_of))) return rcp;
    c->d_user_of = S.user_of;
  }
  // planar index tables
  std::vector<int> triv(3 * (size_t) T), bendv(4 * (size_t) E);
  for (int t = 0; t < T; t++) for (int k = 0; k < 3; k++) triv[(size_t) k * T + t] = H.tri[3 * t + k];
  for (int e = 0; e < E; e++) for (int k = 0; k < 4; k++) bendv[(size_t) k * E + e] = H.bend_v[4 * e + k];
  std::vector<float> bendnw(2 * (size_t) E), dinv(N);
  for (int e = 0; e < E; e++) { bendnw[2 * e] = (float) H.bend_n[e]; bendnw[2 * e + 1] = (float) H.bend_w2[e]; }
  for (int i = 0; i < N; i++) {
    double d = 0;
    for (int k = H.P_ptr[i]; k < H.P_ptr[i + 1]; k++) if (H.P_col[k] == i) d = H.P_val[k];
    dinv[i] = (float) (1.0 / d);
  }
  std::vector<int> att_of(N, -1);
  for (int a = 0; a < Af; a++) att_of[H.att_vertex[a]] = a;
  int rc;
  const float *f4;
  if ((rc = upload<int>(c, &S.tri_v, triv))) return rc;
  if ((rc = upload<float>(c, &f4, H.tri_D))) return rc;
  S.tri_D = (const float4 *) f4;
  if ((rc = upload<float>(c, &S.tri_w2, H.tri_w2))) return rc;
  if ((rc = upload<int>(c, &S.bend_v, bendv))) return rc;
  if ((rc = upload<float>(c, &f4, H.bend_w))) return rc;
  S.bend_w = (const float4 *) f4;
  if ((rc = upload<float>(c, &f4, bendnw))) return rc;
  S.bend_nw = (const float2 *) f4;
  if ((rc = upload<int>(c, &S.att_vertex, H.att_vertex))) return rc;
  if ((rc = upload<int>(c, &S.att_of_vertex, att_of))) return rc;
  if ((rc = upload<float>(c, &S.mass, H.mass))) return rc;
  if ((rc = upload<float>(c, &S.dinv, dinv))) return rc;
  {  // fp64 rest-shape tables of the adjoint's fp64 operator (dc_adjoint64.h), planar
    std::vector<double> d4(4 * (size_t) T), w4(4 * (size_t) E), nw2(2 * (size_t) E);
    for (int t = 0; t < T; t++) for (int k = 0; k < 4; k++) d4[(size_t) k * T + t] = H.tri_D[4 * (size_t) t + k];
    for (int e = 0; e < E; e++) {
      for (int k = 0; k < 4; k++) w4[(size_t) k * E + e] = H.bend_w[4 * (size_t) e + k];
      nw2[e] = H.bend_n[e]; nw2[(size_t) E + e] = H.bend_w2[e];
    }
    if ((rc = upload<double>(c, &S.tri_D64, d4))) return rc;
    if ((rc = upload<double>(c, &S.tri_w2_64, H.tri_w2))) return rc;
    if ((rc = upload<double>(c, &S.bend_w64, w4))) return rc;
    if ((rc = upload<double>(c, &S.bend_nw64, nw2))) return rc;
    if ((rc = upload<double>(c, &S.mass64, H.mass))) return rc;
    std::vector<float> dlo(4 * (size_t) T), blo(4 * (size_t) E);
    for (size_t k = 0; k < dlo.size(); k++) dlo[k] = (float) (H.tri_D[k] - (double) (float) H.tri_D[k]);
    for (int e = 0; e < E; e++) {
      for (int k = 1; k < 4; k++) blo[4 * (size_t) e + k - 1] = (float) (H.bend_w[4 * (size_t) e + k] - (double) (float) H.bend_w[4 * (size_t) e + k]);
      blo[4 * (size_t) e + 3] = (float) (H.bend_n[e] - (double) (float) H.bend_n[e]);
    }
    if ((rc = upload<float>(c, &f4, dlo))) return rc;
    S.tri_Dlo = (const float4 *) f4;
    if ((rc = upload<float>(c, &f4, blo))) return rc;
    S.bend_lo = (const float4 *) f4;
  }
  if ((rc = upload<int>(c, &S.P_ptr, H.P_ptr))) return rc;
  if ((rc = upload<int>(c, &S.P_col, H.P_col))) return rc;
  if ((rc = upload<float>(c, &S.P_val, H.P_val))) return rc;
  if ((rc = upload<int>(c, &S.inc_ptr, H.inc_ptr))) return rc;
  if ((rc = upload<int>(c, &S.inc_idx, H.inc_idx))) return rc;
  if ((rc = upload<float>(c, &S.radii, H.radii))) return rc;
  if ((rc = upload<int>(c, &S.conn_ptr, H.conn_ptr))) return rc;
  if ((rc = upload<int>(c, &S.conn_idx, H.conn_idx))) return rc;
  {
    double mr = H.radii.empty() ? 0.0 : H.radii[0];
    for (double r : H.radii) mr = std::max(mr, r);
    S.max_radii = (float) mr;
    // capacity of the per-rollout contact list: the caller's, or sized from the mesh (a fold brings every vertex of the upper
    // layer into contact with one of the lower: ~N/2 pairs). Slots of the working set are 16-bit: at most 16000 pairs.
    S.self_cap = std::min(p.max_self_contacts > 0 ? p.max_self_contacts : std::max(2048, N), 16000);
    { const char *envs = getenv("DC_SELF_LDS"); S.self_lds = !(envs && envs[0] == '0'); }   // development switch: 0 = global-memory layer passes
  }
  {  // wave-sliced ELL copy of P for the LDS-resident PCG
    const int nchunks = (N + 63) / 64;
    std::vector<int> eptr(nchunks), ew(nchunks);
    std::vector<int> flat;   // (col, value bits) pairs
    for (int ch = 0; ch < nchunks; ch++) {
      int w = 0;
      for (int r = 64 * ch; r < std::min(N, 64 * ch + 64); r++) w = std::max(w, H.P_ptr[r + 1] - H.P_ptr[r]);
      eptr[ch] = (int) (flat.size() / 2); ew[ch] = w;
      flat.resize(flat.size() + (size_t) 2 * 64 * w);
      for (int s = 0; s < w; s++)
        for (int l = 0; l < 64; l++) {
          const int r = 64 * ch + l;
          int col = std::min(r, N - 1);
          float val = 0.f;
          if (r < N && H.P_ptr[r] + s < H.P_ptr[r + 1]) { col = H.P_col[H.P_ptr[r] + s]; val = (float) H.P_val[H.P_ptr[r] + s]; }
          int bits;
          std::memcpy(&bits, &val, sizeof(int));
          const size_t o = 2 * ((size_t) eptr[ch] + (size_t) s * 64 + l);
          flat[o] = col; flat[o + 1] = bits;
        }
    }
    const int *ellp;
    if ((rc = upload<int>(c, &ellp, flat))) return rc;
    S.ell = (const int2 *) ellp;
    if ((rc = upload<int>(c, &S.ell_ptr, eptr))) return rc;
    if ((rc = upload<int>(c, &S.ell_w, ew))) return rc;
  }
  {  // element windows: the local step and the adjoint's element pass run inside LDS
    HostWindows HW;
    const char *envw = getenv("DC_WINDOWS");       // development switch: DC_WINDOWS=0 keeps the global-memory corner passes
    S.win_ok = 0;
    if (!(envw && envw[0] == '0') && HW.build(H, (size_t) 150 * 1024)) {
      const int *ip; const float *fp;
      if ((rc = upload<int>(c, &ip, HW.win))) return rc;
      S.win = (const int4 *) ip;
      if ((rc = upload<int>(c, &ip, HW.tri_rec))) return rc;
      S.wtri_rec = (const int4 *) ip;
      if ((rc = upload<float>(c, &fp, HW.tri_D))) return rc;
      S.wtri_D = (const float4 *) fp;
      if ((rc = upload<int>(c, &ip, HW.bend_rec))) return rc;
      S.wbend_rec = (const int4 *) ip;
      if ((rc = upload<float>(c, &fp, HW.bend_w))) return rc;
      S.wbend_w = (const float4 *) fp;
      if ((rc = upload<float>(c, &fp, HW.tri_Dlo))) return rc;
      S.wtri_Dlo = (const float4 *) fp;
      if ((rc = upload<float>(c, &fp, HW.bend_lo))) return rc;
      S.wbend_lo = (const float4 *) fp;
      if ((rc = upload<int>(c, &ip, HW.inc))) return rc;
      S.winc = (const int4 *) ip;
      if ((rc = upload<int>(c, &S.winc_ptr, HW.inc_ptr))) return rc;
      if ((rc = upload<int>(c, &S.winc_n, HW.inc_n))) return rc;
      S.nwin = HW.nwin; S.win_vcap = HW.vcap; S.win_nrcap = HW.nrcap; S.win_lds_bytes = (int) HW.lds_bytes;
      S.win_ok = 1;
    }
  }
  {  // packet-ELL copy of the scaled matrix for dc_forward_pk.hip (dc_packets.h)
    HostPackets HP;
    S.pk_ok = 0; S.pk_vpt = 0; S.pk = nullptr; S.pk_ptr = nullptr; S.pk_n = nullptr; S.sq_dinv = nullptr;
    if (HP.build(H)) {
      const int *pkp;
      if ((rc = upload<int>(c, &pkp, HP.pk))) return rc;
      S.pk = (const int4 *) pkp;
      if ((rc = upload<int>(c, &S.pk_ptr, HP.pk_ptr))) return rc;
      if ((rc = upload<int>(c, &S.pk_n, HP.pk_n))) return rc;
      if ((rc = upload<float>(c, &S.sq_dinv, HP.sq_dinv))) return rc;
      S.pk_vpt = HP.vpt; S.pk_threads = HP.threads; S.pk_ok = 1;
    } else {
      // no packet tables (matrix bandwidth beyond the +-511 of their column deltas: the reference's 17 562-vertex dress, 647 after
      // renumbering): the scaling D^-1/2 alone, for the coarse level of the ADJOINT's preconditioner (dc_adjoint64.h), which such a mesh needs
      std::vector<float> sq((size_t) round64(N), 0.f);
      for (int r = 0; r < N; r++)
        for (int k = H.P_ptr[r]; k < H.P_ptr[r + 1]; k++)
          if (H.P_col[k] == r) sq[r] = (float) (1.0 / std::sqrt(H.P_val[k]));
      if ((rc = upload<float>(c, &S.sq_dinv, sq))) return rc;
    }
  }
  {  // irregular garments: the 16 lowest eigenvectors of the scaled matrix as a deflation space of the forward solve (dc_deflate.h)
    HostDeflation *HDp = nullptr;
    S.defl_u = nullptr; S.defl_au = nullptr; S.defl_g = nullptr; c->defl_k = 0; c->defl_probe = 0;
    const int want = deflation_want(p);
    // (the deflated FORWARD kernels exist for 512 threads x >= 4 rows: meshes of more than 1536 vertices, dc_forward_pk_defl.hip; smaller meshes
    //  solve their forward step with the explicit inverse and use the space for the adjoint's coarse level only)
    S.fwd_defl = 0;
    if (S.win_ok && build_deflation_cached(c, H, want, S.pk_ok ? S.pk_threads * S.pk_vpt : round64(N), &HDp)) {
      const HostDeflation &HD = *HDp;
      if ((rc = upload<float>(c, &S.defl_u, HD.U))) return rc;
      if ((rc = upload<float>(c, &S.defl_au, HD.AU))) return rc;
      if ((rc = upload<float>(c, &S.defl_g, HD.G))) return rc;
      c->defl_k = HD.k;
      // (round 6: a mesh without packet tables runs the global-memory kernel, which projects too — dc_devlib.h: deflate_global)
      S.fwd_defl = ((S.pk_ok && S.pk_threads == 512 && S.pk_vpt >= 4) || !S.pk_ok) ? 1 : 0;
    }
    static const char *envc = getenv("DC_ADJ_COARSE");      // development switch: 0 = block preconditioner only in the adjoint's fall-back
    S.adj_coarse = (S.defl_u && !(envc && atoi(envc) == 0)) ? 1 : 0;
    c->defl_probe = HDp ? HDp->probe_iterations : 0;
  }
  {  // small meshes: explicit inverse of the scaled matrix (dc_dense.h) for the forward global step
    HostDense HD;
    const char *envd = getenv("DC_DENSE_MAX_N");   // development switch: 0 disables, other values move the size limit
    const int max_n = envd ? atoi(envd) : 768;          // 2.4 MB: the matrix must stay in every XCD's 4 MB L2 next to the other tables
    S.dense_inv = nullptr; S.dense_ld = 0;
    if (S.pk_ok && S.win_ok && HD.build(H, max_n)) {
      if ((rc = upload<float>(c, &S.dense_inv, HD.inv))) return rc;
      S.dense_ld = HD.ld;
    }
  }
  S.h = (float) p.time_step; S.k_att = (float) p.k_att;
  S.k_stretch = (float) p.k_stretch; S.k_bend = (float) p.k_bend; S.density = (float) p.density;
  S.gx = p.gravity_enabled ? (float) p.gravity[0] : 0.f;
  S.gy = p.gravity_enabled ? (float) p.gravity[1] : 0.f;
  S.gz = p.gravity_enabled ? (float) p.gravity[2] : 0.f;
  S.h64 = p.time_step; S.k_att64 = p.k_att; S.k_stretch64 = p.k_stretch; S.k_bend64 = p.k_bend; S.density64 = p.density;
  for (int k = 0; k < 3; k++) S.g64[k] = p.gravity_enabled ? p.gravity[k] : 0.0;
  S.contact_enabled = p.contact_enabled; S.self_enabled = p.selfcollision_enabled;
  S.nprim = (int) c->prims.size(); S.ngroups = std::max(c->ngroups, 1);
  S.dsph_tri = nullptr; S.dsph_ntri = 0;
  for (int k = 0; k < S.nprim; k++) {
    const dc_primitive &q = c->prims[k];
    if (q.kind == DC_PRIM_SPHERE_DISCRETIZED) {      // the face table of the sphere's own mesh (Sphere::Sphere, Primitive.cpp:133-216)
      const int res = q.length >= 3 ? (int) q.length : 40;
      const std::vector<double> tab = sphere_mesh_table(q.radius, res);
      if ((rc = upload<double>(c, &S.dsph_tri, tab))) return rc;
      S.dsph_ntri = (int) tab.size() / 12;
    }
    DevPrim &d = S.prims[k];
    d.kind = q.kind; d.group = c->group_of_prim[k]; d.rotates = q.rotates; d.pad = 0;
    d.cx = (float) q.center[0]; d.cy = (float) q.center[1]; d.cz = (float) q.center[2]; d.radius = (float) q.radius;
    d.tx = (float) q.top_offset[0]; d.ty = (float) q.top_offset[1]; d.tz = (float) q.top_offset[2]; d.length = (float) q.length;
    d.ux = (float) q.corner2[0]; d.uy = (float) q.corner2[1]; d.uz = (float) q.corner2[2]; d.pad2 = 0.f;
  }
  {  // device-resident copy of the descriptor itself (kernels take a pointer to it)
    DevSystem *dS = nullptr;
    if ((rc = dev_alloc(c, c->table_allocs, &dS, 1))) return rc;
    S.self_dev = dS;
    HIPCHK(c, hipMemcpy(dS, &S, sizeof(DevSystem), hipMemcpyHostToDevice));
  }
  c->built = true;
  // a batch allocated for a different system size is no longer valid
  if (c->B > 0) {
    int B = c->B, tape = c->tape;
    c->B = 0;
    return dc_alloc_batch(c, B, tape);
  }
  return DC_OK;
}

int dc_set_solver(dc_ctx *c, double forward_tol, double backward_tol, int gradient_clipping, double clip_threshold, int force_direct_adjoint) {
  if (!c) return DC_ERR_INVALID;
  c->params.forward_tol = forward_tol; c->params.backward_tol = backward_tol;
  c->params.gradient_clipping = gradient_clipping; c->params.gradient_clipping_threshold = clip_threshold;
  c->params.adjoint_mode = force_direct_adjoint ? 1 : 0;
  return DC_OK;   // solver knobs are kernel arguments: no rebuild, the batch and its tape stay valid
}

int dc_set_flags(dc_ctx *c, int gravity_enabled, int contact_enabled, int selfcollision_enabled) {
  if (!c) return DC_ERR_INVALID;
  c->params.gravity_enabled = gravity_enabled; c->params.contact_enabled = contact_enabled;
  c->params.selfcollision_enabled = selfcollision_enabled;
  if (!c->built || c->host_only) return DC_OK;
  DevSystem &S = c->S;
  S.gx = gravity_enabled ? (float) c->params.gravity[0] : 0.f;
  S.gy = gravity_enabled ? (float) c->params.gravity[1] : 0.f;
  S.gz = gravity_enabled ? (float) c->params.gravity[2] : 0.f;
  for (int k = 0; k < 3; k++) S.g64[k] = gravity_enabled ? c->params.gravity[k] : 0.0;
  S.contact_enabled = contact_enabled; S.self_enabled = selfcollision_enabled;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy((void *) S.self_dev, &S, sizeof(DevSystem), hipMemcpyHostToDevice));
  return DC_OK;
}

int dc_get_counts(const dc_ctx *c, int *out6) {
  if (!c || !out6 || !c->mesh_set) return DC_ERR_INVALID;
  out6[0] = c->host.N; out6[1] = c->host.T; out6[2] = c->host.E; out6[3] = (int) c->host.att_vertex.size();
  out6[4] = (int) c->host.P_col.size(); out6[5] = c->host.rows();
  return DC_OK;
}
int dc_get_system_matrix(const dc_ctx *c, int *row_ptr, int *col, double *val) {
  if (!c || !c->built) return DC_ERR_STATE;
  const HostSystem &H = c->host;
  if (c->user_of.empty()) {
    std::memcpy(row_ptr, H.P_ptr.data(), sizeof(int) * H.P_ptr.size());
    std::memcpy(col, H.P_col.data(), sizeof(int) * H.P_col.size());
    std::memcpy(val, H.P_val.data(), sizeof(double) * H.P_val.size());
    return DC_OK;
  }
  // back to the caller's numbering: row u = device row dev_of[u], columns relabelled and sorted
  row_ptr[0] = 0;
  std::vector<std::pair<int, double>> row;
  for (int u = 0; u < H.N; u++) {
    const int i = c->dev_of[u];
    row.clear();
    for (int k = H.P_ptr[i]; k < H.P_ptr[i + 1]; k++) row.push_back({c->user_of[H.P_col[k]], H.P_val[k]});
    std::sort(row.begin(), row.end());
    for (size_t k = 0; k < row.size(); k++) { col[row_ptr[u] + k] = row[k].first; val[row_ptr[u] + k] = row[k].second; }
    row_ptr[u + 1] = row_ptr[u] + (int) row.size();
  }
  return DC_OK;
}
int dc_get_vertex_data(const dc_ctx *c, double *mass, double *area, double *radii) {
  if (!c || !c->built) return DC_ERR_STATE;
  const int N = c->host.N;
  for (int i = 0; i < N; i++) {
    const int u = c->user_of.empty() ? i : c->user_of[i];
    if (mass) mass[u] = c->host.mass[i];
    if (area) area[u] = c->host.area[i];
    if (radii) radii[u] = c->host.radii[i];
  }
  return DC_OK;
}

int dc_alloc_batch(dc_ctx *c, int B, int tape) {
  if (!c) return DC_ERR_INVALID;
  if (c->host_only) return fail(c, DC_ERR_STATE, "host-only context (dc_create(-1)): no device, and there is no CPU compute path");
  if (!c->built) return fail(c, DC_ERR_STATE, "dc_alloc_batch: dc_build has not been called");
  if (B <= 0 || tape <= 0) return fail(c, DC_ERR_INVALID, "dc_alloc_batch: batch and tape_steps must be > 0");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  free_pool(c->batch_allocs);
  c->B = B; c->tape = tape; c->start_slot = 0;
  const int N = c->host.N, Af = (int) c->host.att_vertex.size(), NC = c->S.NC, G = c->S.ngroups;
  const size_t se = (size_t) B * 3 * N, slots = (size_t) tape + 1;
  auto &pool = c->batch_allocs;
  int rc;
  if ((rc = dev_alloc(c, pool, &c->X, se * slots))) return rc;
  if ((rc = dev_alloc(c, pool, &c->V, se * slots))) return rc;
  if ((rc = dev_alloc(c, pool, &c->F, se * slots))) return rc;
  if ((rc = dev_alloc(c, pool, &c->R, se * slots))) return rc;
  if ((rc = dev_alloc(c, pool, &c->NRM, se * slots))) return rc;
  if ((rc = dev_alloc(c, pool, &c->PRIM, (size_t) B * N * slots))) return rc;
  if ((rc = dev_alloc(c, pool, &c->W.g, se))) return rc;
  if ((rc = dev_alloc(c, pool, &c->W.vnow, se))) return rc;
  if ((rc = dev_alloc(c, pool, &c->W.vbest, se))) return rc;
  if ((rc = dev_alloc(c, pool, &c->W.cg_r, se))) return rc;
  if ((rc = dev_alloc(c, pool, &c->W.cg_p, se))) return rc;
  if ((rc = dev_alloc(c, pool, &c->W.cg_ap, se))) return rc;
  if ((rc = dev_alloc(c, pool, &c->W.cg_x, se))) return rc;
  if ((rc = dev_alloc(c, pool, &c->W.corner, (size_t) B * 3 * NC))) return rc;
  if ((rc = dev_alloc(c, pool, &c->W.ap4, (size_t) B * N))) return rc;
  if ((rc = dev_alloc(c, pool, &c->W.pre_p, se))) return rc;
  if ((rc = dev_alloc(c, pool, &c->W.pre_s, se))) return rc;
  if ((rc = dev_alloc(c, pool, &c->W.minv, (size_t) B * 9 * N))) return rc;
  if ((rc = dev_alloc(c, pool, &c->W.u64, se))) return rc;
  if ((rc = dev_alloc(c, pool, &c->W.r64, se))) return rc;
  if ((rc = dev_alloc(c, pool, &c->W.y64, se))) return rc;
  if ((rc = dev_alloc(c, pool, &c->W.x64, se))) return rc;
  if ((rc = dev_alloc(c, pool, &c->W.c64, (size_t) B * 3 * NC))) return rc;
  for (int k = 0; k < 6; k++) if ((rc = dev_alloc(c, pool, &c->W.k64[k], se))) return rc;
  {
    const int cap = c->S.self_cap;
    c->self_cap = cap;
    if ((rc = dev_alloc(c, pool, &c->SC_pair, (size_t) B * cap * slots))) return rc;
    if ((rc = dev_alloc(c, pool, &c->SC_nrm, (size_t) B * cap * slots))) return rc;
    if ((rc = dev_alloc(c, pool, &c->SC_d, (size_t) B * cap * slots))) return rc;
    if ((rc = dev_alloc(c, pool, &c->SC_meta, (size_t) B * kMetaStride * slots))) return rc;
    if ((rc = dev_alloc(c, pool, &c->SC_verts, (size_t) B * 2 * cap * slots))) return rc;
    if ((rc = dev_alloc(c, pool, &c->W.sd_cell, (size_t) B * N))) return rc;
    if ((rc = dev_alloc(c, pool, &c->W.sd_order, (size_t) B * N))) return rc;
    if ((rc = dev_alloc(c, pool, &c->W.sd_sx, se))) return rc;
    if ((rc = dev_alloc(c, pool, &c->W.sd_rawpair, (size_t) B * cap))) return rc;
    if ((rc = dev_alloc(c, pool, &c->W.sd_rawn, (size_t) B * cap))) return rc;
    if ((rc = dev_alloc(c, pool, &c->W.sd_tmp, (size_t) B * self_tmp_ints(cap)))) return rc;
  }
  if ((rc = dev_alloc(c, pool, &c->xf_cur, (size_t) B * 3 * Af))) return rc;
  if ((rc = dev_alloc(c, pool, &c->XF, (size_t) B * 3 * Af * slots))) return rc;
  if ((rc = dev_alloc(c, pool, &c->DPAR, (size_t) B * 8 * slots))) return rc;
  if ((rc = dev_alloc(c, pool, &c->mu, (size_t) B * G))) return rc;
  if ((rc = dev_alloc(c, pool, &c->fu, (size_t) B * 3))) return rc;
  if ((rc = dev_alloc(c, pool, &c->fv, (size_t) B * 3 * N))) return rc;
  if ((rc = dev_alloc(c, pool, &c->fv2, (size_t) B * 3 * N))) return rc;
  if ((rc = dev_alloc(c, pool, &c->GX, se))) return rc;
  if ((rc = dev_alloc(c, pool, &c->GV, se))) return rc;
  if ((rc = dev_alloc(c, pool, &c->IX, se))) return rc;
  if ((rc = dev_alloc(c, pool, &c->IV, se))) return rc;
  if ((rc = dev_alloc(c, pool, &c->DXF, (size_t) B * 3 * Af * slots))) return rc;
  if ((rc = dev_alloc(c, pool, &c->FU_S, (size_t) B * 3 * slots))) return rc;
  if ((rc = dev_alloc(c, pool, &c->FVS_S, (size_t) B * slots))) return rc;
  c->SEEDX = c->SEEDV = nullptr;
  c->INJ_X = c->INJ_F = c->INJ_N = c->INJ_SN = c->INJ_SD = nullptr; c->inj_slot = -1;
  c->YS = nullptr; c->keep_y = false;
  c->sched_xf.assign(slots + 1, 0); c->sched_fu.assign(slots + 1, 0); c->sched_fvs.assign(slots + 1, 0); c->sched_seed.assign(slots + 1, 0);
  if ((rc = dev_alloc(c, pool, &c->DMU, (size_t) B * G))) return rc;
  if ((rc = dev_alloc(c, pool, &c->target, (size_t) 3 * N))) return rc;
  if ((rc = dev_alloc(c, pool, &c->fstats, (size_t) B * slots))) return rc;
  if ((rc = dev_alloc(c, pool, &c->bstats, (size_t) B * slots))) return rc;
  c->stage_elems = se;
  for (int k = 0; k < 4; k++) if ((rc = dev_alloc(c, pool, &c->stage[k], se))) return rc;
  c->fu_set = false; c->fv_set = false; c->fv2_set = false;
  // default fixed-point targets = rest positions of the attached vertices (FixedPoint::pos = pos_rest)
  if (Af > 0) {
    std::vector<float> xf((size_t) B * 3 * Af);
    for (int b = 0; b < B; b++)
      for (int a = 0; a < Af; a++)
        for (int d = 0; d < 3; d++) xf[((size_t) b * 3 + d) * Af + a] = (float) c->host.rest[3 * c->host.att_vertex[a] + d];
    HIPCHK(c, hipMemcpy(c->xf_cur, xf.data(), xf.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  if ((rc = choose_cluster(c))) return rc;
  return dc_set_mu(c, nullptr);
}

int dc_set_mu(dc_ctx *c, const double *mu) {
  if (!c || c->B <= 0) return fail(c, DC_ERR_STATE, "dc_set_mu: no batch");
  const int G = c->S.ngroups;
  std::vector<float> m((size_t) c->B * G, 0.f);
  for (int b = 0; b < c->B; b++)
    for (int g = 0; g < G; g++) {
      double v = 0;
      if (mu) v = mu[(size_t) b * G + g];
      else for (size_t k = 0; k < c->prims.size(); k++) if (c->group_of_prim[k] == g) { v = c->prims[k].mu; break; }
      m[(size_t) b * G + g] = (float) v;
    }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(c->mu, m.data(), m.size() * sizeof(float), hipMemcpyHostToDevice));
  return DC_OK;
}

int dc_set_uniform_force(dc_ctx *c, const double *f) {
  if (!c || c->B <= 0) return fail(c, DC_ERR_STATE, "dc_set_uniform_force: no batch");
  if (!f) { c->fu_set = false; return DC_OK; }
  std::vector<float> v((size_t) c->B * 3);
  for (size_t k = 0; k < v.size(); k++) v[k] = (float) f[k];
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(c->fu, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  c->fu_set = true;
  return DC_OK;
}

int dc_set_vertex_force_field(dc_ctx *c, const double *f) {
  if (!c || c->B <= 0) return fail(c, DC_ERR_STATE, "dc_set_vertex_force_field: no batch");
  if (!f) { c->fv2_set = false; return DC_OK; }
  HIPCHK(c, hipSetDevice(c->device));
  int rc = h2d_planar(c, f, c->fv2, c->host.N, 0, true);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->fv2_set = true;
  return DC_OK;
}

int dc_set_vertex_forces(dc_ctx *c, const double *f) {
  if (!c || c->B <= 0) return fail(c, DC_ERR_STATE, "dc_set_vertex_forces: no batch");
  if (!f) { c->fv_set = false; return DC_OK; }
  HIPCHK(c, hipSetDevice(c->device));
  int rc = h2d_planar(c, f, c->fv, c->host.N, 0, true);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->fv_set = true;
  return DC_OK;
}

int dc_get_force_gradient(dc_ctx *c, double *dL_df) {
  int rc = check_batch(c, 0, 0);
  if (rc) return rc;
  if (!dL_df) return fail(c, DC_ERR_INVALID, "dc_get_force_gradient: null output");
  // the adjoint kernel leaves y = (I + dr_df)^T u* of its last step in the work vector it shares with the forward kernel
  if ((rc = d2h_planar(c, c->W.vbest, dL_df, c->host.N, 0, true))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const double h2 = c->params.time_step * c->params.time_step;
  const size_t n = (size_t) c->B * 3 * c->host.N;
  for (size_t k = 0; k < n; k++) dL_df[k] *= h2;
  return DC_OK;
}

int dc_keep_force_gradients(dc_ctx *c, int keep) {
  int rc = check_batch(c, 0, 0);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  if (keep && !c->YS && (rc = dev_alloc(c, c->batch_allocs, &c->YS, slot_elems(c) * ((size_t) c->tape + 1)))) return rc;
  c->keep_y = keep != 0;
  return DC_OK;
}

int dc_get_force_gradients(dc_ctx *c, int slot0, int nslots, double *dL_df) {
  int rc = check_batch(c, slot0, slot0 + nslots - 1);
  if (rc) return rc;
  if (slot0 < 1 || nslots < 1 || !dL_df) return fail(c, DC_ERR_INVALID, "dc_get_force_gradients: slot 0 has no record");
  if (!c->YS) return fail(c, DC_ERR_STATE, "dc_get_force_gradients: dc_keep_force_gradients(1) was not set before the backward sweep");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t se = slot_elems(c);
  const double h2 = c->params.time_step * c->params.time_step;
  for (int k = 0; k < nslots; k++) {
    if ((rc = d2h_planar(c, c->YS + se * (slot0 + k), dL_df + se * k, c->host.N, k & 3, true))) return rc;
    if ((k & 3) == 3) HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (size_t q = 0; q < se * (size_t) nslots; q++) dL_df[q] *= h2;
  return cluster_check(c);
}

int dc_set_state(dc_ctx *c, int slot, const double *x, const double *v) {
  int rc = check_batch(c, slot, slot);
  if (rc) return rc;
  if (!x || !v) return fail(c, DC_ERR_INVALID, "dc_set_state: null state");
  HIPCHK(c, hipSetDevice(c->device));
  if (slot == c->inj_slot || slot + 1 == c->inj_slot) c->inj_slot = -1;      // a record handed in with dc_set_record described the state overwritten here
  const size_t se = slot_elems(c);
  if ((rc = h2d_planar(c, x, c->X + se * slot, c->host.N, 0, true))) return rc;
  if ((rc = h2d_planar(c, v, c->V + se * slot, c->host.N, 1, true))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));   // host buffers may be reused by the caller
  return DC_OK;
}

int dc_get_state(dc_ctx *c, int slot, double *x, double *v) {
  int rc = check_batch(c, slot, slot);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t se = slot_elems(c);
  if (x && (rc = d2h_planar(c, c->X + se * slot, x, c->host.N, 0, true))) return rc;
  if (v && (rc = d2h_planar(c, c->V + se * slot, v, c->host.N, 1, true))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return cluster_check(c);          // a state written by split kernels whose exchange timed out is not a state
}

int dc_step_forward(dc_ctx *c, int slot, const double *fixed_pts, dc_step_stats *stats) {
  int rc = check_batch(c, slot, slot + 1);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  const int Af = c->S.Af;
  if (fixed_pts && Af > 0) {
    if ((rc = h2d_planar(c, fixed_pts, c->xf_cur, Af, 2, false))) return rc;
    c->sched_xf[slot + 1] = 0;               // explicit targets win over a schedule entry of this step
  }
  if (Af > 0 && !c->sched_xf[slot + 1]) HIPCHK(c, hipMemcpyAsync(c->XF + (size_t) c->B * 3 * Af * (slot + 1), c->xf_cur, sizeof(float) * c->B * 3 * Af, hipMemcpyDeviceToDevice, c->stream));
  if (c->S.contact_enabled && c->S.self_enabled) launch_self_detect(c->S, c->W, fwd_args(c, slot), c->B, c->stream);
  if ((rc = cluster_begin(c))) return rc;
  if ((rc = enqueue_pd_step(c, fwd_args(c, slot)))) return rc;
  if (stats) {
    HIPCHK(c, hipMemcpyAsync(stats, c->fstats + (size_t) c->B * (slot + 1), sizeof(dc_step_stats) * c->B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if ((rc = cluster_check(c))) return rc;
    if ((rc = check_self_overflow(c, stats, slot + 1))) return rc;
  } else if (fixed_pts && Af > 0) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return DC_OK;
}

int dc_get_record(dc_ctx *c, int slot, double *f, double *r) {
  int rc = check_batch(c, slot, slot);
  if (rc) return rc;
  if (slot < 1) return fail(c, DC_ERR_INVALID, "dc_get_record: slot 0 has no record");
  const size_t se = slot_elems(c);
  if (f && (rc = d2h_planar(c, c->F + se * slot, f, c->host.N, 0, true))) return rc;
  if (r && (rc = d2h_planar(c, c->R + se * slot, r, c->host.N, 1, true))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return cluster_check(c);
}

int dc_get_contacts(dc_ctx *c, int slot, int *prim_group, double *normal) {
  int rc = check_batch(c, slot, slot);
  if (rc) return rc;
  if (slot < 1) return fail(c, DC_ERR_INVALID, "dc_get_contacts: slot 0 has no record");
  const size_t se = slot_elems(c), sp = (size_t) c->B * c->host.N;
  if (normal && (rc = d2h_planar(c, c->NRM + se * slot, normal, c->host.N, 0, true))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (prim_group) {
    std::vector<int> dev(sp);
    HIPCHK(c, hipMemcpy(dev.data(), c->PRIM + sp * slot, sp * sizeof(int), hipMemcpyDeviceToHost));
    const int N = c->host.N;
    for (int b = 0; b < c->B; b++)
      for (int i = 0; i < N; i++) {
        const int g = dev[(size_t) b * N + i];
        prim_group[(size_t) b * N + (c->user_of.empty() ? i : c->user_of[i])] = g >= 0 ? c->prims[g].group : g;
      }
  }
  return DC_OK;
}

int dc_get_self_contacts(dc_ctx *c, int slot, int rollout, int cap, int *count, int *num_layers, int *pairs, int *layer, double *normal) {
  int rc = check_batch(c, slot, slot);
  if (rc) return rc;
  if (slot < 1 || rollout < 0 || rollout >= c->B) return fail(c, DC_ERR_INVALID, "dc_get_self_contacts: bad slot / rollout");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<int> meta_v(kMetaStride); int *meta = meta_v.data();
  HIPCHK(c, hipMemcpy(meta, c->SC_meta + ((size_t) c->B * slot + rollout) * kMetaStride, sizeof(int) * kMetaStride, hipMemcpyDeviceToHost));
  const int C = meta[0], nl = meta[1];
  if (count) *count = C;
  if (num_layers) *num_layers = nl;
  const int n = std::min(C, cap);
  if (n <= 0) return DC_OK;
  const size_t base = ((size_t) c->B * slot + rollout) * c->self_cap;
  std::vector<int2> pr(n);
  std::vector<float4> nr(n);
  HIPCHK(c, hipMemcpy(pr.data(), c->SC_pair + base, sizeof(int2) * n, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(nr.data(), c->SC_nrm + base, sizeof(float4) * n, hipMemcpyDeviceToHost));
  for (int k = 0; k < n; k++) {
    if (pairs) {
      pairs[2 * k] = c->user_of.empty() ? pr[k].x : c->user_of[pr[k].x];
      pairs[2 * k + 1] = c->user_of.empty() ? pr[k].y : c->user_of[pr[k].y];
    }
    if (normal) { normal[3 * k] = nr[k].x; normal[3 * k + 1] = nr[k].y; normal[3 * k + 2] = nr[k].z; }
    if (layer) { int l = 0; while (l + 1 < nl && k >= meta[2 + l + 1]) l++; layer[k] = l; }
  }
  return DC_OK;
}

int dc_set_record(dc_ctx *c, int slot, const dc_record *rec) {
  int rc = check_batch(c, slot, slot);
  if (rc) return rc;
  if (slot < 1) return fail(c, DC_ERR_INVALID, "dc_set_record: slot 0 has no record");
  if (!rec || !rec->x || !rec->v || !rec->f || !rec->prim || !rec->normal) return fail(c, DC_ERR_INVALID, "dc_set_record: x, v, f, prim and normal are required");
  if (rec->self_count && (!rec->self_pairs || !rec->self_layer || !rec->self_normal || !rec->self_d)) return fail(c, DC_ERR_INVALID, "dc_set_record: incomplete self-contact lists");
  HIPCHK(c, hipSetDevice(c->device));
  const int N = c->host.N, B = c->B, Af = c->S.Af, cap = c->self_cap, np = (int) c->prims.size();
  const size_t se = slot_elems(c), sp = (size_t) B * N;
  if (!c->INJ_X) {
    if ((rc = dev_alloc(c, c->batch_allocs, &c->INJ_X, se))) return rc;
    if ((rc = dev_alloc(c, c->batch_allocs, &c->INJ_F, se))) return rc;
    if ((rc = dev_alloc(c, c->batch_allocs, &c->INJ_N, se))) return rc;
    if ((rc = dev_alloc(c, c->batch_allocs, &c->INJ_SN, (size_t) B * cap * 3))) return rc;
    if ((rc = dev_alloc(c, c->batch_allocs, &c->INJ_SD, (size_t) B * cap * 3))) return rc;
  }
  c->inj_slot = -1;
  // per-vertex part: fp32 tape entries + fp64 planes (device numbering)
  struct { const double *src; float *dst32; double *dst64; } planes[5] = {
      {rec->x, c->X + se * slot, c->INJ_X}, {rec->v, c->V + se * slot, nullptr}, {rec->f, c->F + se * slot, c->INJ_F},
      {rec->r, c->R + se * slot, nullptr}, {rec->normal, c->NRM + se * slot, c->INJ_N}};
  for (auto &pl : planes) {
    if (!pl.src) continue;
    HIPCHK(c, hipMemcpyAsync(c->stage[0], pl.src, se * sizeof(double), hipMemcpyHostToDevice, c->stream));
    launch_f64i_to_f32p(c->stage[0], pl.dst32, B, N, c->d_user_of, c->stream);
    if (pl.dst64) launch_f64i_to_f64p(c->stage[0], pl.dst64, B, N, c->d_user_of, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  {
    std::vector<int> prim(sp);
    for (int b = 0; b < B; b++)
      for (int i = 0; i < N; i++) {
        const int q = rec->prim[(size_t) b * N + (c->user_of.empty() ? i : c->user_of[i])];
        if (q >= np) return fail(c, DC_ERR_INVALID, "dc_set_record: primitive index out of range");
        prim[(size_t) b * N + i] = q < 0 ? -1 : q;
      }
    HIPCHK(c, hipMemcpy(c->PRIM + sp * slot, prim.data(), sp * sizeof(int), hipMemcpyHostToDevice));
  }
  if (rec->x_fixed && Af > 0) {
    if ((rc = h2d_planar(c, rec->x_fixed, c->XF + (size_t) B * 3 * Af * slot, Af, 2, false))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  {  // self contacts: the record layout the detection kernel leaves (dc_selflib.h): contacts by layer, pairs in device numbering (.x = the
     // caller's smaller id), working-set slots = rank of the caller's ids among the contact vertices, offsets and counts in the meta block
    std::vector<int> meta((size_t) B * kMetaStride, 0), verts((size_t) B * 2 * cap, 0);
    std::vector<int2> pair((size_t) B * cap, make_int2(0, 0));
    std::vector<float4> nrm((size_t) B * cap, make_float4(0, 0, 0, 0)), dv((size_t) B * cap, make_float4(0, 0, 0, 0));
    std::vector<double> sn((size_t) B * cap * 3, 0.0), sd((size_t) B * cap * 3, 0.0);
    size_t at = 0;
    for (int b = 0; b < B && rec->self_count; b++) {
      const int C = rec->self_count[b];
      if (C < 0 || C > cap) return fail(c, DC_ERR_CAPACITY, "dc_set_record: more self contacts than max_self_contacts = " + std::to_string(cap));
      int *m = meta.data() + (size_t) b * kMetaStride;
      std::vector<int> ids, in_layer((size_t) N, -1);      // (in_layer: the last layer a vertex appeared in)
      int nl = 0;
      for (int k = 0; k < C; k++) {
        const int p1 = rec->self_pairs[2 * (at + k)], p2 = rec->self_pairs[2 * (at + k) + 1], l = rec->self_layer[at + k];
        if (p1 < 0 || p2 >= N || p1 >= p2) return fail(c, DC_ERR_INVALID, "dc_set_record: self contact pair must satisfy 0 <= id1 < id2 < N");
        if (l < 0 || l >= kMaxLayers || (k > 0 && l < rec->self_layer[at + k - 1])) return fail(c, DC_ERR_INVALID, "dc_set_record: self contacts must come in layer order");
        // the contacts of a layer are applied in parallel (Simulation::contactSorting, Simulation.cpp:422-624, builds them vertex-disjoint)
        if (in_layer[p1] == l || in_layer[p2] == l) return fail(c, DC_ERR_INVALID, "dc_set_record: rollout " + std::to_string(b) + ": a vertex appears twice in self-contact layer " + std::to_string(l));
        in_layer[p1] = in_layer[p2] = l;
        nl = std::max(nl, l + 1);
        ids.push_back(p1); ids.push_back(p2);
      }
      std::sort(ids.begin(), ids.end());
      ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
      const int M = (int) ids.size();
      m[0] = C; m[1] = C > 0 ? nl : 0;
      for (int k = 0; k < C; k++) m[2 + rec->self_layer[at + k] + 1]++;
      for (int l = 0; l < nl; l++) m[2 + l + 1] += m[2 + l];
      m[kMetaStride - 1] = M; m[kMetaStride - 2] = 0; m[kMetaStride - 3] = C;
      for (int q = 0; q < M; q++) verts[(size_t) b * 2 * cap + q] = c->dev_of.empty() ? ids[q] : c->dev_of[ids[q]];
      for (int k = 0; k < C; k++) {
        const int p1 = rec->self_pairs[2 * (at + k)], p2 = rec->self_pairs[2 * (at + k) + 1];
        const int s1 = (int) (std::lower_bound(ids.begin(), ids.end(), p1) - ids.begin()), s2 = (int) (std::lower_bound(ids.begin(), ids.end(), p2) - ids.begin());
        const size_t o = (size_t) b * cap + k;
        pair[o] = make_int2(c->dev_of.empty() ? p1 : c->dev_of[p1], c->dev_of.empty() ? p2 : c->dev_of[p2]);
        const int slots = s1 | (s2 << 16);
        float w; std::memcpy(&w, &slots, sizeof(float));
        nrm[o] = make_float4((float) rec->self_normal[3 * (at + k)], (float) rec->self_normal[3 * (at + k) + 1], (float) rec->self_normal[3 * (at + k) + 2], w);
        dv[o] = make_float4((float) rec->self_d[3 * (at + k)], (float) rec->self_d[3 * (at + k) + 1], (float) rec->self_d[3 * (at + k) + 2], 0.f);
        for (int d = 0; d < 3; d++) { sn[3 * o + d] = rec->self_normal[3 * (at + k) + d]; sd[3 * o + d] = rec->self_d[3 * (at + k) + d]; }
      }
      at += C;
    }
    const size_t sc = (size_t) B * cap * slot;
    HIPCHK(c, hipMemcpy(c->SC_meta + (size_t) B * kMetaStride * slot, meta.data(), meta.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->SC_pair + sc, pair.data(), pair.size() * sizeof(int2), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->SC_nrm + sc, nrm.data(), nrm.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->SC_d + sc, dv.data(), dv.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->SC_verts + 2 * sc, verts.data(), verts.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->INJ_SN, sn.data(), sn.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->INJ_SD, sd.data(), sd.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  c->inj_slot = slot;
  return DC_OK;
}

int dc_step_backward(dc_ctx *c, int slot, const double *dL_dxnew, const double *dL_dvnew, const double *dL_dxinit,
                     const double *dL_dvinit, int is_start, double *dL_dx, double *dL_dv, double *dL_dxfixed,
                     double *dL_dmu, dc_bwd_stats *stats) {
  int rc = check_batch(c, slot, slot);
  if (rc) return rc;
  if (slot < 1) return fail(c, DC_ERR_INVALID, "dc_step_backward: slot 0 has no record");
  if (!dL_dxnew || !dL_dvnew || !dL_dx || !dL_dv) return fail(c, DC_ERR_INVALID, "dc_step_backward: null gradient");
  HIPCHK(c, hipSetDevice(c->device));
  const int N = c->host.N, Af = c->S.Af, G = c->S.ngroups;
  if ((rc = h2d_planar(c, dL_dxnew, c->GX, N, 0, true))) return rc;
  if ((rc = h2d_planar(c, dL_dvnew, c->GV, N, 1, true))) return rc;
  const bool with_init = dL_dxinit && dL_dvinit;
  if (with_init) {
    if ((rc = h2d_planar(c, dL_dxinit, c->IX, N, 2, true))) return rc;
    if ((rc = h2d_planar(c, dL_dvinit, c->IV, N, 3, true))) return rc;
  }
  HIPCHK(c, hipMemsetAsync(c->DMU, 0, sizeof(float) * c->B * G, c->stream));
  if (Af > 0) HIPCHK(c, hipMemsetAsync(c->DXF + (size_t) c->B * 3 * Af * slot, 0, sizeof(float) * c->B * 3 * Af, c->stream));
  if ((rc = cluster_begin(c))) return rc;
  {
    BwdArgs BA = bwd_args(c, slot, is_start != 0, with_init);
    if (!with_init) { BA.ix = nullptr; BA.iv = nullptr; BA.slot_ix = 0; }     // the per-step call takes its seeds from its arguments only
    if ((rc = enqueue_adjoint_step(c, BA))) return rc;
  }
  if ((rc = d2h_planar(c, c->GX, dL_dx, N, 0, true))) return rc;
  if ((rc = d2h_planar(c, c->GV, dL_dv, N, 1, true))) return rc;
  if (dL_dxfixed && Af > 0 && (rc = d2h_planar(c, c->DXF + (size_t) c->B * 3 * Af * slot, dL_dxfixed, Af, 2, false))) return rc;
  std::vector<float> dmu((size_t) c->B * G);
  HIPCHK(c, hipMemcpyAsync(dmu.data(), c->DMU, dmu.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (stats) HIPCHK(c, hipMemcpyAsync(stats, c->bstats + (size_t) c->B * slot, sizeof(dc_bwd_stats) * c->B, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if ((rc = cluster_check(c))) return rc;
  if (dL_dmu) for (size_t k = 0; k < dmu.size(); k++) dL_dmu[k] = dmu[k];
  return DC_OK;
}

// ---- device-pointer boundary: the per-step calls for callers whose tensors live on this GPU
//      (the RL / controller-training loop: functional.py:20-102, hatController.py:78-105 run one stepNN + stepBackwardNN per step).
//      Buffers are DEVICE pointers in the caller's layout — xyz interleaved, B rollouts concatenated, fp32 (is_f32 = 1, torch's default)
//      or fp64 — converted on the device; nothing crosses PCIe and nothing synchronises: every call is enqueued on the context's stream.
int dc_use_stream(dc_ctx *c, void *hip_stream) {
  if (!c || c->host_only) return fail(c, DC_ERR_STATE, "dc_use_stream: no device context");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->stream = hip_stream ? (hipStream_t) hip_stream : c->own_stream;
  return DC_OK;
}

int dc_set_state_dev(dc_ctx *c, int slot, const void *d_x, const void *d_v, int is_f32) {
  int rc = check_batch(c, slot, slot);
  if (rc) return rc;
  if (!d_x || !d_v) return fail(c, DC_ERR_INVALID, "dc_set_state_dev: null state");
  HIPCHK(c, hipSetDevice(c->device));
  if (slot == c->inj_slot || slot + 1 == c->inj_slot) c->inj_slot = -1;      // (see dc_set_state)
  const size_t se = slot_elems(c);
  launch_dev_to_planar(d_x, is_f32, c->X + se * slot, c->B, c->host.N, c->d_user_of, c->stream);
  launch_dev_to_planar(d_v, is_f32, c->V + se * slot, c->B, c->host.N, c->d_user_of, c->stream);
  HIPCHK(c, hipGetLastError());
  return DC_OK;
}

int dc_get_state_dev(dc_ctx *c, int slot, void *d_x, void *d_v, int is_f32) {
  int rc = check_batch(c, slot, slot);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t se = slot_elems(c);
  if (d_x) launch_planar_to_dev(c->X + se * slot, d_x, is_f32, c->B, c->host.N, c->d_user_of, c->stream);
  if (d_v) launch_planar_to_dev(c->V + se * slot, d_v, is_f32, c->B, c->host.N, c->d_user_of, c->stream);
  HIPCHK(c, hipGetLastError());
  return DC_OK;
}

int dc_step_forward_dev(dc_ctx *c, int slot, const void *d_fixed_pts, int is_f32) {
  int rc = check_batch(c, slot, slot + 1);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  const int Af = c->S.Af;
  if (d_fixed_pts && Af > 0) {
    launch_dev_to_planar(d_fixed_pts, is_f32, c->xf_cur, c->B, Af, nullptr, c->stream);
    c->sched_xf[slot + 1] = 0;
  }
  if (Af > 0 && !c->sched_xf[slot + 1]) HIPCHK(c, hipMemcpyAsync(c->XF + (size_t) c->B * 3 * Af * (slot + 1), c->xf_cur, sizeof(float) * c->B * 3 * Af, hipMemcpyDeviceToDevice, c->stream));
  if (c->S.contact_enabled && c->S.self_enabled) launch_self_detect(c->S, c->W, fwd_args(c, slot), c->B, c->stream);
  if ((rc = cluster_begin(c))) return rc;
  return enqueue_pd_step(c, fwd_args(c, slot));
}

int dc_step_backward_dev(dc_ctx *c, int slot, const void *d_dL_dxnew, const void *d_dL_dvnew, const void *d_dL_dxinit, const void *d_dL_dvinit,
                         int is_start, void *d_dL_dx, void *d_dL_dv, void *d_dL_dxfixed, void *d_dL_dmu, int is_f32) {
  int rc = check_batch(c, slot, slot);
  if (rc) return rc;
  if (slot < 1) return fail(c, DC_ERR_INVALID, "dc_step_backward_dev: slot 0 has no record");
  if (!d_dL_dxnew || !d_dL_dvnew || !d_dL_dx || !d_dL_dv) return fail(c, DC_ERR_INVALID, "dc_step_backward_dev: null gradient");
  HIPCHK(c, hipSetDevice(c->device));
  const int N = c->host.N, Af = c->S.Af, G = c->S.ngroups;
  launch_dev_to_planar(d_dL_dxnew, is_f32, c->GX, c->B, N, c->d_user_of, c->stream);
  launch_dev_to_planar(d_dL_dvnew, is_f32, c->GV, c->B, N, c->d_user_of, c->stream);
  const bool with_init = d_dL_dxinit && d_dL_dvinit;
  if (with_init) {
    launch_dev_to_planar(d_dL_dxinit, is_f32, c->IX, c->B, N, c->d_user_of, c->stream);
    launch_dev_to_planar(d_dL_dvinit, is_f32, c->IV, c->B, N, c->d_user_of, c->stream);
  }
  HIPCHK(c, hipMemsetAsync(c->DMU, 0, sizeof(float) * c->B * G, c->stream));
  if (Af > 0) HIPCHK(c, hipMemsetAsync(c->DXF + (size_t) c->B * 3 * Af * slot, 0, sizeof(float) * c->B * 3 * Af, c->stream));
  if ((rc = cluster_begin(c))) return rc;
  {
    BwdArgs BA = bwd_args(c, slot, is_start != 0, with_init);
    if (!with_init) { BA.ix = nullptr; BA.iv = nullptr; BA.slot_ix = 0; }
    if ((rc = enqueue_adjoint_step(c, BA))) return rc;
  }
  launch_planar_to_dev(c->GX, d_dL_dx, is_f32, c->B, N, c->d_user_of, c->stream);
  launch_planar_to_dev(c->GV, d_dL_dv, is_f32, c->B, N, c->d_user_of, c->stream);
  if (d_dL_dxfixed && Af > 0) launch_planar_to_dev(c->DXF + (size_t) c->B * 3 * Af * slot, d_dL_dxfixed, is_f32, c->B, Af, nullptr, c->stream);
  if (d_dL_dmu) launch_copy_cast(c->DMU, d_dL_dmu, is_f32, (long) c->B * G, c->stream);
  HIPCHK(c, hipGetLastError());
  return DC_OK;
}

int dc_rollout_forward(dc_ctx *c, int slot, int nsteps) {
  int rc = check_batch(c, slot, slot + nsteps);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventRecord(c->ev_a, c->stream));
  const bool self_on = c->S.contact_enabled && c->S.self_enabled;
  static const bool fuse_ok = !(getenv("DC_FUSE_STEPS") && getenv("DC_FUSE_STEPS")[0] == '0');     // development switch
  const bool fused = fuse_ok && nsteps > 1 && (use_cluster_fwd(c) || pd_step_fusable(c->S));
  if ((rc = cluster_begin(c))) return rc;
  if (fused) {
    // all steps of a rollout run inside ONE launch (self-collision detection inlined per step), so a rollout never waits
    // for the slowest rollout of the batch between steps
    // a schedule (dc_set_*_schedule) has to cover all steps of a fused sweep or none of them
    int nxf = 0, nfu = 0, nfvs = 0;
    for (int k = 1; k <= nsteps; k++) { nxf += c->sched_xf[slot + k]; nfu += c->sched_fu[slot + k]; nfvs += c->sched_fvs[slot + k]; }
    if ((nxf % nsteps) || (nfu % nsteps) || (nfvs % nsteps))
      return fail(c, DC_ERR_INVALID, "dc_rollout_forward: a fixed-point / force schedule covers only part of the steps " + std::to_string(slot) + " .. " + std::to_string(slot + nsteps));
    for (int k = 0; k < nsteps && c->S.Af > 0 && nxf == 0; k++)
      HIPCHK(c, hipMemcpyAsync(c->XF + (size_t) c->B * 3 * c->S.Af * (slot + k + 1), c->xf_cur, sizeof(float) * c->B * 3 * c->S.Af, hipMemcpyDeviceToDevice, c->stream));
    FwdArgs A = fwd_args(c, slot);           // (points x_fixed / fu / fv_scale at the first step's schedule entries)
    if (nxf) A.slot_xfix = (size_t) c->B * 3 * c->S.Af;
    if (nfu) A.slot_fu = (size_t) c->B * 3;
    if (nfvs && A.fv_scale) A.slot_fvs = (size_t) c->B;
    A.nsteps = nsteps; A.inline_detect = self_on ? 1 : 0;
    if ((rc = enqueue_pd_step(c, A))) return rc;
  } else {
    for (int k = 0; k < nsteps; k++) {
      if (c->S.Af > 0 && !c->sched_xf[slot + k + 1]) HIPCHK(c, hipMemcpyAsync(c->XF + (size_t) c->B * 3 * c->S.Af * (slot + k + 1), c->xf_cur, sizeof(float) * c->B * 3 * c->S.Af, hipMemcpyDeviceToDevice, c->stream));
      if (self_on) launch_self_detect(c->S, c->W, fwd_args(c, slot + k), c->B, c->stream);
      if ((rc = enqueue_pd_step(c, fwd_args(c, slot + k)))) return rc;
    }
  }
  if (c->S.Af > 0 && c->sched_xf[slot + nsteps])      // later unscheduled steps continue from the last scheduled targets
    HIPCHK(c, hipMemcpyAsync(c->xf_cur, c->XF + (size_t) c->B * 3 * c->S.Af * (slot + nsteps), sizeof(float) * c->B * 3 * c->S.Af, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipEventRecord(c->ev_b, c->stream));
  HIPCHK(c, hipGetLastError());
  // kernel-time accounting is resolved lazily in dc_kernel_times / dc_sync
  HIPCHK(c, hipEventSynchronize(c->ev_b));
  float ms = 0;
  HIPCHK(c, hipEventElapsedTime(&ms, c->ev_a, c->ev_b));
  const int chunks = use_cluster_fwd(c) ? (c->B + c->cl.nb - 1) / c->cl.nb : 1;
  c->fwd_ms += ms; c->fwd_launches += (fused ? 1 : nsteps) * chunks;
  return cluster_check(c);
}

int dc_seed_gradient(dc_ctx *c, int slot, const double *target, double scale_x) {
  int rc = check_batch(c, slot, slot);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  const int N = c->host.N;
  std::vector<float> t(3 * (size_t) N);
  for (int i = 0; i < N; i++)
    for (int d = 0; d < 3; d++)
      t[(size_t) d * N + i] = (float) (target ? target[3 * (size_t) (c->user_of.empty() ? i : c->user_of[i]) + d] : c->host.rest[3 * i + d]);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(c->target, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
  launch_seed_gradient(c->X + slot_elems(c) * slot, c->target, c->GX, c->GV, c->B, N, (float) scale_x, c->stream);
  HIPCHK(c, hipMemsetAsync(c->DMU, 0, sizeof(float) * c->B * c->S.ngroups, c->stream));
  return DC_OK;
}

int dc_rollout_backward(dc_ctx *c, int slot, int nsteps) {
  int rc = check_batch(c, slot - nsteps + 1, slot);
  if (rc) return rc;
  if (slot - nsteps + 1 < 1) return fail(c, DC_ERR_INVALID, "dc_rollout_backward: would run past slot 1");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventRecord(c->ev_a, c->stream));
  static const bool fuse_ok = !(getenv("DC_FUSE_STEPS") && getenv("DC_FUSE_STEPS")[0] == '0');     // development switch
  if ((rc = cluster_begin(c))) return rc;
  if (c->SEEDX) {
    int ns = 0;
    for (int k = 0; k < nsteps; k++) ns += c->sched_seed[slot - k - 1];
    if (ns % nsteps) return fail(c, DC_ERR_INVALID, "dc_rollout_backward: the seed schedule covers only part of the slots " + std::to_string(slot - nsteps) + " .. " + std::to_string(slot - 1));
  }
  if (c->S.Af > 0) HIPCHK(c, hipMemsetAsync(c->DXF + (size_t) c->B * 3 * c->S.Af * (slot - nsteps + 1), 0, sizeof(float) * c->B * 3 * c->S.Af * nsteps, c->stream));
  const bool inj_inside = c->inj_slot >= slot - nsteps + 1 && c->inj_slot <= slot;      // (dc_set_record: that step gets a launch of its own)
  const bool fused_bwd = fuse_ok && nsteps > 1 && !inj_inside;
  if (fused_bwd) {
    BwdArgs A = bwd_args(c, slot, slot == c->start_slot + 1, false);
    A.nsteps = nsteps;                       // the whole sweep of a rollout in one launch
    if ((rc = enqueue_adjoint_step(c, A))) return rc;
  } else {
    for (int k = 0; k < nsteps; k++) {
      const int s = slot - k;
      if ((rc = enqueue_adjoint_step(c, bwd_args(c, s, s == c->start_slot + 1, false)))) return rc;   // isStart: Simulation.cpp:3947
    }
  }
  HIPCHK(c, hipEventRecord(c->ev_b, c->stream));
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventSynchronize(c->ev_b));
  float ms = 0;
  HIPCHK(c, hipEventElapsedTime(&ms, c->ev_a, c->ev_b));
  const int chunks = use_cluster_bwd(c) ? (c->B + c->cl.nb - 1) / c->cl.nb : 1;
  c->bwd_ms += ms; c->bwd_launches += (fused_bwd ? 1 : nsteps) * chunks;
  return cluster_check(c);
}

int dc_set_trajectory_start(dc_ctx *c, int start_slot) {
  if (!c) return DC_ERR_INVALID;
  if (start_slot < -1) return fail(c, DC_ERR_INVALID, "dc_set_trajectory_start: start_slot must be >= -1");
  c->start_slot = start_slot;
  return DC_OK;
}

int dc_get_gradient(dc_ctx *c, double *dL_dx, double *dL_dv, double *dL_dmu) {
  int rc = check_batch(c, 0, 0);
  if (rc) return rc;
  const int N = c->host.N, G = c->S.ngroups;
  if (dL_dx && (rc = d2h_planar(c, c->GX, dL_dx, N, 0, true))) return rc;
  if (dL_dv && (rc = d2h_planar(c, c->GV, dL_dv, N, 1, true))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (dL_dmu) {
    std::vector<float> dmu((size_t) c->B * G);
    HIPCHK(c, hipMemcpy(dmu.data(), c->DMU, dmu.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < dmu.size(); k++) dL_dmu[k] = dmu[k];
  }
  return DC_OK;
}

// ---- device-resident schedules of the fused rollouts (stepFixPoints / fillForces / the per-frame loss seeds
// of runBackwardTask, Simulation.cpp:55-116, 964-1018, 3938-3952, as streams on the device) ----
int dc_set_fixed_point_schedule(dc_ctx *c, int slot0, int nsteps, const double *xf) {
  int rc = check_batch(c, slot0, slot0 + nsteps);
  if (rc) return rc;
  if (nsteps < 1 || !xf) return fail(c, DC_ERR_INVALID, "dc_set_fixed_point_schedule: null schedule");
  const int Af = c->S.Af;
  if (Af <= 0) return DC_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t per = (size_t) c->B * 3 * Af;
  for (int k = 0; k < nsteps; k++) {
    if ((rc = h2d_planar(c, xf + per * k, c->XF + per * (slot0 + k + 1), Af, k & 3, false))) return rc;
    c->sched_xf[slot0 + k + 1] = 1;
    if ((k & 3) == 3) HIPCHK(c, hipStreamSynchronize(c->stream));      // the four staging buffers are reused
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return DC_OK;
}

int dc_set_force_schedule(dc_ctx *c, int slot0, int nsteps, const double *fu, const double *fv_scale) {
  int rc = check_batch(c, slot0, slot0 + nsteps);
  if (rc) return rc;
  if (nsteps < 1) return fail(c, DC_ERR_INVALID, "dc_set_force_schedule: nsteps < 1");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (fu) {
    std::vector<float> v((size_t) nsteps * c->B * 3);
    for (size_t k = 0; k < v.size(); k++) v[k] = (float) fu[k];
    HIPCHK(c, hipMemcpy(c->FU_S + (size_t) c->B * 3 * (slot0 + 1), v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  if (fv_scale) {
    std::vector<float> v((size_t) nsteps * c->B);
    for (size_t k = 0; k < v.size(); k++) v[k] = (float) fv_scale[k];
    HIPCHK(c, hipMemcpy(c->FVS_S + (size_t) c->B * (slot0 + 1), v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  for (int k = 1; k <= nsteps; k++) { c->sched_fu[slot0 + k] = fu ? 1 : 0; c->sched_fvs[slot0 + k] = fv_scale ? 1 : 0; }
  return DC_OK;
}

int dc_set_seed_schedule(dc_ctx *c, int slot0, int nslots, const double *dL_dx, const double *dL_dv) {
  int rc = check_batch(c, slot0, slot0 + nslots - 1);
  if (rc) return rc;
  if (nslots < 1 || !dL_dx) return fail(c, DC_ERR_INVALID, "dc_set_seed_schedule: null schedule");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t se = slot_elems(c), slots = (size_t) c->tape + 1;
  if (!c->SEEDX) {
    if ((rc = dev_alloc(c, c->batch_allocs, &c->SEEDX, se * slots))) return rc;
    if ((rc = dev_alloc(c, c->batch_allocs, &c->SEEDV, se * slots))) return rc;
  }
  const int N = c->host.N;
  for (int k = 0; k < nslots; k++) {
    if ((rc = h2d_planar(c, dL_dx + (size_t) 3 * N * c->B * k, c->SEEDX + se * (slot0 + k), N, 0, true))) return rc;
    if (dL_dv) { if ((rc = h2d_planar(c, dL_dv + (size_t) 3 * N * c->B * k, c->SEEDV + se * (slot0 + k), N, 1, true))) return rc; }
    else HIPCHK(c, hipMemsetAsync(c->SEEDV + se * (slot0 + k), 0, se * sizeof(float), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->sched_seed[slot0 + k] = 1;
  }
  return DC_OK;
}

int dc_clear_schedules(dc_ctx *c) {
  if (!c || c->B <= 0) return fail(c, DC_ERR_STATE, "dc_clear_schedules: no batch");
  std::fill(c->sched_xf.begin(), c->sched_xf.end(), 0); std::fill(c->sched_fu.begin(), c->sched_fu.end(), 0);
  std::fill(c->sched_fvs.begin(), c->sched_fvs.end(), 0); std::fill(c->sched_seed.begin(), c->sched_seed.end(), 0);
  return DC_OK;
}

int dc_set_gradient(dc_ctx *c, const double *dL_dx, const double *dL_dv) {
  int rc = check_batch(c, 0, 0);
  if (rc) return rc;
  if (!dL_dx || !dL_dv) return fail(c, DC_ERR_INVALID, "dc_set_gradient: null gradient");
  HIPCHK(c, hipSetDevice(c->device));
  if ((rc = h2d_planar(c, dL_dx, c->GX, c->host.N, 0, true))) return rc;
  if ((rc = h2d_planar(c, dL_dv, c->GV, c->host.N, 1, true))) return rc;
  HIPCHK(c, hipMemsetAsync(c->DMU, 0, sizeof(float) * c->B * c->S.ngroups, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return DC_OK;
}

int dc_get_states(dc_ctx *c, int slot0, int nslots, double *x, double *v) {
  int rc = check_batch(c, slot0, slot0 + nslots - 1);
  if (rc) return rc;
  if (nslots < 1) return fail(c, DC_ERR_INVALID, "dc_get_states: nslots < 1");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t se = slot_elems(c);
  for (int k = 0; k < nslots; k++) {
    if (x && (rc = d2h_planar(c, c->X + se * (slot0 + k), x + se * k, c->host.N, 0, true))) return rc;
    if (v && (rc = d2h_planar(c, c->V + se * (slot0 + k), v + se * k, c->host.N, 1, true))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return DC_OK;
}

int dc_get_dxfixed(dc_ctx *c, int slot0, int nslots, double *dL_dxfixed) {
  int rc = check_batch(c, slot0, slot0 + nslots - 1);
  if (rc) return rc;
  if (slot0 < 1 || nslots < 1 || !dL_dxfixed) return fail(c, DC_ERR_INVALID, "dc_get_dxfixed: slot 0 has no record");
  const int Af = c->S.Af;
  if (Af <= 0) return DC_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t per = (size_t) c->B * 3 * Af;
  for (int k = 0; k < nslots; k++) {
    if ((rc = d2h_planar(c, c->DXF + per * (slot0 + k), dL_dxfixed + per * k, Af, 2, false))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return DC_OK;
}

int dc_get_param_gradients(dc_ctx *c, int slot, double *out) {
  int rc = check_batch(c, slot, slot);
  if (rc) return rc;
  if (slot < 1 || !out) return fail(c, DC_ERR_INVALID, "dc_get_param_gradients: slot 0 has no record");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<float> v((size_t) c->B * 8);
  HIPCHK(c, hipMemcpy(v.data(), c->DPAR + (size_t) c->B * 8 * slot, v.size() * sizeof(float), hipMemcpyDeviceToHost));
  for (size_t k = 0; k < v.size(); k++) out[k] = v[k];
  return DC_OK;
}

int dc_get_stats(dc_ctx *c, int slot, dc_step_stats *fwd, dc_bwd_stats *bwd) {
  int rc = check_batch(c, slot, slot);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (fwd) HIPCHK(c, hipMemcpy(fwd, c->fstats + (size_t) c->B * slot, sizeof(dc_step_stats) * c->B, hipMemcpyDeviceToHost));
  if (fwd && slot >= 1 && (rc = check_self_overflow(c, fwd, slot))) return rc;
  if (bwd) HIPCHK(c, hipMemcpy(bwd, c->bstats + (size_t) c->B * slot, sizeof(dc_bwd_stats) * c->B, hipMemcpyDeviceToHost));
  return DC_OK;
}

int dc_sync(dc_ctx *c) {
  if (!c) return DC_ERR_INVALID;
  if (c->host_only) return DC_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return cluster_check(c);
}

int dc_get_deflation(const dc_ctx *c, int *vectors, int *probe_iterations) {
  if (!c) return DC_ERR_INVALID;
  if (!c->built) return DC_ERR_STATE;
  if (vectors) *vectors = c->defl_k;
  if (probe_iterations) *probe_iterations = c->defl_probe;
  return DC_OK;
}

int dc_get_layout(const dc_ctx *c, int *out6) {
  if (!c || !out6) return DC_ERR_INVALID;
  if (!c->built) return DC_ERR_STATE;
  out6[0] = c->user_of.empty() ? 0 : 1; out6[1] = c->bandwidth; out6[2] = c->S.pk_ok; out6[3] = c->S.win_ok; out6[4] = c->S.nwin;
  out6[5] = c->S.dense_inv ? 1 : 0;
  return DC_OK;
}

// ---- collective for C++ callers (the L-BFGS side sums loss + parameter gradients over the ranks) ------------------
// RCCL is bound at run time (dlopen) the first time one of these entry points is used: the library has no link-time dependency on it,
// and a process that already carries an RCCL (torch.distributed) gets that same copy by its soname.
extern "C++" {
namespace {
struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};
RcclApi &rccl() {
  static RcclApi api;
  if (api.lib || !api.error.empty()) return api;
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (api.lib) break;
  }
  if (!api.lib) {
    const char *why = dlerror();      // (one call: dlerror() clears the message it returns)
    api.error = std::string("RCCL not found (dlopen librccl.so.1): ") + (why ? why : "");
    return api;
  }
  api.GetUniqueId = (decltype(api.GetUniqueId)) dlsym(api.lib, "ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank)) dlsym(api.lib, "ncclCommInitRank");
  api.AllReduce = (decltype(api.AllReduce)) dlsym(api.lib, "ncclAllReduce");
  api.CommDestroy = (decltype(api.CommDestroy)) dlsym(api.lib, "ncclCommDestroy");
  api.GetErrorString = (decltype(api.GetErrorString)) dlsym(api.lib, "ncclGetErrorString");
  if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) { api.error = "RCCL library lacks the expected entry points"; api.lib = nullptr; }
  return api;
}
int rccl_fail(dc_ctx *c, const char *what, ncclResult_t r) {
  RcclApi &R = rccl();
  return fail(c, DC_ERR_HIP, std::string(what) + ": " + (R.GetErrorString ? R.GetErrorString(r) : "RCCL error") + " (" + std::to_string((int) r) + ")");
}
}  // namespace
}  // extern "C++"

int dc_comm_unique_id(char *id128) {
  if (!id128) return DC_ERR_INVALID;
  RcclApi &R = rccl();
  if (!R.lib) return DC_ERR_HIP;
  ncclUniqueId id;
  if (R.GetUniqueId(&id) != ncclSuccess) return DC_ERR_HIP;
  std::memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return DC_OK;
}

int dc_comm_init(dc_ctx *c, int nranks, int rank, const char *id128) {
  if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, DC_ERR_INVALID, "dc_comm_init: bad arguments");
  if (c->host_only) return fail(c, DC_ERR_STATE, "dc_comm_init: host-only context");
  RcclApi &R = rccl();
  if (!R.lib) return fail(c, DC_ERR_HIP, R.error);
  if (c->comm) { int rc = dc_comm_destroy(c); if (rc) return rc; }
  HIPCHK(c, hipSetDevice(c->device));
  ncclUniqueId id;
  std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t comm = nullptr;
  ncclResult_t r = R.CommInitRank(&comm, nranks, id, rank);
  if (r != ncclSuccess) return rccl_fail(c, "ncclCommInitRank", r);
  c->comm = (void *) comm; c->comm_ranks = nranks;
  return DC_OK;
}

int dc_allreduce_sum(dc_ctx *c, double *inout, int count) {
  if (!c || !inout || count < 1) return fail(c, DC_ERR_INVALID, "dc_allreduce_sum: bad arguments");
  if (!c->comm) return fail(c, DC_ERR_STATE, "dc_allreduce_sum: dc_comm_init has not been called");
  RcclApi &R = rccl();
  HIPCHK(c, hipSetDevice(c->device));
  double *buf = nullptr;
  HIPCHK(c, hipMalloc((void **) &buf, sizeof(double) * (size_t) count));
  hipError_t e = hipMemcpyAsync(buf, inout, sizeof(double) * (size_t) count, hipMemcpyHostToDevice, c->stream);
  ncclResult_t r = ncclSuccess;
  if (e == hipSuccess) r = R.AllReduce(buf, buf, (size_t) count, ncclFloat64, ncclSum, (ncclComm_t) c->comm, c->stream);
  if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(inout, buf, sizeof(double) * (size_t) count, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void) hipFree(buf);
  if (r != ncclSuccess) return rccl_fail(c, "ncclAllReduce", r);
  HIPCHK(c, e);
  return DC_OK;
}

int dc_comm_destroy(dc_ctx *c) {
  if (!c) return DC_ERR_INVALID;
  if (!c->comm) return DC_OK;
  RcclApi &R = rccl();
  if (R.lib) (void) R.CommDestroy((ncclComm_t) c->comm);
  c->comm = nullptr; c->comm_ranks = 0;
  return DC_OK;
}

int dc_get_cluster(const dc_ctx *c, int *workgroups_per_rollout, int *rollouts_per_launch) {
  if (!c) return DC_ERR_INVALID;
  if (workgroups_per_rollout) *workgroups_per_rollout = c->cl.ok ? c->cl.K : 1;
  if (rollouts_per_launch) *rollouts_per_launch = c->cl.ok ? c->cl.nb : c->B;
  return DC_OK;
}

int dc_timer_start(dc_ctx *c) {
  if (!c) return DC_ERR_INVALID;
  HIPCHK(c, hipEventRecord(c->ev_t0, c->stream));
  return DC_OK;
}
int dc_timer_stop(dc_ctx *c, float *ms) {
  if (!c || !ms) return DC_ERR_INVALID;
  HIPCHK(c, hipEventRecord(c->ev_t1, c->stream));
  HIPCHK(c, hipEventSynchronize(c->ev_t1));
  HIPCHK(c, hipEventElapsedTime(ms, c->ev_t0, c->ev_t1));
  return DC_OK;
}
int dc_kernel_times(dc_ctx *c, float *fwd_ms, int *fwd_launches, float *bwd_ms, int *bwd_launches, int reset) {
  if (!c) return DC_ERR_INVALID;
  if (fwd_ms) *fwd_ms = c->fwd_ms;
  if (fwd_launches) *fwd_launches = c->fwd_launches;
  if (bwd_ms) *bwd_ms = c->bwd_ms;
  if (bwd_launches) *bwd_launches = c->bwd_launches;
  if (reset) { c->fwd_ms = c->bwd_ms = 0; c->fwd_launches = c->bwd_launches = 0; }
  return DC_OK;
}

}  // extern "C"
