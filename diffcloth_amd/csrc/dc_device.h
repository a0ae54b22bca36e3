// Device-side data descriptors shared by the C-ABI layer (dc_engine.hip) and the kernels (dc_kernels.hip).
// All device vectors are fp32, component-planar per rollout: v[b][c][i] at ((b*3 + c)*N + i).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/diffcloth_hip.h"

namespace dc {

constexpr int kMaxPrims = 8;
constexpr int kMaxLayers = 4088;   // self-contact layers per step (contactSorting: a chain of L contacts takes L layers)
// ints per (slot, rollout): count, nlayers, layer offsets[nlayers + 1] ... and at the end [kMetaStride - 3] = pairs found
// (before clamping to the list capacity), [- 2] = overflow flags (1: more pairs than max_self_contacts, 2: more layers than
// kMaxLayers), [- 1] = number of distinct vertices in the contacts
constexpr int kMetaStride = 4096;

// Table pointers of DevSystem carry the GLOBAL address space in device code: the struct itself lives in global memory
// and is read through a pointer, so the compiler cannot infer the address space of the pointers stored in it and would
// emit FLAT loads (which also tick the LDS counter and so serialise against every LDS wait) and vector loads for
// wave-uniform table entries instead of global / scalar loads.
// (only in the kernel translation units, which define DC_KERNEL_TU before including this header: the host code of
// dc_engine.hip is parsed in the device pass too and assigns plain pointers)
#if defined(__HIP_DEVICE_COMPILE__) && defined(DC_KERNEL_TU)
#define DC_G __attribute__((address_space(1)))
#define DC_C __attribute__((address_space(4)))   // constant address space: wave-uniform entries become scalar loads
#else
#define DC_G
#define DC_C
#endif

constexpr int kMaxDevices = 64;       // launchers cache per-device kernel attributes

struct DevPrim {
  int kind, group, rotates, pad;
  float cx, cy, cz, radius;
  float tx, ty, tz, length;     // capsule: top offset; plane: corner upperLeft relative to the centre
  float ux, uy, uz, pad2;       // plane: corner upperRight relative to the centre
};

// Shared (per-context) tables: topology + matrices, identical for every rollout of the batch.
struct DevSystem {
  int N, T, E, Af, NC;          // NC = 3T + 4E constraint corners
  int nprim, ngroups, pad0;
  const int DC_G *tri_v;             // [3][T]
  const float4 DC_G *tri_D;          // [T]  inv_deltaUV (d00,d01,d10,d11)
  const float DC_G *tri_w2;          // [T]  area * k_stretch
  const int DC_G *bend_v;            // [4][E]
  const float4 DC_G *bend_w;         // [E]  cotan weights
  const float2 DC_G *bend_nw;        // [E]  (rest norm n, weight^2)
  const int DC_G *att_vertex;        // [Af]
  const int DC_G *att_of_vertex;     // [N]  fixed-point index or -1
  const float DC_G *mass;            // [N]
  const float DC_G *dinv;            // [N]  1 / P_ii  (block-Jacobi preconditioner; blocks are scalar * I3)
  const int DC_G *P_ptr;             // [N+1]
  const int DC_G *P_col;             // [nnz]
  const float DC_G *P_val;           // [nnz]
  const int DC_G *inc_ptr;           // [N+1] vertex -> constraint corners
  const int DC_G *inc_idx;
  // P again, in wave-sliced ELL for the resident PCG: chunk c = rows 64c..64c+63, entry (s, lane) at
  // ell[ell_ptr[c] + 64 s + lane] = (column, float bits of the value); padded entries are (row, 0.0f)
  const int2 DC_G *ell;
  const int DC_G *ell_ptr;           // [ceil(N/64)]
  const int DC_G *ell_w;             // [ceil(N/64)] width of the chunk
  // P once more, symmetrically scaled to unit diagonal (D^-1/2 P D^-1/2) and packed for dc_forward_pk.hip: per
  // 64-row chunk pk_n[c] 16-byte packets per row (a multiple of 4), packet (s, lane) at pk[pk_ptr[c] + 64 s + lane] =
  // {v0, v1, v2, d0 | d1 << 10 | d2 << 20}, d = column - row + 512; chunks cover pk_threads * pk_vpt rows
  const int4 DC_G *pk;
  const int DC_C *pk_ptr;
  const int DC_C *pk_n;
  const float DC_G *sq_dinv;         // [pk_threads * pk_vpt] sqrt(1 / P_ii), 0 for padding rows
  int pk_vpt, pk_ok;            // rows per thread of the packet kernel; 0 = tables not usable (bandwidth > 511 or N too large)
  int pk_threads;               // threads of the packet kernel the tables are padded for (512 or 768)
  int adj_coarse;               // the adjoint's fp64 fall-back adds the coarse correction over the deflation space (dc_adjoint64.h)
  const double DC_G *dsph_tri;  // discretised sphere (DC_PRIM_SPHERE_DISCRETIZED): [dsph_ntri][12] = p0, p1, p2, face normal of its mesh, creation order
  int dsph_ntri;
  int fwd_defl;                 // the forward kernels use the deflation space (the deflated instances exist for 512 threads x >= 4 rows; the adjoint's
                                // coarse level needs only the tables)
  // spectral deflation of the forward solve (dc_deflate.h): 16 lowest eigenvectors of the scaled matrix, [pk rows][16] row-major; null = none
  const float DC_G *defl_u;
  const float DC_G *defl_au;         // Ahat U, same layout
  const float DC_C *defl_g;          // [16][16] (U^T Ahat U)^-1
  // explicit inverse of the scaled matrix for small meshes (dc_dense.h): [N + pad][dense_ld] fp32, null = not built
  const float DC_G *dense_inv;
  int dense_ld;
  // vertex renumbering (bandwidth reduction, dc_engine.hip): device index <-> caller's index; null = identity
  const int DC_G *user_of;           // [N] device -> caller
  const int DC_G *dev_of;            // [N] caller -> device
  // element windows (dc_windows.h / dc_winlib.h): the per-constraint passes run window by window inside LDS
  const int4 DC_C *win;              // [2 * nwin]: {v0, v1, lo, vs}, {tri_off, ntri, bend_off, nbend}
  const int4 DC_G *wtri_rec;         // per window-triangle: j0 | j1 << 16, j2, bits(w^2), triangle id
  const float4 DC_G *wtri_D;
  const int4 DC_G *wbend_rec;        // per window-flap: j0 | j1 << 16, j2 | j3 << 16, bits(rest norm), bits(w^2)
  const float4 DC_G *wbend_w;
  const float4 DC_G *wtri_Dlo;       // low-order parts of wtri_D / of the cotan weights 1..3 and the rest norm (precise record pass)
  const float4 DC_G *wbend_lo;
  const int4 DC_G *winc;             // vertex -> (result vector, coefficient) pair packets, wave-sliced by 64-vertex chunk
  const int DC_C *winc_ptr;
  const int DC_C *winc_n;
  int nwin, win_vcap, win_nrcap, win_ok;
  int win_lds_bytes, pad2;
  // self-collision (Simulation.cpp:194-220, 225-373): collision radii, connected-pair table (share a triangle)
  const float DC_G *radii;           // [N]
  const int DC_G *conn_ptr;          // [N+1]
  const int DC_G *conn_idx;          // sorted neighbours incl. self
  float max_radii;
  int self_cap;                 // capacity of the per-rollout self-contact list
  int self_lds;                 // 1: the layered self-contact passes run in LDS when their working set fits (dc_devlib.h)
  float h, k_att, gx, gy, gz;
  float k_stretch, k_bend, density;
  int contact_enabled, self_enabled, pad1;
  // fp64 copies of the rest-shape tables and parameters: the adjoint's fp64 operator / gradient assembly (dc_adjoint64.h).
  // The reference computes these in fp64 (Triangle.cpp:587-645, TriangleBending.cpp:186-239, Simulation.cpp:2894-2966); the fp32
  // tables above are their roundings.
  const double DC_G *tri_D64;        // [4][T] planar: inv_deltaUV d00, d01, d10, d11
  const double DC_G *tri_w2_64;      // [T]
  const double DC_G *bend_w64;       // [4][E] planar cotan weights
  const double DC_G *bend_nw64;      // [2][E] planar: rest norm, weight^2
  const double DC_G *mass64;         // [N]
  const float4 DC_G *tri_Dlo;        // [T] low-order parts of inv_deltaUV (value - fl32(value)): fp64-strain element operators on the global tables
  const float4 DC_G *bend_lo;        // [E] low-order parts of the cotan weights 1..3 and of the rest norm
  double h64, k_att64, k_stretch64, k_bend64, density64;
  double g64[3];
  DevPrim prims[kMaxPrims];
  // device-resident copy of this struct: kernels receive THIS pointer and read fields with scalar loads on
  // demand (passing the ~450-byte struct by value cost > 100 spilled SGPRs per kernel)
  const DevSystem *self_dev;
};

// Per-batch work buffers (one set, reused by forward and backward kernels).
struct DevWork {
  float *g;        // [B][3][N]  M (s_n - x_n) / h                    | backward: carried-in gradient copy
  float *vnow;     // [B][3][N]  current velocity iterate             | backward: u
  float *vbest;    // [B][3][N]  best iterate                          | backward: y = u + dr_df^T u
  float *cg_r, *cg_p, *cg_ap, *cg_x;   // [B][3][N] each
  float *corner;   // [B][3][NC] per-constraint-corner contributions
  float4 *ap4;     // [B][N] per-vertex float4 scratch of the resident PCG (A p, one 16-byte access per vertex)
  // adjoint, direct solve: preconditioned search direction / residual (M^-1 p, M^-1 s) and the 3 x 3 block inverses (dc_adjprecond.h)
  float *pre_p, *pre_s;   // [B][3][N]
  float *minv;            // [B][9][N]
  // adjoint in mixed precision (dc_adjoint64.h): solution, true residual, y = (I + dr_df)^T z, and the six further vectors of the
  // fp64 fall-back BiCGSTAB; [B][3][N] doubles each
  double *u64, *r64, *y64, *x64, *k64[6];
  double *c64;            // [B][3][NC] per-constraint-corner results of the fp64 element pass
  // self-collision detection / layering scratch (k_self_detect)
  int *sd_cell, *sd_order;      // [B][N]
  float *sd_sx;                 // [B][3][N] positions in cell-sorted order
  int2 *sd_rawpair;             // [B][cap]
  float4 *sd_rawn;              // [B][cap]
  int *sd_tmp;                  // [B][self_tmp_ints(cap)] layering structures (dc_selflib.h)
};

// Self contacts of one record, per rollout b: pair[b*cap + k] = (particleId1 < particleId2), nrm = contact normal,
// dvec = d of the last friction evaluation (SelfCollisionInformation::d), meta[b*64 + ...] = {count, nlayers,
// layer_offset[0..nlayers]} with the contacts stored layer by layer.
struct SelfRec {
  int2 *pair;
  float4 *nrm;        // .w carries the two working-set slots of the contact: slotA | slotB << 16 (as int bits)
  float4 *dvec;
  int *meta;          // [kMetaStride - 1] = number of distinct vertices in the contacts (working-set size)
  int *verts;         // [2 * cap] vertex of every working-set slot
};

struct FwdArgs {
  const float *x_in, *v_in;     // slot k   [B][3][N]
  float *x_out, *v_out;         // slot k+1
  float *rec_f, *rec_r, *rec_n; // record k+1 [B][3][N]
  int *rec_prim;                // record k+1 [B][N]   flattened primitive index or -1
  const float *x_fixed;         // [B][3][Af] fixed-point targets for this step
  const float *mu;              // [B][ngroups]
  const float *fu;              // [B][3] uniform extra force or nullptr
  const float *fv;              // [B][3][N] per-vertex extra force (wind with fall-off, constant force field) or nullptr
  const float *fv_scale;        // [B] factor on fv for this step (per-step wind factor of a force schedule) or nullptr = 1
  const float *fv2;             // [B][3][N] second per-vertex term with factor 1 (the constant force field next to a wind with fall-off and its own
                                // time factor, Simulation.cpp:91-93 + :99-106) or nullptr
  dc_step_stats *stats;         // [B]
  SelfRec self;                 // record k+1 (filled by k_self_detect before the step kernel runs)
  float fwd_tol, cg_tol;
  int pd_cap, cg_max, stall_window;
  int self_full;                // 1 = rebuild the right-hand side of all vertices after the self-contact layers (development switch DC_FWD_SELFFULL)
  int cg_seed;                  // packet / split kernels: first search direction of a solve = the previous PD iteration's correction
  // several consecutive steps in one launch (packet kernel only): step s uses tape slot k + s
  int nsteps, inline_detect;    // inline_detect: run the self-collision detection of every step inside the kernel
  size_t slot_state, slot_prim, slot_stats;     // slot strides of the [B][3][N] arrays, the [B][N] array, the stats
  size_t slot_self, slot_meta;                  // ... of the self-contact lists and their meta blocks
  // device-resident schedules (dc_set_*_schedule): per-step strides of x_fixed, fu and fv_scale; 0 = the same values in every step
  size_t slot_xfix, slot_fu, slot_fvs;
};

struct BwdArgs {
  const float *x_new;           // slot k [B][3][N]
  const float *rec_f, *rec_n;   // record k
  const int *rec_prim;
  SelfRec self;                 // record k
  const float *mu;
  float *gx, *gv;               // carried gradient, in: dL_dxnew/dL_dvnew, out: dL_dx/dL_dv   [B][3][N]
  const float *ix, *iv;         // dL_dxinit / dL_dvinit or nullptr
  float *d_xfixed;              // [B][3][Af] out (overwritten) or nullptr
  float *d_mu;                  // [B][ngroups] accumulated (+=) or nullptr
  float *d_param;               // [B][8] per-step parameter gradients (dk_stretch, dk_bend, dk_att, ddensity, h^2 sum y) or nullptr
  const float *x_fixed;         // [B][3][Af] fixed-point targets used by the step that produced the record
  const float *x_prev, *v_prev; // slot k-1 state [B][3][N]
  const float *v_new;           // slot k velocity [B][3][N]
  dc_bwd_stats *stats;          // [B]
  float bwd_tol, cg_tol, clip_thr, rel_tol;
  int mode;                     // 0: reference fixed-point iteration (+ direct fallback), 1: direct Krylov solve
  int it_cap, cg_max, is_start, clip, stall_window;
  int block_pre;                // direct solve: 1 = block-Jacobi from K's own diagonal blocks (dc_adjprecond.h), 0 = Jacobi from diag(P)
  int fp32_only;                // direct solve: 1 = the fp32 Krylov solve alone (no fp64 residual, no refinement, no fp64 fall-back)
  int dense_y;                  // 1 = form y = (I + dr_df)^T z over all vertices in every operator application (development switch DC_ADJ_DENSEY)
  int verify_all;               // direct solve: 1 = evaluate the fp64 residual after EVERY correction solve (development switch DC_ADJ_VERIFY)
  int cg_first;                 // direct solve, diag(P) preconditioner: correction solves by CG, BiCGSTAB once a CG cycle fails to contract the fp64 residual (DC_ADJ_CG, default 1)
  int warm;                     // direct solve inside a fused sweep: start step s > 0 from gamma u*(step s - 1) (DC_ADJ_WARM, dc_adjoint.hip)
  int ycap, ybase;              // entries of the contact vertices' y list in the dynamic LDS and its start in floats (set by the launch: dc_adjoint.hip, AdjCtx::ylist)
  // several consecutive steps of the backward sweep in one launch: step s differentiates tape slot `slot` - s
  int nsteps, slot;
  int start_at;                 // the record whose backward step is the trajectory's isStart step (dc_set_trajectory_start: start slot + 1; 0 = none)
  size_t slot_state, slot_prim, slot_self, slot_meta, slot_param, slot_xf, slot_stats;   // per-slot strides (elements); d_xfixed steps by slot_xf too
  size_t slot_ix;               // seed schedule: ix / iv of step s are ix - s * slot_ix (0: none / the same buffer)
  // record handed in from outside (dc_set_record): the fp64 values of x_new, f, the primitive-contact normals ([B][3][N] planar) and of
  // the self contacts' normals / d ([B][cap][3]) for the fp64 operator; all null for a record the forward kernels made
  const double *inj_x, *inj_f, *inj_n, *inj_sn, *inj_sd;
  float *ys;                    // y = (I + dr_df)^T u* of the step, kept per tape slot (dc_keep_force_gradients; steps by slot_state) or nullptr
};

void launch_pd_step(const DevSystem &S, const DevWork &W, const FwdArgs &A, int B, hipStream_t st);
// LDS/register-resident variant (dc_forward_res.hip); returns false when N is too large for it.
bool pd_step_fusable(const DevSystem &S);   // launch_pd_step would take the packet kernel, which honours FwdArgs::nsteps
bool launch_pd_step_packet(const DevSystem &S, const DevWork &W, const FwdArgs &A, int B, hipStream_t st);
bool launch_pd_step_resident(const DevSystem &S, const DevWork &W, const FwdArgs &A, int B, hipStream_t st, int variant);
void launch_self_detect(const DevSystem &S, const DevWork &W, const FwdArgs &A, int B, hipStream_t st);
void launch_adjoint_step(const DevSystem &S, const DevWork &W, const BwdArgs &A, int B, hipStream_t st);
void launch_f64i_to_f32p(const double *src, float *dst, int B, int n, const int *user_of, hipStream_t st);
void launch_f32p_to_f64i(const float *src, double *dst, int B, int n, const int *user_of, hipStream_t st);
void launch_f64i_to_f64p(const double *src, double *dst, int B, int n, const int *user_of, hipStream_t st);
void launch_dev_to_planar(const void *src, int is_f32, float *dst, int B, int n, const int *user_of, hipStream_t st);
void launch_planar_to_dev(const float *src, void *dst, int is_f32, int B, int n, const int *user_of, hipStream_t st);
void launch_copy_cast(const float *src, void *dst, int is_f32, long total, hipStream_t st);
void launch_seed_gradient(const float *x, const float *target, float *gx, float *gv, int B, int N, float scale, hipStream_t st);

}  // namespace dc
