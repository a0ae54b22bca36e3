/* diffcloth_hip.h — C-ABI of the MI355X-native DiffCloth stepper (libdiffcloth_hip.so).
 *
 * Drop-in boundary for the hot path of omegaiota/DiffCloth: the Projective-Dynamics forward step with
 * Signorini–Coulomb dry friction and its adjoint backward step. Each entry point names the reference
 * interface it replaces (paths relative to /root/reference/src/code/simulation/). The C++ host class
 * `Simulation` (diffcloth_amd/csrc/host/) keeps the reference's step()/stepNN()/stepBackward()/
 * stepBackwardNN() signatures and forwards to these functions; INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - plain C, opaque handle, int status (0 = DC_OK); no exceptions cross the boundary;
 *    dc_last_error() returns a message for the last failing call on that context.
 *  - host vectors are float64, xyz-interleaved, length 3N (or 3*Af) PER ROLLOUT, rollouts concatenated:
 *    exactly the reference's VecXd layout (Simulation.h: ForwardInformation::x etc.), batched.
 *  - the device state is fp32, component-planar [B][3][N]; the step computes in fp32 with fp64 where a difference
 *    of nearly equal quantities decides the result: element strains in the forward step, residual / fall-back solve / gradient
 *    sums in the adjoint (dc_params::adjoint_fp32_only).
 *  - a context owns one HIP stream; calls are enqueued in order; calls that return host data synchronise.
 *  - "slot" k of the tape holds the state after k steps; slot 0 is the initial state.
 */
#ifndef DIFFCLOTH_HIP_H
#define DIFFCLOTH_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dc_ctx dc_ctx;

enum { DC_OK = 0, DC_ERR_INVALID = 1, DC_ERR_HIP = 2, DC_ERR_STATE = 3, DC_ERR_TOPOLOGY = 4, DC_ERR_CAPACITY = 5 };

/* Primitive kinds (Primitive.h PrimitiveType; only the analytic isInContact family is on the hot path). */
enum { DC_PRIM_SPHERE = 0, DC_PRIM_CAPSULE = 1, DC_PRIM_PLANE = 2, DC_PRIM_BOWL = 3,
       DC_PRIM_SPHERE_DISCRETIZED = 4   /* Sphere with discretized = true (Primitive.cpp:230-253; the BIG_SPHERE scene, Simulation.cpp:1905-1911):
                                           contact normal = face normal of the sphere's own latitude / longitude mesh; `length` = its
                                           resolution (0 = the reference's 40). At most one per context. */ };

/* One analytic obstacle. A LowerLeg (Primitive.cpp:383-418) is passed as its three children
 * (joint sphere, foot capsule, leg capsule) sharing one `group`; friction uses the group's mu and the
 * first child in contact wins, as LowerLeg::isInContact does.                                          */
typedef struct dc_primitive {
  int kind;            /* DC_PRIM_*                                                              */
  int group;           /* primitive id seen by the caller (index into Simulation::primitives)    */
  double center[3];    /* world position tested against: center_prim (+ child centerInit)        */
  double top_offset[3];/* capsule: globalRotation * (0,length,0)  (Primitive.cpp:582);
                          plane: corner upperLeft relative to the centre (Plane ctor, Primitive.cpp:13-21) */
  double corner2[3];   /* plane: corner upperRight relative to the centre (the rectangle's other two corners are
                          the negatives; Plane::isInContact Primitive.cpp:66-130). Unused otherwise           */
  double radius;       /* sphere, capsule; bowl: the hemisphere shell Bowl::isInContact tests (Primitive.cpp:362-381) */
  double length;       /* capsule length                                                         */
  double mu;           /* Primitive::mu (default for rollouts without a per-rollout override)     */
  int rotates;         /* Sphere::rotates (Primitive.cpp:255-257)                                 */
} dc_primitive;

/* Scene / solver parameters. Mirrors the process-global statics of the reference
 * (Simulation.h:325-338, Simulation.cpp:9-22) and SceneConfiguration / FabricConfiguration.           */
typedef struct dc_params {
  double time_step;                 /* sceneConfig.timeStep                                       */
  double density;                   /* fabric.density                                             */
  double k_stretch, k_bend, k_att;  /* Triangle::k_stiff, TriangleBending::k_stiff, AttachmentSpring::k_stiff */
  double gravity[3];                /* Simulation::gravity                                        */
  double forward_tol;               /* Simulation::forwardConvergenceThreshold                    */
  double backward_tol;              /* Simulation::backwardConvergenceThreshold                   */
  int gravity_enabled, contact_enabled, selfcollision_enabled;
  int gradient_clipping;            /* Simulation::gradientClipping                               */
  double gradient_clipping_threshold;
  int pd_iter_cap;                  /* <0: (-log10(forward_tol))*150 as Simulation.cpp:1182       */
  int adjoint_iter_cap;             /* <=0: 400 as Simulation.cpp:1562                            */
  /* inner block-Jacobi PCG (replaces SimplicialLLT::solve, Simulation.cpp:1267 / :1577) */
  double cg_rel_tol;                /* stop when |r|_{D^-1} <= cg_rel_tol * |r0|_{D^-1}; <=0: 1e-4 */
  int cg_max_iter;                  /* <=0: 500                                                   */
  /* optional fp32-floor guard, OFF by default (<= 0): leave the PD / adjoint loop when the update norm has set no new
   * minimum for this many consecutive iterations, return the best iterate and report converged = 2. The scene tables
   * ask for 1e-9..1e-10, below fp32 resolution for stiff scenes, where the loop otherwise runs to the cap and
   * reverts to the best iterate exactly as the reference does (Simulation.cpp:1357-1367). It is off by default
   * because |x_new - x_now| is NOT monotone under contact mode switching: plateaus of > 100 iterations occur in the
   * reference's own T-shirt run and a short window would stop early there (tests/test_gpu_parity.py, golden rollout). */
  int stall_window;
  /* adjoint solver: 0 = the reference's fixed-point iteration u <- P^-1 (g + dP^T u) (Simulation.cpp:1561-1600)
   * with a direct solve when the cap is hit (:1589-1594); 1 = always the direct solve, i.e. the semantics of
   * Simulation::backwardGradientForceDirectSolver / solveDirect (:1431-1440). On the GPU the direct solve is a
   * block-Jacobi preconditioned BiCGSTAB on (P - dP^T) run to adjoint_rel_tol (relative residual; <=0: 1e-6). */
  int adjoint_mode;
  double adjoint_rel_tol;
  /* direct adjoint solve: 1 (default) = block-Jacobi preconditioner from the 3x3 diagonal blocks of K = P - dP^T itself, rebuilt
   * from x_new every backward step (in-plane stiff / normal soft, sticking contacts decoupled); 0 = the forward solve's Jacobi
   * diag(P)^-1. Changes the iteration count only, not what the solve converges to.                                      */
  int adjoint_block_precond;
  /* direct adjoint solve, precision: 0 (default) = mixed — fp32 Krylov solves (CG first, BiCGSTAB once CG stalls; round 6) for corrections of the residual g - K u evaluated
   * in fp64 from fp64 rest-shape tables, until |g - K u| <= adjoint_rel_tol |g| holds in fp64; when the fp32 solve makes no
   * progress (adjoint systems beyond fp32, e.g. a compressed fine garment) a block-Jacobi BiCGSTAB in fp64 on the same operator
   * takes over — the role of the reference's fp64 SparseLU (Simulation.cpp:1431-1440), which always returns a solution.
   * 1 = the fp32 Krylov solve alone (round-2 behaviour: gradients at eps_fp32 * cond(K), 1-3e-4 on stiff / large scenes).   */
  int adjoint_fp32_only;
  int max_self_contacts;            /* capacity of the per-rollout self-contact list of one step; <=0: sized from the mesh,
                                       max(2048, N) pairs (at most 16000); overflow is reported, never silent: dc_step_stats */
  /* spectral deflation for ill-conditioned meshes (csrc/dc_deflate.h): < 0 (default) = decided in dc_build by a probe solve (16 eigenvectors of
   * the scaled system matrix when Jacobi-PCG needs more than 80 iterations on a smooth right-hand side: the reference's 7 742-vertex dress
   * needs 337, the hat 171; the cloth / T-shirt / sock / dress-3634 meshes 15 ... 39 and get none), 0 = never, > 0 = always.
   * Forward step (meshes of more than 1536 vertices; smaller ones solve with the explicit inverse): every PCG solve starts with a Galerkin
   * projection onto those vectors, after its recycled first direction. Adjoint (direct solve with the block preconditioner): the vectors are
   * the coarse level of the BiCGSTAB preconditioner. Stopping rules and what the solves converge to are unchanged. dc_get_deflation reports
   * what dc_build decided.                                                                                                            */
  int forward_deflation;
} dc_params;

/* Per-rollout statistics of one forward step (ForwardInformation::converged/convergeIter). */
typedef struct dc_step_stats {
  int converged;         /* 1: tolerance met; 2: stalled at the fp32 floor, best iterate returned; 0: cap hit */
  int pd_iters;
  int cg_iters;          /* total inner PCG iterations over the step */
  int prim_contacts;
  int self_contacts;
  float last_xdiff;      /* |x_new - x_now|_2 / N of the final iteration */
  int self_overflow;     /* 0: the self-contact list is complete (the reference has no limit, Simulation.cpp:281-352, 422-624);
                            bit 0: more pairs than dc_params::max_self_contacts were found and the list was cut (the step's
                            result is NOT the reference's: raise max_self_contacts and repeat the step); bit 1: more layers
                            than the layer table holds (4088). Calls that return these statistics fail with DC_ERR_CAPACITY */
} dc_step_stats;

/* Per-rollout statistics of one backward step (BackwardInformation::converged/backwardIters). */
typedef struct dc_bwd_stats {
  int converged;         /* 1: tolerance met; 2: stalled at the fp32 floor; 0: cap hit (reference falls back to SparseLU) */
  int adjoint_iters;
  int cg_iters;
  int clipped;
  int used_direct;       /* 1 when the direct (Krylov) solve ran; adjoint_iters then counts its iterations too */
  float last_udiff;      /* mode 0: |u_new - u|_2 / N; direct solve: relative residual |g - K u| / |g| — mixed precision: the TRUE
                            residual evaluated in fp64 (residual_verified = 1), or, when the last fp32 correction solve was accepted
                            without a further fp64 evaluation, an upper bound: its recurrence residual + twice the measured fp32
                            operator error (residual_verified = 0); adjoint_fp32_only: the fp32 recurrence's                 */
  int refine_cycles;     /* direct solve, mixed precision: fp32 solves run (each followed by an fp64 residual evaluation)    */
  int fp64_iters;        /* BiCGSTAB iterations of the fp64 fall-back (0: the fp32 corrections were enough)                 */
  int residual_verified; /* direct solve, mixed precision: 1 = last_udiff is an fp64-evaluated residual, 0 = the bound described there */
  int workgroups;        /* workgroups that ran this rollout's backward step. Smaller than dc_get_cluster's K when the split adjoint kernel
                            does not implement the configuration — adjoint_mode 0 (the reference's fixed-point iteration, Simulation.cpp:
                            1569-1600) runs on ONE workgroup per rollout although the forward steps are split — reported, not silent   */
} dc_bwd_stats;

/* ---- lifetime ------------------------------------------------------------------------------------ */
/* device_id >= 0: HIP device; fails with DC_ERR_HIP when no device is present — there is no CPU compute path.
 * device_id == -1: host-only context for building / inspecting the constraint system (dc_set_*, dc_build,
 * dc_get_counts/system_matrix/vertex_data); every batch or step call on it fails with DC_ERR_STATE.      */
int dc_create(int device_id, dc_ctx **out);
int dc_destroy(dc_ctx *ctx);
const char *dc_last_error(const dc_ctx *ctx);
const char *dc_version(void);

/* ---- system definition: replaces createClothMeshFromModel/…FromConfig (Simulation.cpp:2170-2255,
 *      2611-2757), createBendingConstraints (:2096-2131), updateCollisionRadii (:2407-2431),
 *      updateAreaMatrix/updateMassMatrix (:2894-2966), initializePrefactoredMatrices (:2969-3059).
 *      The caller passes the already oriented / normalised rest positions.                           */
int dc_set_mesh(dc_ctx *ctx, int num_vertices, const double *rest_pos /*3N*/, int num_triangles, const int *tris /*3T*/);
int dc_set_attachments(dc_ctx *ctx, int num_fixed, const int *vertex /*Af*/);   /* SystemMatrix::attachments */
int dc_set_params(dc_ctx *ctx, const dc_params *p);
int dc_set_primitives(dc_ctx *ctx, int count, const dc_primitive *prims);       /* Simulation::primitives */
int dc_build(dc_ctx *ctx);      /* (re)assemble A, P = M + h^2 A^T A and upload; call after any of the setters */
void dc_default_params(dc_params *p);
/* The reference reads its process-global statics at every step (Simulation::forwardConvergenceThreshold,
 * backwardConvergenceThreshold, gradientClipping[Threshold], backwardGradientForceDirectSolver, and the
 * gravity/contact/self-collision switches of setWindAncCollision): these two calls change them between steps
 * without rebuilding the system or invalidating the batch / tape.                                          */
int dc_set_solver(dc_ctx *ctx, double forward_tol, double backward_tol, int gradient_clipping, double clip_threshold,
                  int force_direct_adjoint);
int dc_set_flags(dc_ctx *ctx, int gravity_enabled, int contact_enabled, int selfcollision_enabled);

/* sizes: N, T, E (bending flaps), Af, nnz(P), constraint rows (6T + 3E + 3Af) */
int dc_get_counts(const dc_ctx *ctx, int *out6);
/* host-side tables for inspection / parity tests: scalar P in CSR; per-vertex mass, area, radii */
int dc_get_system_matrix(const dc_ctx *ctx, int *row_ptr /*N+1*/, int *col /*nnz*/, double *val /*nnz*/);
int dc_get_vertex_data(const dc_ctx *ctx, double *mass, double *area, double *radii);

/* ---- batch state ---------------------------------------------------------------------------------- */
/* B independent rollouts sharing the system; tape_steps = how many forward steps can be recorded. */
int dc_alloc_batch(dc_ctx *ctx, int batch, int tape_steps);
/* Simulation::resetSystem / stepNN's state injection (Simulation.cpp:1020-1030): write slot `slot`. */
int dc_set_state(dc_ctx *ctx, int slot, const double *x /*B*3N*/, const double *v /*B*3N*/);
int dc_get_state(dc_ctx *ctx, int slot, double *x, double *v);
/* per-rollout friction coefficient per primitive group: mu[b*num_groups + g]; NULL restores the defaults */
int dc_set_mu(dc_ctx *ctx, const double *mu);
/* uniform external force added to every vertex of rollout b during the NEXT forward steps (wind*windNorm*
 * windFactor of fillForces, Simulation.cpp:96-105): f[b*3+d]; NULL = none                              */
int dc_set_uniform_force(dc_ctx *ctx, const double *f);
/* per-vertex external force added during the NEXT forward steps: f[b*3N + 3i + d] (xyz interleaved like the states). Carries
 * what fillForces adds per vertex beyond gravity and the uniform wind: wind * windNorm * windFactor (.) windFallOff for
 * WIND_SIN_AND_FALLOFF / WIND_FACTOR_PER_STEP and the constant force field (Simulation.cpp:87-105); NULL = none       */
int dc_set_vertex_forces(dc_ctx *ctx, const double *f /*B*3N or NULL*/);
/* a SECOND per-vertex external force, added with factor 1 in every step: the constant force field of fillForces
 * (`if (enableConstantForcefield) f_ext += external_force_field`, Simulation.cpp:91-93) next to a wind with fall-off whose per-step
 * factor travels with dc_set_vertex_forces + dc_set_force_schedule (fv_scale) — the two per-vertex terms of a step have different
 * time factors, and a fused rollout needs both on the device. Same layout as dc_set_vertex_forces; NULL = none                    */
int dc_set_vertex_force_field(dc_ctx *ctx, const double *f /*B*3N or NULL*/);

/* ---- the hot path --------------------------------------------------------------------------------- */
/* Simulation::step()/stepNN() (Simulation.cpp:1020-1428): advance slot -> slot+1 for all rollouts.
 * fixed_pts: rlFixedPointPos per rollout, B*3Af (NULL: keep the previous targets / rest positions).
 * stats (optional, length B) is filled after synchronising; pass NULL to stay asynchronous.            */
int dc_step_forward(dc_ctx *ctx, int slot, const double *fixed_pts, dc_step_stats *stats);
/* ForwardInformation fields of the step that produced `slot` (f, r): B*3N each; NULL to skip. */
int dc_get_record(dc_ctx *ctx, int slot, double *f, double *r);
/* per-vertex primitive contact of that step: group id or -1 (B*N) and contact normal (B*3N) */
int dc_get_contacts(dc_ctx *ctx, int slot, int *prim_group, double *normal);

/* layered self contacts of that step for ONE rollout (ForwardInformation::collisionInfos.second, the output of
 * contactSorting, Simulation.cpp:422-624): pairs (particleId1 < particleId2) in layer order, layer index and normal
 * per contact; at most `cap` entries are written, *count receives the total.                                  */
int dc_get_self_contacts(dc_ctx *ctx, int slot, int rollout, int cap, int *count, int *num_layers, int *pairs /*2*cap*/,
                         int *layer /*cap*/, double *normal /*3*cap*/);

/* A forward record handed in from OUTSIDE. Simulation::stepBackward differentiates the ForwardInformation it is given, whoever made it
 * (Simulation.cpp:1455-1551 reads forwardInfo_new.x / .x_prev / .v_prev / .f / .x_fixedpoints / .collisionInfos; records also come from
 * disk, resetForwardRecordsFromFolder): dc_set_record writes record `slot` of all B rollouts — the tape entries in fp32 AND the values
 * as passed (fp64) for the adjoint's fp64 operator, so that dc_step_backward(slot) solves the adjoint system of exactly this record
 * (teacher-forced parity tests upload the fp64 oracle's record; the forward kernels are not involved). The state the step started from
 * (x_prev, v_prev) is slot - 1 (dc_set_state). The fp64 copy belongs to this one slot and is dropped when a forward step overwrites the
 * slot, when the slot's state or the system is rewritten (dc_set_state*, dc_build), when another record is set, or by dc_alloc_batch; a
 * fused backward sweep over an injected slot runs step by step. Self contacts come in layer order and the contacts of one layer must be
 * vertex-disjoint (they are applied in parallel, as contactSorting builds them: Simulation.cpp:422-624) — DC_ERR_INVALID otherwise.       */
typedef struct dc_record {
  const double *x, *v;          /* B*3N  ForwardInformation::x, ::v — the state after the step                                       */
  const double *f;              /* B*3N  ForwardInformation::f (b~ - C v of the last PD iteration: the contact vectors d derive from it) */
  const double *r;              /* B*3N  ForwardInformation::r, may be NULL (the adjoint does not read it)                           */
  const int *prim;              /* B*N   primitive in contact with the vertex: index into the dc_set_primitives array (LowerLeg children
                                          flattened), -1 = none (PrimitiveCollisionInformation::primitiveId)                          */
  const double *normal;         /* B*3N  contact normal at the vertices in contact (ignored elsewhere)                               */
  const double *x_fixed;        /* B*3Af ForwardInformation::x_fixedpoints of the step; NULL keeps the slot's                        */
  /* layered self contacts (collisionInfos.second, the output of contactSorting): self_count[b] contacts per rollout, concatenated in
   * rollout order; within a rollout in layer order (self_layer non-decreasing, starting at 0)                                       */
  const int *self_count;        /* B, or NULL = no self contacts                                                                     */
  const int *self_pairs;        /* 2 per contact: particleId1 < particleId2                                                          */
  const int *self_layer;        /* 1 per contact: layerId                                                                            */
  const double *self_normal;    /* 3 per contact                                                                                     */
  const double *self_d;         /* 3 per contact: SelfCollisionInformation::d of the last friction evaluation                        */
} dc_record;
int dc_set_record(dc_ctx *ctx, int slot, const dc_record *rec);

/* Simulation::stepBackward()/stepBackwardNN() (Simulation.cpp:1443-1780) through the step that produced
 * `slot` (forwardInfo_new = record `slot`). Inputs B*3N each; dL_dxinit/dL_dvinit may be NULL (zeros).
 * Outputs: dL_dx, dL_dv (B*3N), dL_dxfixed (B*3Af, may be NULL), dL_dmu (B*num_groups, may be NULL).    */
int dc_step_backward(dc_ctx *ctx, int slot, const double *dL_dxnew, const double *dL_dvnew,
                     const double *dL_dxinit, const double *dL_dvinit, int is_start, double *dL_dx,
                     double *dL_dv, double *dL_dxfixed, double *dL_dmu, dc_bwd_stats *stats);

/* Parameter gradients of the LAST backward step through record `slot` (Simulation.cpp:1672-1764), 8 doubles per
 * rollout: [0..2] this step's contribution to dL/dk of {stretch, bending, attachment} (dL_dk_pertype), [3] to
 * dL/ddensity (adddr_dd = false), [4..6] h^2 * sum_i ((I + dr_df)^T u*)_i = the summed dL_dfext_vec from which the
 * caller forms dL_dfext / dL_dwind (multiply by windFactor, cos(...) etc. as :1730-1760), [7] unused.            */
int dc_get_param_gradients(dc_ctx *ctx, int slot, double *out /*B*8*/);
/* dL/df per vertex of the LAST backward step: h^2 ((I + dr_df)^T u*)_i, B*3N xyz interleaved — the vector the reference
 * calls dL_dfext_vec (Simulation.cpp:1700-1760), from which dL_dconstantForceField (sum over the steps), dL_dwindtimestep
 * (dot with (wind * windNorm) (.) windFallOff) and the fall-off variants of dL_dfext / dL_dwind are formed.            */
int dc_get_force_gradient(dc_ctx *ctx, double *dL_df /*B*3N*/);
/* The same vector of EVERY step of a backward sweep: with keep = 1 the backward kernels also store y = (I + dr_df)^T u* of the step through
 * record `slot` in a tape-sized array (allocated on first use, one [B][3][N] fp32 plane per slot), so that a fused sweep
 * (dc_rollout_backward: all steps in one launch) leaves what Simulation::stepBackward forms per step from dL_dfext_vec — dL_dconstantForceField
 * (the sum over the steps), dL_dwindtimestep[step] (its dot with the wind field), the fall-off variants of dL_dfext / dL_dwind
 * (Simulation.cpp:1700-1764) — readable afterwards: dc_get_force_gradients returns h^2 y of the records slot0 .. slot0 + nslots - 1. */
int dc_keep_force_gradients(dc_ctx *ctx, int keep);
int dc_get_force_gradients(dc_ctx *ctx, int slot0, int nslots, double *dL_df /*nslots*B*3N*/);

/* ---- device-pointer boundary: the per-step calls for callers whose tensors already live on this GPU (torch-ROCm: the controller /
 * RL training loops, reference src/python_code/pySim/functional.py:20-102 and hatController.py:78-105, call stepNN + stepBackwardNN
 * once per time step). Same semantics as dc_set_state / dc_get_state / dc_step_forward / dc_step_backward, but every buffer is a
 * DEVICE pointer in the caller's layout — xyz interleaved, B rollouts concatenated, fp32 (is_f32 = 1, torch's default dtype) or
 * fp64 (0) — converted to / from the planar device layout by a kernel: no host copy, no synchronisation; the calls are enqueued on
 * the context's stream and return at once. dc_use_stream makes that stream the caller's (e.g. torch.cuda.current_stream()), so that
 * the calls are ordered with the caller's own kernels; NULL returns to the context's own stream. Statistics of such steps: dc_get_stats. */
int dc_use_stream(dc_ctx *ctx, void *hip_stream);
int dc_set_state_dev(dc_ctx *ctx, int slot, const void *d_x /*B*3N*/, const void *d_v /*B*3N*/, int is_f32);
int dc_get_state_dev(dc_ctx *ctx, int slot, void *d_x /*B*3N or NULL*/, void *d_v /*B*3N or NULL*/, int is_f32);
int dc_step_forward_dev(dc_ctx *ctx, int slot, const void *d_fixed_pts /*B*3Af or NULL*/, int is_f32);
int dc_step_backward_dev(dc_ctx *ctx, int slot, const void *d_dL_dxnew, const void *d_dL_dvnew, const void *d_dL_dxinit /*or NULL*/,
                         const void *d_dL_dvinit /*or NULL*/, int is_start, void *d_dL_dx, void *d_dL_dv, void *d_dL_dxfixed /*B*3Af or NULL*/,
                         void *d_dL_dmu /*B*num_groups or NULL*/, int is_f32);

/* ---- device-resident rollouts (no host copies inside; used by bench.py and batched callers) --------- */
/* nsteps forward steps slot -> slot+nsteps; fixed points held at their current values. Returns when the sweep has finished on the
 * device (its kernel time and the split kernels' error word are read back; a self-contact overflow of a step is reported by
 * dc_get_stats for that slot: DC_ERR_CAPACITY). When the packet
 * kernel is in use all steps of a rollout run inside ONE launch (each rollout advances on its own, self-collision
 * detection inlined per step); results are bitwise those of nsteps dc_step_forward calls.                          */
int dc_rollout_forward(dc_ctx *ctx, int slot, int nsteps);
/* Seed the carried gradient (dL_dx, dL_dv) on the device: g_x = scale_x * (x[slot] - target), g_v = 0
 * (the MATCH-shape loss gradient of Simulation.cpp:3237-3488); target NULL means the rest shape.        */
int dc_seed_gradient(dc_ctx *ctx, int slot, const double *target /*3N or NULL*/, double scale_x);
/* nsteps backward steps from record `slot` down to slot-nsteps+1, carrying (dL_dx, dL_dv) on the device
 * exactly as Simulation::runBackwardTask does (Simulation.cpp:3938-3952), all steps in one launch. dL_dmu accumulates
 * over the steps; the per-step parameter gradients (dc_get_param_gradients) and dL_dx_fixed of every step (dc_get_dxfixed) stay
 * readable per slot. Returns when the sweep has finished on the device.                                              */
int dc_rollout_backward(dc_ctx *ctx, int slot, int nsteps);
int dc_get_gradient(dc_ctx *ctx, double *dL_dx, double *dL_dv, double *dL_dmu /*B*num_groups or NULL*/);
/* Carried gradient of the backward sweep set from the host (the loss gradient w.r.t. the last state, Simulation.cpp:3925-3936);
 * also clears the accumulated dL_dmu. */
int dc_set_gradient(dc_ctx *ctx, const double *dL_dx /*B*3N*/, const double *dL_dv /*B*3N*/);

/* ---- device-resident schedules: what the host loop of Simulation::runBackwardTask (Simulation.cpp:3853-3961) feeds into every
 * step — stepFixPoints targets (:964-1018), fillForces terms (:55-116), the per-frame loss gradients dL_dxinit / dL_dvinit
 * (:3938-3952) — uploaded once per rollout, so that a whole loss + gradient evaluation is dc_rollout_forward + dc_rollout_backward.
 * A schedule entry belongs to a tape slot and stays until it is overwritten, dc_clear_schedules or dc_alloc_batch; the per-step
 * calls honour force / fixed-point entries of their slot as well (explicit fixed_pts of dc_step_forward win). A fused sweep needs
 * each kind of schedule on all of its steps or on none (DC_ERR_INVALID otherwise). */
/* targets of the steps slot0+k -> slot0+k+1, k = 0..nsteps-1: xf[k*B*3Af ...] laid out like dc_step_forward's fixed_pts */
int dc_set_fixed_point_schedule(dc_ctx *ctx, int slot0, int nsteps, const double *xf /*nsteps*B*3Af*/);
/* external forces of those steps: fu = uniform force per rollout (wind * windNorm * windFactor(t)), fv_scale = factor on the
 * per-vertex force of dc_set_vertex_forces (windFactor(t) for a wind with fall-off; pass the factor-free field there). Either may
 * be NULL (= the current dc_set_uniform_force value / factor 1). */
int dc_set_force_schedule(dc_ctx *ctx, int slot0, int nsteps, const double *fu /*nsteps*B*3 or NULL*/, const double *fv_scale /*nsteps*B or NULL*/);
/* loss gradient w.r.t. the state AT slot slot0+k (k = 0..nslots-1): added by dc_rollout_backward when its sweep arrives at that
 * state, i.e. it is the dL_dxinit / dL_dvinit argument of the step through record slot0+k+1. dL_dv may be NULL (zeros).
 * Costs two more tape-sized arrays, allocated on first use. dc_step_backward ignores it (its seeds are its arguments). */
int dc_set_seed_schedule(dc_ctx *ctx, int slot0, int nslots, const double *dL_dx /*nslots*B*3N*/, const double *dL_dv /*or NULL*/);
int dc_clear_schedules(dc_ctx *ctx);
/* states of nslots consecutive slots in one call: x[k*B*3N ...], v likewise (either may be NULL) */
int dc_get_states(dc_ctx *ctx, int slot0, int nslots, double *x, double *v);
/* dL_dxfixed of the steps through records slot0 .. slot0+nslots-1 of the last backward sweep / step (B*3Af each) */
int dc_get_dxfixed(dc_ctx *ctx, int slot0, int nslots, double *dL_dxfixed /*nslots*B*3Af*/);
/* Which tape slot holds the trajectory's INITIAL state (default 0): the backward step through record start_slot + 1 is the reference's isStart step
 * (Simulation.cpp:3947, :1534: dL_dx of the initial state takes no dL_dv / h term) in dc_rollout_backward. -1: this tape holds a LATER segment of a
 * trajectory — no step of it is the start. With it a trajectory can be run as a chain of fused segments over several contexts (e.g. one context per
 * attachment set, SceneConfiguration::customAttachmentVertexIdx, Simulation.cpp:1053-1068): state handed on with dc_get_state_dev / dc_set_state_dev,
 * the carried gradient on the way back with dc_get_gradient / dc_set_gradient (tests/test_gpu_schedules.py). dc_step_backward takes is_start itself. */
int dc_set_trajectory_start(dc_ctx *ctx, int start_slot);
int dc_get_stats(dc_ctx *ctx, int slot, dc_step_stats *fwd /*B or NULL*/, dc_bwd_stats *bwd /*B or NULL*/);
int dc_sync(dc_ctx *ctx);
/* Split execution: with fewer rollouts than compute units (BASELINE C4 sharded over 8 GPUs: 32 per GPU; hatController.py: 20 rollouts)
 * or a mesh too large for one workgroup's LDS, a rollout is run by K workgroups that own K contiguous vertex ranges and exchange
 * boundary rows and partial sums inside the launch (csrc/dc_cluster.h). K is chosen in dc_alloc_batch from the batch size, the mesh
 * and the device (environment DC_CLUSTER=k forces k; 0 or 1 = one workgroup per rollout). Results agree with the one-workgroup
 * kernels to solver tolerance (different summation order), not bitwise. Reports K and how many rollouts one launch covers. */
int dc_get_cluster(const dc_ctx *ctx, int *workgroups_per_rollout, int *rollouts_per_launch);
/* Which kernel set dc_build chose for this system (works on a host-only context too): out6 = {vertices renumbered on the device
 * (0 / 1), bandwidth of the system matrix in device numbering, packet-matrix forward kernel usable (needs bandwidth <= 511),
 * LDS element windows usable, number of element windows, explicit-inverse solve (small meshes)}. A mesh that fails the packet /
 * window conditions runs on the global-memory fallback kernels — correct, but several times slower. */
int dc_get_layout(const dc_ctx *ctx, int *out6);
/* Deflation space of the forward solve dc_build chose (works on a host-only context): number of vectors (0 = none) and the iteration
 * count of the probe solve that decided (Jacobi-PCG to 1e-4 on a smooth right-hand side).                                            */
int dc_get_deflation(const dc_ctx *ctx, int *vectors, int *probe_iterations);

/* ---- collective for C++ callers: one context (= one GPU) per rank; the optimiser sums [loss, dL/dtheta] over the ranks once per
 * evaluation, the only exchange of the rollout-sharded job. RCCL is bound at run time when these are first used.
 * rank 0 obtains a 128-byte id (dc_comm_unique_id) and hands it to the other ranks by whatever the caller has (MPI_Bcast, a file, a
 * socket); every rank then calls dc_comm_init with the same id. dc_allreduce_sum: in-place sum of `count` doubles over the ranks
 * through the context's stream (host buffer in, host buffer out; blocks). Python callers use torch.distributed instead
 * (diffcloth_amd/distributed.py). */
int dc_comm_unique_id(char *id128 /*128 bytes out*/);
int dc_comm_init(dc_ctx *ctx, int nranks, int rank, const char *id128);
int dc_allreduce_sum(dc_ctx *ctx, double *inout, int count);
int dc_comm_destroy(dc_ctx *ctx);
/* HIP-event timing of everything enqueued between the two calls on the context's stream (ms). */
int dc_timer_start(dc_ctx *ctx);
int dc_timer_stop(dc_ctx *ctx, float *ms);
/* accumulated device time (ms) and launch count of the forward / backward step kernels since the last
 * reset, measured with HIP events on the context's stream; used by bench.py's roofline block. A dc_rollout_* call
 * may run all its time steps in ONE launch (every rollout advances on its own), so launches <= steps.            */
int dc_kernel_times(dc_ctx *ctx, float *fwd_ms, int *fwd_launches, float *bwd_ms, int *bwd_launches, int reset);

#ifdef __cplusplus
}
#endif
#endif /* DIFFCLOTH_HIP_H */
