/*
 * mi_elasticity.h -- C-ABI of the MI355X-native structural-elasticity hot path.
 *
 * This is the drop-in boundary for the hot path of precice/dealii-adapter's nonlinear solver
 * (source/nonlinear_elasticity/nonlinear_elasticity.cc): one Newmark step =
 * Newton iterations x (cell tangent/residual assembly + CG) + Newmark vector updates.
 * The host-side Solid<dim>/Adapter (dealii-adapter_amd/host/) call exactly these entry points at the
 * sites cited below; nothing else crosses the boundary.  All device memory is owned by the opaque
 * context; the host only passes interface-sized buffers and scalars (plain pointers + sizes).
 *
 * Conventions: every function returns MI_OK (0) or a negative MI_E* code; mi_last_error() gives text.
 * One host thread per context; a context is not re-entrant.  DoF numbering is node-major:
 * dof = dim*node + component; nodes are lexicographic (x fastest) on the (p*reps+1)^dim lattice.
 */
#ifndef MI_ELASTICITY_H
#define MI_ELASTICITY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_OK 0
#define MI_EINVAL (-1)      /* bad argument / unsupported (dim, degree)            */
#define MI_EHIP (-2)        /* HIP runtime error                                   */
#define MI_ENOCONV_LIN (-3) /* CG hit max iterations (SolverControl::NoConvergence) */
#define MI_ENOCONV_NR (-4)  /* "No convergence in nonlinear solver!" (nonlinear_elasticity.cc:497) */
#define MI_ECOMM (-5)       /* RCCL error                                          */

/* role of each side of the box, order x-,x+,y-,y+,z-,z+ (deal.II colorize ids 0..5,
 * nonlinear_elasticity.cc:237-241), values = the ids the reference relabels to (:255-278, .h:256-257) */
#define MI_FACE_FREE 0
#define MI_FACE_CLAMPED 1   /* clamped_boundary_id: u = 0                     */
#define MI_FACE_INTERFACE 7 /* boundary_interface_id: coupling traction       */
#define MI_FACE_ZCLAMP 8    /* out_of_plane_clamped_mesh_id: u_z = 0 (3D)     */

typedef struct mi_ctx mi_ctx;

/* replaces make_grid (nonlinear_elasticity.cc:171-301) + system_setup (:305-380) inputs */
typedef struct
{
  int32_t       dim;       /* 2 or 3 (the reference's -DDIM, CMakeLists.txt:15-18)        */
  int32_t       degree;    /* FE_Q degree ("Polynomial degree", parameters.cc:110-113)    */
  int32_t       reps[3];   /* subdivided_hyper_rectangle repetitions                      */
  double        lo[3], hi[3];
  int32_t       face_role[6];
  const double *vertex_perturbation; /* optional nverts*dim offsets (tests), else NULL    */
} mi_mesh_desc;

/* parameters.cc:35-50 */
typedef struct
{
  double mu, nu, rho;
  double body_force[3];
} mi_material_desc;

/* parameters.cc:120-125, parameters.cc:12-15; alpha_1..6 derived as nonlinear_elasticity.h:242-250 */
typedef struct
{
  double beta, gamma, delta_t;
} mi_newmark_desc;

/* domain decomposition into `size` slabs along one box direction (`cut_axis`), one slab per process/GPU with ghost planes
 * exchanged by ncclSend/ncclRecv and scalars by ncclAllReduce (RCCL over xGMI).  size==1 (or a NULL descriptor):
 * single GPU.  rank == -1: all `size` slabs are created inside this process on one device and advance in
 * lockstep (test mode: lets the decomposition be checked on a single-GPU box).  All entry points keep speaking
 * GLOBAL arrays in every mode.  The reference has no counterpart (adapter.h:152-154 hard-codes one rank). */
typedef struct
{
  int32_t     rank, size;
  const void *nccl_unique_id; /* 128-byte ncclUniqueId from mi_comm_unique_id(), the same on all ranks */
  int32_t     cut_axis;       /* direction the slabs are cut along: 0 = automatic (the one with most cell layers; ties: the
                                 last, i.e. z-slabs of a cube), 1 / 2 / 3 = x / y / z.  The reference's flap is 18 x 3 (x 1)
                                 cells (nonlinear_elasticity.cc:189-205): only x can be cut into more than 3 parts */
} mi_comm_desc;

/* host-only description of one slab (no device needed; used by the multi-process CPU tests) */
typedef struct
{
  int32_t z0, z1, local_layers;      /* owned cell layers [z0,z1); layers of the local box (incl. ghost layer) */
  int64_t plane_nodes, node_offset;  /* nodes per lattice plane; global id of local node 0                    */
  int64_t nnodes_global, nnodes_local;
  int64_t own_begin, own_end;        /* owned LOCAL node range                                                */
  int64_t up_send, up_send_n, up_recv, up_recv_n;         /* LOCAL node ranges exchanged with rank+1          */
  int64_t down_send, down_send_n, down_recv, down_recv_n; /* ... and with rank-1                              */
  int32_t local_reps[3];
  double  local_lo[3], local_hi[3];
  int32_t local_face_role[6];
} mi_partition_info;
int mi_partition_describe(const mi_mesh_desc *mesh, int rank, int size, mi_partition_info *out);
/* host-only: the row order of the slab's SpMV (sliced-ELL, 64 rows per slice).  rows[] receives the LOCAL node of
 * every slot (-1 = padding) if it is not NULL and capacity suffices; slots [0, 64*n_interior_slices) hold the rows
 * without ghost columns (computed while the halo exchange is in flight), the rest the rows that wait for it. */
int mi_partition_spmv_rows(const mi_mesh_desc *mesh, int rank, int size, int64_t *n_slices, int64_t *n_interior_slices,
                           int32_t *rows, int64_t capacity);
int mi_comm_unique_id(void *out128); /* ncclGetUniqueId */
/* slabs of the decomposition this context belongs to, and ncclCommCount of its RCCL communicator (0: none) */
int mi_comm_info(const mi_ctx *ctx, int *team_size, int *rccl_ranks);
/* values[0..n) of rank 0 to every rank of the decomposition (collective; no-op in one process).  The reference couples
 * through ONE process (adapter.h:152-154, 213-225): with several GPUs only rank 0 owns the precice::Participant and what
 * it reads -- coupling data (:346-361), isCouplingOngoing, getMaxTimeStepSize, the checkpoint requests (:447-489) --
 * travels to the other ranks through this call (host/include/adapter/rank_zero_participant.h). */
int mi_comm_broadcast(mi_ctx *ctx, double *values, int32_t n);

/* parameters.cc:61-99 ("Solver" subsection) */
typedef struct
{
  double  tol_lin;            /* "Residual": CG stops at ||r|| <= tol_lin*||rhs||  (:1171-1172) */
  double  max_iterations_lin; /* "Max iteration multiplier" (x n_dofs)             (:1169-1170) */
  int32_t max_iterations_NR;  /* :436 */
  double  tol_f, tol_u;       /* :459-463 */
} mi_solver_desc;

typedef struct
{
  int32_t newton_iterations; /* linear solves done                          */
  int32_t assemblies;        /* = newton_iterations + 1 when converged      */
  int32_t lin_its_total;
  int32_t converged;
  double  res_norm, res_abs, upd_norm, upd_abs; /* last Newton table row (:489-494) */
  int32_t lin_its[16];       /* per Newton iteration (first 16)             */
  double  lin_res[16];
} mi_step_info;

/* ---- lifetime ---------------------------------------------------------------------------- */
int         mi_ctx_create(const mi_mesh_desc *mesh, const mi_material_desc *mat, const mi_newmark_desc *nm,
                          int device_id, const mi_comm_desc *comm, mi_ctx **out);
void        mi_ctx_destroy(mi_ctx *ctx);
const char *mi_last_error(const mi_ctx *ctx); /* ctx may be NULL: error of the last failed create */

/* ---- sizes / topology (global numbers, identical on every rank) --------------------------- */
int64_t mi_n_dofs(const mi_ctx *ctx);
int64_t mi_n_nodes(const mi_ctx *ctx);
int64_t mi_n_cells(const mi_ctx *ctx);
int64_t mi_nnz(const mi_ctx *ctx);   /* scalar non-zeros of the tangent (= dim^2 * stored blocks) */
int     mi_n_colours(const mi_ctx *ctx);
int     mi_get_node_coords(const mi_ctx *ctx, double *xyz /* n_nodes*dim */);
int     mi_get_constrained(const mi_ctx *ctx, uint8_t *flags /* n_dofs */);

/* ---- coupling interface: replaces Adapter::format_* (adapter.h:389-443) ------------------- */
/* interface nodes = nodes on MI_FACE_INTERFACE sides in ascending node (= x-dof) order (adapter.h:313-321);
 * coords interleaved [x0,y0,(z0),x1,...] exactly as passed to precice::setMeshVertices (adapter.h:305-326) */
int mi_n_interface_nodes(const mi_ctx *ctx);
int mi_get_interface_nodes(const mi_ctx *ctx, int32_t *node_ids, double *coords);
/* external_stress[interface dofs] = vals (format_precice_to_deal, adapter.h:421-443); vals [x0,y0,(z0),...] */
int mi_set_interface_traction(mi_ctx *ctx, int n, const double *vals);
/* vals = total_displacement[interface dofs] (format_deal_to_precice, adapter.h:389-417) */
int mi_get_interface_displacement(mi_ctx *ctx, int n, double *vals);

/* ---- the hot path, piecewise (call sites in solve_nonlinear_timestep, :410-499) ----------- */
int mi_newton_begin_step(mi_ctx *ctx);              /* solution_delta = 0 (:121); newton_update = 0 (:419) */
int mi_update_acceleration(mi_ctx *ctx);            /* :444 -> :592-599                                   */
int mi_assemble(mi_ctx *ctx, double *res_norm);     /* :446 -> :1044-1087 incl. Neumann :791-859, scatter
                                                       :760-774; *res_norm = get_error_residual :549-560  */
/* the same system_rhs and norm WITHOUT the tangent (bit-identical residual; tangent, Jacobi diagonal and multigrid
 * hierarchy keep the state of the last mi_assemble): for the assembly that only feeds the convergence test of
 * :459-463 -- callers use it when the update criterion already holds and fall back to mi_assemble if the residual
 * criterion fails.  No reference counterpart (the reference always builds both, :1078-1084). */
int mi_assemble_residual(mi_ctx *ctx, double *res_norm);
int mi_cg_solve(mi_ctx *ctx, double rel_tol, int64_t max_it, int *its, double *res);
                                                    /* :472 -> :1153-1191 (Jacobi-PCG, warm start) + :1208 */
/* "Solver type = Direct" (SparseDirectUMFPACK re-factorised in every Newton iteration, :1192-1200; the reference's shipped
 * default, parameters.prm:43): banded Cholesky factorisation of the current tangent + substitution on the device (one
 * workgroup; nodes renumbered with the shortest lattice direction fastest), then :1208.  *res = 0 as in :1199.  Meant for
 * the sizes of the reference's own geometries: MI_EINVAL ("too large ...") beyond ~3e8 flops of factorisation or on a
 * decomposed mesh -- callers then use mi_cg_solve at a tolerance of 1e-12.  MI_ENOCONV_LIN: non-positive pivot.
 * mi_set_tuning("solver_type", 1) makes mi_newmark_step and mi_linear_step solve this way (with that fallback); the
 * linear model factorises its constant matrix once (linear_elasticity.cc:553-559). */
int mi_direct_solve(mi_ctx *ctx, double *res);
int mi_apply_newton_update(mi_ctx *ctx, double *upd_norm); /* get_error_update :564-576; delta += update :487; then
                                                       the consumed update is cleared, so that the next solve of the
                                                       step starts from zero -- unless mi_set_tuning("cg_warm_start", 1)
                                                       asks for the reference's start vector, the previous update
                                                       (:419, :472-473: 28 instead of 22 CG iterations per step), or
                                                       "cg_warm_start" 2 for the solution of the same solve of the
                                                       previous time step (what the executable sets: 19-20) */
int mi_newmark_finish_step(mi_ctx *ctx);            /* :139-144: u += delta; a, v updates; old := new      */
/* the whole of solve_nonlinear_timestep + :139-144 with the reference's convergence logic */
int mi_newmark_step(mi_ctx *ctx, const mi_solver_desc *s, mi_step_info *info);

/* implicit-coupling checkpoint of the 6 state vectors (adapter.h:447-489), device-to-device */
int mi_state_save(mi_ctx *ctx);
int mi_state_restore(mi_ctx *ctx);

/* ---- linear model: ElastoDynamics (source/linear_elasticity/linear_elasticity.cc) -------------------------- */
/* state vectors of the linear model live in the same slots as the nonlinear ones */
enum
{
  MI_L_DISPLACEMENT     = 0, /* MI_V_TOTAL_DISPLACEMENT      */
  MI_L_OLD_DISPLACEMENT = 1, /* MI_V_TOTAL_DISPLACEMENT_OLD  */
  MI_L_VELOCITY         = 2,
  MI_L_OLD_VELOCITY     = 3,
  MI_L_OLD_STRESS       = 4, /* load vector F_n (assemble_rhs :402-409) */
  MI_L_STRESS           = 6, /* MI_V_EXTERNAL_STRESS: coupling data at the interface dofs */
  MI_L_SYSTEM_RHS       = 9
};
/* assemble_system (:248-374): K, M (QGauss(p+1)) once; stepping matrix M + theta^2 dt^2 K with the zero
 * boundary values of :426-451 eliminated; consistent-load operator of :458-521; body-force vector (:358-373) */
int mi_linear_setup(mi_ctx *ctx, double theta);
/* one pass of the time loop body (:676-684): assemble_rhs (:378-454, data_consistent: 1 "Stress" face integral,
 * 0 "Force" nodal forces), solve (:525-575; Jacobi-PCG, absolute tolerance, warm start from the previous
 * velocity), update_displacement (:579-586) */
int mi_linear_step(mi_ctx *ctx, int data_consistent, double abs_tol, int64_t max_it, int *its, double *res);
/* K (0), M (1) or the constrained stepping matrix (2) as scalar CSR, for parity tests */
int mi_linear_matrix_get_csr(mi_ctx *ctx, int which, int64_t *rowptr, int32_t *col, double *val);

/* per-vector device snapshots: what `old_state_data[i] = *state_variables[i]` (adapter.h:457-460) and its
 * inverse (:481-482) become when VectorType is a handle to a device-resident vector */
typedef struct mi_snapshot mi_snapshot;
int  mi_snapshot_create(mi_ctx *ctx, mi_snapshot **out);
void mi_snapshot_destroy(mi_ctx *ctx, mi_snapshot *s);
int  mi_snapshot_store(mi_ctx *ctx, mi_snapshot *s, int which); /* snapshot := vector `which` (MI_V_*) */
int  mi_snapshot_load(mi_ctx *ctx, const mi_snapshot *s, int which); /* vector `which` := snapshot   */

/* ---- inspection hooks (tests, bench) ------------------------------------------------------ */
enum
{
  MI_V_TOTAL_DISPLACEMENT = 0,
  MI_V_TOTAL_DISPLACEMENT_OLD,
  MI_V_VELOCITY,
  MI_V_VELOCITY_OLD,
  MI_V_ACCELERATION,
  MI_V_ACCELERATION_OLD,
  MI_V_EXTERNAL_STRESS,
  MI_V_SOLUTION_DELTA,
  MI_V_NEWTON_UPDATE,
  MI_V_SYSTEM_RHS,
  MI_V_COUNT
};
int mi_vec_get(mi_ctx *ctx, int which, double *host, int64_t n);
int mi_vec_set(mi_ctx *ctx, int which, const double *host, int64_t n);
/* tangent as scalar CSR (host arrays: rowptr n_dofs+1 (int64), col nnz (int32), val nnz) */
int mi_matrix_get_csr(mi_ctx *ctx, int64_t *rowptr, int32_t *col, double *val);
int mi_spmv(mi_ctx *ctx, const double *x_host, double *y_host); /* y = K x through the device kernel */
/* the dim x dim diagonal block of every node of the current tangent (host array [n_nodes][dim * dim], row-major, node order of
 * the reference) under the constraint rule of the assembly (nonlinear_elasticity.cc:760-774): out of the assembled tangent or,
 * with "fine_level" 1, as formed from the point records.  One process only (single or emulated slabs). */
int mi_get_diagonal_blocks(mi_ctx *ctx, double *blocks);

/* per-kernel-class device timings from HIP events on the context's stream */
enum
{
  MI_T_ASSEMBLE_CELLS = 0, /* all colours of one assembly                     */
  MI_T_ASSEMBLE_TOTAL,     /* memset + cells + faces + diag + norm            */
  MI_T_SPMV,
  MI_T_CG_VECTOR,          /* the two fused vector kernels of a CG iteration  */
  MI_T_CG_TOTAL,
  MI_T_NEWMARK,
  MI_T_STEP,               /* whole mi_newmark_step                           */
  MI_T_ASSEMBLE_DIAG,      /* matrix-free fine level ("fine_level" 1): diagonal blocks from the point records, per tangent
                              (this slot was the block-CSR -> sliced-ELL copy until round 2; no such copy exists)  */
  MI_T_ASSEMBLE_RESIDUAL,  /* all colours of one residual-only pass (mi_assemble_residual)              */
  MI_T_SPMV_PRECOND,       /* fine-level products of the multigrid preconditioner, per product                */
  MI_T_EBE_LAUNCH,         /* single launches of ebe_spmv (one colour of one element-tangent product), timed from
                              the dispatch itself on a sample of the products (every 6th)                      */
  MI_T_COUNT
};
typedef struct
{
  double  ms[MI_T_COUNT];    /* accumulated milliseconds   */
  int64_t count[MI_T_COUNT]; /* launches / calls           */
} mi_timings;
int mi_set_profiling(mi_ctx *ctx, int enable);
/* Run-time switches of a context (all its slabs).  None is needed for production use: the defaults are what bench.py and
 * the executables run, except "cg_warm_start", which both set to 2.  Unknown key or value out of range: MI_EINVAL.
 * The library reads NO environment variable (round 6): the column "env" names the hook of the EXPERIMENTS build only
 * (-DMI_EXPERIMENTS, make EXPERIMENTS=1 -> libmi_elasticity_exp.so: A/B runs of tools/), read once at context creation there.
 *
 *  key                  values (default first)     meaning                                                            env
 *  -------------------  -------------------------  -----------------------------------------------------------------  ----------------
 *  SOLVER POLICIES
 *  precond              1 | 0                      CG preconditioner: multigrid V-cycle (default from 75 k dofs) |      MI_PRECOND
 *                                                  Jacobi (default below)
 *  solver_type          0 | 1                      mi_newmark_step / mi_linear_step solve by PCG | banded Cholesky      -
 *                                                  ("Solver type = Direct", nonlinear_elasticity.cc:1192-1200)
 *  cg_warm_start        0 | 1 | 2 | 3              start vector of a solve: zero | the previous Newton update (the      MI_CG_WARM_START
 *                                                  reference, :419,472) | the same solve of the previous time step |
 *                                                  ... extrapolated over two steps; see mi_apply_newton_update
 *  cg_speculate         1 | 0                      multigrid-PCG: iterations up to (count of the same solve one step    -
 *                                                  earlier) - 2 are enqueued without polling the convergence flag, the
 *                                                  device decides; same iterates bit by bit | poll every iteration
 *  mg_lag               1 | 0                      coarse operators kept over the Newton iterations of a step | rebuilt  -
 *                                                  after every assembly
 *  mg_refresh_every     8 (1..1000)                coarse operators rebuilt at the first solve of every k-th step, or   MI_MG_REFRESH_EVERY
 *                                                  earlier when a solve needs 25 % (>= 2) more iterations than the
 *                                                  first one after a rebuild
 *  correct_face_F       0 | 1                      Neumann pull-back with F of CELL point fq (the reference's quirk,    MI_CORRECT_FACE_F
 *                                                  :825-827) | with F at the face point
 *  OPERATOR FORMS
 *  fine_level           0 | 1                      3D Q2: the fine level assembled (global tangent + sell_spmv: the     -
 *                                                  north-star path) | matrix-free end to end: a tangent assembly writes
 *                                                  point records, residual and the nodes' diagonal blocks only; the CG's
 *                                                  product, residual / start-vector products and the smoother run on
 *                                                  mf_spmv; the assembled tangent's memory is released.  Same results
 *                                                  (nonlinear_elasticity.cc:1044-1087, 1153-1191); excludes
 *                                                  "solver_type" 1, "precond_storage" 32 and matrix export
 *  mf_diag_lag          0 | 1                      "fine_level" 1: diagonal blocks (smoother's D, Jacobi diagonal) at     -
 *                                                  every tangent | at the first tangent of a time step, kept over its
 *                                                  Newton iterations (what bench.py and the executable set)
 *  face_slots           1 | 0                      Neumann term: all interface cells in one launch, contributions into    -
 *                                                  slots, summed per node in entry order | eight colour launches
 *  asm_box_geometry     1 | 0                      assemble_q2sf on meshes of axis-parallel boxes: 1/h and the volume     -
 *                                                  (as mf_spmv) | the trilinear map inverted at every point
 *  mf_point_slots       1 | 0                      "fine_level" 1: the point pass (records + residual) over all cells in  -
 *                                                  one launch, residual through the product's slots | eight colour launches
 *  smoother_operator    2 | 1 | 0                  fine-level products of the smoother on 3D Q2 slabs > 100 k nodes:    MI_EBE
 *                                                  matrix-free from the assembly's point records | stored element
 *                                                  tangents | assembled matrix
 *  cg_operator          0 | 1                      the CG's own product on the assembled matrix (north star) | in the   -
 *                                                  smoother's unassembled form (A/B)
 *  cg_r0_operator       1 | 0                      A h of a predicted start vector h ("cg_warm_start" 2 / 3) by the     -
 *                                                  matrix-free product where the point records exist | assembled
 *  precond_storage      64 | 32                    smoother multiplies with the fp64 matrices | an fp32-rounded copy    -
 *                                                  (opt-in; arithmetic, CG product, residuals stay fp64)
 *  mf_single_launch     1 | 0                      matrix-free product: one launch + gather | eight colour launches     MI_MF_SINGLE_LAUNCH
 *  cell_lattice         1 | 0                      mf_spmv: node ids of a cell by lattice arithmetic | from conn        MI_CELL_LATTICE
 *  element_tangents     2 | 1                      tests: keep point records / element tangents whatever the size       -
 *  DECOMPOSITION
 *  halo_overlap         1 | 0                      ghost planes exchanged on the communication stream next to the       MI_HALO_OVERLAP
 *                                                  interior rows | in line (same bits)
 *  mf_halo_overlap      1 | 0                      a slab's matrix-free product in launches over cell layers: the       -
 *                                                  layers away from the ghost planes while the halo travels | one
 *                                                  launch after the exchange (same bits)
 *  halo_skip            1 | 0                      no exchange before a product whose operand's ghost planes are        -
 *                                                  current (first post-smoothing step) | always exchange (same bits)
 *  smoother_quadrature  3 | 4                      Gauss points per direction of the multigrid smoother's fine-level       -
 *                                                  operator (3D Q2, where it multiplies matrix-free): the full-order
 *                                                  3 x 3 x 3 rule (mf_spmv27, two cells per wave, records of its own; the
 *                                                  V-cycle's residual and eigenvalue estimate use it too) | the assembly's
 *                                                  4 x 4 x 4 (nonlinear_elasticity.cc:74).  Preconditioner side only, fp64:
 *                                                  the CG's operator, residuals and the assembly always integrate with 4
 *  smoother_precision   64 | 32                    the smoother's matrix-free fine-level products in fp64 | in fp32      -
 *                                                  arithmetic on fp32 point records (opt-in; everything else stays fp64;
 *                                                  takes effect with the next tangent assembly)
 *  cg_single_reduction  -1 | 0 | 1                multigrid-PCG with ONE all-reduce per iteration (r.z, z.Az, ||r||^2;      MI_CG_SINGLE_REDUCTION
 *                                                  Chronopoulos-Gear form): on teams of several slabs | never | always
 *  cg_speculate_margin  0 | 1..16                  expected iterations of a solve left to polled ones (0: two)          -
 *  mg_dist_nodes        -1 | n                     node count from which the first coarsened multigrid level of a team   MI_MG_DIST_NODES
 *                                                  is cut into slabs of its own (default 65,536); an existing hierarchy
 *                                                  is rebuilt
 *  mg_coarsest          4 | 1..64                  cells per direction at which the multigrid coarsening stops            MI_MG_COARSEST
 *  mg_dense             1 | 0                      exact dense solve on the coarsest level (<= 384 dofs) | polynomial     MI_MG_DENSE
 *                                                  (both keys rebuild an existing hierarchy)
 *  mg_restrict_fuse     1 | 0                      the restriction takes the coarse level's first smoother step | a     MI_MG_RESTRICT_FUSE
 *                                                  launch of its own (same bits)
 *  KERNEL A/B (timing, tests)
 *  spmv_variant         3 | 1 | 4 | 11..14         sliced-ELL LDS-DMA product | row-per-wave cross-check | mi_spmv      MI_SPMV_VARIANT
 *                                                  through the unassembled form | streaming calibration kernels
 *  sell_icol            1 | 0                      column indices generated from the rows' column boxes | read          MI_SELL_ICOL
 *  sell_unroll, spmv_grid, xcd_remap               launch shape of the sliced-ELL product (MI_SELL_SPLIT=0: never the   MI_SELL_UNROLL
 *                                                  one-workgroup-per-slice kernel on short launches)
 *  cg_fused_dot         1 | 0                      p.q partials in the product's epilogue | separate reduction          MI_CG_FUSED_DOT
 *  small_cg             1 | 0                      matrices <= 1 MiB: whole Jacobi-PCG in one launch | three launches   MI_SMALL_CG
 *                                                  per iteration
 *  asm_variant          0 | 9 | 1,2 | 3 | 4-8      3D Q2 element kernel: sum factorised | node-pair form | its chunk    -
 *                                                  sizes | experiments build only: the sum-factorised kernel as of
 *                                                  round 4 | round-5 A/B combinations (profiles/r05/asm_ab_*.txt)
 *  asm_split            0 | 1 | 2                  experiments build only: the tangent as point pass + tangent kernel    -
 *                                                  from the point records (profiles/r06/asm_split_ab_n59.txt: slower)
 *  mg_fuse              1 | 0 | 2                  smoother update fused into the product on small levels | never |     MI_MG_FUSE
 *                                                  always
 *  mf_slots_cell_major  -1 | 0 | 1 | 2             result slots of the matrix-free kernels: follows smoother_quadrature     -
 *                                                  (3: cell-major, a cell's 81 results one contiguous run, the gathers read
 *                                                  through an index; 4: node-major) | forced (A/B; 2: line-major within an
 *                                                  x-row of cells, lattice meshes -- measured slower)
 *  spmv_as_smoother     0 | 1 | 2                  tests: mi_spmv / mi_bench_spmv apply the smoother's form of the operator -
 *                                                  | 2: mi_spmv applies M^-1, one V-cycle of the last solve's hierarchy
 *  mg_scale_lmax_percent 10..400                   tests: spoil the eigenvalue estimates once                           -
 *
 * Further switches of the experiments build only (read at creation; diagnostics): MI_MG_NU, MI_MG_NU_COARSE, MI_MG_RATIO, MI_MG_KIND,
 * MI_MG_BLOCK, MI_MG_THREE_TERM, MI_MG_COARSEST, MI_MG_DENSE, MI_MG_FACTOR, MI_MG_SAFETY, MI_MG_POWER_ITS,
 * MI_MG_COARSE_DEGREE, MI_MG_COARSE_RATIO, MI_MG_FUSE_MAX_NODES, MI_MG_VERBOSE (multigrid parameters);
 * MI_MF_XCD (XCD-aware cell order of mf_spmv), MI_ASM_CELL_LATTICE, MI_ASM_STAMPS / MI_MF_STAMPS / MI_MF_DBG (phase stamps
 * and timing-only ablations of the two element kernels). */
int mi_set_tuning(mi_ctx *ctx, const char *key, int value);
/* Read back.  Counters since the last mi_reset_timings (what a solve costs in latency-bound events; counted on one slab
 * as well, where the collectives themselves are no-ops): "count_scalar_allreduce", "count_scalar_allreduce_cg" (those inside
 * the linear solves), "count_vector_allreduce", "count_halo_exchange", "count_cg_host_sync", "count_cg_iterations", "count_cg_solves", "count_mg_refresh" (rebuilds of
 * the preconditioner's coarse operators).  State: "smoother_operator_active" (2 / 1: the smoother's fine-level products
 * are matrix-free / use the stored element tangents, 0: the assembled matrix), "precond", "spmv_variant",
 * "mf_single_launch", "cell_lattice", "mg_refresh_every", "cg_speculate", "halo_skip", "cut_axis" (1 / 2 / 3: the slabs
 * are cut along x / y / z, 0: not decomposed), "cg_single_reduction_active", "mg_distributed_levels" (multigrid levels cut
 * into slabs; 0 on one slab). */
int mi_get_tuning(mi_ctx *ctx, const char *key, int *value);
int mi_reset_timings(mi_ctx *ctx);
int mi_get_timings(mi_ctx *ctx, mi_timings *out);
/* isolated kernel benches on the current matrix/state: average ms per launch over reps */
int mi_bench_spmv(mi_ctx *ctx, int reps, double *ms_per_launch);
int mi_bench_assemble(mi_ctx *ctx, int reps, double *ms_per_assembly);

#ifdef __cplusplus
}
#endif
#endif /* MI_ELASTICITY_H */
