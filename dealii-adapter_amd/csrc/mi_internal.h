// mi_internal.h -- context object shared by the translation units of the library (not part of the C-ABI)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mi_elasticity.h"
#include "mi_kernels.h"
#include "mi_mesh.hpp"

namespace mi_detail
{
  constexpr int MAX_PART = 16384; // upper bound of per-workgroup reduction partials
  constexpr int CG_BATCH = 16;    // CG iterations enqueued between two host polls of the convergence flag

  enum
  {
    W_R = 0, // CG residual
    W_P,     // CG search direction
    W_Q,     // CG A*p
    W_DINV,  // Jacobi
    W_Z,     // preconditioned residual (multigrid)
    W_S,     // single-reduction CG: A*p by recurrence
    W_COUNT
  };
  // what an SpMV launch fuses on top of y = K x: partials of dotv . y, skipped altogether when *done != 0
  struct SpmvFusion
  {
    const double  *dotv;
    double        *partials;
    const int32_t *done;
  };
  constexpr int64_t SMALL_CG_MAX_MATRIX_BYTES = 1 << 20;
  constexpr double  DIRECT_MAX_FLOPS = 3.0e8; // n * hbw^2 of the banded Cholesky one workgroup is asked to do (a few ms)
  // Chebyshev-Jacobi step fused into the product (multigrid smoother): see mi::SellParams
  struct ChebFusion
  {
    const double *b, *dinv;
    double       *d, *xout;
    double        c1, c2;
    int           blk = 0; // dinv = DxD blocks per node
    int           inplace = 0; // matrix-free single-launch product only: the gather applies the step itself, x += d in
                               // place (the product is complete by then); d == null: y = b - K x
    // ... in three-term form (mf_gather_cheb3): x'' = x' + c1 (x' - xprev) + c2 D^-1 (b - K x') written to xnext (may be
    // xprev's buffer; xprev == null: zero), D^-1 from its symmetric half
    const double *dinv6 = nullptr, *xprev = nullptr;
    double       *xnext = nullptr;
  };
  struct LinearModel; // mi_linear.cpp
  struct Multigrid;   // mi_mg.cpp

  // the slab contexts that advance together (1 unless decomposed), see mi_ctx.cpp
  struct Team
  {
    std::vector<mi_ctx *> members; // local slab contexts (1 unless emulated)
    int                   size     = 1;
    bool                  emulated = false;
    bool                  owns_stream = true;
    void                 *nccl     = nullptr; // ncclComm_t
    hipStream_t           stream   = nullptr;
    hipStream_t           comm_stream = nullptr; // RCCL halo exchange next to the interior rows of the SpMV
    hipEvent_t            ev_ready = nullptr, ev_halo = nullptr;
    int                   overlap  = 1;          // 0: halo exchange in line on `stream`
    int                   mf_overlap = 1;        // the matrix-free product of a slab in two launches around its halo exchange
                                                 // (inner layers while the ghost planes travel); 0: one launch after it
    int                   halo_skip = 1;         // 1: products whose operand's ghost planes are known to be current (the
                                                 // first post-smoothing step after a prolongation that filled them) run
                                                 // without an exchange; 0: every product exchanges (A/B, bitwise the same)
    int                   device   = 0;
    int                   dim      = 0;
    std::vector<int>      cuts;                  // cell layers [cuts[r], cuts[r+1]) of slab r; empty: the balanced split
    double               *d_acc    = nullptr;    // RCCL: receive scratch of team_halo_accumulate
    size_t                acc_cap  = 0;
    int64_t               n_global = 0, nnodes_global = 0;
    std::vector<int64_t>  iface_global; // ascending global node ids
    std::vector<double>   iface_xyz;    // their coordinates
    double               *d_gbuf  = nullptr; // global-vector scratch (n_global doubles), on demand
    double               *d_ifbuf = nullptr; // global interface scratch (n_if * dim doubles)
    double              **d_sc_ptrs = nullptr; // emulated all-reduce: the members' scalar blocks
    mi_mesh_desc          md{};                // the undecomposed mesh AS THE LATTICE SEES IT (coarse multigrid levels are global)
    // decomposition along a direction other than the last: the lattice lies rotated over the box (mi::AxisMap).  e2i /
    // i2e: node of the reference's order (x fastest; what the C-ABI speaks) <-> node of the internal lattice; empty when
    // the two agree.  coord_of_axis goes to HostMesh::build of every context of the team (levels included).
    mi::AxisMap           amap;
    std::vector<int64_t>  e2i, i2e;
    std::vector<double>   perturb_int;         // vertex perturbation in internal vertex order (tests)
    int64_t ext_node(int64_t internal) const { return i2e.empty() ? internal : i2e[size_t(internal)]; }
    // what a solve costs in latency-bound events since the last mi_reset_timings: scalar all-reduces, vector all-reduces,
    // halo exchanges (complete for a decomposed team; a single slab skips the last two before their counters), host
    // synchronisations and iterations inside cg_run
    int64_t n_scalar_allreduce = 0, n_vector_allreduce = 0, n_halo = 0, n_cg_sync = 0, n_cg_its = 0, n_cg_solves = 0;
    int64_t n_scalar_allreduce_cg = 0; // those of n_scalar_allreduce made inside cg_run (the rest: Newton-level norms)
  };
} // namespace mi_detail

struct mi_ctx
{
  std::string   err;
  int           device = 0;
  hipStream_t   stream = nullptr;
  mi::HostMesh  mesh;
  mi::Tables1D  tab;
  int           dim = 0, degree = 0;
  int64_t       n = 0; // dofs
  mi_material_desc mat{};
  mi_newmark_desc  nm{};
  double        kappa = 0;
  double        alpha[7] = {0, 0, 0, 0, 0, 0, 0};

  // device memory
  int32_t  *d_conn = nullptr, *d_rowptr = nullptr, *d_col = nullptr, *d_diagpos = nullptr, *d_iface_nodes = nullptr,
          *d_faces = nullptr, *d_flags = nullptr;
  // Neumann faces in one launch (round 6): the cells' face contributions [entries][npc * dim] and, per interface node, the
  // list of (entry, local node) pairs that touch it (CSR over d_fn_ids), in entry order
  double   *d_face_slots = nullptr;
  int32_t  *d_fn_ids = nullptr, *d_fn_start = nullptr, *d_fn_src = nullptr;
  int       n_fn = 0;
  int       face_slots = 1; // tuning "face_slots": 1 one launch + gather (default), 0 eight colour launches
  int32_t  *d_rowinfo = nullptr; // [nnodes][2] where the row of a node starts in d_vals and its g-stride (mi::HostMesh::rowinfo)
  uint8_t  *d_rowwx = nullptr;   // [nnodes] x-width of the row's column box
  int32_t  *d_sell_wx = nullptr; // [nslices]
  double   *d_cverts = nullptr, *d_tab = nullptr, *d_vals = nullptr, *d_vecs = nullptr, *d_work = nullptr,
         *d_saved = nullptr, *d_part = nullptr, *d_sc = nullptr, *d_iface_buf = nullptr;
  int32_t  *d_sell_perm = nullptr, *d_sell_len = nullptr, *d_sell_col = nullptr, *d_sell_box = nullptr;
  int       sell_icol = 1; // 1: the SpMV generates the column indices from the rows' column boxes, 0: reads them
  int64_t  *d_sell_off = nullptr;
  // d_vals IS the sliced-ELL matrix (slice-interleaved block rows): the element scatter writes what the SpMV reads
  double   *d_dinv_blk = nullptr; // inverse diagonal blocks (block-Jacobi smoother), allocated when it is switched on
  double   *d_dinv_sym6 = nullptr; // 3D: their symmetric halves [nnodes][6] (xx yy zz xy xz yz) for the matrix-free smoother step
  bool      want_dinv_blk = false;
  double   *d_ke = nullptr;   // unassembled element tangents (3D Q2, single slab): the multigrid smoother's operator
  uint32_t *d_node_first = nullptr; // per cell: bit a = first touch of local node a (see HostMesh::node_first)
  float    *d_qrec32 = nullptr; // the same records in fp32 (opt-in "smoother_precision" 32)
  int       smoother_precision = 64;
  // tuning "smoother_quadrature" 3 (round 6; 3D Q2 with point records): the multigrid smoother's fine-level products, the
  // V-cycle's fine residual and the eigenvalue estimate integrate the tangent with 3 x 3 x 3 Gauss points (mf_spmv27: two
  // cells per wave) from records of their own; 4 (default): the assembly's 4 x 4 x 4 rule.  The CG's operator stays exact.
  int       smoother_points = 3; // (library default since round 6: same Newton tables and iteration counts, -11 % per step)
  double   *d_qrec27 = nullptr;  // [ncells][MF_NREC][27]
  double   *d_tab27  = nullptr;  // 1D tables of the 3-point rule
  bool      qrec27_valid = false;
  int       spmv_as_smoother = 0; // tests: mi_spmv applies the smoother's form of the operator
  bool      qrec32_valid = false; // d_qrec32 belongs to the current tangent
  double   *d_qrec = nullptr; // quadrature-point records of the last tangent assembly (3D Q2): the matrix-free form of the smoother's operator
  double   *d_mf_yc = nullptr;        // matrix-free product in one launch: per-(cell, node) contributions ...
  int32_t  *d_mf_dst = nullptr;       // ... their slots [ncells][27] ...
  int32_t  *d_mf_slot_base = nullptr; // ... and the first slot of every node [nnodes+1]
  int32_t  *d_mf_src = nullptr;       // cell-major slots: position of every contribution, node by node (MfParams::slot_src)
  int       slots_layout = 0;         // what build_slot_tables made: 0 node-major, 1 cell-major, 2 line-major
  int       slots_cell_major = -1;    // -1: follows the smoother's quadrature (3: cell-major, 4: node-major: what each measured faster
                                      // with); 0 / 1: forced (tuning "mf_slots_cell_major", A/B)
  int       mf_slots = 1;             // tuning "mf_single_launch": 1 one launch + gather (default), 0 eight colour launches
  mi::CellLattice lat;                // 3D Q2: node ids of a cell by arithmetic (ncol == 0: unavailable / switched off)
  mi::CellLattice lat_built;          // ... as built at creation (tuning "cell_lattice" 0 / 1 switches lat)
  mi::CellLatticeRow *d_lat_rows = nullptr; // its per-colour rows
  std::vector<mi::CellLatticeRow> lat_rows_host; // (host copy: layer ranges of a slab's product, enqueue_spmv)
  double   *d_cellbox = nullptr; // with d_qrec when every local cell is an axis-parallel box: [ncells][4] = 1/h, volume
  bool      ke_valid = false; // d_ke / d_qrec belong to the current tangent
  int64_t   ebe_products = 0; // element-tangent products so far (profiling samples every 6th)
  // "cg_warm_start" 2 / 3: the start vector of the j-th linear solve of a time step is the solution of the j-th solve
  // of the previous step (2) or its linear extrapolation over the last two steps (3); history per slab
  static constexpr int NPRED = 4;
  double   *d_pred[NPRED][2] = {}, *d_pred_saved[NPRED][2] = {}; // (saved: mi_state_save / mi_state_restore)
  int       pred_count[NPRED] = {}, pred_count_saved[NPRED] = {}, solves_this_step = 0;
  int       cg_operator = 0;   // A/B: 1 = the CG's own product on the element tangents too (no sliced-ELL copy); default 0:
                               // the assembled matrix, the kernel north_star names
  int       cg_warm_start = 0; // 0 (library default): every solve starts from zero, 1: later solves of a step start from the previous Newton
                               // update, as the reference's do, 2 / 3: from the same solve of the previous time step(s)
  int       cg_fused_dot = 1; // 1: p.q partials in the epilogue of the CG's product, 0: separate reduction (A/B)
  int       correct_face_F = 0; // tuning "correct_face_F": the Neumann pull-back with F at the face point (default: the reference's quirk)
  int       cg_r0_unassembled = 1; // A h of a predicted start vector by the matrix-free product where available ("cg_r0_operator")
  bool      unassembled_now = false; // (set around that one product)
  int64_t   mg_dist_nodes = -1;       // multigrid: node count from which the first coarsened level of a team is distributed
                                      // (-1: the default of mi_mg.cpp; read by mg_setup; tuning "mg_dist_nodes")
  int64_t   mg_coarsest = -1, mg_dense = -1; // tuning "mg_coarsest" / "mg_dense": Multigrid::coarsest_reps / dense (-1: defaults)
  int       cg_single_reduction = -1; // multigrid-PCG in the single-reduction form (one all-reduce per iteration): -1 = on
                                      // teams of several slabs, 0 never, 1 always (cg_run)
  int       cg_speculate_margin = 0;  // expected iterations left to polled ones: 0 = the default (two)
  int       cg_speculate = 1; // multigrid-PCG: enqueue the iterations the previous step's same solve needed (minus two)
                              // without polling the convergence flag in between (tuning "cg_speculate" 0: poll every one)
  int       pred_its[NPRED] = {}; // iterations of the j-th solve of the previous time step (0: unknown)
  // Matrix-free fine level (round 6; tuning "fine_level" 1; 3D Q2): the level keeps NO assembled tangent.  A tangent
  // assembly is the residual pass that also writes the point records (assemble_q2sf<true>) + the nodes' diagonal blocks
  // from those records (mf_diag); the CG's product, start-vector and residual products and the smoother all run on
  // mf_spmv; d_vals is released.  Same results as the assembled level [REF nonlinear_elasticity.cc:1044-1087, 1153-1191].
  int       mf_fine = 0;
  // "mf_diag_lag" 1 (what bench.py and the executable set beside "fine_level" 1): the diagonal blocks -- the smoother's D,
  // the Jacobi diagonal -- are formed at the FIRST tangent of a time step and kept over its Newton iterations, as the coarse
  // operators are ("mg_lag"); a preconditioner-side policy: the operator (records) and the residual are always current
  int       mf_point_slots = 1; // the level's point pass (records + residual) over all cells in ONE launch, residual through the
                                // product's slots (0: eight colour launches; the cells' sums in the same order)
  int       mf_diag_lag = 0;
  bool      mf_diag_fresh = false; // the blocks belong to this time step (cleared by mi_newton_begin_step)
  double   *d_diag_blk   = nullptr; // [nnodes][9] diagonal blocks under the assembled matrix's constraint rule
  double   *d_diag_slots = nullptr; // [ncells * 27][6] the cells' contributions (slot order = processing order)
  int32_t  *d_diagpos_mf = nullptr; // [nnodes] the node's own id where it has a row here, else -1: d_diag_blk read as `vals`
  size_t    vals_doubles = 0;       // size of d_vals (released while mf_fine, allocated again with "fine_level" 0)
  int       ebe = 2;          // tuning "smoother_operator": 2 matrix-free from the quadrature-point records, 1 element
                              // tangents (both where available), 0 assembled matrix
  float    *d_sell_vals32 = nullptr; // fp32-rounded copy for the multigrid smoother (tuning "precond_storage" 32)
  int       precond_storage = 64;
  int       small_cg = 1; // matrices up to SMALL_CG_MAX_MATRIX_BYTES on one slab: whole Jacobi-PCG in one launch
  // banded Cholesky factor of a small tangent ("Solver type = Direct"), allocated on first use
  double   *d_band = nullptr, *d_band_work = nullptr;
  int32_t  *d_band_perm = nullptr;
  int       band_hbw = 0;
  int       solver_direct = 0; // tuning "solver_type" 1: mi_newmark_step / mi_linear_step solve with the banded Cholesky
  uint16_t *d_off   = nullptr;
  uint8_t  *d_cmask = nullptr;
  double   *h_pinned = nullptr; // pinned host scratch (scalars, flags, interface buffer)
  size_t    h_pinned_doubles = 0;
  bool      have_saved = false;
  bool      newton_update_is_zero = false; // MI_V_NEWTON_UPDATE was cleared by the library and not written since
  bool      cg_breakdown = false; // the last solve stopped on a non-finite residual or p.Ap <= 0

  int grid_gdot = 1; // workgroups (= partials) of mf_gather_dot
  int grid_vec = 0, grid_spmv = 0, grid_spmv_int = 0, grid_spmv_bnd = 0; // grid_spmv = _int + _bnd (partials)
  bool split_int = false, split_bnd = false; // small launch: one workgroup per slice (mi::sell_spmv_split)
  int spmv_variant = 3, maxrow = 0, sell_unroll = 5, xcd_remap = 0; // tuning: SpMV kernel (3 = sliced-ELL); longest block row

  // profiling
  bool profiling = false;
  struct Stamp
  {
    hipEvent_t a, b;
    int        cls;
  };
  std::vector<Stamp> stamps;
  size_t             stamps_used = 0;
  mi_timings         timings{};

  // matrix / diagonal the SpMV and CG currently act on (tangent by default, linear-model operators otherwise)
  const double *active_sell_vals = nullptr;
  const double *active_dinv      = nullptr;
  std::vector<double> h_iface;   // host copy of the last interface values (linear model: consistent loading)
  mi_detail::LinearModel *linear = nullptr;

  // domain decomposition (z-slabs, mi_mesh.hpp SlabPartition); single rank: part.size == 1, everything is owned
  mi_detail::Team     *team = nullptr;
  mi::SlabPartition    slab;
  int64_t              own0 = 0, own_n = 0; // owned dof offset / count inside the local vectors
  std::vector<int32_t> iface_slot;          // global interface slot of every local interface node
  int32_t             *d_own_if_nodes = nullptr, *d_own_if_slots = nullptr; // owned interface nodes -> global slots
  int                  n_own_if = 0;

  // multigrid preconditioner of this slab (mi_mg.cpp); precond: 0 Jacobi, 1 multigrid V-cycle
  mi_detail::Multigrid *mg = nullptr;
  int                   precond = 1;
  bool                  vals32_stale = false; // the fp32-rounded copy (opt-in) is older than the tangent
  bool                  mg_stale = true; // the coarse operators belong to an older state than the fine tangent
  bool                  mg_force = true; // rebuild them at the next solve (set at the start of every time step)
  int                   asm_variant = 0;
  int                   asm_box_geometry = 1; // assemble_q2sf on a mesh of axis-parallel boxes: 1/h and the volume instead of the trilinear map
  int                   asm_split = 0;   // experiments build: 3D Q2 with point records: the tangent in two kernels (1 / 2; measured
                                         // slower than the fused kernel, profiles/r06/asm_split_ab_n59.txt)
  int                   mg_lag   = 1;    // 1: keep the coarse operators over the Newton iterations of one step
  // ... and over time steps: refreshed at the first solve of every k-th step, or before the next solve when one
  // needed a quarter (at least 2) more iterations than the first solve after the last refresh (mg_its_ref)
  int                   mg_refresh_every = 8, mg_steps_since_refresh = 0, mg_its_ref = 0;
  int64_t               n_mg_refresh = 0;

  double *vec(int which) { return d_vecs + size_t(which) * size_t(n); }
  double *work(int which) { return d_work + size_t(which) * size_t(n); }
  double *part(int which) { return d_part + size_t(which) * mi_detail::MAX_PART; }
};

namespace mi_detail
{
  int  fail(mi_ctx *c, int code, const char *fmt, ...);
  int  tic(mi_ctx *c, int cls, bool ext = false);
  void toc(mi_ctx *c, int id);
  int  sync(mi_ctx *c);
  // banded Cholesky on the device (small undecomposed problems): factor the matrix `vals` (tangent layout) and/or solve
  // x = K^-1 b; MI_EINVAL when the problem is too large for it (the caller then iterates)
  int            direct_prepare(mi_ctx *c);
  int            direct_factor_solve(mi_ctx *c, const double *vals, const double *b, double *x, bool factor, bool solve);
  void           refresh_vals32(mi_ctx *c); // fp32-rounded copy of the current tangent (opt-in smoother storage), if stale
  int            element_form(const mi_ctx *c); // 2 quadrature-point records, 1 element tangents, 0 none (current tangent)
  bool           mf_gather_fusable(const mi_ctx *c); // the smoother's product is the single-launch matrix-free form
  mi::SellParams sell_params(mi_ctx *c, const double *x, double *y, const double *dotv, double *partials,
                             const int32_t *done);
  // smoother: the product belongs to the multigrid preconditioner and may use the fp32-rounded copy of the values
  void enqueue_spmv(mi_ctx *c, const double *x, double *y, const double *dotv, double *partials, const int32_t *done,
                    int part = 0, bool smoother = false, const ChebFusion *cheb = nullptr);
  int  set_precond_storage(mi_ctx *c, int bits); // 64 | 32, for the context and its multigrid levels
  // Jacobi-PCG on the active matrix (all slabs of the team): x = vector x_id (warm start), b = vector b_id;
  // tol >= 0 relative to ||b||, tol < 0 absolute (-tol)
  // x_is_zero: the start vector is known to be zero (r0 = b without a product)
  int  cg_run(mi_ctx *c, int x_id, int b_id, double tol, int64_t max_it, int *its, double *res, bool x_is_zero = false,
              bool scale_start = false, // scale_start: the start vector is a prediction h, replaced by its best multiple
              int expected_its = 0);    // multigrid-PCG: iterations the same solve took one time step earlier (0: unknown)
  int  team_size(const mi_ctx *c);
  void linear_destroy(mi_ctx *c);
  int  create_member(Team &T, const mi_mesh_desc *md, const mi_material_desc *mat, const mi_newmark_desc *nm, int rank,
                     mi_ctx **out);
  void destroy_team(Team *T);
  int  enqueue_assembly(mi_ctx *c, bool residual_only = false);
  int  ensure_element_tangents(mi_ctx *c);
  // multigrid (mi_mg.cpp)
  int  mg_setup(mi_ctx *c);
  void mg_reset_estimates(Team &T); // eigenvalue estimates of all levels from scratch at the next operator update
  void mg_scale_estimates(Team &T, double f); // tests: spoil the current estimates // build the level hierarchy of a slab (once)
  void mg_destroy(mi_ctx *c);
  int  mg_update(Team &T);  // re-assemble the coarse operators for the current state (team-wide)
  int  mg_distributed_levels(const mi_ctx *c);
  int  mg_apply(Team &T);   // W_Z = V-cycle(W_R) on every slab of the team (team-wide, collective)
  bool mg_active(const mi_ctx *c);
  // team collectives (mi_ctx.cpp)
  int team_allreduce(Team &T, int off, int cnt);
  // ctx_of: whose slab geometry the vector has (null: the member's own; else e.g. its distributed multigrid level)
  int team_halo(Team &T, const std::function<double *(mi_ctx *)> &vec,
                const std::function<mi_ctx *(mi_ctx *)> &ctx_of = nullptr);
  int team_halo_begin(Team &T, const std::function<double *(mi_ctx *)> &vec,
                      const std::function<mi_ctx *(mi_ctx *)> &ctx_of = nullptr);
  int team_halo_end(Team &T);
  int team_spmv(Team &T, const std::function<mi_ctx *(mi_ctx *)> &ctx_of, const std::function<double *(mi_ctx *)> &x_of,
                const std::function<double *(mi_ctx *)> &y_of, const SpmvFusion *fusion, bool smoother = false,
                const ChebFusion *cheb = nullptr, // cheb: one entry per member
                bool ghosts_current = false);     // the ghost planes of x are up to date: no exchange (Team::halo_skip)
  int mg_set_storage(mi_ctx *c, int bits); // mi_mg.cpp: forwards to the level contexts
  int mg_set_fuse(mi_ctx *c, int fuse);
  int mg_set_restrict_fuse(mi_ctx *c, int on);
  int team_allreduce_vectors(Team &T, const std::function<double *(mi_ctx *)> &vec, size_t n);
  // the adjoint of team_halo: what every slab holds on its GHOST planes is added to the owner's copy of those planes
  int team_halo_accumulate(Team &T, const std::function<double *(mi_ctx *)> &vec,
                           const std::function<mi_ctx *(mi_ctx *)> &ctx_of);

#define HIPCHK(ctx, call)                                                                                   \
  do                                                                                                        \
    {                                                                                                       \
      hipError_t e_ = (call);                                                                               \
      if (e_ != hipSuccess)                                                                                 \
        return mi_detail::fail(ctx, MI_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                               __LINE__);                                                                   \
    }                                                                                                       \
  while (0)

  template <typename T>
  int upload(mi_ctx *c, T **dst, const std::vector<T> &src)
  {
    const size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    HIPCHK(c, hipMalloc((void **)dst, bytes));
    if (!src.empty())
      HIPCHK(c, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return MI_OK;
  }
} // namespace mi_detail
