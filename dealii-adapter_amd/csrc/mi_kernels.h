// mi_kernels.h -- kernel argument blocks and launchers shared by mi_kernels.hip and mi_ctx.cpp
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

namespace mi
{
  // Environment switches are EXPERIMENT hooks (A/B runs, diagnostics of tools/): the release library reads none -- its
  // numerics and solver path do not change with the caller's environment; every supported switch is a mi_set_tuning key.
  // -DMI_EXPERIMENTS (make EXPERIMENTS=1 -> libmi_elasticity_exp.so, what tools/profile_round.sh builds) compiles the hooks
  // and the extra kernel instantiations they select.
#ifdef MI_EXPERIMENTS
  inline const char *exp_env(const char *name) { return getenv(name); }
#else
  inline const char *exp_env(const char *) { return nullptr; }
#endif

  // Node ids of a cell WITHOUT reading the connectivity (round 4).  The mesh is a lattice: the cell at colour-sorted
  // position pos belongs to colour col (begin[col] <= pos < begin[col+1]), is the r-th cell (x fastest) of that colour's
  // sub-lattice of mx x my x mz cells, and its first node is  base + rx sx + ry sy + rz sz;  local node (i, j, k) of a
  // 3D cell then is  node0 + i + nn0 j + nn01 k.  The two divisions of r are multiplications by floor(2^42 / d) + 1
  // (exact for r < 2^22, d < 2^20: r e < 2^42 with e = M d - 2^42 <= d), so a workgroup knows its nodes after a handful
  // of scalar instructions and ONE scalar load of its colour's row (32 bytes of a 256-byte table every workgroup reads:
  // scalar-cache resident) instead of after a round trip to memory for the connectivity (conn -> values was a chain of
  // two dependent vector loads at the head of mf_spmv and assemble_q2sf).  Built and checked against conn by
  // build_cell_lattice (mi_ctx.cpp); ncol == 0: not available (the kernels then read conn).  The per-colour rows live in
  // device memory, not in the argument block: a by-value table indexed by the colour made the compiler copy the whole
  // block to scratch at the head of every wave (0.72 instead of 0.40 ms per product, measured).
  struct CellLatticeRow
  {
    int32_t  mx, mxy, base, pz; // pz: parity of the colour's cells along the last lattice direction
    uint64_t magic_mx, magic_mxy;
  };
  struct CellLattice
  {
    int32_t               ncol = 0;
    int32_t               begin[9] = {};
    int32_t               sx = 0, sy = 0, sz = 0, nn0 = 0, nn01 = 0;
    const CellLatticeRow *rows = nullptr; // [8] device
  };

  // arguments of assemble_cells / neumann_faces (device pointers unless noted)
  struct AsmParams
  {
    const int32_t  *conn;   // [ncells][npc]       colour-sorted
    const double   *cverts; // [ncells][2^dim][dim]
    const uint16_t *off;    // [ncells][npc][npc]
    const int2     *rowinfo; // [nnodes] {base, gstride}: block (g, kx) of the node's row sits at base + g * gstride + kx of
                             // `vals` (mi::HostMesh::rowinfo; off[] holds g << 4 | kx); base -1: no row here (ghost
                             // node of a slab) -> nothing stored
    const uint8_t  *cmask;  // [nnodes]
    const double   *tab1d;  // N1[nq1][np1], dN1[nq1][np1], qw[nq1], qx[nq1]
    const double   *u, *du, *acc, *stress;
    double         *rhs;
    double         *vals;   // tangent, slice-interleaved block rows (mi_mesh.hpp) -- the layout the SpMV reads
    uint32_t        zero_blk, trash_blk; // two blocks behind the matrix: one that stays zero (what a first touch "reads"), one that
                                         // takes the stores of nodes without a row here (assemble_q2sf's branch-free scatter)
    double          mu, kappa, rho, alpha1;
    double          body[3];
    int64_t         cell_begin;
    int32_t         cell_count;
    int32_t         variant; // kernel variant for A/B timing
    int32_t         residual_only; // 1: residual without the tangent (Newton convergence check), same numbers
    double         *qrec;   // optional (3D Q2): the quadrature-point records the tangent is made of, [cell][MF_NREC][64] (see mf_spmv)
    float          *qrec32; // optional: the same records rounded to fp32 (opt-in fp32 smoother product)
    int32_t         axmap;    // how the lattice lies over the box (mi::AxisMap): bits 2d..2d+1 the physical coordinate lattice
                              // direction d runs along, bit 6+d set if backwards; 0x24 = aligned.  Only the Neumann term needs it
                              // (the reference pairs face and cell quadrature points by their index in PHYSICAL order)
    double         *inverted; // set to 1.0 by any quadrature point with det F <= 0 (the reference asserts det F > 0 there,
                              // nonlinear_elasticity.cc:935); sits next to the residual norm in the context's scalar block
    unsigned long long *stamps; // diagnostic (null in production): [cells of the launch][8] shader-clock stamps of one
                                // tangent wave at the phase boundaries of assemble_q2sf (mi_bench_assemble, MI_ASM_STAMPS)
    double         *ke;     // optional (3D Q2): the cell's masked element tangent, lower-triangle node-pair blocks, stored
                            // [cell][e = 0..8][block = a(a+1)/2 + b] -- the multigrid smoother's operator (see ebe_spmv)
    CellLattice     lat;    // 3D Q2: node ids by arithmetic (ncol == 0: read conn)
    double         *face_slots; // Neumann faces in one launch: [entries][npc * dim] (see neumann_faces); null: colour by colour
    double         *res_slots; // matrix-free fine level, point pass in one launch: [nslots][3] the cells' residual entries ...
    const int32_t  *slot_dst;  // ... at slot_dst[cell][27] (MfParams::dst); xcd_chunk: cells per XCD of that launch
    int32_t         xcd_chunk;
    const double   *cellbox;  // [ncells][4] = 1/hx, 1/hy, 1/hz, hx hy hz when every local cell is an axis-parallel box, else null
    int32_t         box_geometry; // assemble_q2sf: take 1/h and the volume from cellbox where present (tuning "asm_box_geometry")
    int32_t         from_records; // 3D Q2, records present: tangent in two kernels -- point pass, then the tangent from its records
    int32_t         correct_face_F; // Neumann term: 0 (default) = the reference's pull-back with the deformation gradient of CELL
                                    // quadrature point fq (nonlinear_elasticity.cc:825-827), 1 = F at the
                                    // face quadrature point itself (the physically consistent pull-back; the oracle's switch)
  };

  // product with the unassembled element tangents (see ebe_spmv in mi_kernels.hip)
  struct EbeParams
  {
    const double  *ke;
    const int32_t  *conn;  // [ncells][27] colour-sorted
    const uint32_t *first; // [ncells] bit a: first cell (in processing order) that contains its local node a -> store
    const double   *x;
    double         *y;     // complete after all colours (no zero fill needed: first touches store)
  };

  // matrix-free product from the quadrature-point records of the last tangent assembly (see mf_spmv in mi_kernels.hip)
  constexpr int MF_NREC = 11; // per point: F[9], J^(-2/3), 1/J -- the state the tangent is linearised at
  struct MfParams
  {
    const double   *qrec;    // [ncells][MF_NREC][64]
    const float    *qrec32;  // the same records rounded to fp32 (opt-in fp32 smoother product), or null
    const double   *qrec27;  // [ncells][MF_NREC][27] the records at the 27 points of the 3-point rule (smoother quadrature 3), or null
    const double   *tab27;   // N1[3][3], dN1[3][3], qw[3], qx[3] of that rule
    const int32_t  *conn;    // [ncells][27] colour-sorted
    const uint32_t *first;   // as EbeParams
    const uint8_t  *cmask;   // [nnodes]
    const double   *vals;    // assembled tangent: diagonal entries of constrained dofs
    const int32_t  *diagpos; // [nnodes] position of block (node,node) in vals (in blocks), -1: no row here
    const double   *tab1d;   // N1[4][3], dN1[4][3], qw[4], qx[4] (the assembly's tables)
    const double   *cverts;  // [ncells][8][3]
    double          mu, kappa;
    const double   *cellbox; // [ncells][4] = 1/hx, 1/hy, 1/hz, hx hy hz when every cell is an axis-parallel box, else null
    const double   *x;
    double         *y;
    double          mass;    // alpha_1 rho
    int32_t         count, xcd_chunk; // set by the launcher: cells of this launch, cells per XCD (0: plain order)
    // single-launch form (null: colour-by-colour update of y): every cell stores its results in its own slots
    double         *yc;        // [nslots][3] contributions
    const int32_t  *dst;       // [ncells][27] slot of (cell, local node)
    const int32_t  *slot_base; // [nnodes+1] first slot of every node (slots of a node in processing order of its cells)
    const int32_t  *slot_src;  // cell-major slots (dst[cell][a] = cell * 27 + a: a cell's 81 results are ONE contiguous run): position
                               // of the k-th contribution of a node, k in [slot_base[n], slot_base[n+1]); null: node-major (dst = k)
    int32_t         slot_inline; // 1: dst[cell][a] = cell * 27 + a, no need to read it (0: another order of the slots behind slot_src)
    CellLattice     lat;       // node ids by arithmetic (ncol == 0: read conn)
    unsigned long long *stamps; // diagnostic (null in production): [cells][8] shader-clock stamps at the stage boundaries
    // slabs, lattice ids only: a launch over the cells of the layers [z_a, z_b) of the last lattice direction alone (the
    // layers that touch no ghost plane of x run while the halo is in flight, the others after it).  Cells are sorted by
    // colour with z slowest, so the layers' cells are ONE contiguous range of positions per colour: workgroup `local` of the
    // launch takes position sel_pos0[c] + (local - sel_begin[c]) of colour c, sel_begin[c] <= local < sel_begin[c + 1].
    // sel_n == 0: all cells, position = local.
    int32_t         sel_n, sel_begin[9], sel_pos0[8];
  };

  // constant operators of the linear model (linear_elasticity.cc:248-374), one launch per colour
  struct LinAsmParams
  {
    const int32_t  *conn;    // [ncells][npc] colour-sorted
    const double   *cverts;  // [ncells][2^dim][dim]
    const uint16_t *off;     // as AsmParams::off
    const int2     *rowinfo; // as AsmParams::rowinfo
    const uint8_t  *cmask;   // [nnodes]
    const double   *tab;     // N1[nq1][np1], dN1[nq1][np1], qw[nq1], qx[nq1] of the LINEAR model's rule (nq1 = p + 1, :61)
    int32_t         np1, nq1;
    double          lambda, mu, rho, ctheta; // Lame parameters, density, theta^2 dt^2
    double          body[3];
    double         *K, *M, *A; // block values in the tangent's layout, zeroed by the caller
    double         *bodyvec;   // [ndofs] rho b integrated against the shape functions (:358-373), or null
    int64_t         cell_begin;
    int32_t         cell_count;
  };

  // row-per-wave cross-check product (tests) and the streaming calibration kernels
  struct SpmvParams
  {
    const int32_t *rowptr;  // [nnodes+1] block pattern
    const int32_t *col;
    const int2    *rowinfo; // [nnodes] as AsmParams::rowinfo
    const uint8_t *rowwx;   // [nnodes] x-width of the row's column box
    const double  *vals;
    const double  *x;
    double        *y;
    const double  *dotv;     // optional: partials of y . dotv
    double        *partials; // [grid]
    const int32_t *done;     // optional early-exit flag
    int64_t        row0, nrows;
    int64_t        nvalblocks; // number of stored blocks (timing-only stream kernels)
  };

  // sliced-ELL SpMV (see mi_kernels.hip)
  struct SellParams
  {
    const int32_t *perm; // [nslices*64]
    const int32_t *len;  // [nslices]
    const int64_t *off;  // [nslices+1]
    const int32_t *col;  // [nblk64*64]
    const int32_t *rowbox; // optional [nslices*64][2]: first column and box widths of every row; when set the kernel
                           // generates the column indices instead of reading `col` (lattice meshes: columns form a box)
    int32_t        nn0, nn1; // lattice points along x and y (column = x + nn0 * (y + nn1 * z))
    const double  *vals; // x-line-interleaved block rows (mi_mesh.hpp)
    const int32_t *wx;   // [nslices] x-width of the rows' column boxes
    const float   *vals32; // same layout, rounded to fp32: the smoother's copy (null: use vals)
    // fused Chebyshev-Jacobi epilogue (multigrid smoother; cheb_d == null: plain product).  Per owned row i:
    //   res = cheb_b - (K x); d = c1 d + c2 dinv res; cheb_xout = x + d   (x itself stays: other rows still gather it)
    // cheb_b set but cheb_d null: residual mode, y = cheb_b - K x
    const double  *cheb_b, *cheb_dinv;
    double        *cheb_d, *cheb_xout;
    double         cheb_c1, cheb_c2;
    int32_t        cheb_blk; // 1: cheb_dinv holds DxD inverse diagonal blocks per node (block-Jacobi)
    const double  *x;
    double        *y;
    const double  *dotv;
    double        *partials;
    const int32_t *done;
    int32_t        nslices;            // slices this launch works on ...
    int32_t        slice0;             // ... starting here (interior / boundary launches of a slab)
    int32_t        part0;              // first slot of `partials` this launch writes (one per workgroup)
    int32_t        own_begin, own_end; // rows (nodes) that contribute to the fused dot product
    int32_t        xcd_remap; // 1: workgroups of one XCD take a contiguous eighth of the slices (measured slower)
    int32_t        split;     // 1: small launch -- one workgroup of SELL_SPLIT_W wavefronts per slice (sell_spmv_split; the
                              // dot partials then count one per SLICE); set from the launch's slice count alone
  };

  struct CgParams
  {
    double       *x, *r, *p, *q;
    double       *s;     // single-reduction form: s = A p by recurrence (q then holds w = A z)
    const double *dinv;
    double       *part_rr, *part_rz, *part_pq;
    double       *sc;    // [8] device scalars
    int32_t      *flags; // [2] done, iterations
    int64_t       n;
    int32_t       npart, npart_pq;
    const double *z;      // preconditioned residual (general preconditioner); null -> Jacobi fused in the kernels
    double       *hist;   // optional [2*it] alpha, [2*it+1] beta of every iteration (Lanczos eigenvalue estimates)
    const double *totals; // distributed: all-reduced [rr, rz, pq, bb]; null -> consumers reduce the partials
  };

  struct NewmarkParams
  {
    double *u, *u_old, *v, *v_old, *a, *a_old;
    const double *du;
    double  alpha1, alpha2, alpha3, alpha4, alpha5, alpha6;
    int64_t n;
  };

  // index-space transfer between two lattices (see lattice_interp / lattice_restrict)
  struct LatticeParams
  {
    int64_t        n_tgt;    // nodes of the target (interp) or coarse (restrict) lattice
    int32_t        nt[3];    // its extents
    int32_t        ns[3];    // extents of the source (interp) / fine (restrict) lattice
    const int32_t *i0[3];    // interp: source index per target index
    const double  *w[3];     //         weight of i0+1
    const int32_t *rstart[3]; // restrict: per coarse index, range into ri/rw
    const int32_t *ri[3];
    const double  *rw[3];
    int32_t        rmax;      // restrict: longest list of any coarse index in any direction (<= 4: the unrolled kernel)
  };

  struct LinearParams
  {
    const double *load, *body;
    double       *f_old, *v, *d, *v_old, *d_old, *rhs, *w;
    double        theta, dt;
    int64_t       n;
  };

  void launch_linear_rhs_prepare(const LinearParams &p, hipStream_t s);
  void launch_linear_rhs_finish(int dim, const LinearParams &p, const double *mv, const double *kw,
                                const uint8_t *cmask, hipStream_t s);
  void launch_linear_update_displacement(const LinearParams &p, hipStream_t s);
  int  launch_assemble_cells(int dim, int degree, const AsmParams &p, hipStream_t s);
  int  launch_neumann_faces(int dim, int degree, const AsmParams &p, const int32_t *faces, int face_begin,
                            int face_count, hipStream_t s);
  void launch_neumann_gather(int dim, const double *slots, const int32_t *node_ids, const int32_t *start, const int32_t *src,
                             const uint8_t *cmask, double *rhs, int nnodes_if, hipStream_t s);
  void launch_spmv(int dim, const SpmvParams &p, int grid, hipStream_t s, int variant, int maxrow);
  void launch_sell_spmv(int dim, const SellParams &p, int grid, hipStream_t s, int unroll);
  void set_next_sell_launch_events(hipEvent_t start, hipEvent_t stop); // profiling: bracket exactly the next launch
  // ev_start / ev_stop (optional): bracket exactly this launch (dispatch-level events, profiling)
  void launch_ebe_spmv(const EbeParams &p, int64_t cell_begin, int32_t cell_count, hipStream_t s,
                       hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
  // (f32: the fp32 form -- production shape with p.qrec32 only, otherwise the fp64 kernel runs)
  void launch_mf_spmv(const MfParams &p, int64_t cell_begin, int32_t cell_count, hipStream_t s,
                      hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
  // smoother quadrature 3 (see mf_spmv27): the product with two cells per wave from the 27-point records, and the kernel that
  // writes those records from u + du
  void launch_mf_spmv27(const MfParams &p, int32_t cell_count, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
  void launch_mf_records27(const MfParams &p, const double *u, const double *du, double *rec27, int32_t cell_count, hipStream_t s);
  void launch_mf_gather(const MfParams &p, int64_t ndofs, hipStream_t s);
  // ... with the partials of dotv . y over the owned dofs (fixed grid): the CG's product on the matrix-free fine level
  void launch_mf_gather_dot(const MfParams &p, int64_t ndofs, const double *dotv, double *partials, int grid, int64_t own0,
                            int64_t own_n, hipStream_t s);
  // matrix-free fine level (round 6): the nodes' diagonal blocks from the point records -- every cell into its own slots
  // [nslots][6] (MfParams::dst), then summed per node under the assembled matrix's constraint rule (see mf_diag)
  void launch_mf_diag(const MfParams &p, double *slots6, int32_t cell_count, hipStream_t s);
  // the matrix-free fine level's point pass over ALL cells in one launch (assemble_q2sf<true> with the residual into slots:
  // AsmParams::res_slots / slot_dst; cell_begin = 0, cell_count = all) and the sum of the slots into system_rhs
  void launch_point_pass_slots(const AsmParams &p, hipStream_t s);
  void launch_residual_gather(const double *slots3, const int32_t *slot_base, const int32_t *slot_src, const uint8_t *cmask, double *rhs,
                              int64_t ndofs, hipStream_t s);
  void launch_mf_diag_gather(const double *slots6, const int32_t *slot_base, const int32_t *slot_src, const uint8_t *cmask,
                             const int32_t *diagpos, double *blk, double *dinv, double *dinv_blk, double *sym6, int64_t nnodes, hipStream_t s);
  // gather fused with the smoother's Chebyshev step (d != null: x += d in place) or residual (d == null: yres = b - K x)
  void launch_mf_gather_cheb3(const MfParams &p, const double *b, const double *dinv6, const double *xprev, const double *xcur,
                              double *xnext, double c1, double c2, int64_t node0, int64_t nnodes, hipStream_t s);
  void launch_mf_gather_cheb(const MfParams &p, const double *b, const double *dinv, double *d, double *xio, double *yres,
                             double c1, double c2, int64_t node0, int64_t nnodes, hipStream_t s);
  // coarsest multigrid level: dense inverse of the level's sliced-ELL matrix (n <= 96, -1 otherwise) and its application
  int  launch_dense_inverse_from_sell(int dim, const SellParams &p, int n, double *out, hipStream_t s);
  void launch_dense_apply(const double *inv, const double *b, double *x, int n, hipStream_t s);
  // banded Cholesky of a small tangent (direct solver): see band_cholesky_solve in mi_kernels.hip
  constexpr int BAND_NB = 16, BAND_MAXH = 512; // block-column width; the half bandwidth (in dofs) stays below BAND_MAXH
  void launch_band_extract(int dim, const SellParams &p, const int32_t *bperm, double *band, int hbw, hipStream_t s);
  int  launch_band_cholesky_solve(int dim, double *band, int n, int hbw, const int32_t *bperm, int nnodes, const double *b,
                                  double *x, double *work, int32_t *flag, bool factor, bool solve, hipStream_t s); // -1: band too wide
  constexpr int SELL_SPLIT_MAX_SLICES = 160; // launches up to this many slices take the k-split kernel (sell_spmv_split)
  constexpr int SELL_WPB = 3;   // wavefronts (= slices in flight) per workgroup of sell_spmv: launch grids and dot partials count in these
  constexpr int EBE_NBLK = 378; // 27 * 28 / 2 node-pair blocks of a 3D Q2 cell
  void launch_vals_to_f32(const double *vals, float *vals32, int64_t n, hipStream_t s); // the smoother's fp32-rounded copy (opt-in)
  void launch_sell_build_cols(const SellParams &p, const int32_t *rowptr, const int32_t *bsr_col, int32_t *sell_col,
                              hipStream_t s);
  void launch_dot_partials(const double *a, const double *b, int64_t n, double *part, int grid, hipStream_t s);
  void launch_finish_sum(const double *part, int n, double *out, hipStream_t s); // out[0] = sum of part[0..n)
  void launch_cheb4_start(double *x, double *d, double *r, const double *b, const double *q, const double *dinv,
                          double s0, int64_t n, hipStream_t s);
  void launch_cheb4_step(double *x, double *d, double *r, const double *q, const double *dinv, double beta, double ca,
                         double cb, int64_t n, hipStream_t s);
  void launch_extract_dinv_blk(int dim, const double *vals, const int32_t *diagpos, double *dinv, double *sym6, int64_t nnodes,
                               hipStream_t s);
  void launch_gather_diag_blocks(int dim, const double *vals, const int32_t *diagpos, double *out, int64_t nnodes, hipStream_t s);
  void launch_blk_apply(int dim, double *out, const double *a, const double *dinv, int64_t nnodes, hipStream_t s);
  void launch_cheb_step_blk(int dim, double *x, double *d, const double *b, const double *q, const double *dinv,
                            double c1, double c2, int64_t node0, int64_t nnodes, hipStream_t s);
  void launch_cheb_step(double *x, double *d, const double *b, const double *q, const double *dinv, double c1, double c2,
                        int64_t n, hipStream_t s);
  void launch_vec_scale_mul(double *dst, const double *a, const double *b, double s, int64_t n, hipStream_t st);
  void launch_vec_residual(double *res, const double *b, const double *q, int64_t n, hipStream_t s);
  void launch_vec_lincomb2(double *x, double a, const double *h1, double b, const double *h2, int64_t n, hipStream_t s);
  void launch_scale_start(double *x, double *q, const double *sc, int64_t n, hipStream_t s);
  void launch_copy_owned(double *y, const double *x, int64_t n, int64_t own0, int64_t own_n, hipStream_t s);
  void launch_lattice_interp(int dim, bool add, const LatticeParams &p, double *tgt, const double *src,
                             const uint8_t *cmask_tgt, hipStream_t s);
  bool launch_lattice_restrict_first_step(int dim, const LatticeParams &p, double *coarse, const double *fine,
                                          const uint8_t *cmask_coarse, double *x, double *d, const double *dinv_blk, double c2,
                                          int64_t node0, int64_t nnodes, hipStream_t s);
  void launch_lattice_restrict(int dim, const LatticeParams &p, double *coarse, const double *fine,
                               const uint8_t *cmask_coarse, hipStream_t s);
  void launch_cg_update_p(const CgParams &c, int it, int grid, hipStream_t s);
  void launch_cg_update_xr(const CgParams &c, int it, int grid, hipStream_t s);
  void launch_cg_update_single(const CgParams &c, int it, int grid, hipStream_t s);
  void launch_cg_init_residual(const CgParams &c, const double *b, double *part_bb, int grid, hipStream_t s);
  void launch_cg_set_tolerance(const CgParams &c, const double *part_bb, double rel_tol, hipStream_t s);
  void launch_cg_final_check(const CgParams &c, int it, hipStream_t s);
  // whole Jacobi-PCG in one single-workgroup launch (small problems, one slab)
  void launch_cg_small(int dim, const SellParams &p, const CgParams &c, const double *b, double rel_tol, int max_it,
                       hipStream_t s);
  void launch_assemble_linear(int dim, const LinAsmParams &p, hipStream_t s);
  void launch_extract_dinv(int dim, const double *vals, const int32_t *diagpos, double *dinv, int64_t nnodes,
                           hipStream_t s);
  void launch_masked_norm(int dim, const double *v, const uint8_t *cmask, int64_t n, double *part, int grid,
                          double *out, hipStream_t s);
  void launch_reduce_to_totals(const double *pa, int na, double *oa, const double *pb, int nb, double *ob,
                               const int32_t *done, hipStream_t s);
  void launch_team_sum(double *const *bufs, int nranks, int off, int cnt, hipStream_t s);
  void launch_gather_to_slots(int dim, const double *v, const int32_t *nodes, const int32_t *slots, int n, double *out,
                              hipStream_t s);
  void launch_zero_constrained(int dim, double *x, const uint8_t *cmask, int64_t n, hipStream_t s);
  void launch_newmark_acceleration(const NewmarkParams &p, hipStream_t s);
  void launch_newmark_finish(const NewmarkParams &p, hipStream_t s);
  void launch_vec_add(double *y, const double *x, int64_t n, hipStream_t s);
  void launch_gather_nodes(int dim, const double *v, const int32_t *nodes, int n, double *out, hipStream_t s);
  void launch_scatter_nodes(int dim, double *v, const int32_t *nodes, int n, const double *in, hipStream_t s);
} // namespace mi
