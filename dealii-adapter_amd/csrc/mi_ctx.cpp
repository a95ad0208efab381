// mi_ctx.cpp -- context object and C-ABI (include/mi_elasticity.h) of the device hot path.
//
// Host logic here is the counterpart of Solid::solve_nonlinear_timestep / assemble_system /
// solve_linear_system (nonlinear_elasticity.cc:410-499, :1044-1087, :1153-1211): it owns the device
// state, orders the kernel launches on one HIP stream and turns HIP failures into status codes.
// There is no CPU fallback: every entry point needs a working HIP device.
//
// Domain decomposition: the mesh is cut into z-slabs (mi_mesh.hpp SlabPartition).  The slab contexts that
// advance together form a Team with one of three communication modes:
//   single   one slab = the whole mesh (no communication; fused reductions)
//   RCCL     one slab per process/GPU; ghost planes by ncclSend/ncclRecv, scalars by ncclAllReduce over xGMI
//   emulated several slabs inside ONE process on one GPU sharing a stream; copies and a sum kernel stand in for
//            the collectives.  Exists so that the decomposition can be parity-tested on a single-GPU box.
// All entry points speak GLOBAL arrays (dofs, interface nodes), whatever the mode.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "mi_internal.h"

namespace mi_detail
{
  std::string g_create_error;

  // slots of the per-context device scalar block d_sc[16]
  enum
  {
    SC_RZ0 = 0, // r.z ping-pong
    SC_RZ1 = 1,
    SC_TOL = 2,
    SC_RES = 3,
    SC_BNORM = 4,
    SC_INVERTED = 7, // set by the element kernels when a quadrature point has det F <= 0; travels with SC_NORM_RHS
    SC_NORM_RHS = 8,
    SC_NORM_UPD = 9,
    SC_TOT = 10, // [rr, rz, pq, bb] all-reduced totals of the distributed CG
    SC_START = 14 // [h.b, h.Ah] of a predicted start vector (cg_run, scale_start)
  };

  int team_size(const mi_ctx *c)
  {
    return c->team ? c->team->size : 1;
  }

  void set_create_error(const char *msg) { g_create_error = msg; } // for the context-free entry points (mi_partition.cpp)

  int fail(mi_ctx *c, int code, const char *fmt, ...)
  {
    char    buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c)
      {
        c->err = buf;
        if (c->team && !c->team->members.empty())
          c->team->members[0]->err = buf;
      }
    else
      g_create_error = buf;
    return code;
  }

#define TNCCL(T) (static_cast<ncclComm_t>((T).nccl))
#define NCCLCHK(ctx, call)                                                                                  \
  do                                                                                                        \
    {                                                                                                       \
      ncclResult_t r_ = (call);                                                                             \
      if (r_ != ncclSuccess)                                                                                \
        return fail(ctx, MI_ECOMM, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
    }                                                                                                       \
  while (0)

  // ---- profiling stamps: HIP events on the context's stream, resolved after a synchronize
  // ext: the events are handed to the launch itself (hipExtLaunchKernelGGL), nothing is recorded here
  int tic(mi_ctx *c, int cls, bool ext)
  {
    if (!c->profiling)
      return -1;
    if (c->stamps_used == c->stamps.size())
      {
        mi_ctx::Stamp s;
        if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess)
          return -1;
        c->stamps.push_back(s);
      }
    mi_ctx::Stamp &s = c->stamps[c->stamps_used];
    s.cls            = cls;
    if (!ext)
      hipEventRecord(s.a, c->stream);
    return int(c->stamps_used++);
  }
  void toc(mi_ctx *c, int id)
  {
    if (id >= 0)
      hipEventRecord(c->stamps[size_t(id)].b, c->stream);
  }
  void resolve_stamps(mi_ctx *c)
  {
    for (size_t i = 0; i < c->stamps_used; ++i)
      {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->stamps[i].a, c->stamps[i].b) == hipSuccess)
          {
            c->timings.ms[c->stamps[i].cls] += ms;
            c->timings.count[c->stamps[i].cls] += 1;
          }
      }
    c->stamps_used = 0;
  }
  int sync(mi_ctx *c)
  {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    resolve_stamps(c->team ? c->team->members[0] : c);
    return MI_OK;
  }

  // Closed form of the connectivity of a 3D lattice mesh (mi::CellLattice), checked cell by cell against conn: any
  // disagreement (or a mesh outside the exactness bounds of the reciprocal multiplications) leaves ncol = 0 and the
  // kernels keep reading conn.  rows: the per-colour table for the device.
  mi::CellLattice build_cell_lattice(const mi::HostMesh &m, std::vector<mi::CellLatticeRow> &rows)
  {
    mi::CellLattice L{};
    rows.assign(8, mi::CellLatticeRow{1, 1, 0, 0, 0, 0});
    if (m.dim != 3 || m.ncolours < 1 || m.ncolours > 8 || m.nnodes >= (int64_t(1) << 31))
      return L;
    const int p = m.p;
    L.nn0  = m.nn[0];
    L.nn01 = m.nn[0] * m.nn[1];
    L.sx   = 2 * p;
    L.sy   = 2 * p * m.nn[0];
    L.sz   = 2 * p * m.nn[0] * m.nn[1];
    for (int c = 0; c < m.ncolours; ++c)
      {
        const int64_t b = m.colour_begin[size_t(c)], e = m.colour_begin[size_t(c) + 1];
        if (e <= b || e - b >= (int64_t(1) << 22))
          return mi::CellLattice{};
        int ci[3];
        mi::HostMesh::split(m.cell_orig[size_t(b)], m.reps, 3, ci); // first cell of the colour: its parities
        const int par[3] = {ci[0] & 1, ci[1] & 1, ci[2] & 1};
        if (ci[0] != par[0] || ci[1] != par[1] || ci[2] != par[2])
          return mi::CellLattice{};
        const int mx = (m.reps[0] - par[0] + 1) / 2, my = (m.reps[1] - par[1] + 1) / 2, mz = (m.reps[2] - par[2] + 1) / 2;
        if (int64_t(mx) * my * mz != e - b || int64_t(mx) * my >= (int64_t(1) << 20))
          return mi::CellLattice{};
        L.begin[c]        = int32_t(b);
        rows[c].mx        = mx;
        rows[c].mxy       = mx * my;
        rows[c].magic_mx  = (uint64_t(1) << 42) / uint64_t(mx) + 1;
        rows[c].magic_mxy = (uint64_t(1) << 42) / uint64_t(mx * my) + 1;
        rows[c].base      = p * (par[0] + m.nn[0] * (par[1] + m.nn[1] * par[2]));
        rows[c].pz        = par[2];
      }
    for (int c = m.ncolours; c <= 8; ++c)
      L.begin[c] = int32_t(m.ncells); // unused colours are empty ranges at the end
    L.ncol = m.ncolours;
    // the check: every node of every cell, with the device's arithmetic
    const int npc = m.npc, np1 = m.np1;
    for (int64_t pos = 0; pos < m.ncells; ++pos)
      {
        int col = 0;
        for (int c = 1; c < 8; ++c)
          col += pos >= L.begin[c] ? 1 : 0;
        const mi::CellLatticeRow &R = rows[size_t(col)];
        const uint32_t r   = uint32_t(pos - L.begin[col]);
        const uint32_t rz  = uint32_t((uint64_t(r) * R.magic_mxy) >> 42);
        const uint32_t rem = r - rz * uint32_t(R.mxy);
        const uint32_t ry  = uint32_t((uint64_t(rem) * R.magic_mx) >> 42);
        const uint32_t rx  = rem - ry * uint32_t(R.mx);
        const int32_t  n0  = R.base + int32_t(rx) * L.sx + int32_t(ry) * L.sy + int32_t(rz) * L.sz;
        for (int a = 0; a < npc; ++a)
          {
            const int i = a % np1, j = (a / np1) % np1, k = a / (np1 * np1);
            if (m.conn[size_t(pos) * npc + a] != n0 + i + j * L.nn0 + k * L.nn01)
              return mi::CellLattice{};
          }
      }
    return L;
  }

  mi::AsmParams asm_params(mi_ctx *c)
  {
    mi::AsmParams p{};
    // node ids by arithmetic: measured 2 % SLOWER in assemble_q2sf (7.93 against 7.79 ms per tangent, same process: three
    // workgroups per CU already hide the connectivity load), 4 % faster in mf_spmv -- so only the product uses it;
    // MI_ASM_CELL_LATTICE=1 switches it on here for A/B
    static const bool asm_lat = mi::exp_env("MI_ASM_CELL_LATTICE") && atoi(mi::exp_env("MI_ASM_CELL_LATTICE")) != 0;
    if (asm_lat)
      p.lat = c->lat;
    p.conn   = c->d_conn;
    p.cverts = c->d_cverts;
    p.off    = c->d_off;
    p.rowinfo = reinterpret_cast<const int2 *>(c->d_rowinfo);
    p.cmask  = c->d_cmask;
    p.tab1d  = c->d_tab;
    p.u      = c->vec(MI_V_TOTAL_DISPLACEMENT);
    p.du     = c->vec(MI_V_SOLUTION_DELTA);
    p.acc    = c->vec(MI_V_ACCELERATION);
    p.stress = c->vec(MI_V_EXTERNAL_STRESS);
    p.rhs    = c->vec(MI_V_SYSTEM_RHS);
    p.vals   = c->d_vals;
    p.zero_blk  = uint32_t(c->mesh.nvalblocks() + 1);
    p.trash_blk = uint32_t(c->mesh.nvalblocks() + 2);
    p.mu     = c->mat.mu;
    p.kappa  = c->kappa;
    p.rho    = c->mat.rho;
    p.alpha1 = c->alpha[1];
    for (int i = 0; i < 3; ++i)
      p.body[i] = c->mat.body_force[i];
    p.variant = c->asm_variant;
    p.ke      = c->d_ke;
    p.qrec    = c->d_qrec;
    p.qrec32  = (c->d_qrec && c->smoother_precision == 32) ? c->d_qrec32 : nullptr;
    p.cellbox = c->d_cellbox;
    p.box_geometry = c->asm_box_geometry;
    p.from_records = c->d_qrec ? c->asm_split : 0;
    if (p.from_records) // both kernels of the pair take the node ids by lattice arithmetic (as mf_spmv)
      p.lat = c->lat;
    p.inverted = c->d_sc + SC_INVERTED;
    p.correct_face_F = c->correct_face_F;
    p.axmap    = 0;
    for (int d = 0; d < 3; ++d)
      p.axmap |= (c->team->amap.ext_axis[d] << (2 * d)) | ((c->team->amap.dir[d] < 0 ? 1 : 0) << (6 + d));
    return p;
  }

  mi::NewmarkParams newmark_params(mi_ctx *c)
  {
    mi::NewmarkParams p{};
    p.u      = c->vec(MI_V_TOTAL_DISPLACEMENT);
    p.u_old  = c->vec(MI_V_TOTAL_DISPLACEMENT_OLD);
    p.v      = c->vec(MI_V_VELOCITY);
    p.v_old  = c->vec(MI_V_VELOCITY_OLD);
    p.a      = c->vec(MI_V_ACCELERATION);
    p.a_old  = c->vec(MI_V_ACCELERATION_OLD);
    p.du     = c->vec(MI_V_SOLUTION_DELTA);
    p.alpha1 = c->alpha[1];
    p.alpha2 = c->alpha[2];
    p.alpha3 = c->alpha[3];
    p.alpha4 = c->alpha[4];
    p.alpha5 = c->alpha[5];
    p.alpha6 = c->alpha[6];
    p.n      = c->n; // whole local vector: pointwise updates keep the ghost copies consistent without communication
    return p;
  }

  mi::SpmvParams spmv_params(mi_ctx *c, const double *x, double *y, const double *dotv, double *partials,
                             const int32_t *done)
  {
    mi::SpmvParams p{};
    p.rowptr           = c->d_rowptr;
    p.col              = c->d_col;
    p.rowinfo          = reinterpret_cast<const int2 *>(c->d_rowinfo);
    p.rowwx            = c->d_rowwx;
    p.vals             = c->active_sell_vals ? c->active_sell_vals : c->d_vals;
    p.x                = x;
    p.y                = y;
    p.dotv             = dotv;
    p.partials         = partials;
    p.done             = done;
    p.row0             = c->slab.own_begin;
    p.nrows            = c->slab.own_end - c->slab.own_begin;
    p.nvalblocks       = c->mesh.nvalblocks();
    return p;
  }

  mi::SellParams sell_params(mi_ctx *c, const double *x, double *y, const double *dotv, double *partials,
                             const int32_t *done)
  {
    mi::SellParams p{};
    p.perm      = c->d_sell_perm;
    p.len       = c->d_sell_len;
    p.off       = c->d_sell_off;
    p.col       = c->d_sell_col;
    p.rowbox    = c->sell_icol ? c->d_sell_box : nullptr;
    p.nn0       = c->mesh.nn[0];
    p.nn1       = c->mesh.nn[1];
    p.vals      = c->active_sell_vals ? c->active_sell_vals : c->d_vals;
    p.wx        = c->d_sell_wx;
    p.x         = x;
    p.y         = y;
    p.dotv      = dotv;
    p.partials  = partials;
    p.done      = done;
    p.nslices   = int32_t(c->mesh.sell_nslices);
    p.own_begin = int32_t(c->slab.own_begin);
    p.own_end   = int32_t(c->slab.own_end);
    p.xcd_remap = c->xcd_remap;
    return p;
  }

  // The tangent is assembled straight into the layout the SpMV reads (mi_mesh.hpp: slice-interleaved block rows), so no
  // copy stands between an assembly and the first product.  Only the opt-in fp32-rounded smoother storage
  // ("precond_storage" 32) keeps a second array, refreshed lazily before the first product with a new tangent.
  void refresh_vals32(mi_ctx *c)
  {
    if (!c->vals32_stale || c->precond_storage != 32 || !c->d_sell_vals32)
      return;
    mi::launch_vals_to_f32(c->d_vals, c->d_sell_vals32, c->mesh.nvalblocks() * int64_t(c->dim * c->dim), c->stream);
    c->vals32_stale = false;
  }

  // ---- direct solver (banded Cholesky in one workgroup) for the sizes of the reference's own geometries
  int direct_prepare(mi_ctx *c)
  {
    if (c->team->size != 1)
      return fail(c, MI_EINVAL, "the direct solver runs on an undecomposed mesh only");
    if (c->d_band)
      return MI_OK;
    std::vector<int32_t> perm;
    const int            hbw = c->mesh.band_perm(perm);
    const double         flops = double(c->n) * double(hbw) * double(hbw);
    if (hbw >= mi::BAND_MAXH || flops > DIRECT_MAX_FLOPS)
      return fail(c, MI_EINVAL, "system too large for the device direct solver (%lld dofs, half bandwidth %d): use the CG",
                  (long long)c->n, hbw);
    c->band_hbw = hbw;
    int rc      = upload(c, &c->d_band_perm, perm);
    if (rc)
      return rc;
    HIPCHK(c, hipMalloc((void **)&c->d_band, size_t(c->n) * size_t(hbw + 1) * sizeof(double)));
    HIPCHK(c, hipMalloc((void **)&c->d_band_work, size_t(c->n) * sizeof(double)));
    return MI_OK;
  }

  int direct_factor_solve(mi_ctx *c, const double *vals, const double *b, double *x, bool factor, bool solve)
  {
    int rc = direct_prepare(c);
    if (rc)
      return rc;
    int32_t *flag = c->d_flags + 3;
    if (factor)
      {
        HIPCHK(c, hipMemsetAsync(flag, 0, sizeof(int32_t), c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_band, 0, size_t(c->n) * size_t(c->band_hbw + 1) * sizeof(double), c->stream));
        mi::SellParams sp = sell_params(c, nullptr, nullptr, nullptr, nullptr, nullptr);
        sp.vals           = vals;
        mi::launch_band_extract(c->dim, sp, c->d_band_perm, c->d_band, c->band_hbw, c->stream);
      }
    if (mi::launch_band_cholesky_solve(c->dim, c->d_band, int(c->n), c->band_hbw, c->d_band_perm, int(c->mesh.nnodes), b, x,
                                       c->d_band_work, flag, factor, solve, c->stream))
      return fail(c, MI_EINVAL, "band too wide for the device direct solver");
    HIPCHK(c, hipGetLastError());
    if (factor)
      {
        int32_t *h = reinterpret_cast<int32_t *>(c->h_pinned + 8);
        HIPCHK(c, hipMemcpyAsync(h, flag, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        if ((rc = sync(c)))
          return rc;
        if (h[0])
          return fail(c, MI_ENOCONV_LIN, "direct solver: the matrix is not positive definite (non-positive pivot in the "
                                         "Cholesky factorisation)");
      }
    return MI_OK;
  }

  // which unassembled form of the current tangent the element products use: 2 quadrature-point records (mf_spmv),
  // 1 element tangents (ebe_spmv), 0 none
  int element_form(const mi_ctx *c)
  {
    if (!c->ke_valid)
      return 0;
    if (c->d_qrec && (c->ebe == 2 || !c->d_ke))
      return 2;
    return c->d_ke ? 1 : 0;
  }

  bool mf_gather_fusable(const mi_ctx *c)
  {
    return element_form(c) == 2 && c->ebe == 2 && c->mf_slots && c->d_mf_yc && c->precond_storage == 64 &&
           !c->active_sell_vals && c->d_dinv_blk;
  }

  // y = K x on the owned rows (+ optional fused dot partials); x and y are whole local vectors.
  // part: 0 all rows, 1 interior rows only (no ghost columns: may run while the halo is in flight), 2 boundary rows
  void enqueue_spmv(mi_ctx *c, const double *x, double *y, const double *dotv, double *partials, const int32_t *done,
                    int part, bool smoother, const ChebFusion *cheb)
  {
    // product with the unassembled element tangents: the smoother's fine-level products (and, for tests, any plain
    // product under "spmv_variant" 4); not for fused epilogues, fused dot products or the linear model's operators
    // (opt-in A/B "cg_operator" 1: the CG's own product as well, with p.q by a separate reduction -- then the
    // sliced-ELL copy of the tangent is never made)
    const int  kind       = element_form(c);
    const bool mf_all     = c->mf_fine && !c->active_sell_vals; // matrix-free fine level: every product of the tangent
    const bool ebe_for_cg = dotv && (c->cg_operator == 1 || mf_all) && kind && !c->active_sell_vals && !cheb;
    const bool cheb_ok    = !cheb || (cheb->inplace && smoother && mf_gather_fusable(c));
    if (ebe_for_cg || (kind && cheb_ok && !dotv && !c->active_sell_vals &&
                       (mf_all || (smoother ? (c->ebe != 0 && c->precond_storage == 64) : (c->spmv_variant == 4 || c->unassembled_now)))))
      {
        // on a slab every local cell (own layers + ghost layer) contributes to owned rows, and the cells are not sorted
        // by layer: the whole product waits for the ghost planes of x (part 2 = after the halo exchange); the rows of
        // the ghost planes receive partial sums that nobody reads
        // ... unless the cells can be told apart by their layer (round 4: matrix-free, one launch, lattice ids): the cells that
        // touch no ghost plane of x -- all but the lowest layer (rank > 0) and the ghost layer (rank < size - 1) -- run as part 1
        // while the halo is in flight, the others as part 2; every cell still writes its own slots, so the sum is the same
        // smoother quadrature 3: the smoother's products (never the CG's) from the 27-point records, two cells per wave, in the
        // same launches over layers
        const bool q27 = smoother && !ebe_for_cg && c->smoother_points == 3 && c->qrec27_valid && kind == 2 && c->mf_slots && c->d_mf_yc &&
                         !(c->smoother_precision == 32 && c->qrec32_valid); // (the opt-in fp32 smoother products keep the 64-point kernel)
        const bool mf_split = part != 0 && kind == 2 && c->mf_slots && c->d_mf_yc && c->lat.ncol > 0 && c->team->mf_overlap &&
                              !(ebe_for_cg && !mf_all);
        if (part == 1 && !mf_split)
          return;
        mi::EbeParams e{c->d_ke, c->d_conn, c->d_node_first, x, y};
        mi::MfParams  f{};
        f.qrec    = c->d_qrec;
        // opt-in "smoother_precision" 32: the SMOOTHER's products in fp32 arithmetic on fp32 records (residuals, start-vector
        // products and mi_spmv keep the fp64 form)
        f.qrec32  = (smoother && c->smoother_precision == 32 && c->qrec32_valid) ? c->d_qrec32 : nullptr;
        f.qrec27  = c->d_qrec27;
        f.tab27   = c->d_tab27;
        f.conn    = c->d_conn;
        f.first   = c->d_node_first;
        f.cmask   = c->d_cmask;
        f.vals    = c->mf_fine ? c->d_diag_blk : c->d_vals;      // (diagonal entries of constrained dofs)
        f.diagpos = c->mf_fine ? c->d_diagpos_mf : c->d_diagpos;
        f.tab1d   = c->d_tab;
        f.cverts  = c->d_cverts;
        f.mu      = c->mat.mu;
        f.kappa   = c->kappa;
        f.cellbox = c->d_cellbox;
        f.x       = x;
        f.y       = y;
        f.mass    = c->alpha[1] * c->mat.rho;
        f.lat     = c->lat;
        const bool one_launch = kind == 2 && c->mf_slots && c->d_mf_yc;
        if (one_launch)
          {
            f.yc        = c->d_mf_yc;
            f.dst       = c->d_mf_dst;
            f.slot_base = c->d_mf_slot_base;
            f.slot_src  = c->d_mf_src;
            f.slot_inline = c->slots_layout == 1;
          }
        // profiling: every 6th product has its launches timed from the dispatch itself (kernel start / end as a
        // profiler reports them), class MI_T_EBE_LAUNCH
        mi_ctx    *c0     = c->team->members[0];
        // (a product that runs in two parts around a halo exchange counts once -- with its second part -- and the
        // launch that holds the bulk of the cells, part 1, is the one that is timed)
        const bool counts = !(mf_split && part == 1);
        // (the SMOOTHER's products: the class times one kernel -- the CG's own matrix-free products and the start-vector
        // products of a matrix-free fine level are 64-point launches beside the smoother's 27-point ones)
        const bool sample = c0->profiling && smoother && (counts ? c->ebe_products++ % 6 == 0 : c->ebe_products % 6 == 0);
        if (one_launch) // all cells at once (no two cells share a slot), then the sum over the slots of every node
          {
            if (mf_split)
              {
                // layers [za, zb) of the slab's cells: one contiguous range of positions per colour (z slowest)
                auto layers = [&](int za, int zb, bool timed) {
                  mi::MfParams g = f;
                  int32_t      n = 0;
                  for (int col = 0; col < 8; ++col)
                    {
                      g.sel_begin[col] = n;
                      g.sel_pos0[col]  = 0;
                      if (col >= c->lat.ncol)
                        continue;
                      const mi::CellLatticeRow &R = c->lat_rows_host[size_t(col)];
                      const int mz = int((c->lat.begin[col + 1] - c->lat.begin[col]) / R.mxy);
                      // cells of the colour have cz = 2 rz + pz: rz in [ceil((za - pz) / 2), ceil((zb - pz) / 2))
                      const int ra = std::min(mz, std::max(0, (za - R.pz + 1) / 2)), rb = std::min(mz, std::max(0, (zb - R.pz + 1) / 2));
                      g.sel_pos0[col] = c->lat.begin[col] + ra * R.mxy;
                      n += std::max(0, rb - ra) * R.mxy;
                    }
                  g.sel_begin[8] = n;
                  g.sel_n        = n;
                  if (n > 0)
                    {
                      const int t = timed ? tic(c0, MI_T_EBE_LAUNCH, true) : -1;
                      if (q27)
                        mi::launch_mf_spmv27(g, n, c->stream, t >= 0 ? c0->stamps[size_t(t)].a : nullptr,
                                             t >= 0 ? c0->stamps[size_t(t)].b : nullptr);
                      else
                        mi::launch_mf_spmv(g, 0, n, c->stream, t >= 0 ? c0->stamps[size_t(t)].a : nullptr,
                                           t >= 0 ? c0->stamps[size_t(t)].b : nullptr);
                    }
                };
                const int nzl = c->mesh.reps[2];
                const int zlo = c->slab.rank > 0 ? 1 : 0, zhi = c->slab.rank + 1 < c->team->size ? nzl - 1 : nzl;
                if (part == 1)
                  {
                    layers(zlo, zhi, sample); // no ghost plane of x in reach: while the halo is in flight
                    return;           // (the slot sum follows the other layers)
                  }
                layers(0, zlo, false);
                layers(zhi, nzl, false);
              }
            else if (q27)
              {
                const int t = sample ? tic(c0, MI_T_EBE_LAUNCH, true) : -1;
                mi::launch_mf_spmv27(f, int32_t(c->mesh.ncells), c->stream, t >= 0 ? c0->stamps[size_t(t)].a : nullptr,
                                     t >= 0 ? c0->stamps[size_t(t)].b : nullptr);
              }
            else
              {
                const int t = sample ? tic(c0, MI_T_EBE_LAUNCH, true) : -1;
                mi::launch_mf_spmv(f, 0, int32_t(c->mesh.ncells), c->stream, t >= 0 ? c0->stamps[size_t(t)].a : nullptr,
                                   t >= 0 ? c0->stamps[size_t(t)].b : nullptr);
              }
            if (cheb && cheb->xnext) // the smoother's step on the owned nodes, straight from the slots (three-term form)
              mi::launch_mf_gather_cheb3(f, cheb->b, cheb->dinv6, cheb->xprev, x, cheb->xnext, cheb->c1, cheb->c2, c->own0 / 3,
                                         c->own_n / 3, c->stream);
            else if (cheb) // ... with the update vector d, in place / the residual
              mi::launch_mf_gather_cheb(f, cheb->b, cheb->dinv, cheb->d, const_cast<double *>(x), y, cheb->c1, cheb->c2,
                                        c->own0 / 3, c->own_n / 3, c->stream);
            else if (ebe_for_cg) // the CG's q = K p: the slot sum and the partials of p.q in one launch
              mi::launch_mf_gather_dot(f, int64_t(c->mesh.nnodes) * 3, dotv, partials, c->grid_gdot, c->own0, c->own_n, c->stream);
            else
              mi::launch_mf_gather(f, int64_t(c->mesh.nnodes) * 3, c->stream);
          }
        else
        for (int col = 0; col < c->mesh.ncolours; ++col)
          {
            const int32_t cnt = int32_t(c->mesh.colour_begin[col + 1] - c->mesh.colour_begin[col]);
            const int     t   = (sample && cnt > 0) ? tic(c0, MI_T_EBE_LAUNCH, true) : -1;
            if (kind == 2)
              mi::launch_mf_spmv(f, c->mesh.colour_begin[col], cnt, c->stream, t >= 0 ? c0->stamps[size_t(t)].a : nullptr,
                                 t >= 0 ? c0->stamps[size_t(t)].b : nullptr);
            else
              mi::launch_ebe_spmv(e, c->mesh.colour_begin[col], cnt, c->stream, t >= 0 ? c0->stamps[size_t(t)].a : nullptr,
                                  t >= 0 ? c0->stamps[size_t(t)].b : nullptr);
          }
        if (ebe_for_cg && !one_launch) // partials of dotv . y over the owned dofs (the early-exit flag is honoured by their consumer)
          mi::launch_dot_partials(dotv + c->own0, y + c->own0, c->own_n, partials, c->grid_vec, c->stream);
        return;
      }
    if (c->spmv_variant == 3 || c->spmv_variant == 4 || c->active_sell_vals) // linear-model operators exist in sliced-ELL form only
      {
        refresh_vals32(c);
        mi::SellParams p   = sell_params(c, x, y, dotv, partials, done);
        if (smoother && c->precond_storage == 32 && c->d_sell_vals32 && !c->active_sell_vals)
          p.vals32 = c->d_sell_vals32;
        if (cheb)
          {
            p.cheb_b    = cheb->b;
            p.cheb_dinv = cheb->dinv;
            p.cheb_d    = cheb->d;
            p.cheb_xout = cheb->xout;
            p.cheb_c1   = cheb->c1;
            p.cheb_c2   = cheb->c2;
            p.cheb_blk  = cheb->blk;
          }
        const int32_t  nin = int32_t(c->mesh.sell_nslices_interior), nbd = int32_t(c->mesh.sell_nslices) - nin;
        if (part != 2 && nin > 0)
          {
            p.slice0  = 0;
            p.nslices = nin;
            p.part0   = 0;
            p.split   = c->split_int;
            mi::launch_sell_spmv(c->dim, p, c->grid_spmv_int, c->stream, c->sell_unroll);
          }
        if (part != 1 && nbd > 0)
          {
            p.slice0  = nin;
            p.nslices = nbd;
            p.part0   = nin > 0 ? c->grid_spmv_int : 0;
            p.split   = c->split_bnd;
            mi::launch_sell_spmv(c->dim, p, c->grid_spmv_bnd, c->stream, c->sell_unroll);
          }
      }
    else if (part != 1) // row-per-wave cross-check kernel: not split, runs after the halo
      mi::launch_spmv(c->dim, spmv_params(c, x, y, dotv, partials, done), c->grid_spmv, c->stream, c->spmv_variant,
                      c->maxrow);
  }

  // y = K x on every slab of the team with the ghost planes of x exchanged on the way: the halo travels (RCCL: on
  // the team's communication stream) while the interior rows are computed; the boundary rows follow it.
  int team_spmv(Team &T, const std::function<mi_ctx *(mi_ctx *)> &ctx_of, const std::function<double *(mi_ctx *)> &x_of,
                const std::function<double *(mi_ctx *)> &y_of, const SpmvFusion *fusion, bool smoother,
                const ChebFusion *cheb, bool ghosts_current)
  {
    auto launch = [&](int part) {
      for (size_t k = 0; k < T.members.size(); ++k)
        {
          mi_ctx *m = T.members[k];
          enqueue_spmv(ctx_of(m), x_of(m), y_of(m), fusion ? fusion[k].dotv : nullptr,
                       fusion ? fusion[k].partials : nullptr, fusion ? fusion[k].done : nullptr, part, smoother,
                       cheb ? &cheb[k] : nullptr);
        }
    };
    if (T.size == 1 || (ghosts_current && T.halo_skip)) // (all rows in one launch where the kernel splits them)
      {
        launch(0);
        return MI_OK;
      }
    int rc = team_halo_begin(T, x_of, ctx_of);
    if (rc)
      return rc;
    launch(1);
    if ((rc = team_halo_end(T)))
      return rc;
    launch(2);
    return MI_OK;
  }

  // ---- team collectives (no-ops for a single slab) -------------------------------------------------------
  // sum over all slabs of d_sc[off .. off+cnt) in place
  int team_allreduce(Team &T, int off, int cnt)
  {
    ++T.n_scalar_allreduce;
    if (T.size == 1)
      return MI_OK;
    mi_ctx *c0 = T.members[0];
    if (T.nccl)
      NCCLCHK(c0, ncclAllReduce(c0->d_sc + off, c0->d_sc + off, size_t(cnt), ncclDouble, ncclSum, TNCCL(T), T.stream));
    else
      mi::launch_team_sum(T.d_sc_ptrs, int(T.members.size()), off, cnt, T.stream);
    return MI_OK;
  }

  // ghost planes of a local vector from the neighbouring slabs, split-phase: _begin starts the exchange of the
  // planes as the team's stream has them at this point, _end makes the stream wait for the ghost values.
  // The transfer (RCCL send/recv; device copies between emulated slabs) runs on the team's communication stream, so
  // work enqueued between _begin and _end that does not touch the ghost planes overlaps with it.
  int team_halo_begin(Team &T, const std::function<double *(mi_ctx *)> &vec,
                      const std::function<mi_ctx *(mi_ctx *)> &ctx_of)
  {
    ++T.n_halo;
    if (T.size == 1)
      return MI_OK;
    const int D = T.dim;
    // the slab geometry is that of ctx_of(member): the member itself or its distributed multigrid level
    auto slab_of = [&](mi_ctx *m) -> const mi::SlabPartition & { return (ctx_of ? ctx_of(m) : m)->slab; };
    mi_ctx           *c  = T.members[0];
    const bool        ov = T.overlap && T.comm_stream;
    const hipStream_t cs = ov ? T.comm_stream : T.stream;
    if (ov)
      {
        HIPCHK(c, hipEventRecord(T.ev_ready, T.stream));
        HIPCHK(c, hipStreamWaitEvent(T.comm_stream, T.ev_ready, 0));
      }
    if (T.nccl)
      {
        const mi::SlabPartition &s = slab_of(c);
        double                  *v = vec(c);
        NCCLCHK(c, ncclGroupStart());
        if (s.up_send_n)
          {
            NCCLCHK(c, ncclSend(v + s.up_send * D, size_t(s.up_send_n) * D, ncclDouble, s.rank + 1, TNCCL(T), cs));
            NCCLCHK(c, ncclRecv(v + s.up_recv * D, size_t(s.up_recv_n) * D, ncclDouble, s.rank + 1, TNCCL(T), cs));
          }
        if (s.down_send_n)
          {
            NCCLCHK(c, ncclSend(v + s.down_send * D, size_t(s.down_send_n) * D, ncclDouble, s.rank - 1, TNCCL(T), cs));
            NCCLCHK(c, ncclRecv(v + s.down_recv * D, size_t(s.down_recv_n) * D, ncclDouble, s.rank - 1, TNCCL(T), cs));
          }
        NCCLCHK(c, ncclGroupEnd());
      }
    else
      for (size_t r = 0; r + 1 < T.members.size(); ++r)
        {
          mi_ctx *a = T.members[r], *b = T.members[r + 1];
          const mi::SlabPartition &sa = slab_of(a), &sb = slab_of(b);
          HIPCHK(a, hipMemcpyAsync(vec(b) + sb.down_recv * D, vec(a) + sa.up_send * D,
                                   size_t(sa.up_send_n) * D * sizeof(double), hipMemcpyDeviceToDevice, cs));
          HIPCHK(a, hipMemcpyAsync(vec(a) + sa.up_recv * D, vec(b) + sb.down_send * D,
                                   size_t(sa.up_recv_n) * D * sizeof(double), hipMemcpyDeviceToDevice, cs));
        }
    if (ov)
      HIPCHK(c, hipEventRecord(T.ev_halo, cs));
    return MI_OK;
  }

  int team_halo_end(Team &T)
  {
    if (T.size > 1 && T.overlap && T.comm_stream)
      HIPCHK(T.members[0], hipStreamWaitEvent(T.stream, T.ev_halo, 0));
    return MI_OK;
  }

  int team_halo(Team &T, const std::function<double *(mi_ctx *)> &vec, const std::function<mi_ctx *(mi_ctx *)> &ctx_of)
  {
    const int rc = team_halo_begin(T, vec, ctx_of);
    return rc ? rc : team_halo_end(T);
  }

  // Ghost planes -> owners, summed: the adjoint of the halo exchange.  A slab's top ghost planes (its up_recv range) are
  // added to the planes the upper neighbour sends down in a halo exchange (its down_send range), its bottom ghost plane
  // (down_recv) to the lower neighbour's up_send plane.  Used where slabs hold PARTIAL results on ghost planes: the
  // restriction to / the state of a distributed multigrid level whose cuts do not coincide with the finer level's
  // (mi_mg.cpp).  The owner adds what comes from below first, then what comes from above: a fixed order.
  int team_halo_accumulate(Team &T, const std::function<double *(mi_ctx *)> &vec,
                           const std::function<mi_ctx *(mi_ctx *)> &ctx_of)
  {
    ++T.n_halo;
    if (T.size == 1)
      return MI_OK;
    const int D       = T.dim;
    auto      slab_of = [&](mi_ctx *m) -> const mi::SlabPartition & { return (ctx_of ? ctx_of(m) : m)->slab; };
    mi_ctx   *c       = T.members[0];
    if (T.nccl)
      {
        const mi::SlabPartition &s = slab_of(c);
        double                  *v = vec(c);
        // what arrives: from below the lower neighbour's top ghost planes (as many as my down_send range), from above the
        // upper neighbour's bottom ghost plane (as many as my up_send range)
        const size_t n_lo = size_t(s.down_send_n) * D, n_hi = size_t(s.up_send_n) * D;
        if (T.acc_cap < n_lo + n_hi)
          {
            if (T.d_acc)
              hipFree(T.d_acc);
            T.d_acc = nullptr;
            HIPCHK(c, hipMalloc((void **)&T.d_acc, (n_lo + n_hi) * sizeof(double)));
            T.acc_cap = n_lo + n_hi;
          }
        NCCLCHK(c, ncclGroupStart());
        if (s.up_send_n)
          {
            NCCLCHK(c, ncclSend(v + s.up_recv * D, size_t(s.up_recv_n) * D, ncclDouble, s.rank + 1, TNCCL(T), T.stream));
            NCCLCHK(c, ncclRecv(T.d_acc + n_lo, n_hi, ncclDouble, s.rank + 1, TNCCL(T), T.stream));
          }
        if (s.down_send_n)
          {
            NCCLCHK(c, ncclSend(v + s.down_recv * D, size_t(s.down_recv_n) * D, ncclDouble, s.rank - 1, TNCCL(T), T.stream));
            NCCLCHK(c, ncclRecv(T.d_acc, n_lo, ncclDouble, s.rank - 1, TNCCL(T), T.stream));
          }
        NCCLCHK(c, ncclGroupEnd());
        if (n_lo)
          mi::launch_vec_add(v + s.down_send * D, T.d_acc, int64_t(n_lo), T.stream);
        if (n_hi)
          mi::launch_vec_add(v + s.up_send * D, T.d_acc + n_lo, int64_t(n_hi), T.stream);
        return MI_OK;
      }
    // emulated slabs: for every slab first the contribution from below, then the one from above (the order of the RCCL branch)
    for (size_t r = 0; r < T.members.size(); ++r)
      {
        mi_ctx                  *m = T.members[r];
        const mi::SlabPartition &s = slab_of(m);
        if (r > 0)
          {
            mi_ctx                  *b  = T.members[r - 1];
            const mi::SlabPartition &sb = slab_of(b);
            mi::launch_vec_add(vec(m) + s.down_send * D, vec(b) + sb.up_recv * D, int64_t(sb.up_recv_n) * D, T.stream);
          }
        if (r + 1 < T.members.size())
          {
            mi_ctx                  *a  = T.members[r + 1];
            const mi::SlabPartition &sa = slab_of(a);
            mi::launch_vec_add(vec(m) + s.up_send * D, vec(a) + sa.down_recv * D, int64_t(sa.down_recv_n) * D, T.stream);
          }
      }
    return MI_OK;
  }

  // sum over all slabs of a replicated vector (every slab holds all n entries): coarse multigrid residuals
  int team_allreduce_vectors(Team &T, const std::function<double *(mi_ctx *)> &vec, size_t n)
  {
    ++T.n_vector_allreduce;
    if (T.size == 1)
      return MI_OK;
    if (T.nccl)
      {
        mi_ctx *c = T.members[0];
        NCCLCHK(c, ncclAllReduce(vec(c), vec(c), n, ncclDouble, ncclSum, TNCCL(T), T.stream));
        return MI_OK;
      }
    mi_ctx *c0 = T.members[0];
    for (size_t r = 1; r < T.members.size(); ++r)
      mi::launch_vec_add(vec(c0), vec(T.members[r]), int64_t(n), T.stream);
    for (size_t r = 1; r < T.members.size(); ++r)
      HIPCHK(c0, hipMemcpyAsync(vec(T.members[r]), vec(c0), n * sizeof(double), hipMemcpyDeviceToDevice, T.stream));
    return MI_OK;
  }

  // sum over all ranks of a device buffer that every rank holds in full (global vector / interface scratch)
  int team_allreduce_buffer(Team &T, double *buf, size_t n)
  {
    if (T.nccl)
      NCCLCHK(T.members[0], ncclAllReduce(buf, buf, n, ncclDouble, ncclSum, TNCCL(T), T.stream));
    return MI_OK; // emulated / single: all slabs wrote into the same buffer already
  }

  int ensure_gbuf(Team &T)
  {
    if (!T.d_gbuf)
      HIPCHK(T.members[0], hipMalloc((void **)&T.d_gbuf, size_t(T.n_global) * sizeof(double)));
    return MI_OK;
  }

  // the enqueue part of assemble_system for one slab (no host synchronisation).
  // residual_only: system_rhs alone (same numbers as the full pass); the tangent, its diagonal and the SpMV-side
  // copy keep the state of the last full assembly
  int enqueue_assembly(mi_ctx *c, bool residual_only)
  {
    // tangent_matrix = 0 (:1054) is implied: the first cell that touches a block stores instead of adding
    // (system_rhs = 0, :1055 -- except where residual_gather writes every entry of it: the one-launch point pass)
    if (!(c->mf_fine && c->mf_point_slots && c->d_mf_yc))
      HIPCHK(c, hipMemsetAsync(c->vec(MI_V_SYSTEM_RHS), 0, size_t(c->n) * sizeof(double), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_sc + SC_INVERTED, 0, sizeof(double), c->stream));
    mi::AsmParams p  = asm_params(c);
    p.residual_only  = residual_only ? 1 : 0;
    // matrix-free fine level: the tangent pass IS the residual kernel, handed the record pointers (it then writes the
    // state its residual was formed at); the diagonal blocks follow from the records below
    const bool mf_tangent = c->mf_fine && !residual_only;
    if (c->mf_fine)
      {
        p.residual_only = 1;
        p.variant       = 0;
        p.ke            = nullptr;
      }
    if (residual_only) // the convergence check's pass leaves the records of the last tangent alone
      p.qrec = nullptr, p.qrec32 = nullptr;
    mi_ctx       *c0 = c->team->members[0];
    const int     t0 = tic(c0, residual_only ? MI_T_ASSEMBLE_RESIDUAL : MI_T_ASSEMBLE_CELLS);
    // matrix-free fine level: the point pass over ALL cells in one launch (residual into the cells' slots, summed per node in
    // processing order: the product's slots and its order), then the Neumann faces colour by colour on the summed vector
    const bool one_launch = c->mf_fine && c->mf_point_slots && c->d_mf_yc;
    const bool faces_one_launch = c->face_slots && c->n_fn > 0;
    if (one_launch)
      {
        p.lat        = c->lat; // node ids by arithmetic, as mf_spmv
        p.cell_begin = 0;
        p.cell_count = int32_t(c->mesh.ncells);
        p.res_slots  = c->d_mf_yc; // (the product's slot array: no product is in flight during an assembly)
        p.slot_dst   = c->d_mf_dst;
        mi::launch_point_pass_slots(p, c->stream);
        mi::launch_residual_gather(c->d_mf_yc, c->d_mf_slot_base, c->d_mf_src, c->d_cmask, c->vec(MI_V_SYSTEM_RHS),
                                   int64_t(c->mesh.nnodes) * 3, c->stream);
      }
    for (int col = 0; col < c->mesh.ncolours; ++col)
      {
        p.cell_begin = c->mesh.colour_begin[col];
        p.cell_count = int32_t(c->mesh.colour_begin[col + 1] - c->mesh.colour_begin[col]);
        if (!one_launch && mi::launch_assemble_cells(c->dim, c->degree, p, c->stream))
          return fail(c, MI_EINVAL, "no assembly kernel for dim=%d degree=%d", c->dim, c->degree);
        const int fb = int(c->mesh.iface_colour_begin[col]);
        const int fc = int(c->mesh.iface_colour_begin[col + 1]) - fb;
        if (!faces_one_launch && mi::launch_neumann_faces(c->dim, c->degree, p, c->d_faces, fb, fc, c->stream))
          return fail(c, MI_EINVAL, "no face kernel for dim=%d degree=%d", c->dim, c->degree);
      }
    if (faces_one_launch) // the Neumann term of ALL interface faces, then added to the summed vector node by node in entry order
      {
        p.face_slots = c->d_face_slots;
        if (mi::launch_neumann_faces(c->dim, c->degree, p, c->d_faces, 0, int(c->mesh.iface_faces.size()), c->stream))
          return fail(c, MI_EINVAL, "no face kernel for dim=%d degree=%d", c->dim, c->degree);
        mi::launch_neumann_gather(c->dim, c->d_face_slots, c->d_fn_ids, c->d_fn_start, c->d_fn_src, c->d_cmask,
                                  c->vec(MI_V_SYSTEM_RHS), c->n_fn, c->stream);
      }
    toc(c0, t0);
    if (residual_only)
      {
        HIPCHK(c, hipGetLastError());
        return MI_OK;
      }
    c->qrec27_valid = false;
    if (c->smoother_points == 3 && c->d_qrec && c->d_qrec27) // the smoother's own records: F, J^(-2/3), 1/J at the 27 points
      {
        mi::MfParams f{};
        f.conn = c->d_conn, f.cverts = c->d_cverts, f.cellbox = c->d_cellbox, f.tab27 = c->d_tab27, f.lat = c->lat;
        mi::launch_mf_records27(f, c->vec(MI_V_TOTAL_DISPLACEMENT), c->vec(MI_V_SOLUTION_DELTA), c->d_qrec27, int32_t(c->mesh.ncells),
                                c->stream);
        c->qrec27_valid = true;
      }
    // which kernel ran: assemble_q2sf (sum factorised; it alone writes the fp32 records) or the node-pair form
    const bool q2sf = c->dim == 3 && c->degree == 2 && (p.variant == 0 || (p.variant >= 3 && p.variant <= 8));
    c->ke_valid     = (c->d_ke && q2sf && !c->mf_fine) || c->d_qrec;
    c->qrec32_valid = p.qrec32 != nullptr && q2sf;
    if (mf_tangent && c->mf_diag_lag && c->mf_diag_fresh) // the step's first tangent formed the blocks: kept ("mf_diag_lag")
      {
        HIPCHK(c, hipGetLastError());
        c->mg_stale = true;
        return MI_OK;
      }
    if (mf_tangent)
      {
        c->mf_diag_fresh = true;
        mi::MfParams f{};
        f.qrec = c->d_qrec, f.tab1d = c->d_tab, f.cverts = c->d_cverts, f.cellbox = c->d_cellbox, f.dst = c->d_mf_dst;
        f.mu = c->mat.mu, f.kappa = c->kappa, f.mass = c->alpha[1] * c->mat.rho;
        const int td = tic(c0, MI_T_ASSEMBLE_DIAG);
        mi::launch_mf_diag(f, c->d_diag_slots, int32_t(c->mesh.ncells), c->stream);
        mi::launch_mf_diag_gather(c->d_diag_slots, c->d_mf_slot_base, c->d_mf_src, c->d_cmask, c->d_diagpos_mf, c->d_diag_blk, c->work(W_DINV),
                                  c->d_dinv_blk, c->d_dinv_sym6, c->mesh.nnodes, c->stream);
        toc(c0, td);
        HIPCHK(c, hipGetLastError());
        c->mg_stale = true;
        return MI_OK;
      }
    mi::launch_extract_dinv(c->dim, c->d_vals, c->d_diagpos, c->work(W_DINV), c->mesh.nnodes, c->stream);
    if (c->want_dinv_blk) // block-Jacobi diagonal for the multigrid smoother
      {
        if (!c->d_dinv_blk)
          HIPCHK(c, hipMalloc((void **)&c->d_dinv_blk, size_t(c->mesh.nnodes) * c->dim * c->dim * sizeof(double)));
        if (!c->d_dinv_sym6 && c->dim == 3)
          HIPCHK(c, hipMalloc((void **)&c->d_dinv_sym6, size_t(c->mesh.nnodes) * 6 * sizeof(double)));
        mi::launch_extract_dinv_blk(c->dim, c->d_vals, c->d_diagpos, c->d_dinv_blk, c->d_dinv_sym6, c->mesh.nnodes, c->stream);
      }
    c->vals32_stale = true; // the opt-in fp32 copy is refreshed by the first product that needs it (enqueue_spmv)
    HIPCHK(c, hipGetLastError());
    c->mg_stale = true; // the coarse operators belong to an older state
    return MI_OK;
  }

  // slots of the matrix-free kernels' results (see MfParams::dst / slot_base / slot_src); rebuilt when the layout key changes
  int build_slot_tables(mi_ctx *c)
  {
    for (int32_t **p : {&c->d_mf_dst, &c->d_mf_slot_base, &c->d_mf_src})
      if (*p)
        {
          hipFree(*p);
          *p = nullptr;
        }

    // slots for the single-launch product: rank of every (cell, local node) among the cells of the node, in
    // processing order (colour-sorted cell order: the order in which the colour-by-colour update adds them)
    const int64_t        nc = c->mesh.ncells, nn = c->mesh.nnodes;
    std::vector<int32_t> base(size_t(nn) + 1, 0), dst(size_t(nc) * 27), fill(size_t(nn), 0);
    for (int64_t e = 0; e < nc; ++e)
      for (int a = 0; a < 27; ++a)
        ++base[size_t(c->mesh.conn[size_t(e) * 27 + a]) + 1];
    for (int64_t n = 0; n < nn; ++n)
      base[size_t(n) + 1] += base[size_t(n)];
    for (int64_t e = 0; e < nc; ++e)
      for (int a = 0; a < 27; ++a)
        {
          const int32_t n = c->mesh.conn[size_t(e) * 27 + a];
          dst[size_t(e) * 27 + a] = base[size_t(n)] + fill[size_t(n)]++;
        }
    // cell-major layout ("mf_slots_cell_major" 1): a cell stores its 81 results as one contiguous run, the gathers read a
    // node's contributions through slot_src (their positions, in processing order)
    // line-major layout ("mf_slots_cell_major" 2, lattice meshes): within an x-row of a colour's cells (mx cells, consecutive
    // positions) the slots are ordered [(k,j)][cell][i] instead of [cell][(k,j)][i] -- the contributions to ONE line of nodes
    // from one row of cells are one contiguous run, in the order of the nodes (the last node of a cell next to the first of
    // its neighbour): a line of nodes gathers from 2 x (1 | 2 | 4) such runs, every fetched line of memory is used by that
    // line of nodes alone (cell-major: a 128-byte line holds pieces of two node lines, fetched through two XCDs' L2);
    // the product stores 72-byte pieces, the pieces of consecutive cells next to each other
    std::vector<int32_t> src;
    int layout = c->slots_cell_major >= 0 ? c->slots_cell_major : (c->smoother_points == 3 ? 1 : 0);
    if (layout == 2 && !(c->lat.ncol > 0 && !c->lat_rows_host.empty()))
      layout = 1;
    c->slots_layout = layout;
    if (layout)
      {
        src.resize(dst.size());
        std::vector<int32_t> place(dst.size()); // slot of (cell, a)
        for (size_t k = 0; k < dst.size(); ++k)
          place[k] = int32_t(k);
        if (layout == 2)
          for (int col = 0; col < c->lat.ncol; ++col)
            {
              const mi::CellLatticeRow &R  = c->lat_rows_host[size_t(col)];
              const int64_t             b0 = c->lat.begin[col], b1 = c->lat.begin[col + 1];
              for (int64_t e = b0; e < b1; ++e)
                {
                  const int64_t r = e - b0, row = r / R.mx, rx = r - row * R.mx;
                  for (int a = 0; a < 27; ++a)
                    place[size_t(e) * 27 + a] = int32_t((b0 + row * R.mx) * 27 + int64_t(a / 3) * 3 * R.mx + rx * 3 + a % 3);
                }
            }
        for (size_t k = 0; k < dst.size(); ++k)
          src[size_t(dst[k])] = place[k];
        dst = place;
      }
    const bool cell_major = layout != 0;
    int rc = upload(c, &c->d_mf_dst, dst);
    if (rc == MI_OK)
      rc = upload(c, &c->d_mf_slot_base, base);
    if (rc == MI_OK && cell_major)
      rc = upload(c, &c->d_mf_src, src);
    if (rc)
      return rc;
    if (!c->d_mf_yc)
      HIPCHK(c, hipMalloc((void **)&c->d_mf_yc, size_t(nc) * 27 * 3 * sizeof(double)));
    
    return MI_OK;
  }

  // the smoother's own records and tables ("smoother_quadrature" 3), beside the assembly's
  int alloc_records27(mi_ctx *c)
  {
    if (c->d_qrec27 || c->smoother_points != 3)
      return MI_OK;
    mi::Tables1D t3;
    t3.build(2, 3);
    const int rc = upload(c, &c->d_tab27, t3.packed());
    if (rc)
      return rc;
    HIPCHK(c, hipMalloc((void **)&c->d_qrec27, size_t(c->mesh.ncells) * mi::MF_NREC * 27 * sizeof(double)));
    return MI_OK;
  }

  // quadrature-point records for the matrix-free product (3D Q2) + the geometry class of the local cells
  int alloc_point_records(mi_ctx *c)
  {
    if (c->d_qrec)
      return alloc_records27(c);
    if (int r27 = alloc_records27(c))
      return r27;
    HIPCHK(c, hipMalloc((void **)&c->d_qrec, size_t(c->mesh.ncells) * mi::MF_NREC * 64 * sizeof(double)));
    bool box = c->dim == 3;
    for (int64_t e = 0; e < c->mesh.ncells && box; ++e)
      {
        const double *cv = &c->mesh.cverts[size_t(e) * 24];
        for (int v = 0; v < 8 && box; ++v)
          for (int d = 0; d < 3; ++d)
            box = box && cv[v * 3 + d] == cv[(((v >> d) & 1) ? 7 : 0) * 3 + d];
      }
    if (int rs = build_slot_tables(c))
      return rs;
    if (box) // 1/h and the volume per cell, so that the product needs no division for its geometry
      {
        std::vector<double> cb(size_t(c->mesh.ncells) * 4);
        for (int64_t e = 0; e < c->mesh.ncells; ++e)
          {
            const double *cv = &c->mesh.cverts[size_t(e) * 24];
            const double  hx = cv[3] - cv[0], hy = cv[7] - cv[1], hz = cv[14] - cv[2]; // vertex v: bit d = upper end along d
            cb[size_t(e) * 4 + 0] = 1.0 / hx;
            cb[size_t(e) * 4 + 1] = 1.0 / hy;
            cb[size_t(e) * 4 + 2] = 1.0 / hz;
            cb[size_t(e) * 4 + 3] = hx * hy * hz;
          }
        return upload(c, &c->d_cellbox, cb);
      }
    return MI_OK;
  }

  // tuning "fine_level": 1 = the fine level matrix-free end to end (see mi_ctx::mf_fine), 0 = assembled (default).  Takes
  // effect with the next tangent assembly; the assembled tangent's 8 bytes per non-zero are released / allocated again.
  int set_fine_level(mi_ctx *c, int on)
  {
    if ((on != 0) == (c->mf_fine != 0))
      return MI_OK;
    if (on)
      {
        if (c->dim != 3 || c->degree != 2)
          return fail(c, MI_EINVAL, "the matrix-free fine level exists for 3D Q2 meshes only");
        if (c->precond_storage != 64 || c->solver_direct)
          return fail(c, MI_EINVAL, "the matrix-free fine level excludes \"precond_storage\" 32 and \"solver_type\" 1");
        int rc = alloc_point_records(c);
        if (rc == MI_OK && !c->d_node_first)
          rc = upload(c, &c->d_node_first, c->mesh.node_first);
        if (rc)
          return rc;
        const size_t nn = size_t(c->mesh.nnodes);
        if (!c->d_diagpos_mf)
          {
            std::vector<int32_t> dp(nn);
            for (size_t n = 0; n < nn; ++n)
              dp[n] = c->mesh.diagpos[n] >= 0 ? int32_t(n) : -1;
            if ((rc = upload(c, &c->d_diagpos_mf, dp)))
              return rc;
            HIPCHK(c, hipMalloc((void **)&c->d_diag_blk, nn * 9 * sizeof(double)));
            HIPCHK(c, hipMalloc((void **)&c->d_diag_slots, size_t(c->mesh.ncells) * 27 * 6 * sizeof(double)));
          }
        if (!c->d_dinv_blk)
          HIPCHK(c, hipMalloc((void **)&c->d_dinv_blk, nn * 9 * sizeof(double)));
        if (!c->d_dinv_sym6)
          HIPCHK(c, hipMalloc((void **)&c->d_dinv_sym6, nn * 6 * sizeof(double)));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(c->d_vals));
        c->d_vals   = nullptr;
        c->ebe      = 2;
        c->mf_slots = 1;
      }
    else
      {
        HIPCHK(c, hipMalloc((void **)&c->d_vals, c->vals_doubles * sizeof(double)));
        HIPCHK(c, hipMemsetAsync(c->d_vals, 0, c->vals_doubles * sizeof(double), c->stream));
      }
    c->mf_fine      = on ? 1 : 0;
    c->mf_diag_fresh = false;
    c->ke_valid     = false;
    c->qrec32_valid = false;
    c->vals32_stale = true;
    c->mg_stale = c->mg_force = true;
    return MI_OK;
  }

  // the tangent's products on a matrix-free fine level need the point records of an assembly
  int check_mf_tangent(mi_ctx *c)
  {
    for (mi_ctx *m : c->team->members)
      if (m->mf_fine && !m->active_sell_vals && element_form(m) != 2)
        return fail(c, MI_EINVAL, "matrix-free fine level: no tangent has been assembled yet (mi_assemble first)");
    return MI_OK;
  }

  // storage for the unassembled element tangents, where the smoother can use them (filled by the next full assembly)
  int ensure_element_tangents(mi_ctx *c)
  {
    const bool want = c->ebe && c->dim == 3 && c->degree == 2 && c->precond == 1 && c->mg &&
                      c->mesh.nnodes > 100000; // below that the smoother runs fused on the assembled matrix
    if (want && (c->ebe == 2 ? !c->d_qrec : !c->d_ke))
      {
        if (c->ebe == 2)
          {
            const int rq = alloc_point_records(c);
            if (rq)
              return rq;
          }
        else
          HIPCHK(c, hipMalloc((void **)&c->d_ke, size_t(c->mesh.ncells) * 9 * mi::EBE_NBLK * sizeof(double)));
        const int rc = c->d_node_first ? MI_OK : upload(c, &c->d_node_first, c->mesh.node_first);
        if (rc)
          return rc;
        c->ke_valid = false;
        c->mg_stale = c->mg_force = true;
      }
    return MI_OK;
  }

  // l2 norm over the unconstrained owned dofs of vector `which`, summed over the team (:549-576)
  // flag (optional): the scalar in the slot BEFORE `slot` travels with the norm (same all-reduce, same copy): the
  // inverted-element flag of the assembly sits in front of the residual norm
  int team_masked_norm(Team &T, int which, int slot, double *out, double *flag = nullptr)
  {
    mi_ctx   *c0 = T.members[0];
    const int ex = flag ? 1 : 0;
    for (mi_ctx *m : T.members)
      mi::launch_masked_norm(m->dim, m->vec(which) + m->own0, m->d_cmask + m->slab.own_begin, m->own_n, m->part(3),
                             m->grid_vec, m->d_sc + slot, m->stream);
    int rc = team_allreduce(T, slot - ex, 1 + ex);
    if (rc)
      return rc;
    HIPCHK(c0, hipGetLastError());
    HIPCHK(c0, hipMemcpyAsync(c0->h_pinned, c0->d_sc + slot - ex, (1 + ex) * sizeof(double), hipMemcpyDeviceToHost, c0->stream));
    if ((rc = sync(c0)))
      return rc;
    if (flag)
      *flag = c0->h_pinned[0];
    *out = std::sqrt(c0->h_pinned[ex]);
    return MI_OK;
  }

  // Jacobi-PCG (deal.II SolverCG semantics: start from x, stop when ||r||_2 <= tolerance) on the active matrix of
  // every slab of the team; followed by constraints.distribute (x[constrained] = 0)
  int cg_run(mi_ctx *c, int x_id, int b_id, double tol, int64_t max_it, int *its, double *res, bool x_is_zero, bool scale_start,
             int expected_its)
  {
    Team      &T    = *c->team;
    mi_ctx    *c0   = T.members[0];
    const bool dist = T.size > 1;
    struct CountCg // the scalar all-reduces made between here and any return belong to the solve
    {
      Team   &T;
      int64_t at;
      ~CountCg() { T.n_scalar_allreduce_cg += T.n_scalar_allreduce - at; }
    } count_cg{T, T.n_scalar_allreduce};
    if (int e = check_mf_tangent(c0))
      return e;
    for (mi_ctx *m : T.members) // outside the timed SpMV launches
      refresh_vals32(m);
    const int  tt   = tic(c0, MI_T_CG_TOTAL);
    std::vector<mi::CgParams> cgs;
    for (mi_ctx *m : T.members)
      {
        mi::CgParams cg{};
        cg.x        = m->vec(x_id) + m->own0;
        cg.r        = m->work(W_R) + m->own0;
        cg.p        = m->work(W_P) + m->own0;
        cg.q        = m->work(W_Q) + m->own0;
        cg.dinv     = (m->active_dinv ? m->active_dinv : m->work(W_DINV)) + m->own0;
        cg.part_rr  = m->part(0);
        cg.part_rz  = m->part(1);
        cg.part_pq  = m->part(2);
        cg.sc       = m->d_sc;
        cg.flags    = m->d_flags;
        cg.n        = m->own_n;
        cg.npart    = m->grid_vec;
        {
          // partials of p.q: one per workgroup of the kernel that forms them -- the fused product's, the matrix-free gather's
          // (its own, wider grid), or the separate reduction's
          const bool elem = (m->cg_operator == 1 || m->mf_fine) && element_form(m) && !m->active_sell_vals;
          const bool one  = elem && element_form(m) == 2 && m->mf_slots && m->d_mf_yc;
          cg.npart_pq     = !m->cg_fused_dot ? m->grid_vec : one ? m->grid_gdot : elem ? m->grid_vec : m->grid_spmv;
        }
        cg.totals   = dist ? m->d_sc + SC_TOT : nullptr;
        cgs.push_back(cg);
      }
    const size_t R = T.members.size();
    std::vector<SpmvFusion> fusion; // q = K p with the partials of p.q
    for (size_t k = 0; k < R; ++k)
      fusion.push_back(SpmvFusion{T.members[k]->work(W_P), cgs[k].part_pq, cgs[k].flags});
    int          rc;
    bool         use_mg = true;
    for (mi_ctx *m : T.members)
      use_mg = use_mg && mg_active(m);
    if (use_mg)
      {
        if (c0->mg_stale && (c0->mg_force || !c0->mg_lag))
          {
            if ((rc = mg_update(T)))
              return rc;
            for (mi_ctx *m : T.members)
              {
                m->mg_force                = false;
                m->mg_steps_since_refresh = 0;
                m->mg_its_ref             = 0;
              }
            ++c0->n_mg_refresh;
          }
        for (size_t k = 0; k < R; ++k)
          cgs[k].z = T.members[k]->work(W_Z) + T.members[k]->own0;
      }
    int32_t *h_flags = reinterpret_cast<int32_t *>(c0->h_pinned + 8);
    // small problems on one slab: the whole Jacobi-PCG in ONE launch (cg_small), one host synchronisation per solve
    // (pays while one CU can stream the matrix from the L2 faster than three launches take: ~130 GB/s vs ~18 us,
    // i.e. up to ~1 MB of matrix values; measured with tools/small_case_latency.py)
    if (!dist && !use_mg && c0->small_cg && max_it > 0 && (c0->spmv_variant == 3 || c0->active_sell_vals) &&
        !(c0->mf_fine && !c0->active_sell_vals) && // (the one-launch solver streams the assembled rows)
        c0->mesh.sell_nblk64 * 64 * int64_t(c0->dim * c0->dim) * 8 <= SMALL_CG_MAX_MATRIX_BYTES)
      {
        mi::launch_cg_small(c0->dim, sell_params(c0, nullptr, nullptr, nullptr, nullptr, nullptr), cgs[0],
                            c0->vec(b_id), tol, int(std::min<int64_t>(max_it, 2000000000)), c0->stream);
        mi::launch_zero_constrained(c0->dim, c0->vec(x_id), c0->d_cmask, c0->n, c0->stream); // :1208
        HIPCHK(c0, hipGetLastError());
        HIPCHK(c0, hipMemcpyAsync(h_flags, c0->d_flags, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c0->stream));
        HIPCHK(c0, hipMemcpyAsync(c0->h_pinned, c0->d_sc, 8 * sizeof(double), hipMemcpyDeviceToHost, c0->stream));
        toc(c0, tt);
        int e = sync(c0);
        if (e)
          return e;
        if (its)
          *its = h_flags[1];
        if (res)
          *res = c0->h_pinned[SC_RES];
        c0->cg_breakdown = h_flags[0] == 2;
        if (h_flags[0] == 2)
          return fail(c0, MI_ENOCONV_LIN, "CG broke down after %d iterations (non-finite residual or p.Ap <= 0; residual %.3e)",
                      int(h_flags[1]), c0->h_pinned[SC_RES]);
        if (!h_flags[0])
          return fail(c0, MI_ENOCONV_LIN, "CG did not reach tolerance %.3e within %lld iterations (residual %.3e)",
                      std::fabs(tol), (long long)max_it, c0->h_pinned[SC_RES]);
        return MI_OK;
      }
    // z = M^-1 r by the team-wide V-cycle, then the partials of r.z (and their team totals)
    // with_rr: the team total of ||r||^2 (partials of the last update) travels in the same all-reduce -- the iterations
    // whose V-cycle is enqueued before their convergence test is known (see `speculate` below); with_init: ... and |b|^2
    // of the initial residual (nobody needs either before the first V-cycle has run)
    auto precondition = [&](bool with_rr, bool with_init) -> int {
      int e = mg_apply(T);
      if (e)
        return e;
      for (size_t k = 0; k < R; ++k)
        {
          mi_ctx *m = T.members[k];
          mi::launch_dot_partials(cgs[k].r, cgs[k].z, m->own_n, cgs[k].part_rz, m->grid_vec, m->stream);
          if (dist)
            mi::launch_reduce_to_totals(cgs[k].part_rz, m->grid_vec, m->d_sc + SC_TOT + 1, with_rr ? cgs[k].part_rr : nullptr,
                                        m->grid_vec, m->d_sc + SC_TOT, with_init ? nullptr : cgs[k].flags, m->stream);
        }
      return with_init ? team_allreduce(T, SC_TOT, 4) : with_rr ? team_allreduce(T, SC_TOT, 2) : team_allreduce(T, SC_TOT + 1, 1);
    };
    const int64_t batch = use_mg ? 1 : CG_BATCH; // a V-cycle costs ~5 SpMVs: poll every iteration, never waste one
    auto         x_of = [x_id](mi_ctx *m) { return m->vec(x_id); };
    auto         p_of = [](mi_ctx *m) { return m->work(W_P); };

    // r0 = b - A x0, tolerance = rel_tol * ||b||  (:1171-1172)
    auto self = [](mi_ctx *m) { return m; };
    auto q_of = [](mi_ctx *m) { return m->work(W_Q); };
    if (x_is_zero) // the start vector is known to be zero: A x0 = 0 without a product
      {
        for (mi_ctx *m : T.members)
          HIPCHK(m, hipMemsetAsync(m->work(W_Q), 0, size_t(m->n) * sizeof(double), m->stream));
      }
    else
      {
        // A h for a predicted start vector h: matrix-free where the point records of the current tangent exist (0.37 instead
        // of 1.28 ms at 5 M dofs; the same operator to 5e-16, fp64 throughout) -- the products of the iteration itself,
        // q = K p, stay on the assembled matrix ("cg_r0_operator" 0: this one too)
        const bool mf_r0 = scale_start && c0->cg_r0_unassembled;
        for (mi_ctx *m : T.members)
          m->unassembled_now = mf_r0 && element_form(m) == 2;
        rc = team_spmv(T, self, x_of, q_of, nullptr); // not under MI_T_SPMV: that class is the fused q = K p only
        for (mi_ctx *m : T.members)
          m->unassembled_now = false;
        if (rc)
          return rc;
        if (scale_start)
          {
            // the start vector is a prediction h (the same solve of the previous time step): take alpha h with
            // alpha = h.b / h.Ah, the multiple with the smallest energy-norm error -- never worse than starting from
            // zero, whatever the load did since (A h is at hand: no further product)
            for (mi_ctx *m : T.members)
              {
                mi::launch_dot_partials(m->vec(x_id) + m->own0, m->vec(b_id) + m->own0, m->own_n, m->part(5), m->grid_vec, m->stream);
                mi::launch_dot_partials(m->vec(x_id) + m->own0, m->work(W_Q) + m->own0, m->own_n, m->part(6), m->grid_vec, m->stream);
                mi::launch_reduce_to_totals(m->part(5), m->grid_vec, m->d_sc + SC_START, m->part(6), m->grid_vec,
                                            m->d_sc + SC_START + 1, nullptr, m->stream);
              }
            if ((rc = team_allreduce(T, SC_START, 2)))
              return rc;
            for (mi_ctx *m : T.members)
              mi::launch_scale_start(m->vec(x_id), m->work(W_Q), m->d_sc + SC_START, m->n, m->stream);
          }
      }
    for (size_t k = 0; k < R; ++k)
      {
        mi_ctx *m = T.members[k];
        mi::launch_cg_init_residual(cgs[k], m->vec(b_id) + m->own0, m->part(4), m->grid_vec, m->stream);
        if (dist)
          {
            if (!use_mg) // (multigrid: ||r0||^2 joins r.z after the first V-cycle)
              mi::launch_reduce_to_totals(cgs[k].part_rr, m->grid_vec, m->d_sc + SC_TOT, cgs[k].part_rz, m->grid_vec,
                                          m->d_sc + SC_TOT + 1, nullptr, m->stream);
            mi::launch_reduce_to_totals(m->part(4), m->grid_vec, m->d_sc + SC_TOT + 3, nullptr, 0, nullptr, nullptr,
                                        m->stream);
          }
      }
    // Single-reduction form of the multigrid-PCG (Chronopoulos & Gear): the product is applied to
    // z = M^-1 r instead of p, A p follows by recurrence, and r.z, z.Az and ||r||^2 travel in ONE all-reduce per iteration
    // (the standard recurrence needs p.Ap between its two updates: two to three).  Default on teams of several slabs, where
    // an all-reduce is a latency the ranks wait out together; one GPU keeps the standard form ("cg_single_reduction" 0 / 1).
    const bool single = use_mg && (c0->cg_single_reduction == 1 || (c0->cg_single_reduction < 0 && dist));
    if (!use_mg && (rc = team_allreduce(T, SC_TOT, 4)))
      return rc;
    if (use_mg && !single && (rc = precondition(true, true)))
      return rc;
    if (single && dist && max_it <= 0)
      {
        // no iteration will bring |b|^2 and ||r0||^2: reduce them now, so that the final check below reports this solve's
        // residual against this solve's tolerance
        for (size_t k = 0; k < R; ++k)
          mi::launch_reduce_to_totals(cgs[k].part_rr, T.members[k]->grid_vec, T.members[k]->d_sc + SC_TOT, nullptr, 0, nullptr, nullptr,
                                      T.members[k]->stream);
        if ((rc = team_allreduce(T, SC_TOT, 4)))
          return rc;
      }
    if (!single || !dist || max_it <= 0) // (single-reduction form on a team: |b|^2 arrives with the first iteration's all-reduce)
      for (size_t k = 0; k < R; ++k)
        mi::launch_cg_set_tolerance(cgs[k], T.members[k]->part(4), tol, T.members[k]->stream);

    int64_t  it      = 0;
    bool     done    = false;
    auto     poll    = [&]() -> int {
      HIPCHK(c0, hipGetLastError());
      HIPCHK(c0, hipMemcpyAsync(h_flags, c0->d_flags, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c0->stream));
      HIPCHK(c0, hipMemcpyAsync(c0->h_pinned, c0->d_sc, 8 * sizeof(double), hipMemcpyDeviceToHost, c0->stream));
      HIPCHK(c0, hipStreamSynchronize(c0->stream));
      ++T.n_cg_sync;
      done = h_flags[0] != 0;
      return MI_OK;
    };
    // Multigrid-PCG: an iteration costs milliseconds, so the host learns the outcome of every convergence test before
    // it enqueues the next V-cycle (one synchronisation per iteration) -- except where the outcome is as good as known:
    // the same solve of the previous time step took expected_its iterations ("cg_warm_start" >= 2), so iterations
    // 1 .. expected_its - 2 are enqueued back to back, V-cycle included, and the test of iteration i is taken on the
    // device by the update of iteration i + 1 (as in the Jacobi batches).  Should the solve converge earlier after all,
    // the flag turns every later CG kernel into a no-op and the V-cycles in the queue are wasted work, nothing else:
    // iterates, iteration count and residual are those of the polled loop bit by bit.  Saves expected_its - 2 host
    // synchronisations and as many scalar all-reduces per solve (||r||^2 then travels with r.z).
    // (margin: how many of the expected iterations are left to polled ones.  One instead of two saves a poll and an
    // all-reduce per solve and wastes a V-cycle + product whenever a solve ends one iteration early -- measured on 8
    // emulated slabs: 0.2 halo exchanges per iteration more, i.e. a wasted V-cycle every third solve: two stays)
    const int64_t margin       = c0->cg_speculate_margin > 0 ? c0->cg_speculate_margin : 2;
    const int64_t speculate_to = (use_mg && c0->cg_speculate) ? std::min<int64_t>(max_it - 1, int64_t(expected_its) - margin) : 0;
    if (single)
      {
        auto z_of = [](mi_ctx *m) { return m->work(W_Z); };
        std::vector<SpmvFusion> zfusion; // w = K z with the partials of z.w
        for (size_t k = 0; k < R; ++k)
          {
            cgs[k].s = T.members[k]->work(W_S) + T.members[k]->own0;
            zfusion.push_back(SpmvFusion{T.members[k]->work(W_Z), cgs[k].part_pq, cgs[k].flags});
          }
        if (dist) // the flags of the previous solve must not switch this one's first product off
          for (mi_ctx *m : T.members)
            HIPCHK(m, hipMemsetAsync(m->d_flags, 0, 2 * sizeof(int32_t), m->stream));
        while (!done && it < max_it)
          {
            ++it;
            if ((rc = mg_apply(T))) // z = M^-1 r
              return rc;
            for (size_t k = 0; k < R; ++k)
              mi::launch_dot_partials(cgs[k].r, cgs[k].z, T.members[k]->own_n, cgs[k].part_rz, T.members[k]->grid_vec,
                                      T.members[k]->stream);
            int t = tic(c0, MI_T_SPMV);
            if ((rc = team_spmv(T, self, z_of, q_of, c0->cg_fused_dot ? zfusion.data() : nullptr))) // w = K z, z.w
              return rc;
            toc(c0, t);
            if (!c0->cg_fused_dot)
              for (size_t k = 0; k < R; ++k)
                mi::launch_dot_partials(cgs[k].z, cgs[k].q, T.members[k]->own_n, cgs[k].part_pq, T.members[k]->grid_vec,
                                        T.members[k]->stream);
            if (dist)
              {
                for (size_t k = 0; k < R; ++k)
                  {
                    mi_ctx *m = T.members[k];
                    mi::launch_reduce_to_totals(cgs[k].part_rr, m->grid_vec, m->d_sc + SC_TOT, cgs[k].part_rz, m->grid_vec,
                                                m->d_sc + SC_TOT + 1, it == 1 ? nullptr : cgs[k].flags, m->stream);
                    mi::launch_reduce_to_totals(cgs[k].part_pq, cgs[k].npart_pq, m->d_sc + SC_TOT + 2, nullptr, 0, nullptr,
                                                it == 1 ? nullptr : cgs[k].flags, m->stream);
                  }
                if ((rc = team_allreduce(T, SC_TOT, it == 1 ? 4 : 3))) // THE reduction of the iteration
                  return rc;
                if (it == 1)
                  for (size_t k = 0; k < R; ++k)
                    mi::launch_cg_set_tolerance(cgs[k], T.members[k]->part(4), tol, T.members[k]->stream);
              }
            t = tic(c0, MI_T_CG_VECTOR);
            for (size_t k = 0; k < R; ++k)
              mi::launch_cg_update_single(cgs[k], int(it), T.members[k]->grid_vec, T.members[k]->stream);
            toc(c0, t);
            if (it <= speculate_to) // no test, no poll: the next update takes the decision
              continue;
            if (dist)
              {
                for (size_t k = 0; k < R; ++k)
                  mi::launch_reduce_to_totals(cgs[k].part_rr, T.members[k]->grid_vec, T.members[k]->d_sc + SC_TOT, nullptr, 0,
                                              nullptr, cgs[k].flags, T.members[k]->stream);
                if ((rc = team_allreduce(T, SC_TOT, 1)))
                  return rc;
              }
            for (size_t k = 0; k < R; ++k)
              mi::launch_cg_final_check(cgs[k], int(it), T.members[k]->stream);
            if ((rc = poll()))
              return rc;
          }
      }
    else
    while (!done && it < max_it)
      {
        const int64_t stop = std::min<int64_t>(max_it, it + batch);
        for (; it < stop;)
          {
            ++it;
            int t = tic(c0, MI_T_CG_VECTOR);
            for (size_t k = 0; k < R; ++k)
              mi::launch_cg_update_p(cgs[k], int(it), T.members[k]->grid_vec, T.members[k]->stream);
            toc(c0, t);
            // the product the roofline figure is quoted on: when it is ONE launch of the production kernel its events
            // come from the dispatch itself (kernel start / end, as rocprofv3 reports them)
            const bool one_launch = !dist && c0->profiling && c0->spmv_variant == 3 && c0->sell_icol && c0->sell_unroll == 5 &&
                                    !c0->active_sell_vals && c0->mesh.sell_nslices_interior == c0->mesh.sell_nslices &&
                                    !((c0->cg_operator == 1 || c0->mf_fine) && element_form(c0));
            t = tic(c0, MI_T_SPMV, one_launch);
            if (one_launch && t >= 0)
              mi::set_next_sell_launch_events(c0->stamps[size_t(t)].a, c0->stamps[size_t(t)].b);
            if ((rc = team_spmv(T, self, p_of, q_of, c0->cg_fused_dot ? fusion.data() : nullptr)))
              return rc;
            if (!one_launch)
              toc(c0, t);
            if (!c0->cg_fused_dot) // A/B: p.q by a separate reduction over the owned dofs
              for (size_t k = 0; k < R; ++k)
                mi::launch_dot_partials(cgs[k].p, cgs[k].q, T.members[k]->own_n, cgs[k].part_pq, T.members[k]->grid_vec,
                                        T.members[k]->stream);
            if (dist)
              {
                for (size_t k = 0; k < R; ++k)
                  mi::launch_reduce_to_totals(cgs[k].part_pq, cgs[k].npart_pq, T.members[k]->d_sc + SC_TOT + 2,
                                              nullptr, 0, nullptr, cgs[k].flags, T.members[k]->stream);
                if ((rc = team_allreduce(T, SC_TOT + 2, 1)))
                  return rc;
              }
            t = tic(c0, MI_T_CG_VECTOR);
            for (size_t k = 0; k < R; ++k)
              mi::launch_cg_update_xr(cgs[k], int(it), T.members[k]->grid_vec, T.members[k]->stream);
            toc(c0, t);
            if (dist && it > speculate_to)
              {
                for (size_t k = 0; k < R; ++k)
                  mi::launch_reduce_to_totals(cgs[k].part_rr, T.members[k]->grid_vec, T.members[k]->d_sc + SC_TOT,
                                              use_mg ? nullptr : cgs[k].part_rz, T.members[k]->grid_vec,
                                              T.members[k]->d_sc + SC_TOT + 1, cgs[k].flags, T.members[k]->stream);
                if ((rc = team_allreduce(T, SC_TOT, use_mg ? 1 : 2)))
                  return rc;
              }
          }
        if (it <= speculate_to) // no test, no poll: the next update takes the decision
          {
            if ((rc = precondition(true, false)))
              return rc;
            continue;
          }
        for (size_t k = 0; k < R; ++k)
          mi::launch_cg_final_check(cgs[k], int(it), T.members[k]->stream);
        if ((rc = poll()))
          return rc;
        if (use_mg && !done && it < max_it && (rc = precondition(false, false)))
          return rc;
      }
    if (max_it <= 0)
      {
        for (size_t k = 0; k < R; ++k)
          mi::launch_cg_final_check(cgs[k], 0, T.members[k]->stream);
        if ((rc = poll()))
          return rc;
      }
    for (mi_ctx *m : T.members) // constraints.distribute (:1208)
      mi::launch_zero_constrained(m->dim, m->vec(x_id), m->d_cmask, m->n, m->stream);
    if ((rc = team_halo(T, x_of))) // ghost copies of the solution
      return rc;
    toc(c0, tt);
    HIPCHK(c0, hipGetLastError());
    if ((rc = sync(c0)))
      return rc;
    if (its)
      *its = h_flags[1];
    ++T.n_cg_solves;
    T.n_cg_its += h_flags[1];
    if (res)
      *res = c0->h_pinned[SC_RES];
    c0->cg_breakdown = h_flags[0] == 2;
    if (h_flags[0] == 2)
      return fail(c0, MI_ENOCONV_LIN, "CG broke down after %d iterations (non-finite residual or p.Ap <= 0; residual %.3e)",
                  int(h_flags[1]), c0->h_pinned[SC_RES]);
    if (!done)
      return fail(c0, MI_ENOCONV_LIN, "CG did not reach tolerance %.3e within %lld iterations (residual %.3e)",
                  std::fabs(tol), (long long)max_it, c0->h_pinned[SC_RES]);
    return MI_OK;
  }

  // fp32-rounded copy of the sliced-ELL values for the multigrid smoother (the CG's own product, residuals and all
  // arithmetic stay fp64); 64 drops it again
  int set_precond_storage(mi_ctx *c, int bits)
  {
    if (bits == 32 && !c->d_sell_vals32)
      {
        const size_t cnt = std::max<size_t>(1, size_t(c->mesh.sell_nblk64) * 64 * size_t(c->dim) * c->dim);
        HIPCHK(c, hipMalloc((void **)&c->d_sell_vals32, cnt * sizeof(float)));
        HIPCHK(c, hipMemsetAsync(c->d_sell_vals32, 0, cnt * sizeof(float), c->stream));
      }
    c->precond_storage = bits;
    c->vals32_stale    = true; // refreshed before the next product
    return mg_set_storage(c, bits);
  }

  void destroy_member(mi_ctx *c)
  {
    if (!c)
      return;
    linear_destroy(c);
    mg_destroy(c);
    for (auto &s : c->stamps)
      {
        hipEventDestroy(s.a);
        hipEventDestroy(s.b);
      }
    void *ptrs[] = {c->d_conn,      c->d_rowptr,    c->d_col,         c->d_diagpos,     c->d_iface_nodes, c->d_faces,
                    c->d_flags,     c->d_cverts,    c->d_tab,         c->d_vals,        c->d_vecs,        c->d_work,
                    c->d_saved,     c->d_part,      c->d_sc,          c->d_iface_buf,   c->d_off,         c->d_cmask,
                    c->d_sell_perm, c->d_sell_len,  c->d_sell_col,    c->d_sell_off,    c->d_rowinfo, c->d_rowwx, c->d_sell_wx, c->d_band, c->d_band_work, c->d_band_perm,
                    c->d_own_if_nodes, c->d_own_if_slots, c->d_sell_vals32, c->d_dinv_blk, c->d_dinv_sym6, c->d_sell_box, c->d_ke, c->d_node_first, c->d_qrec, c->d_qrec32, c->d_cellbox, c->d_mf_yc, c->d_mf_dst, c->d_mf_slot_base, c->d_mf_src, c->d_lat_rows,
                    c->d_qrec27, c->d_tab27, c->d_diag_blk, c->d_diag_slots, c->d_diagpos_mf, c->d_face_slots, c->d_fn_ids, c->d_fn_start, c->d_fn_src,
                    c->d_pred[0][0], c->d_pred[0][1], c->d_pred[1][0], c->d_pred[1][1], c->d_pred[2][0], c->d_pred[2][1], c->d_pred[3][0], c->d_pred[3][1],
                    c->d_pred_saved[0][0], c->d_pred_saved[0][1], c->d_pred_saved[1][0], c->d_pred_saved[1][1], c->d_pred_saved[2][0],
                    c->d_pred_saved[2][1], c->d_pred_saved[3][0], c->d_pred_saved[3][1]};
    for (void *p : ptrs)
      if (p)
        hipFree(p);
    if (c->h_pinned)
      hipHostFree(c->h_pinned);
    delete c;
  }

  // one slab context: local box mesh (own layers + ghost layer), device arrays, launch geometry
  int create_member(Team &T, const mi_mesh_desc *md, const mi_material_desc *mat, const mi_newmark_desc *nm, int rank,
                    mi_ctx **out)
  {
    mi_ctx *c = new mi_ctx;
    *out      = c;
    c->team   = &T;
    c->device = T.device;
    c->stream = T.stream;
    c->dim    = md->dim;
    c->degree = md->degree;
    c->mat    = *mat;
    c->nm     = *nm;
    try
      {
        c->slab = mi::make_slab_partition(md->dim, md->degree, md->reps, md->lo, md->hi, md->face_role, rank, T.size,
                                          T.cuts.empty() ? nullptr : T.cuts.data());
        const mi::SlabPartition &s = c->slab;
        const double *perturb = md->vertex_perturbation ? md->vertex_perturbation + s.vertex_offset * md->dim : nullptr;
        if (T.size == 1)
          c->mesh.build(md->dim, md->degree, md->reps, md->lo, md->hi, md->face_role, perturb, 0, 0, 0, -1,
                        T.amap.identity ? nullptr : T.amap.ext_axis); // replicated multigrid levels of a rotated lattice
        else
          c->mesh.build(md->dim, md->degree, s.local_reps, md->lo, md->hi, s.local_face_role, perturb, s.z0,
                        md->reps[md->dim - 1], s.own_begin, s.own_end, T.amap.identity ? nullptr : T.amap.ext_axis);
        c->tab.build(md->degree, md->degree + 2); // qf_cell(p+2), qf_face(p+2): nonlinear_elasticity.cc:74-75
      }
    catch (const std::exception &e)
      {
        return fail(c, MI_EINVAL, "%s", e.what());
      }
    const mi::HostMesh &m = c->mesh;
    if (m.dim == 3 && m.p == 2) // the kernels that use it (assemble_q2sf, mf_spmv)
      {
        std::vector<mi::CellLatticeRow> rows;
        c->lat_built = build_cell_lattice(m, rows);
        if (c->lat_built.ncol > 0)
          {
            const int rl = upload(c, &c->d_lat_rows, rows);
            if (rl)
              return rl;
            c->lat_built.rows = c->d_lat_rows;
            c->lat_rows_host  = rows;
          }
        if (!(mi::exp_env("MI_CELL_LATTICE") && atoi(mi::exp_env("MI_CELL_LATTICE")) == 0))
          c->lat = c->lat_built;
      }
    c->n     = m.ndofs;
    c->own0  = c->slab.own_begin * c->dim;
    c->own_n = (c->slab.own_end - c->slab.own_begin) * c->dim;
    c->kappa = (2.0 * mat->mu * (1.0 + mat->nu)) / (3.0 * (1.0 - 2.0 * mat->nu)); // neo_hook_material.h:20
    // nonlinear_elasticity.h:242-250
    c->alpha[1] = 1. / (nm->beta * std::pow(nm->delta_t, 2));
    c->alpha[2] = 1. / (nm->beta * nm->delta_t);
    c->alpha[3] = (1 - (2 * nm->beta)) / (2 * nm->beta);
    c->alpha[4] = nm->gamma / (nm->beta * nm->delta_t);
    c->alpha[5] = 1 - (nm->gamma / nm->beta);
    c->alpha[6] = (1 - (nm->gamma / (2 * nm->beta))) * nm->delta_t;

    int rc;
#define UP(dst, src)                    \
  if ((rc = upload(c, &(dst), (src))))  \
    return rc;
    UP(c->d_conn, m.conn)
    UP(c->d_cverts, m.cverts)
    UP(c->d_off, m.off)
    UP(c->d_rowptr, m.rowptr)
    UP(c->d_col, m.colidx)
    UP(c->d_diagpos, m.diagpos)
    UP(c->d_rowinfo, m.rowinfo)
    UP(c->d_rowwx, m.rowwx)
    UP(c->d_sell_wx, m.sell_wx)
    UP(c->d_cmask, m.cmask)
    UP(c->d_iface_nodes, m.iface_nodes)
    {
      std::vector<int32_t> f;
      for (const auto &x : m.iface_faces)
        {
          f.push_back(x.cell);
          f.push_back(x.face);
        }
      UP(c->d_faces, f)
      // per interface node: which (entry, local node) pairs contribute, in entry order (neumann_gather)
      const int np1 = m.np1, npc = m.npc, dim = m.dim, p = m.p;
      std::vector<std::pair<int32_t, int32_t>> pairs; // (node, entry * npc + a)
      for (size_t e = 0; e < m.iface_faces.size(); ++e)
        for (int a = 0; a < npc; ++a)
          {
            const int ai[3] = {a % np1, (a / np1) % np1, dim == 3 ? a / (np1 * np1) : 0};
            bool      on = false;
            for (int fc = 0; fc < 2 * dim; ++fc)
              if ((m.iface_faces[e].face >> fc) & 1)
                on = on || ai[fc >> 1] == ((fc & 1) ? p : 0);
            if (on)
              pairs.push_back({m.conn[size_t(m.iface_faces[e].cell) * npc + a], int32_t(e * npc + a)});
          }
      std::stable_sort(pairs.begin(), pairs.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
      std::vector<int32_t> ids, start(1, 0), src;
      for (size_t k = 0; k < pairs.size(); ++k)
        {
          if (k == 0 || pairs[k].first != pairs[k - 1].first)
            {
              if (k)
                start.push_back(int32_t(k));
              ids.push_back(pairs[k].first);
            }
          src.push_back(pairs[k].second);
        }
      if (!pairs.empty())
        start.push_back(int32_t(pairs.size()));
      c->n_fn = int(ids.size());
      UP(c->d_fn_ids, ids)
      UP(c->d_fn_start, start)
      UP(c->d_fn_src, src)
      HIPCHK(c, hipMalloc((void **)&c->d_face_slots, std::max<size_t>(1, m.iface_faces.size() * size_t(npc) * dim) * sizeof(double)));
    }
    UP(c->d_tab, c->tab.packed())
    UP(c->d_sell_perm, m.sell_perm)
    UP(c->d_sell_len, m.sell_len)
    UP(c->d_sell_off, m.sell_off)
    UP(c->d_sell_box, m.sell_box)
    // local interface nodes -> slots of the global interface list; owned ones feed the displacement gather
    {
      std::vector<int32_t> own_nodes, own_slots;
      for (int32_t ln : m.iface_nodes)
        {
          const int64_t g  = T.ext_node(ln + c->slab.node_offset); // the list is in the reference's node order
          const auto    it = std::lower_bound(T.iface_global.begin(), T.iface_global.end(), g);
          if (it == T.iface_global.end() || *it != g)
            return fail(c, MI_EINVAL, "internal error: interface node %lld missing from the global list", (long long)g);
          const int32_t slot = int32_t(it - T.iface_global.begin());
          c->iface_slot.push_back(slot);
          if (ln >= c->slab.own_begin && ln < c->slab.own_end)
            {
              own_nodes.push_back(ln);
              own_slots.push_back(slot);
            }
        }
      c->n_own_if = int(own_nodes.size());
      UP(c->d_own_if_nodes, own_nodes)
      UP(c->d_own_if_slots, own_slots)
    }
#undef UP
    const size_t dd = size_t(c->dim) * c->dim;
    HIPCHK(c, hipMalloc((void **)&c->d_sell_col, std::max<size_t>(1, size_t(m.sell_nblk64) * 64) * sizeof(int32_t)));
    // the tangent: ONE array, written by the element scatter, read by the SpMV (padding rows of the last slice of a
    // length class included: they are zeroed once and never written)
    const size_t nvals = std::max<size_t>(1, size_t(m.nvalblocks()) * dd);
    // (+ the ZERO and TRASH blocks of the element kernel's branch-free scatter: blocks nvalblocks + 1 and + 2)
    c->vals_doubles = nvals + 2 + 4 * dd;
    HIPCHK(c, hipMalloc((void **)&c->d_vals, c->vals_doubles * sizeof(double)));
    HIPCHK(c, hipMalloc((void **)&c->d_vecs, size_t(MI_V_COUNT) * size_t(c->n) * sizeof(double)));
    HIPCHK(c, hipMalloc((void **)&c->d_work, size_t(W_COUNT) * size_t(c->n) * sizeof(double)));
    HIPCHK(c, hipMalloc((void **)&c->d_saved, size_t(6) * size_t(c->n) * sizeof(double)));
    HIPCHK(c, hipMalloc((void **)&c->d_part, size_t(8) * MAX_PART * sizeof(double)));
    HIPCHK(c, hipMalloc((void **)&c->d_sc, 16 * sizeof(double)));
    HIPCHK(c, hipMalloc((void **)&c->d_flags, 4 * sizeof(int32_t)));
    const size_t nif = std::max<size_t>(m.iface_nodes.size(), 1) * size_t(c->dim);
    HIPCHK(c, hipMalloc((void **)&c->d_iface_buf, nif * sizeof(double)));
    c->h_pinned_doubles = std::max(nif, T.iface_global.size() * size_t(c->dim)) + 64;
    HIPCHK(c, hipHostMalloc((void **)&c->h_pinned, c->h_pinned_doubles * sizeof(double), hipHostMallocDefault));
    HIPCHK(c, hipMemsetAsync(c->d_vals, 0, c->vals_doubles * sizeof(double), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_vecs, 0, size_t(MI_V_COUNT) * size_t(c->n) * sizeof(double), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_work, 0, size_t(W_COUNT) * size_t(c->n) * sizeof(double), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_sc, 0, 16 * sizeof(double), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_flags, 0, 4 * sizeof(int32_t), c->stream));
    mi::launch_sell_build_cols(sell_params(c, nullptr, nullptr, nullptr, nullptr, nullptr), c->d_rowptr, c->d_col,
                               c->d_sell_col, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));

    // launch geometry: vector kernels use a fixed grid so that reduction partials are deterministic;
    // SpMV: one wavefront per 64-row slice (measured best), grid-stride above MAX_PART workgroups
    c->grid_vec  = int(std::max<int64_t>(1, std::min<int64_t>(1024, (c->own_n + 255) / 256)));
    // (mf_gather_dot walks two dependent loads per dof: eight dofs per thread keep enough of them in flight)
    c->grid_gdot = int(std::max<int64_t>(1, std::min<int64_t>(MAX_PART / 2, (c->n + 2047) / 2048)));
    {
      const int64_t nin = m.sell_nslices_interior, nbd = m.sell_nslices - nin;
      constexpr int64_t W = mi::SELL_WPB; // one wavefront per slice, W wavefronts per workgroup ...
      // ... unless the launch is small: then one WORKGROUP per slice (sell_spmv_split), decided by the slice count alone
      const bool use_split = !(mi::exp_env("MI_SELL_SPLIT") && atoi(mi::exp_env("MI_SELL_SPLIT")) == 0);
      const int  split_max = mi::exp_env("MI_SELL_SPLIT_MAX") ? atoi(mi::exp_env("MI_SELL_SPLIT_MAX")) : mi::SELL_SPLIT_MAX_SLICES; // A/B
      c->split_int      = use_split && nin > 0 && nin <= split_max;
      c->split_bnd      = use_split && nbd > 0 && nbd <= split_max;
      c->grid_spmv_bnd  = c->split_bnd ? int(nbd) : int(std::min<int64_t>(MAX_PART / 4, (nbd + W - 1) / W));
      c->grid_spmv_int  = c->split_int ? int(nin) : int(std::min<int64_t>(MAX_PART - c->grid_spmv_bnd, (nin + W - 1) / W));
      c->grid_spmv      = std::max(1, c->grid_spmv_int + c->grid_spmv_bnd);
      if (nin == 0 && nbd == 0)
        c->grid_spmv_int = 1;
    }
    for (int64_t nd = 0; nd < m.nnodes; ++nd)
      c->maxrow = std::max(c->maxrow, int(m.rowptr[size_t(nd) + 1] - m.rowptr[size_t(nd)]));
    if (const char *v = mi::exp_env("MI_SPMV_VARIANT"))
      c->spmv_variant = atoi(v);
    if (const char *v = mi::exp_env("MI_SELL_UNROLL"))
      c->sell_unroll = atoi(v);
    if (const char *v = mi::exp_env("MI_SELL_ICOL"))
      c->sell_icol = atoi(v) != 0;
    if (const char *v = mi::exp_env("MI_CG_WARM_START"))
      c->cg_warm_start = std::min(3, std::max(0, atoi(v)));
    if (const char *v = mi::exp_env("MI_SMALL_CG"))
      c->small_cg = atoi(v) != 0;
    if (const char *v = mi::exp_env("MI_CG_SINGLE_REDUCTION"))
      c->cg_single_reduction = std::min(1, std::max(-1, atoi(v)));
    return MI_OK;
  }

  void destroy_team(Team *T)
  {
    if (!T)
      return;
    hipSetDevice(T->device);
    if (T->stream)
      hipStreamSynchronize(T->stream);
    for (mi_ctx *m : T->members)
      destroy_member(m);
    if (T->nccl)
      ncclCommDestroy(static_cast<ncclComm_t>(T->nccl));
    for (void *p : {(void *)T->d_gbuf, (void *)T->d_ifbuf, (void *)T->d_sc_ptrs, (void *)T->d_acc})
      if (p)
        hipFree(p);
    if (T->comm_stream)
      {
        hipStreamSynchronize(T->comm_stream);
        hipStreamDestroy(T->comm_stream);
      }
    for (hipEvent_t e : {T->ev_ready, T->ev_halo})
      if (e)
        hipEventDestroy(e);
    if (T->stream && T->owns_stream)
      hipStreamDestroy(T->stream);
    delete T;
  }
} // namespace mi_detail

using namespace mi_detail;

extern "C" {

const char *mi_last_error(const mi_ctx *ctx)
{
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

void mi_ctx_destroy(mi_ctx *c)
{
  if (c)
    destroy_team(c->team);
}

int mi_comm_unique_id(void *out128)
{
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess)
    return fail(nullptr, MI_ECOMM, "ncclGetUniqueId failed");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  std::memcpy(out128, &id, sizeof(id));
  return MI_OK;
}

int mi_ctx_create(const mi_mesh_desc *md, const mi_material_desc *mat, const mi_newmark_desc *nm, int device_id,
                  const mi_comm_desc *comm, mi_ctx **out)
{
  if (!md || !mat || !nm || !out)
    return fail(nullptr, MI_EINVAL, "null argument");
  *out = nullptr;
  if (md->dim != 2 && md->dim != 3)
    return fail(nullptr, MI_EINVAL, "dim must be 2 or 3");
  if (!(mat->nu > -1.0 && mat->nu < 0.5) || !(mat->mu > 0.0) || mat->rho < 0.0)
    return fail(nullptr, MI_EINVAL, "material out of range (mu>0, -1<nu<0.5, rho>=0)");
  if (!(nm->beta > 0.0) || !(nm->delta_t > 0.0))
    return fail(nullptr, MI_EINVAL, "Newmark beta and time step must be positive");
  const int  nranks   = comm ? comm->size : 1;
  const bool emulated = comm && comm->size > 1 && comm->rank < 0;
  if (nranks < 1 || (comm && !emulated && (comm->rank < 0 || comm->rank >= nranks)))
    return fail(nullptr, MI_EINVAL, "bad communicator description (rank %d of %d)", comm ? comm->rank : 0, nranks);
  if (nranks > 1 && !emulated && !comm->nccl_unique_id)
    return fail(nullptr, MI_EINVAL, "a ncclUniqueId is required for %d ranks", nranks);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, MI_EHIP, "no HIP device available (the hot path has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev)
    return fail(nullptr, MI_EINVAL, "device %d out of range (%d devices)", device_id, ndev);

  Team *T     = new Team;
  T->size     = nranks;
  T->emulated = emulated;
  T->device   = device_id;
  T->dim      = md->dim;
  T->md       = *md;
  const mi_mesh_desc *md_ext = md; // the box as the caller describes it (reference order: x fastest)
  mi_mesh_desc        md_int = *md;
  if (nranks > 1)
    {
      // which direction the slabs cut; any but the last: the lattice is laid over the box rotated (mi::AxisMap)
      const int want = comm ? comm->cut_axis : 0;
      if (want < 0 || want > md->dim)
        {
          g_create_error = "cut_axis must be 0 (automatic) or 1..dim";
          delete T;
          return MI_EINVAL;
        }
      try
        {
          T->amap = mi::make_axis_map(md->dim, md->reps, want - 1);
        }
      catch (const std::exception &e)
        {
          g_create_error = e.what();
          delete T;
          return MI_EINVAL;
        }
      if (!T->amap.identity)
        {
          mi::rotate_box(T->amap, md->dim, md->reps, md->lo, md->hi, md->face_role, md_int.reps, md_int.lo, md_int.hi,
                         md_int.face_role);
          int     nn_ext[3] = {1, 1, 1}, nv_ext[3] = {1, 1, 1};
          int64_t nnodes = 1, nverts = 1;
          for (int d = 0; d < md->dim; ++d)
            {
              nn_ext[d] = md->degree * md->reps[d] + 1;
              nv_ext[d] = md->reps[d] + 1;
              nnodes *= nn_ext[d];
              nverts *= nv_ext[d];
            }
          T->e2i.resize(size_t(nnodes));
          T->i2e.resize(size_t(nnodes));
          for (int64_t g = 0; g < nnodes; ++g)
            {
              const int64_t gi     = mi::ext_to_int_point(T->amap, md->dim, nn_ext, g);
              T->e2i[size_t(g)]    = gi;
              T->i2e[size_t(gi)]   = g;
            }
          if (md->vertex_perturbation) // tests: the offsets in internal vertex order (components stay physical)
            {
              T->perturb_int.resize(size_t(nverts) * md->dim);
              for (int64_t v = 0; v < nverts; ++v)
                {
                  const int64_t vi = mi::ext_to_int_point(T->amap, md->dim, nv_ext, v);
                  for (int k = 0; k < md->dim; ++k)
                    T->perturb_int[size_t(vi) * md->dim + k] = md->vertex_perturbation[size_t(v) * md->dim + k];
                }
              md_int.vertex_perturbation = T->perturb_int.data();
            }
          md    = &md_int;
          T->md = md_int;
        }
    }
  T->md.vertex_perturbation = nullptr; // coarse levels use the unperturbed box
  auto bail   = [&](int code, const std::string &msg) {
    g_create_error = msg;
    destroy_team(T);
    return code;
  };
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&T->stream, hipStreamNonBlocking) != hipSuccess)
    return bail(MI_EHIP, "cannot create a HIP stream on the selected device");
  try
    {
      if (md->degree < 1 || md->degree > 4)
        throw std::invalid_argument("polynomial degree must be in 1..4");
      for (int d = 0; d < md->dim; ++d)
        if (md->reps[d] < 1)
          throw std::invalid_argument("repetitions must be >= 1");
      // the coupling interface in the reference's order: ascending node id of the box as the caller describes it
      T->iface_global = mi::global_interface_nodes(md_ext->dim, md_ext->degree, md_ext->reps, md_ext->face_role);
      const mi::SlabPartition s0 =
        mi::make_slab_partition(md->dim, md->degree, md->reps, md->lo, md->hi, md->face_role, 0, nranks);
      T->nnodes_global = s0.nnodes_global;
      T->n_global      = s0.nnodes_global * md->dim;
    }
  catch (const std::exception &e)
    {
      return bail(MI_EINVAL, e.what());
    }
  if (comm && !emulated && comm->nccl_unique_id) // also for one rank: exercises communicator + all-reduce
    {
      ncclUniqueId id;
      std::memcpy(&id, comm->nccl_unique_id, sizeof(id));
      ncclComm_t         nc = nullptr;
      const ncclResult_t r  = ncclCommInitRank(&nc, nranks, id, comm->rank);
      T->nccl               = nc;
      if (r != ncclSuccess)
        return bail(MI_ECOMM, std::string("ncclCommInitRank failed: ") + ncclGetErrorString(r));
    }
  if (nranks > 1)
    {
      // second stream + events for the halo exchange that overlaps with the interior rows of the SpMV
      if (hipStreamCreateWithFlags(&T->comm_stream, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&T->ev_ready, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&T->ev_halo, hipEventDisableTiming) != hipSuccess)
        return bail(MI_EHIP, "cannot create the communication stream");
      if (const char *e = mi::exp_env("MI_HALO_OVERLAP"))
        T->overlap = atoi(e) != 0;
    }
  const int first = emulated ? 0 : (comm ? comm->rank : 0), count = emulated ? nranks : 1;
  for (int r = first; r < first + count; ++r)
    {
      mi_ctx   *m  = nullptr;
      const int rc = create_member(*T, md, mat, nm, r, &m);
      T->members.push_back(m);
      if (rc != MI_OK)
        return bail(rc, m->err);
    }
  mi_ctx *c0 = T->members[0];
  {
    // preconditioner: the multigrid V-cycle pays off from ~75k dofs (a V-cycle is ~100 small launches, i.e. a
    // fixed ~0.8 ms per CG iteration whatever the size; measured crossover with Jacobi-PCG at 73k dofs,
    // tools/small_case_latency.py); below that Jacobi.  mi_set_tuning("precond") / MI_PRECOND override.
    int precond = T->n_global >= 75000 ? 1 : 0;
    if (const char *e = mi::exp_env("MI_PRECOND"))
      precond = atoi(e) != 0;
    for (mi_ctx *m : T->members)
      {
        m->precond = precond;
        if (const char *e = mi::exp_env("MI_MG_REFRESH_EVERY"))
          m->mg_refresh_every = std::max(1, atoi(e));
        if (precond == 1)
          {
            const int rc = mg_setup(m);
            if (rc != MI_OK)
              return bail(rc, m->err);
          }
      }
  }
  // element tangents for the smoother (see ebe_spmv): 3D Q2 with the multigrid preconditioner, on every slab that is big
  // enough for the unfused smoother (small problems run the fused smoother on the assembled matrix)
  for (mi_ctx *m : T->members)
    {
      if (const char *e = mi::exp_env("MI_EBE"))
        m->ebe = std::max(0, std::min(2, atoi(e)));
      if (const char *e = mi::exp_env("MI_MF_SINGLE_LAUNCH"))
        m->mf_slots = atoi(e) != 0;
      if (const char *e = mi::exp_env("MI_CORRECT_FACE_F")) // the executables' "--correct-face-F" (SURVEY section 9); default: the reference's quirk
        m->correct_face_F = atoi(e) != 0;
      const int rc = ensure_element_tangents(m);
      if (rc != MI_OK)
        return bail(rc, m->err);
    }
  // global interface scratch + coordinates of the global interface nodes (summed over the owners)
  {
    const size_t nifg = std::max<size_t>(1, T->iface_global.size() * size_t(md->dim));
    if (hipMalloc((void **)&T->d_ifbuf, nifg * sizeof(double)) != hipSuccess)
      return bail(MI_EHIP, "hipMalloc of the interface scratch failed");
    std::vector<double> xyz(nifg, 0.0);
    for (mi_ctx *m : T->members)
      for (size_t i = 0; i < m->mesh.iface_nodes.size(); ++i)
        {
          const int32_t ln = m->mesh.iface_nodes[i];
          if (ln >= m->slab.own_begin && ln < m->slab.own_end)
            for (int k = 0; k < md->dim; ++k)
              xyz[size_t(m->iface_slot[i]) * md->dim + k] = m->mesh.node_xyz[size_t(ln) * md->dim + k];
        }
    if (T->nccl)
      {
        if (hipMemcpy(T->d_ifbuf, xyz.data(), nifg * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
            ncclAllReduce(T->d_ifbuf, T->d_ifbuf, nifg, ncclDouble, ncclSum, static_cast<ncclComm_t>(T->nccl), T->stream) != ncclSuccess ||
            hipStreamSynchronize(T->stream) != hipSuccess ||
            hipMemcpy(xyz.data(), T->d_ifbuf, nifg * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
          return bail(MI_ECOMM, "exchange of the interface coordinates failed");
      }
    T->iface_xyz = xyz;
  }
  if (emulated)
    {
      std::vector<double *> ptrs;
      for (mi_ctx *m : T->members)
        ptrs.push_back(m->d_sc);
      if (hipMalloc((void **)&T->d_sc_ptrs, ptrs.size() * sizeof(double *)) != hipSuccess ||
          hipMemcpy(T->d_sc_ptrs, ptrs.data(), ptrs.size() * sizeof(double *), hipMemcpyHostToDevice) != hipSuccess)
        return bail(MI_EHIP, "hipMalloc of the team pointer table failed");
    }
  *out = c0;
  return MI_OK;
}

int64_t mi_n_dofs(const mi_ctx *c)
{
  return c->team->n_global;
}
int64_t mi_n_nodes(const mi_ctx *c)
{
  return c->team->nnodes_global;
}
int64_t mi_n_cells(const mi_ctx *c)
{
  // cells of the undecomposed box
  const mi::SlabPartition &s = c->slab;
  int64_t                  n = (s.nnodes_global / s.plane_nodes - 1) / s.p; // layers in the decomposed direction
  for (int d = 0; d + 1 < c->dim; ++d)
    n *= c->mesh.reps[d];
  return n;
}
int64_t mi_nnz(const mi_ctx *c)
{
  // scalar non-zeros of the undecomposed tangent: sum of (coupled box size) over all lattice nodes, closed form
  int64_t prod = int64_t(c->dim) * c->dim;
  int     reps[3] = {1, 1, 1};
  for (int d = 0; d + 1 < c->dim; ++d)
    reps[d] = c->mesh.reps[d];
  reps[c->dim - 1] = int((c->slab.nnodes_global / c->slab.plane_nodes - 1) / c->slab.p);
  const int p      = c->degree;
  for (int d = 0; d < c->dim; ++d)
    {
      // along one direction: interior cell-boundary nodes couple 2p+1, the two end nodes and cell-interior nodes p+1
      const int64_t n = reps[d];
      prod *= (n - 1) * (2 * p + 1) + 2 * (p + 1) + n * (p - 1) * (p + 1);
    }
  return prod;
}
int mi_n_colours(const mi_ctx *c)
{
  return c->mesh.ncolours;
}
int mi_n_interface_nodes(const mi_ctx *c)
{
  return int(c->team->iface_global.size());
}
int mi_get_interface_nodes(const mi_ctx *c, int32_t *node_ids, double *coords)
{
  const Team &T = *c->team;
  for (size_t i = 0; i < T.iface_global.size(); ++i)
    {
      if (node_ids)
        node_ids[i] = int32_t(T.iface_global[i]);
      if (coords)
        for (int k = 0; k < c->dim; ++k)
          coords[i * c->dim + k] = T.iface_xyz[i * c->dim + k];
    }
  return MI_OK;
}

// global <- owned parts of a per-node host quantity (dim values per node), summed over ranks when needed
static int gather_global_host(mi_ctx *c, const std::function<double(mi_ctx *, int64_t, int)> &val, double *out)
{
  Team &T = *c->team;
  HIPCHK(c, hipSetDevice(c->device));
  std::fill(out, out + T.n_global, 0.0);
  for (mi_ctx *m : T.members)
    for (int64_t ln = m->slab.own_begin; ln < m->slab.own_end; ++ln)
      for (int k = 0; k < m->dim; ++k)
        out[T.ext_node(ln + m->slab.node_offset) * m->dim + k] = val(m, ln, k);
  if (T.nccl)
    {
      int rc = ensure_gbuf(T);
      if (rc)
        return rc;
      HIPCHK(c, hipMemcpy(T.d_gbuf, out, size_t(T.n_global) * sizeof(double), hipMemcpyHostToDevice));
      if ((rc = team_allreduce_buffer(T, T.d_gbuf, size_t(T.n_global))))
        return rc;
      HIPCHK(c, hipStreamSynchronize(T.stream));
      HIPCHK(c, hipMemcpy(out, T.d_gbuf, size_t(T.n_global) * sizeof(double), hipMemcpyDeviceToHost));
    }
  return MI_OK;
}

int mi_get_node_coords(const mi_ctx *cc, double *xyz)
{
  mi_ctx *c = const_cast<mi_ctx *>(cc);
  return gather_global_host(
    c, [](mi_ctx *m, int64_t ln, int k) { return m->mesh.node_xyz[size_t(ln) * m->dim + k]; }, xyz);
}
int mi_get_constrained(const mi_ctx *cc, uint8_t *flags)
{
  mi_ctx             *c = const_cast<mi_ctx *>(cc);
  std::vector<double> tmp(size_t(c->team->n_global));
  const int           rc = gather_global_host(
    c, [](mi_ctx *m, int64_t ln, int k) { return double((m->mesh.cmask[size_t(ln)] >> k) & 1); }, tmp.data());
  for (int64_t i = 0; i < c->team->n_global; ++i)
    flags[i] = tmp[size_t(i)] != 0.0;
  return rc;
}

int mi_set_interface_traction(mi_ctx *c, int n, const double *vals)
{
  Team &T = *c->team;
  if (n != int(T.iface_global.size()))
    return fail(c, MI_EINVAL, "expected %d interface nodes, got %d", int(T.iface_global.size()), n);
  if (n == 0)
    return MI_OK;
  HIPCHK(c, hipSetDevice(c->device));
  c->h_iface.assign(vals, vals + size_t(n) * c->dim);
  for (mi_ctx *m : T.members)
    {
      const int nl = int(m->mesh.iface_nodes.size());
      if (nl == 0)
        continue;
      double *stage = m->h_pinned + 64;
      for (int i = 0; i < nl; ++i)
        for (int k = 0; k < m->dim; ++k)
          stage[i * m->dim + k] = vals[size_t(m->iface_slot[size_t(i)]) * m->dim + k];
      HIPCHK(m, hipMemcpyAsync(m->d_iface_buf, stage, size_t(nl) * m->dim * sizeof(double), hipMemcpyHostToDevice,
                               m->stream));
      mi::launch_scatter_nodes(m->dim, m->vec(MI_V_EXTERNAL_STRESS), m->d_iface_nodes, nl, m->d_iface_buf, m->stream);
    }
  HIPCHK(c, hipGetLastError());
  return sync(c); // the pinned staging buffers are reused by the next call
}

int mi_get_interface_displacement(mi_ctx *c, int n, double *vals)
{
  Team &T = *c->team;
  if (n != int(T.iface_global.size()))
    return fail(c, MI_EINVAL, "expected %d interface nodes, got %d", int(T.iface_global.size()), n);
  if (n == 0)
    return MI_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t bytes = size_t(n) * c->dim * sizeof(double);
  HIPCHK(c, hipMemsetAsync(T.d_ifbuf, 0, bytes, T.stream));
  for (mi_ctx *m : T.members)
    mi::launch_gather_to_slots(m->dim, m->vec(MI_V_TOTAL_DISPLACEMENT), m->d_own_if_nodes, m->d_own_if_slots, m->n_own_if,
                               T.d_ifbuf, T.stream);
  HIPCHK(c, hipGetLastError());
  int rc = team_allreduce_buffer(T, T.d_ifbuf, size_t(n) * c->dim);
  if (rc)
    return rc;
  HIPCHK(c, hipMemcpyAsync(c->h_pinned + 64, T.d_ifbuf, bytes, hipMemcpyDeviceToHost, T.stream));
  if ((rc = sync(c)))
    return rc;
  std::memcpy(vals, c->h_pinned + 64, bytes);
  return MI_OK;
}

// Values of rank 0 to every rank of the team (round 4): what the ONE preCICE-facing process has learnt from the coupling
// library -- read data, isCouplingOngoing, the checkpoint decisions -- reaches the other ranks through the library's own
// communicator (host/include/adapter/rank_zero_participant.h).  Ranks > 0 contribute zeros to an all-reduce of the
// interface scratch: x + 0 + ... + 0 = x bit by bit (a -0.0 arrives as +0.0).  One process (single or emulated slabs): no-op.
int mi_comm_broadcast(mi_ctx *c, double *values, int32_t n)
{
  if (!c || (n > 0 && !values) || n < 0)
    return fail(c, MI_EINVAL, "mi_comm_broadcast: bad arguments");
  Team &T = *c->team;
  if (!T.nccl || n == 0)
    return MI_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t cap  = std::max<size_t>(1, T.iface_global.size() * size_t(c->dim));
  const size_t capp = std::min(cap, c->h_pinned_doubles > 64 ? c->h_pinned_doubles - 64 : size_t(0));
  if (capp == 0)
    return fail(c, MI_EINVAL, "mi_comm_broadcast: no staging buffer");
  const bool root = c->slab.rank == 0;
  for (size_t off = 0; off < size_t(n); off += capp)
    {
      const size_t cnt = std::min(capp, size_t(n) - off);
      if (root)
        {
          std::memcpy(c->h_pinned + 64, values + off, cnt * sizeof(double));
          HIPCHK(c, hipMemcpyAsync(T.d_ifbuf, c->h_pinned + 64, cnt * sizeof(double), hipMemcpyHostToDevice, T.stream));
        }
      else
        HIPCHK(c, hipMemsetAsync(T.d_ifbuf, 0, cnt * sizeof(double), T.stream));
      int rc = team_allreduce_buffer(T, T.d_ifbuf, cnt);
      if (rc)
        return rc;
      HIPCHK(c, hipMemcpyAsync(c->h_pinned + 64, T.d_ifbuf, cnt * sizeof(double), hipMemcpyDeviceToHost, T.stream));
      if ((rc = sync(c)))
        return rc;
      std::memcpy(values + off, c->h_pinned + 64, cnt * sizeof(double));
    }
  return MI_OK;
}

int mi_newton_begin_step(mi_ctx *c)
{
  HIPCHK(c, hipSetDevice(c->device));
  for (mi_ctx *m : c->team->members)
    {
      HIPCHK(m, hipMemsetAsync(m->vec(MI_V_SOLUTION_DELTA), 0, size_t(m->n) * sizeof(double), m->stream));
      HIPCHK(m, hipMemsetAsync(m->vec(MI_V_NEWTON_UPDATE), 0, size_t(m->n) * sizeof(double), m->stream));
      m->mf_diag_fresh = false; // ("mf_diag_lag": the step's first tangent forms the diagonal blocks again)
      // new time step: refresh the coarse operators at its first solve ("mg_refresh_every" k: at every k-th step)
      if (m->mg_steps_since_refresh + 1 >= m->mg_refresh_every)
        m->mg_force = true;
      else
        ++m->mg_steps_since_refresh;
    }
  c->team->members[0]->newton_update_is_zero = true;
  c->team->members[0]->solves_this_step      = 0;
  return MI_OK;
}

int mi_update_acceleration(mi_ctx *c)
{
  HIPCHK(c, hipSetDevice(c->device));
  mi_ctx   *c0 = c->team->members[0];
  const int t  = tic(c0, MI_T_NEWMARK);
  for (mi_ctx *m : c->team->members)
    mi::launch_newmark_acceleration(newmark_params(m), m->stream);
  toc(c0, t);
  HIPCHK(c, hipGetLastError());
  return MI_OK;
}

static int assemble_impl(mi_ctx *c, bool residual_only, double *res_norm)
{
  HIPCHK(c, hipSetDevice(c->device));
  Team     &T  = *c->team;
  mi_ctx   *c0 = T.members[0];
  const int t  = tic(c0, MI_T_ASSEMBLE_TOTAL);
  for (mi_ctx *m : T.members)
    {
      const int rc = enqueue_assembly(m, residual_only);
      if (rc)
        return rc;
    }
  toc(c0, t);
  double    nrm = 0, inverted = 0;
  const int rc  = team_masked_norm(T, MI_V_SYSTEM_RHS, SC_NORM_RHS, &nrm, &inverted);
  if (res_norm)
    *res_norm = nrm;
  if (rc == MI_OK && inverted > 0.0) // the reference's debug build stops here too: Assert(det_F > 0) (:935)
    return fail(c, MI_EINVAL, "inverted element: det F <= 0 at a quadrature point (the displacement has folded a cell; the "
                              "reference asserts det F > 0, nonlinear_elasticity.cc:935)");
  return rc;
}

int mi_assemble(mi_ctx *c, double *res_norm)
{
  return assemble_impl(c, false, res_norm);
}

int mi_assemble_residual(mi_ctx *c, double *res_norm)
{
  return assemble_impl(c, true, res_norm);
}

int mi_comm_info(const mi_ctx *c, int *team_size, int *rccl_ranks)
{
  if (!c)
    return MI_EINVAL;
  const Team &T = *c->team;
  if (team_size)
    *team_size = T.size;
  if (rccl_ranks)
    {
      *rccl_ranks = 0;
      if (T.nccl)
        {
          int n = 0;
          if (ncclCommCount(static_cast<ncclComm_t>(T.nccl), &n) != ncclSuccess)
            return MI_ECOMM;
          *rccl_ranks = n;
        }
    }
  return MI_OK;
}

int mi_cg_solve(mi_ctx *c, double rel_tol, int64_t max_it, int *its, double *res)
{
  HIPCHK(c, hipSetDevice(c->device));
  if (rel_tol < 0)
    return fail(c, MI_EINVAL, "negative tolerance");
  for (mi_ctx *m : c->team->members)
    {
      m->active_sell_vals = nullptr; // tangent
      m->active_dinv      = nullptr;
    }
  // SolverCG starts from the passed vector (:1184-1187): whatever MI_V_NEWTON_UPDATE holds (see mi_apply_newton_update);
  // when the library itself has just cleared it, r0 = b needs no product
  mi_ctx *c0     = c->team->members[0];
  bool    x_zero = c0->newton_update_is_zero;
  c0->newton_update_is_zero = false;
  // "cg_warm_start" 2 (what the executable and bench.py set) / 3: when the library has just cleared the update, the j-th solve of a time step starts
  // from the solution of the j-th solve of the previous step (3: extrapolated linearly over the last two) instead of
  // zero -- the loads of a time-stepping run change little from step to step.  Same stopping rule (:1155-1156, a
  // residual norm relative to |rhs|); costs one product for r0, saves about one iteration in seven on the headline run.
  const int  pj   = c0->solves_this_step++;
  const bool pred = c0->cg_warm_start >= 2 && pj < mi_ctx::NPRED;
  bool       predicted = false;
  if (pred && x_zero && c0->pred_count[pj] >= 1)
    {
      const bool two = c0->cg_warm_start == 3 && c0->pred_count[pj] >= 2;
      for (mi_ctx *m : c->team->members)
        mi::launch_vec_lincomb2(m->vec(MI_V_NEWTON_UPDATE), two ? 2.0 : 1.0, m->d_pred[pj][0], two ? -1.0 : 0.0, m->d_pred[pj][1],
                                m->n, m->stream);
      x_zero    = false;
      predicted = true;
    }
  // A multigrid-preconditioned solve needs 7-15 iterations; one that has not converged after 300 has stalled, and
  // iterating on to max_it (dofs x multiplier, i.e. millions) would be a hang in all but name.
  const bool    mg     = mg_active(c);
  const int64_t mg_cap = mg ? std::min<int64_t>(max_it, 300) : max_it;
  int           my_its = 0;
  // (the iteration count of the same solve one step earlier tells cg_run how far it may enqueue without polling)
  const int expect = (pred && predicted) ? c0->pred_its[pj] : 0;
  int rc = cg_run(c, MI_V_NEWTON_UPDATE, MI_V_SYSTEM_RHS, rel_tol, mg_cap, &my_its, res, x_zero, predicted, expect);
  if (pred)
    c0->pred_its[pj] = rc == MI_OK ? my_its : 0;
  // a breakdown (NaN state, indefinite tangent) is final, as with deal.II's SolverControl: no second attempt from a
  // poisoned iterate
  bool broke = rc == MI_ENOCONV_LIN && c->team->members[0]->cg_breakdown;
  if (mg && rc == MI_OK)
    {
      // coarse operators kept over several time steps ("mg_refresh_every"): the first solve after a refresh sets the
      // mark, a later solve that needs a quarter (at least 2) more iterations asks for a refresh before the next one
      if (c0->mg_its_ref == 0)
        c0->mg_its_ref = my_its;
      else if (my_its > c0->mg_its_ref + std::max(2, c0->mg_its_ref / 4))
        for (mi_ctx *m : c->team->members)
          m->mg_force = true;
    }
  if (rc == MI_ENOCONV_LIN && mg && !broke && my_its < max_it)
    {
      // first suspect: a smoother interval that ends below lambda_max (its modes are amplified).  Estimate the
      // eigenvalues of every level from scratch and continue from the current iterate.
      mg_reset_estimates(*c->team);
      int more = 0;
      rc       = cg_run(c, MI_V_NEWTON_UPDATE, MI_V_SYSTEM_RHS, rel_tol, std::min<int64_t>(max_it - my_its, mg_cap), &more, res);
      my_its += more;
      broke = rc == MI_ENOCONV_LIN && c->team->members[0]->cg_breakdown;
    }
  if (its)
    *its = my_its;
  if (rc == MI_ENOCONV_LIN && mg && !broke && my_its < max_it)
    {
      // safety net: should the V-cycle still stall, continue from the current iterate with the Jacobi preconditioner
      // instead of giving up
      max_it -= my_its;
      const int done_its = its ? *its : 0;
      for (mi_ctx *m : c->team->members)
        m->precond = 0;
      rc = cg_run(c, MI_V_NEWTON_UPDATE, MI_V_SYSTEM_RHS, rel_tol, max_it, its, res);
      for (mi_ctx *m : c->team->members)
        {
          m->precond  = 1;
          m->mg_force = true; // rebuild the hierarchy before it is used again
        }
      if (its)
        *its += done_its;
    }
  if (pred && rc == MI_OK) // this solution predicts the same solve of the next step
    {
      for (mi_ctx *m : c->team->members)
        {
          for (int h = 0; h < 2; ++h)
            if (!m->d_pred[pj][h])
              HIPCHK(m, hipMalloc((void **)&m->d_pred[pj][h], size_t(m->n) * sizeof(double)));
          std::swap(m->d_pred[pj][0], m->d_pred[pj][1]);
          HIPCHK(m, hipMemcpyAsync(m->d_pred[pj][0], m->vec(MI_V_NEWTON_UPDATE), size_t(m->n) * sizeof(double),
                                   hipMemcpyDeviceToDevice, m->stream));
        }
      ++c0->pred_count[pj];
    }
  return rc;
}

// SparseDirectUMFPACK of the reference (nonlinear_elasticity.cc:1192-1200): factorise the current tangent and solve
int mi_direct_solve(mi_ctx *c, double *res)
{
  HIPCHK(c, hipSetDevice(c->device));
  for (mi_ctx *m : c->team->members)
    {
      m->active_sell_vals = nullptr;
      m->active_dinv      = nullptr;
    }
  c->newton_update_is_zero = false;
  if (c->mf_fine)
    return fail(c, MI_EINVAL, "the direct solver factorises the assembled tangent (\"fine_level\" 0)");
  int rc = direct_factor_solve(c, c->d_vals, c->vec(MI_V_SYSTEM_RHS), c->vec(MI_V_NEWTON_UPDATE), true, true);
  if (rc)
    return rc;
  mi::launch_zero_constrained(c->dim, c->vec(MI_V_NEWTON_UPDATE), c->d_cmask, c->n, c->stream); // :1208
  HIPCHK(c, hipGetLastError());
  if (res)
    *res = 0.0; // lin_res = 0 (:1199)
  return sync(c);
}

int mi_apply_newton_update(mi_ctx *c, double *upd_norm)
{
  HIPCHK(c, hipSetDevice(c->device));
  Team     &T   = *c->team;
  double    nrm = 0;
  const int rc  = team_masked_norm(T, MI_V_NEWTON_UPDATE, SC_NORM_UPD, &nrm);
  if (rc)
    return rc;
  for (mi_ctx *m : T.members) // :487, on the whole local vector (ghost copies of the update are consistent)
    {
      mi::launch_vec_add(m->vec(MI_V_SOLUTION_DELTA), m->vec(MI_V_NEWTON_UPDATE), m->n, m->stream);
      // The reference keeps newton_update over the Newton iterations of a step, so every later solve of the step starts
      // from the PREVIOUS update (:419, :472-473).  That start vector is ~1000x larger than the new solution, and the
      // solve then has to reduce its residual by as much more: 10-11 instead of 7-8 iterations per solve at 5 M dofs
      // (28 instead of 22 per step).  Unless the reference's start vector is asked for ("cg_warm_start" 1), the
      // consumed update is cleared and the next solve starts from zero; the stopping rule is the same.
      if (m->cg_warm_start != 1)
        HIPCHK(m, hipMemsetAsync(m->vec(MI_V_NEWTON_UPDATE), 0, size_t(m->n) * sizeof(double), m->stream));
    }
  T.members[0]->newton_update_is_zero = T.members[0]->cg_warm_start != 1;
  HIPCHK(c, hipGetLastError());
  if (upd_norm)
    *upd_norm = nrm;
  return MI_OK;
}

int mi_newmark_finish_step(mi_ctx *c)
{
  HIPCHK(c, hipSetDevice(c->device));
  mi_ctx   *c0 = c->team->members[0];
  const int t  = tic(c0, MI_T_NEWMARK);
  for (mi_ctx *m : c->team->members)
    mi::launch_newmark_finish(newmark_params(m), m->stream);
  toc(c0, t);
  HIPCHK(c, hipGetLastError());
  return MI_OK;
}

// solve_nonlinear_timestep (:410-499) followed by :139-144
int mi_newmark_step(mi_ctx *c, const mi_solver_desc *s, mi_step_info *info)
{
  if (!s || !info)
    return fail(c, MI_EINVAL, "null argument");
  const auto t_begin = std::chrono::steady_clock::now();
  std::memset(info, 0, sizeof(*info));
  int rc = mi_newton_begin_step(c);
  if (rc)
    return rc;
  // Errors() default/reset value is 1.0 (nonlinear_elasticity.h:293-315)
  double error_residual = 1.0, error_residual_0 = 1.0, error_residual_norm = 1.0;
  double error_update = 1.0, error_update_0 = 1.0, error_update_norm = 1.0;
  int    newton_iteration = 0;
  for (; newton_iteration < s->max_iterations_NR; ++newton_iteration) // :436
    {
      if ((rc = mi_update_acceleration(c))) // :444
        return rc;
      // :446-449.  The convergence test (:459-463) needs the update criterion AND the residual criterion; the first
      // is known before the assembly.  Only when it holds can this assembly be the last one of the step, whose tangent
      // is never multiplied: then the residual alone is formed first (same numbers), and the tangent follows only if
      // the residual criterion fails.
      const bool update_ok = newton_iteration > 0 && (error_update_norm <= s->tol_u || error_update <= 1e-15);
      if ((rc = update_ok ? mi_assemble_residual(c, &error_residual) : mi_assemble(c, &error_residual)))
        return rc;
      info->assemblies++;
      if (newton_iteration == 0)
        error_residual_0 = error_residual;
      error_residual_norm = error_residual;
      if (error_residual_0 != 0.0)
        error_residual_norm /= error_residual_0;
      if (update_ok && (error_residual_norm <= s->tol_f || error_residual <= 5e-9)) // :459-463
        {
          info->converged = 1;
          break;
        }
      if (update_ok && (rc = mi_assemble(c, &error_residual))) // not converged after all: the tangent is needed
        return rc;
      int    its = 0;
      double res = 0;
      if (c->solver_direct) // "Solver type = Direct" (:1192-1200); too large for the device factorisation: CG at 1e-12
        {
          its = 1;
          rc  = mi_direct_solve(c, &res);
          if (rc == MI_EINVAL)
            rc = mi_cg_solve(c, std::min(s->tol_lin, 1e-12), int64_t(double(mi_n_dofs(c)) * std::max(10.0, s->max_iterations_lin)),
                             &its, &res);
        }
      else
        rc = mi_cg_solve(c, s->tol_lin, int64_t(double(mi_n_dofs(c)) * s->max_iterations_lin), &its, &res); // :472
      if (info->newton_iterations < 16)
        {
          info->lin_its[info->newton_iterations] = its;
          info->lin_res[info->newton_iterations] = res;
        }
      info->lin_its_total += its;
      info->newton_iterations++;
      if (rc)
        return rc;
      if ((rc = mi_apply_newton_update(c, &error_update))) // :476-487
        return rc;
      if (newton_iteration == 0)
        error_update_0 = error_update;
      error_update_norm = error_update;
      if (error_update_0 != 0.0)
        error_update_norm /= error_update_0;
    }
  info->res_norm = error_residual_norm;
  info->res_abs  = error_residual;
  info->upd_norm = error_update_norm;
  info->upd_abs  = error_update;
  if (!(newton_iteration < s->max_iterations_NR)) // :497
    return fail(c, MI_ENOCONV_NR, "No convergence in nonlinear solver!");
  if ((rc = mi_newmark_finish_step(c)))
    return rc;
  if ((rc = sync(c)))
    return rc;
  mi_ctx *c0 = c->team->members[0];
  c0->timings.ms[MI_T_STEP] +=
    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  c0->timings.count[MI_T_STEP] += 1;
  return MI_OK;
}

int mi_state_save(mi_ctx *c)
{
  HIPCHK(c, hipSetDevice(c->device));
  // the six state vectors are the first six of the vector block (nonlinear_elasticity.cc:370-375)
  for (mi_ctx *m : c->team->members)
    {
      HIPCHK(m, hipMemcpyAsync(m->d_saved, m->d_vecs, size_t(6) * size_t(m->n) * sizeof(double), hipMemcpyDeviceToDevice,
                               m->stream));
      m->have_saved = true;
      // the start-vector history of the linear solves belongs to the state: a restored run repeats the saved one
      for (int j = 0; j < mi_ctx::NPRED; ++j)
        for (int h = 0; h < 2; ++h)
          if (m->d_pred[j][h])
            {
              if (!m->d_pred_saved[j][h])
                HIPCHK(m, hipMalloc((void **)&m->d_pred_saved[j][h], size_t(m->n) * sizeof(double)));
              HIPCHK(m, hipMemcpyAsync(m->d_pred_saved[j][h], m->d_pred[j][h], size_t(m->n) * sizeof(double),
                                       hipMemcpyDeviceToDevice, m->stream));
            }
      for (int j = 0; j < mi_ctx::NPRED; ++j)
        m->pred_count_saved[j] = c->team->members[0]->pred_count[j];
    }
  return MI_OK;
}
int mi_state_restore(mi_ctx *c)
{
  if (!c->have_saved)
    return fail(c, MI_EINVAL, "state_variables are not the same as previously saved.");
  HIPCHK(c, hipSetDevice(c->device));
  for (mi_ctx *m : c->team->members)
    {
      HIPCHK(m, hipMemcpyAsync(m->d_vecs, m->d_saved, size_t(6) * size_t(m->n) * sizeof(double), hipMemcpyDeviceToDevice,
                               m->stream));
      for (int j = 0; j < mi_ctx::NPRED; ++j)
        for (int h = 0; h < 2; ++h)
          if (m->d_pred[j][h] && m->d_pred_saved[j][h])
            HIPCHK(m, hipMemcpyAsync(m->d_pred[j][h], m->d_pred_saved[j][h], size_t(m->n) * sizeof(double),
                                     hipMemcpyDeviceToDevice, m->stream));
    }
  for (int j = 0; j < mi_ctx::NPRED; ++j)
    c->team->members[0]->pred_count[j] = c->team->members[0]->pred_count_saved[j];
  return MI_OK;
}

struct mi_snapshot
{
  std::vector<double *> d; // one buffer per slab of the team
};

int mi_snapshot_create(mi_ctx *c, mi_snapshot **out)
{
  if (!out)
    return fail(c, MI_EINVAL, "null argument");
  HIPCHK(c, hipSetDevice(c->device));
  mi_snapshot *s = new mi_snapshot;
  for (mi_ctx *m : c->team->members)
    {
      double    *p = nullptr;
      hipError_t e = hipMalloc((void **)&p, size_t(m->n) * sizeof(double));
      if (e != hipSuccess)
        {
          for (double *q : s->d)
            hipFree(q);
          delete s;
          return fail(c, MI_EHIP, "hipMalloc of a snapshot failed: %s", hipGetErrorString(e));
        }
      s->d.push_back(p);
    }
  *out = s;
  return MI_OK;
}
void mi_snapshot_destroy(mi_ctx *c, mi_snapshot *s)
{
  if (!s)
    return;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  for (double *p : s->d)
    hipFree(p);
  delete s;
}
int mi_snapshot_store(mi_ctx *c, mi_snapshot *s, int which)
{
  if (!s || which < 0 || which >= MI_V_COUNT)
    return fail(c, MI_EINVAL, "bad snapshot or vector id");
  HIPCHK(c, hipSetDevice(c->device));
  for (size_t k = 0; k < c->team->members.size(); ++k)
    {
      mi_ctx *m = c->team->members[k];
      HIPCHK(m, hipMemcpyAsync(s->d[k], m->vec(which), size_t(m->n) * sizeof(double), hipMemcpyDeviceToDevice, m->stream));
    }
  return MI_OK;
}
int mi_snapshot_load(mi_ctx *c, const mi_snapshot *s, int which)
{
  if (!s || which < 0 || which >= MI_V_COUNT)
    return fail(c, MI_EINVAL, "bad snapshot or vector id");
  if (which == MI_V_NEWTON_UPDATE)
    c->team->members[0]->newton_update_is_zero = false;
  HIPCHK(c, hipSetDevice(c->device));
  for (size_t k = 0; k < c->team->members.size(); ++k)
    {
      mi_ctx *m = c->team->members[k];
      HIPCHK(m, hipMemcpyAsync(m->vec(which), s->d[k], size_t(m->n) * sizeof(double), hipMemcpyDeviceToDevice, m->stream));
    }
  return MI_OK;
}

// global dof vectors cross the C-ABI in the reference's node order (x fastest); inside a team whose lattice lies rotated
// over the box (decomposition along a direction other than the last, mi::AxisMap) they are permuted on the way
static int gbuf_to_host(mi_ctx *c, double *host)
{
  Team &T = *c->team;
  if (T.e2i.empty())
    {
      HIPCHK(c, hipMemcpy(host, T.d_gbuf, size_t(T.n_global) * sizeof(double), hipMemcpyDeviceToHost));
      return MI_OK;
    }
  std::vector<double> tmp(size_t(T.n_global));
  HIPCHK(c, hipMemcpy(tmp.data(), T.d_gbuf, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
  const int D = T.dim;
  for (int64_t g = 0; g < T.nnodes_global; ++g)
    for (int k = 0; k < D; ++k)
      host[g * D + k] = tmp[size_t(T.e2i[size_t(g)]) * D + k];
  return MI_OK;
}
static const double *to_internal_order(const Team &T, const double *host, std::vector<double> &buf)
{
  if (T.e2i.empty())
    return host;
  const int D = T.dim;
  buf.resize(size_t(T.n_global));
  for (int64_t g = 0; g < T.nnodes_global; ++g)
    for (int k = 0; k < D; ++k)
      buf[size_t(T.e2i[size_t(g)]) * D + k] = host[g * D + k];
  return buf.data();
}

int mi_vec_get(mi_ctx *c, int which, double *host, int64_t n)
{
  Team &T = *c->team;
  if (which < 0 || which >= MI_V_COUNT || n != T.n_global)
    return fail(c, MI_EINVAL, "bad vector id or length");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (T.size == 1)
    {
      HIPCHK(c, hipMemcpy(host, c->vec(which), size_t(n) * sizeof(double), hipMemcpyDeviceToHost));
      return MI_OK;
    }
  int rc = ensure_gbuf(T);
  if (rc)
    return rc;
  HIPCHK(c, hipMemsetAsync(T.d_gbuf, 0, size_t(n) * sizeof(double), T.stream));
  for (mi_ctx *m : T.members)
    HIPCHK(m, hipMemcpyAsync(T.d_gbuf + (m->slab.node_offset + m->slab.own_begin) * m->dim, m->vec(which) + m->own0,
                             size_t(m->own_n) * sizeof(double), hipMemcpyDeviceToDevice, T.stream));
  if ((rc = team_allreduce_buffer(T, T.d_gbuf, size_t(n))))
    return rc;
  HIPCHK(c, hipStreamSynchronize(T.stream));
  return gbuf_to_host(c, host);
}
int mi_vec_set(mi_ctx *c, int which, const double *host, int64_t n)
{
  Team &T = *c->team;
  if (which < 0 || which >= MI_V_COUNT || n != T.n_global)
    return fail(c, MI_EINVAL, "bad vector id or length");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (which == MI_V_NEWTON_UPDATE)
    T.members[0]->newton_update_is_zero = false;
  std::vector<double> rot;
  host = to_internal_order(T, host, rot);
  for (mi_ctx *m : T.members) // every slab takes its local range (ghost planes included) from the global array
    HIPCHK(m, hipMemcpy(m->vec(which), host + m->slab.node_offset * m->dim, size_t(m->n) * sizeof(double),
                        hipMemcpyHostToDevice));
  return MI_OK;
}

int mi_matrix_get_csr(mi_ctx *c, int64_t *rowptr, int32_t *col, double *val)
{
  if (c->team->size != 1)
    return fail(c, MI_EINVAL, "matrix export is only available on an undecomposed mesh");
  if (c->mf_fine)
    return fail(c, MI_EINVAL, "matrix export: the matrix-free fine level keeps no assembled tangent (\"fine_level\" 0)");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int           D = c->dim, DD = D * D;
  const mi::HostMesh &m = c->mesh;
  std::vector<double> bv(size_t(m.nvalblocks()) * DD);
  HIPCHK(c, hipMemcpy(bv.data(), c->d_vals, bv.size() * sizeof(double), hipMemcpyDeviceToHost));
  int64_t k = 0;
  for (int64_t nd = 0; nd < m.nnodes; ++nd)
    for (int i = 0; i < D; ++i)
      {
        rowptr[nd * D + i] = k;
        for (int32_t b = m.rowptr[size_t(nd)]; b < m.rowptr[size_t(nd) + 1]; ++b)
          for (int j = 0; j < D; ++j)
            {
              col[k] = m.colidx[size_t(b)] * D + j;
              val[k] = bv[size_t(m.valpos(nd, int(b - m.rowptr[size_t(nd)]))) * DD + i * D + j];
              ++k;
            }
      }
  rowptr[m.nnodes * D] = k;
  return MI_OK;
}

// the dim x dim diagonal block of every node of the current tangent, [n_nodes][dim * dim] row-major in the reference's node
// order: out of the assembled tangent, or -- matrix-free fine level -- what mf_diag formed from the point records
int mi_get_diagonal_blocks(mi_ctx *c, double *blocks)
{
  Team &T = *c->team;
  HIPCHK(c, hipSetDevice(c->device));
  if (int e = check_mf_tangent(c))
    return e;
  const int DD = c->dim * c->dim;
  if (T.nccl)
    return fail(c, MI_EINVAL, "mi_get_diagonal_blocks: one process only (single or emulated slabs)");
  std::fill(blocks, blocks + T.nnodes_global * DD, 0.0);
  for (mi_ctx *m : T.members)
    {
      const size_t        cnt = size_t(m->mesh.nnodes) * DD;
      std::vector<double> h(cnt);
      if (m->mf_fine)
        {
          HIPCHK(m, hipStreamSynchronize(m->stream));
          HIPCHK(m, hipMemcpy(h.data(), m->d_diag_blk, cnt * sizeof(double), hipMemcpyDeviceToHost));
        }
      else
        {
          double *tmp = nullptr;
          HIPCHK(m, hipMalloc((void **)&tmp, cnt * sizeof(double)));
          mi::launch_gather_diag_blocks(m->dim, m->d_vals, m->d_diagpos, tmp, m->mesh.nnodes, m->stream);
          const hipError_t e1 = hipStreamSynchronize(m->stream);
          const hipError_t e2 = hipMemcpy(h.data(), tmp, cnt * sizeof(double), hipMemcpyDeviceToHost);
          hipFree(tmp);
          HIPCHK(m, e1);
          HIPCHK(m, e2);
        }
      for (int64_t ln = m->slab.own_begin; ln < m->slab.own_end; ++ln)
        for (int k = 0; k < DD; ++k)
          blocks[T.ext_node(ln + m->slab.node_offset) * DD + k] = h[size_t(ln) * DD + k];
    }
  return MI_OK;
}

// y = K x through the device kernels (global arrays)
int mi_spmv(mi_ctx *c, const double *x_host, double *y_host)
{
  Team &T = *c->team;
  HIPCHK(c, hipSetDevice(c->device));
  if (int e = check_mf_tangent(c))
    return e;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<double> rot;
  x_host = to_internal_order(T, x_host, rot);
  // tests, "spmv_as_smoother" 2: y = M^-1 x, one V-cycle of the multigrid hierarchy as the last linear solve left it
  const bool vcycle = c->spmv_as_smoother == 2;
  if (vcycle && (!c->mg || T.members[0]->precond != 1))
    return fail(c, MI_EINVAL, "\"spmv_as_smoother\" 2 needs the multigrid preconditioner of a linear solve that has run");
  const int w_in = vcycle ? W_R : W_P, w_out = vcycle ? W_Z : W_Q;
  for (mi_ctx *m : T.members)
    HIPCHK(m, hipMemcpy(m->work(w_in), x_host + m->slab.node_offset * m->dim, size_t(m->n) * sizeof(double),
                        hipMemcpyHostToDevice));
  if (vcycle)
    {
      if (int e = mg_apply(T))
        return e;
    }
  else
    for (mi_ctx *m : T.members)
      enqueue_spmv(m, m->work(W_P), m->work(W_Q), nullptr, nullptr, nullptr, 0, m->spmv_as_smoother != 0);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (T.size == 1)
    {
      HIPCHK(c, hipMemcpy(y_host, c->work(w_out), size_t(c->n) * sizeof(double), hipMemcpyDeviceToHost));
      return MI_OK;
    }
  int rc = ensure_gbuf(T);
  if (rc)
    return rc;
  HIPCHK(c, hipMemsetAsync(T.d_gbuf, 0, size_t(T.n_global) * sizeof(double), T.stream));
  for (mi_ctx *m : T.members)
    HIPCHK(m, hipMemcpyAsync(T.d_gbuf + (m->slab.node_offset + m->slab.own_begin) * m->dim, m->work(w_out) + m->own0,
                             size_t(m->own_n) * sizeof(double), hipMemcpyDeviceToDevice, T.stream));
  if ((rc = team_allreduce_buffer(T, T.d_gbuf, size_t(T.n_global))))
    return rc;
  HIPCHK(c, hipStreamSynchronize(T.stream));
  return gbuf_to_host(c, y_host);
}

int mi_set_tuning(mi_ctx *c, const char *key, int value)
{
  const std::string k(key ? key : "");
  for (mi_ctx *m : c->team->members)
    {
      if (m->mf_fine && ((k == "smoother_operator" && value != 2) || (k == "mf_single_launch" && value != 1) ||
                         (k == "precond_storage" && value != 64) || (k == "solver_type" && value != 0) ||
                         (k == "spmv_variant" && value != 3 && value != 4) || (k == "element_tangents" && value != 2)))
        return fail(c, MI_EINVAL, "tuning '%s' %d needs the assembled fine level (\"fine_level\" 0)", k.c_str(), value);
      if (k == "mf_slots_cell_major" && value >= -1 && value <= 2) // A/B: -1 follows the smoother's quadrature
        {
          m->slots_cell_major = value;
          if (m->d_mf_dst)
            {
              HIPCHK(m, hipStreamSynchronize(m->stream));
              const int rc = build_slot_tables(m);
              if (rc)
                return rc;
            }
          continue;
        }
      if (k == "spmv_as_smoother" && value >= 0 && value <= 2) // tests: mi_spmv through the smoother's form of the operator (2: the V-cycle)
        {
          m->spmv_as_smoother = value;
          continue;
        }
      if (k == "smoother_quadrature" && (value == 3 || value == 4))
        {
          // (matters where the smoother multiplies matrix-free from point records: 3D Q2 meshes above 100 k nodes, or after
          // "element_tangents" 2 / "fine_level" 1; elsewhere the key is accepted and has no effect)
          const bool change = m->smoother_points != value;
          m->smoother_points = value;
          if (m->d_qrec && change) // the records exist already: the smoother's own, and the slot layout that goes with the rule
            {
              HIPCHK(m, hipStreamSynchronize(m->stream));
              int rc = alloc_records27(m);
              if (rc == MI_OK && m->slots_cell_major < 0)
                rc = build_slot_tables(m);
              if (rc)
                return rc;
            }
          m->qrec27_valid = false;
          m->ke_valid     = false; // takes effect with the next tangent
          m->mg_stale = m->mg_force = true;
          continue;
        }
      if (k == "asm_box_geometry" && (value == 0 || value == 1)) // assemble_q2sf on meshes of boxes: geometry from 1/h | the trilinear map
        {
          m->asm_box_geometry = value;
          continue;
        }
      if (k == "face_slots" && (value == 0 || value == 1)) // Neumann faces: one launch + gather (1) | colour by colour (0)
        {
          m->face_slots = value;
          continue;
        }
      if (k == "mf_point_slots" && (value == 0 || value == 1)) // matrix-free fine level: point pass in one launch (1) | eight colour launches
        {
          m->mf_point_slots = value;
          continue;
        }
      if (k == "mf_diag_lag" && (value == 0 || value == 1))
        {
          m->mf_diag_lag   = value;
          m->mf_diag_fresh = false;
          continue;
        }
      if (k == "fine_level" && (value == 0 || value == 1))
        {
          const int rc = set_fine_level(m, value);
          if (rc)
            return fail(c, rc, "%s", m->err.c_str());
          continue;
        }
      if (k == "solver_type" && (value == 0 || value == 1))
        {
          m->solver_direct = value;
          continue;
        }
      if (k == "spmv_variant" && (value == 1 || value == 3 || value == 4 || (value >= 11 && value <= 14)))
        m->spmv_variant = value;
      else if (k == "smoother_operator" && value >= 0 && value <= 2)
        {
          m->ebe = value;
          if (value && m->precond == 1)
            {
              const int rc = ensure_element_tangents(m);
              if (rc)
                return rc;
            }
        }
      else if (k == "element_tangents" && (value == 1 || value == 2)) // tests: keep them (1) / the quadrature-point
        {                                                                // records (2) whatever the size / preconditioner
          if (m->dim != 3 || m->degree != 2)
            return fail(c, MI_EINVAL, "element tangents exist for 3D Q2 meshes only");
          if (value == 1 && !m->d_ke)
            HIPCHK(m, hipMalloc((void **)&m->d_ke, size_t(m->mesh.ncells) * 9 * mi::EBE_NBLK * sizeof(double)));
          if (value == 2)
            {
              const int rq = alloc_point_records(m);
              if (rq)
                return rq;
            }
          if (!m->d_node_first)
            {
              const int rc = upload(m, &m->d_node_first, m->mesh.node_first);
              if (rc)
                return rc;
            }
          m->ebe      = value;
          m->ke_valid = false;
        }
      else if (k == "mg_scale_lmax_percent" && value >= 10 && value <= 400) // tests: spoil the eigenvalue estimates once
        {
          if (m == c->team->members[0])
            mg_scale_estimates(*c->team, 0.01 * value);
        }
      else if (k == "mf_single_launch" && (value == 0 || value == 1))
        m->mf_slots = value;
      else if (k == "xcd_remap" && (value == 0 || value == 1))
        m->xcd_remap = value;
      else if (k == "cell_lattice" && (value == 0 || value == 1)) // A/B: node ids by arithmetic (1, default) or from conn
        m->lat = value ? m->lat_built : mi::CellLattice{};
      else if (k == "sell_unroll" && value >= -2 && value <= 8 && value != 0)
        m->sell_unroll = value;
      else if (k == "spmv_grid" && value >= 1 && value + m->grid_spmv_bnd <= MAX_PART)
        {
          m->split_int = m->split_bnd = false; // an explicit grid means the one-wavefront-per-slice kernel
          if (m->mesh.sell_nslices_interior > 0)
            m->grid_spmv_int = value;
          else
            m->grid_spmv_bnd = value;
          m->grid_spmv = m->grid_spmv_int + m->grid_spmv_bnd;
        }
      else if (k == "sell_icol" && (value == 0 || value == 1))
        m->sell_icol = value;
      else if (k == "small_cg" && (value == 0 || value == 1))
        m->small_cg = value;
      else if (k == "halo_overlap" && (value == 0 || value == 1))
        c->team->overlap = value;
      else if (k == "asm_variant" && value >= 0 && value <= 9)
        {
#ifndef MI_EXPERIMENTS
          if (value >= 3 && value <= 8)
            return fail(c, MI_EINVAL, "asm_variant %d is an A/B instantiation of the experiments build (make EXPERIMENTS=1)", value);
#endif
          m->asm_variant = value;
        }
      else if (k == "asm_split" && value >= 0 && value <= 2) // 3D Q2 with point records: the fused kernel (0) | point pass + tangent
        {                                                     // from the records, all waves (1) / wave 0 (2): profiles/r06/asm_split_ab_n59.txt
#ifndef MI_EXPERIMENTS
          if (value != 0)
            return fail(c, MI_EINVAL, "asm_split %d is an A/B instantiation of the experiments build (make EXPERIMENTS=1)", value);
#endif
          m->asm_split = value;
        }
      else if (k == "mg_refresh_every" && value >= 1 && value <= 1000)
        m->mg_refresh_every = value;
      else if (k == "mg_lag" && (value == 0 || value == 1))
        m->mg_lag = value;
      else if (k == "cg_fused_dot" && (value == 0 || value == 1))
        m->cg_fused_dot = value;
      else if (k == "cg_warm_start" && value >= 0 && value <= 3)
        m->cg_warm_start = value;
      else if (k == "cg_operator" && (value == 0 || value == 1))
        m->cg_operator = value;
      else if (k == "correct_face_F" && (value == 0 || value == 1)) // default 0 reproduces :825-827
        m->correct_face_F = value;
      else if (k == "cg_r0_operator" && (value == 0 || value == 1))
        m->cg_r0_unassembled = value;
      else if (k == "cg_speculate" && (value == 0 || value == 1))
        m->cg_speculate = value;
      else if (k == "cg_single_reduction" && value >= -1 && value <= 1)
        m->cg_single_reduction = value;
      else if ((k == "mg_dist_nodes" && value >= -1) || (k == "mg_coarsest" && value >= 1 && value <= 64) ||
               (k == "mg_dense" && (value == 0 || value == 1)))
        {
          // shape of the multigrid hierarchy: node count from which a team's first coarsened level is cut into slabs, cells per
          // direction at which the coarsening stops, exact solve on the coarsest level.  An existing hierarchy is rebuilt.
          (k == "mg_dist_nodes" ? m->mg_dist_nodes : k == "mg_coarsest" ? m->mg_coarsest : m->mg_dense) = value;
          if (m->mg)
            {
              const int rc = mg_setup(m);
              if (rc)
                return fail(c, rc, "%s", m->err.c_str());
            }
        }
      else if (k == "cg_speculate_margin" && value >= 0 && value <= 16)
        m->cg_speculate_margin = value;
      else if (k == "halo_skip" && (value == 0 || value == 1))
        c->team->halo_skip = value;
      else if (k == "mf_halo_overlap" && (value == 0 || value == 1))
        c->team->mf_overlap = value;
      else if (k == "mg_restrict_fuse" && (value == 0 || value == 1))
        {
          const int rc = mg_set_restrict_fuse(m, value);
          if (rc)
            return fail(c, rc, "mg_restrict_fuse: no multigrid hierarchy (set \"precond\" 1 first)");
        }
      else if (k == "mg_fuse" && value >= 0 && value <= 2)
        {
          const int rc = mg_set_fuse(m, value);
          if (rc)
            return fail(c, rc, "mg_fuse needs the multigrid preconditioner");
        }
      else if (k == "smoother_precision" && (value == 64 || value == 32))
        {
          // 32: the matrix-free smoother product in fp32 (3D Q2 meshes with point records; takes effect with the next tangent)
          if (value == 32 && !m->d_qrec32 && m->dim == 3 && m->degree == 2)
            HIPCHK(m, hipMalloc((void **)&m->d_qrec32, size_t(m->mesh.ncells) * mi::MF_NREC * 64 * sizeof(float)));
          m->smoother_precision = value;
          m->qrec32_valid       = false;
        }
      else if (k == "precond_storage" && (value == 64 || value == 32))
        {
          const int rc = set_precond_storage(m, value);
          if (rc)
            return rc;
        }
      else if (k == "precond" && (value == 0 || value == 1))
        {
          m->precond = value;
          if (value == 1 && !m->mg)
            {
              const int rc = mg_setup(m);
              if (rc)
                return fail(c, rc, "%s", m->err.c_str());
            }
          if (value == 1)
            {
              const int rc = ensure_element_tangents(m);
              if (rc)
                return rc;
            }
        }
      else
        return fail(c, MI_EINVAL, "unknown tuning key '%s' or value %d out of range", k.c_str(), value);
    }
  return MI_OK;
}

int mi_get_tuning(mi_ctx *c, const char *key, int *value)
{
  const std::string k(key ? key : "");
  mi_ctx           *m = c->team->members[0];
  if (!value)
    return fail(c, MI_EINVAL, "null argument");
  if (k == "smoother_operator_active") // 1: the smoother's fine-level products run on the element tangents
    *value = (m->ebe && m->precond == 1 && m->precond_storage == 64 && ((m->ebe == 2 && m->d_qrec) || m->d_ke)) ?
               ((m->d_qrec && (m->ebe == 2 || !m->d_ke)) ? 2 : 1) :
               0;
  else if (k == "mf_single_launch")
    *value = (m->mf_slots && m->d_mf_yc) ? 1 : 0;
  else if (k == "fine_level")
    *value = m->mf_fine;
  else if (k == "mf_diag_lag")
    *value = m->mf_diag_lag;
  else if (k == "smoother_quadrature")
    *value = m->smoother_points;
  else if (k == "smoother_quadrature_active") // 3: the smoother's fine-level products run on the 27-point records of the current tangent
    *value = (m->smoother_points == 3 && m->qrec27_valid && element_form(m) == 2) ? 3 : 4;
  else if (k == "experiments") // 1: built with -DMI_EXPERIMENTS (environment hooks and A/B kernel instantiations compiled in)
#ifdef MI_EXPERIMENTS
    *value = 1;
#else
    *value = 0;
#endif
  else if (k == "sell_icol")
    *value = m->sell_icol;
  else if (k == "asm_variant")
    *value = m->asm_variant;
  else if (k == "cell_lattice")
    *value = m->lat.ncol > 0 ? 1 : 0;
  else if (k == "cut_axis") // the box direction the slabs are cut along: 1 / 2 / 3 = x / y / z, 0: not decomposed
    *value = c->team->size > 1 ? c->team->amap.ext_axis[c->team->dim - 1] + 1 : 0;
  else if (k == "cg_speculate")
    *value = m->cg_speculate;
  else if (k == "smoother_precision")
    *value = m->smoother_precision;
  else if (k == "cg_single_reduction")
    *value = m->cg_single_reduction;
  else if (k == "mg_distributed_levels") // levels of the multigrid hierarchy that are cut into slabs (0: no hierarchy)
    *value = mg_distributed_levels(m);
  else if (k == "cg_single_reduction_active") // what cg_run will do with a multigrid-preconditioned solve
    *value = (m->cg_single_reduction == 1 || (m->cg_single_reduction < 0 && c->team->size > 1)) ? 1 : 0;
  else if (k == "cg_speculate_margin")
    *value = m->cg_speculate_margin;
  else if (k == "halo_skip")
    *value = c->team->halo_skip;
  else if (k == "precond")
    *value = m->precond;
  else if (k == "spmv_variant")
    *value = m->spmv_variant;
  else if (k == "count_scalar_allreduce")
    *value = int(c->team->n_scalar_allreduce);
  else if (k == "count_scalar_allreduce_cg")
    *value = int(c->team->n_scalar_allreduce_cg);
  else if (k == "count_vector_allreduce")
    *value = int(c->team->n_vector_allreduce);
  else if (k == "count_halo_exchange")
    *value = int(c->team->n_halo);
  else if (k == "count_cg_host_sync")
    *value = int(c->team->n_cg_sync);
  else if (k == "count_cg_iterations")
    *value = int(c->team->n_cg_its);
  else if (k == "count_mg_refresh")
    *value = int(c->team->members[0]->n_mg_refresh);
  else if (k == "mg_refresh_every")
    *value = c->mg_refresh_every;
  else if (k == "count_cg_solves")
    *value = int(c->team->n_cg_solves);
  else
    return fail(c, MI_EINVAL, "unknown tuning key '%s'", k.c_str());
  return MI_OK;
}

int mi_set_profiling(mi_ctx *c, int enable)
{
  c->team->members[0]->profiling = enable != 0;
  return MI_OK;
}
int mi_reset_timings(mi_ctx *c)
{
  int rc = sync(c);
  Team &T = *c->team;
  T.n_scalar_allreduce = T.n_vector_allreduce = T.n_halo = T.n_cg_sync = T.n_cg_its = T.n_cg_solves = 0;
  T.n_scalar_allreduce_cg = 0;
  T.members[0]->n_mg_refresh = 0;
  std::memset(&c->team->members[0]->timings, 0, sizeof(mi_timings));
  return rc;
}
int mi_get_timings(mi_ctx *c, mi_timings *out)
{
  int rc = sync(c);
  *out   = c->team->members[0]->timings;
  return rc;
}

int mi_bench_spmv(mi_ctx *c, int reps, double *ms_per_launch)
{
  HIPCHK(c, hipSetDevice(c->device));
  hipEvent_t a, b;
  HIPCHK(c, hipEventCreate(&a));
  HIPCHK(c, hipEventCreate(&b));
  if (int e = check_mf_tangent(c))
    return e;
  auto once = [&]() {
    for (mi_ctx *m : c->team->members)
      {
        const bool plain = m->spmv_variant == 4 && !m->mf_fine; // the unassembled forms carry no fused dot product
        if (m->spmv_as_smoother) // (tests / tools: the smoother's form of the operator)
          enqueue_spmv(m, m->work(W_P), m->work(W_Q), nullptr, nullptr, nullptr, 0, true);
        else
          enqueue_spmv(m, m->work(W_P), m->work(W_Q), plain ? nullptr : m->work(W_P), plain ? nullptr : m->part(2), nullptr);
      }
  };
  once(); // warm-up
  HIPCHK(c, hipEventRecord(a, c->stream));
  for (int i = 0; i < reps; ++i)
    once();
  HIPCHK(c, hipEventRecord(b, c->stream));
  HIPCHK(c, hipEventSynchronize(b));
  float ms = 0;
  HIPCHK(c, hipEventElapsedTime(&ms, a, b));
  hipEventDestroy(a);
  hipEventDestroy(b);
  *ms_per_launch = double(ms) / std::max(1, reps);
  if (mi::exp_env("MI_MF_STAMPS") && c->spmv_variant == 4 && element_form(c) == 2 && c->mf_slots && c->d_mf_yc && c->d_cellbox)
    {
      // diagnostic: where a wavefront of the matrix-free product spends its life (shader-clock stamps of lane 0 at the
      // stage boundaries), averaged over all cells of one launch
      const int64_t       ncell = c->mesh.ncells;
      unsigned long long *d_st  = nullptr;
      HIPCHK(c, hipMalloc((void **)&d_st, size_t(ncell) * 8 * sizeof(unsigned long long)));
      HIPCHK(c, hipMemsetAsync(d_st, 0, size_t(ncell) * 8 * sizeof(unsigned long long), c->stream));
      mi::MfParams f{};
      f.qrec = c->d_qrec, f.conn = c->d_conn, f.first = c->d_node_first, f.cmask = c->d_cmask, f.vals = c->d_vals;
      f.diagpos = c->d_diagpos, f.tab1d = c->d_tab, f.cverts = c->d_cverts, f.mu = c->mat.mu, f.kappa = c->kappa;
      f.cellbox = c->d_cellbox, f.x = c->work(W_P), f.y = c->work(W_Q), f.mass = c->alpha[1] * c->mat.rho, f.lat = c->lat;
      f.yc = c->d_mf_yc, f.dst = c->d_mf_dst, f.slot_base = c->d_mf_slot_base, f.slot_src = c->d_mf_src, f.stamps = d_st;
      mi::launch_mf_spmv(f, 0, int32_t(ncell), c->stream);
      std::vector<unsigned long long> st(size_t(ncell) * 8);
      HIPCHK(c, hipStreamSynchronize(c->stream));
      HIPCHK(c, hipMemcpy(st.data(), d_st, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      hipFree(d_st);
      double             sum[5] = {0, 0, 0, 0, 0};
      unsigned long long tmin = ~0ull, tmax = 0;
      for (int64_t e = 0; e < ncell; ++e)
        {
          for (int i = 0; i < 5; ++i)
            sum[i] += double(st[size_t(e) * 8 + i + 1] - st[size_t(e) * 8 + i]);
          tmin = std::min(tmin, st[size_t(e) * 8]);
          tmax = std::max(tmax, st[size_t(e) * 8 + 5]);
        }
      const char *name[5] = {"start -> x gathered", "E1-E3 (gradients at the points)", "point stage (incl. wait for the records)",
                             "I3 + I2", "I1 + stores issued"};
      double      tot     = 0;
      for (int i = 0; i < 5; ++i)
        tot += sum[i];
      fprintf(stderr, "mf_spmv stages (%lld cells; launch spans %.0f ticks; %.0f ticks per wave):\n", (long long)ncell,
              double(tmax - tmin), tot / double(ncell));
      for (int i = 0; i < 5; ++i)
        fprintf(stderr, "  %-42s %8.0f ticks  %5.1f %%\n", name[i], sum[i] / double(ncell), 100.0 * sum[i] / tot);
    }
  return MI_OK;
}

int mi_bench_assemble(mi_ctx *c, int reps, double *ms_per_assembly)
{
  HIPCHK(c, hipSetDevice(c->device));
  hipEvent_t a, b;
  HIPCHK(c, hipEventCreate(&a));
  HIPCHK(c, hipEventCreate(&b));
  int  rc   = MI_OK;
  auto once = [&]() {
    for (mi_ctx *m : c->team->members)
      if (rc == MI_OK)
        rc = enqueue_assembly(m);
  };
  once(); // warm-up
  HIPCHK(c, hipEventRecord(a, c->stream));
  for (int i = 0; i < reps; ++i)
    once();
  if (rc)
    return rc;
  HIPCHK(c, hipEventRecord(b, c->stream));
  HIPCHK(c, hipEventSynchronize(b));
  float ms = 0;
  HIPCHK(c, hipEventElapsedTime(&ms, a, b));
  hipEventDestroy(a);
  hipEventDestroy(b);
  *ms_per_assembly = double(ms) / std::max(1, reps);
  if (mi::exp_env("MI_ASM_STAMPS") && !c->mf_fine && c->dim == 3 && c->degree == 2 && (c->asm_variant == 0 || (c->asm_variant >= 3 && c->asm_variant <= 8)))
    {
      // diagnostic: where a workgroup of the sum-factorised element kernel spends its life (shader-clock stamps of one
      // tangent wave at the phase boundaries), averaged over the cells of the first colour
      const int64_t       ncell = c->mesh.colour_begin[1] - c->mesh.colour_begin[0];
      unsigned long long *d_st  = nullptr;
      HIPCHK(c, hipMalloc((void **)&d_st, size_t(ncell) * 16 * sizeof(unsigned long long)));
      HIPCHK(c, hipMemsetAsync(d_st, 0, size_t(ncell) * 16 * sizeof(unsigned long long), c->stream));
      mi::AsmParams p = asm_params(c);
      p.cell_begin    = c->mesh.colour_begin[0];
      p.cell_count    = int32_t(ncell);
      p.stamps        = d_st;
      mi::launch_assemble_cells(c->dim, c->degree, p, c->stream);
      std::vector<unsigned long long> st(size_t(ncell) * 16);
      HIPCHK(c, hipStreamSynchronize(c->stream));
      HIPCHK(c, hipMemcpy(st.data(), d_st, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      hipFree(d_st);
      double sum[7] = {0, 0, 0, 0, 0, 0, 0}, pro[5] = {0, 0, 0, 0, 0};
      for (int64_t e = 0; e < ncell; ++e)
        {
          const unsigned long long *t = &st[size_t(e) * 16];
          for (int i = 0; i < 6; ++i)
            sum[i] += double(t[i + 1] - t[i]);
          for (int i = 0; i < 5; ++i) // stamps of the working waves inside the prologue, relative to the workgroup's start
            pro[i] += t[8 + i] ? double(t[8 + i] - t[0]) : 0.0;
        }
      const char *name[6] = {"prologue (wave 0; tangent waves wait)", "contractions", "wait at barrier (2)", "image",
                             "block table", "scatter"};
      double      tot     = 0;
      for (int i = 0; i < 6; ++i)
        tot += sum[i];
      fprintf(stderr, "assemble_q2sf phases, first colour (%lld cells):\n", (long long)ncell);
      for (int i = 0; i < 6; ++i)
        fprintf(stderr, "  %-40s %8.0f ticks  %5.1f %%\n", name[i], sum[i] / double(ncell), 100.0 * sum[i] / tot);
      fprintf(stderr, "  inside the prologue, ticks since the start: gradients %0.f, material %.0f, fields %.0f (wave 0); "
                      "row info %.0f, block table %.0f (wave 3)\n",
              pro[0] / double(ncell), pro[1] / double(ncell), pro[2] / double(ncell), pro[3] / double(ncell),
              pro[4] / double(ncell));
    }
  return sync(c);
}

} // extern "C"
