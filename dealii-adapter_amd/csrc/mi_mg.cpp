// mi_mg.cpp -- geometric multigrid V-cycle used as the CG preconditioner (the reference
// preconditions with SSOR, which is sequential; Jacobi needs ~450 iterations per solve at 5 M DoFs).
//
// Everything is built from pieces that already exist on the device:
//   * level 0 is the fine problem itself, distributed over the slabs of the team exactly like the CG (owned rows,
//     ghost planes by halo exchange);
//   * levels >= 1 are ordinary device contexts on coarser lattices: first p-coarsening to Q1 on the same cells,
//     then index-space coarsening of the cells by 2 down to a single cell.  The Q1 level on the same cells (8x
//     fewer dofs, 12x fewer non-zeros) shares the slab decomposition of the fine level -- same cell layers, same
//     ownership rule, halo exchange instead of sums -- while the levels on coarsened cells (another 8x smaller each)
//     live on the UNDECOMPOSED box and are REPLICATED on every slab/GPU: the only collective besides halos that a
//     V-cycle adds is one all-reduce of the first replicated level's restricted residual;
//   * coarse operators are RE-ASSEMBLED by the same element kernel at the current state (displacement
//     interpolated to the level), so no sparse triple products are needed and the Newmark mass term, the
//     Dirichlet rows and the material are treated exactly as on the fine level;
//   * smoothers are Chebyshev-Jacobi polynomials on the level's sliced-ELL SpMV, with the largest eigenvalue of
//     D^-1 A from a power iteration; the coarsest level (one cell) is "solved" by a longer polynomial;
//   * transfers are tensor-product linear interpolation in lattice index space and its transpose.
// The cycle is symmetric (same polynomial before and after the coarse correction), so it is a valid CG
// preconditioner, and it is the same operator for any number of slabs up to rounding.
#include <cmath>
#include <cstring>

#include "mi_internal.h"

namespace mi_detail
{
  struct MgTransfer
  {
    mi::LatticeParams prolong{}, restrict_{}, state{};
    std::vector<void *> dev; // device tables to free
  };

  struct MgLevel
  {
    mi_ctx *ctx  = nullptr; // level 0: the slab itself (not owned)
    Team   *team = nullptr; // levels >= 1 own a private single-slab team (shared stream)
    double *ws   = nullptr; // workspace: b, x, d, q, ev, r, x'  (7 local vectors; ev = running eigenvector estimate)
    bool    x_swapped = false; // the fused smoother step writes x' = x + d into the other of the two x buffers
    bool    ev_ready = false;
    double  lmax = 0.0;     // estimate of the largest eigenvalue of D^-1 A
    double  lam_power = 0.0, lam_boost = 1.0; // the power iteration's last value; what the Krylov estimate of the first
                                              // estimate added to it (estimate_lmax)
    double *dense_inv = nullptr; // coarsest level, n <= 96: the inverse of the level matrix (exact coarse solve in one launch)
    MgTransfer to_coarse;   // to level l+1
    // level 0 works on the CG's own vectors: its right-hand side IS the residual W_R (only owned entries are ever read)
    // and its first x buffer IS W_Z, so a V-cycle neither copies its input nor (unless an odd number of fused steps
    // left the result in the second buffer) its output
    double *b_ext = nullptr, *x_ext = nullptr;
    double *x_first() const { return x_ext ? x_ext : ws + ctx->n; }
    double *b() const { return b_ext ? b_ext : ws; }
    double *x() const { return x_swapped ? ws + 6 * ctx->n : x_first(); }
    double *x_other() const { return x_swapped ? x_first() : ws + 6 * ctx->n; }
    double *d() const { return ws + 2 * ctx->n; }
    double *q() const { return ws + 3 * ctx->n; }
    double *ev() const { return ws + 4 * ctx->n; }
    double *r() const { return ws + 5 * ctx->n; }
  };

  struct Multigrid
  {
    std::vector<MgLevel> levels;
    size_t n_dist        = 1;    // levels [0, n_dist) are distributed over the slabs of the team, the others replicated
    size_t n_aligned     = 1;    // levels [0, n_aligned) share the cuts of the fine level (same cells): transfers between them
                                 // stay inside a slab's local box.  A distributed level beyond (the first coarsened level of
                                 // a team whose slabs are big enough, round 5) has the cuts the finer level's cuts induce, and
                                 // its transfers go through partial results on ghost planes (team_halo_accumulate)
    int64_t dist_nodes   = 65536; // distribute the first coarsened level when it has at least this many nodes (and every
                                  // slab gets a cell layer of it); MI_MG_DIST_NODES / tuning "mg_dist_nodes"
    int    nu            = 3;    // Chebyshev degree of the pre- and post-smoother on the finest level
    int    nu_coarse     = 2;    // ... on the coarser levels
    int    nu_level1     = 0;    // ... on level 1 alone (0: nu_coarse; MI_MG_NU_L1, experiment of round 4)
    int    restrict_fuse = 1;    // the restriction takes the first smoother step of the coarse level (MI_MG_RESTRICT_FUSE=0: never)
    int    fuse          = 1;    // Chebyshev update / residual in the epilogue of the product (one launch instead of two):
                                 // 0 never, 1 on the latency-bound levels (<= fuse_max_nodes), 2 on every level
    int64_t fuse_max_nodes = 100000;
    int    three_term    = 1;    // level 0, matrix-free smoother: the Chebyshev step in three-term form (MI_MG_THREE_TERM=0: with d)
    int    block         = 1;    // 1: block-Jacobi diagonal (DxD node blocks) inside the Chebyshev smoother, 0: point Jacobi
    int    kind          = 1;    // smoother polynomial: 1 = Chebyshev 1st kind on [lmax/ratio, lmax], 4 = 4th kind, optimised
    double smooth_ratio  = 20.0; // smoother targets [lmax/ratio, lmax]
    int    coarse_degree = 12;   // polynomial degree on the coarsest level
    int    coarsest_reps = 4;    // stop coarsening once no direction has more cells than this (round 5: 4^dim cells, 375 dofs
                                 // in 3D, solved exactly by a dense inverse -- a level and its seven launches per V-cycle
                                 // fewer than with 2^dim cells (MI_MG_COARSEST=2: rounds 2-4); MI_MG_COARSEST=1 + MI_MG_DENSE=0:
                                 // round 1's one-cell level with a degree-12 polynomial)
    int    dense         = 1;    // exact solve on the coarsest level when it has <= 384 dofs
    int    coarsen_factor = 2;   // cells per direction shrink by this factor from level to level
    double coarse_ratio  = 60.0;
    int    power_its     = 15;   // first estimate
    int    krylov_its    = 40;   // ... and the steps of its Krylov companion (krylov_lmax)
    int    power_its_update = 2; // refresh, continuing from the previous eigenvector (4 change nothing, 2.5 ms per step)
    double lmax_safety   = 1.15;
  };

  static inline bool is_dist(const Team &T, size_t l)
  {
    return T.size > 1 && l < T.members[0]->mg->n_dist;
  }

  int mg_distributed_levels(const mi_ctx *c)
  {
    return c->mg ? (c->team->size > 1 ? int(c->mg->n_dist) : 0) : 0;
  }

  bool mg_active(const mi_ctx *c)
  {
    return c->precond == 1 && c->mg && c->mg->levels.size() > 1 && !c->active_sell_vals;
  }

  namespace
  {
    template <typename T>
    int to_device(mi_ctx *c, MgTransfer &t, const std::vector<T> &h, const T **out)
    {
      T *d = nullptr;
      HIPCHK(c, hipMalloc((void **)&d, std::max<size_t>(1, h.size()) * sizeof(T)));
      if (!h.empty())
        HIPCHK(c, hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
      t.dev.push_back(d);
      *out = d;
      return MI_OK;
    }

    // 1D linear interpolation table: for every target index t of a lattice with nt points, the source index
    // i0 and the weight w of i0+1 on a lattice with ns points covering the same interval
    void interp_table(int nt, int ns, std::vector<int32_t> &i0, std::vector<double> &w)
    {
      i0.resize((size_t)nt);
      w.resize((size_t)nt);
      for (int t = 0; t < nt; ++t)
        {
          const double s = (nt > 1) ? double(t) * double(ns - 1) / double(nt - 1) : 0.0;
          int          i = int(std::floor(s + 1e-12));
          if (i > ns - 2)
            i = std::max(0, ns - 2);
          double wt = s - i;
          if (std::fabs(wt) < 1e-12)
            wt = 0.0;
          if (std::fabs(wt - 1.0) < 1e-12)
            wt = 1.0;
          if (ns == 1)
            {
              i  = 0;
              wt = 0.0;
            }
          i0[size_t(t)] = i;
          w[size_t(t)]  = wt;
        }
    }

    // Would build_transfer accept the slabs of EVERY rank when the coarser level is cut along `induced`?  The decision to
    // distribute that level must be the same on all ranks (each builds its own hierarchy; a rank that fell back to the
    // replicated level while the others did not would pair up the wrong collectives), so it is taken from the cuts alone, for
    // all ranks, with build_transfer's own index arithmetic along the cut direction: every OWNED finer plane prolongates
    // from and restricts into the coarse slab's box, and a coarse node's state is served without a plane beyond the finer
    // slab.  (explicit Team::cuts or lattices that do not nest can break the cover; the level then stays
    // replicated instead of failing the multigrid setup.)
    bool induced_cuts_cover(int dim, const mi_mesh_desc &g, const int *fine_reps, const int *fine_cuts, const int *coarse_reps,
                            const std::vector<int> &induced, int size)
    {
      try
        {
          for (int r = 0; r < size; ++r)
            {
              const mi::SlabPartition F = mi::make_slab_partition(dim, 1, fine_reps, g.lo, g.hi, g.face_role, r, size, fine_cuts);
              const mi::SlabPartition C = mi::make_slab_partition(dim, 1, coarse_reps, g.lo, g.hi, g.face_role, r, size, induced.data());
              const int nf_loc = int(F.nnodes_local / F.plane_nodes), zoff = int(F.node_offset / F.plane_nodes),
                        nf = int(F.nnodes_global / F.plane_nodes), own_lo = int(F.own_begin / F.plane_nodes),
                        own_hi = int(F.own_end / F.plane_nodes);
              const int nc = int(C.nnodes_local / C.plane_nodes), coff = int(C.node_offset / C.plane_nodes),
                        nc_g = int(C.nnodes_global / C.plane_nodes);
              std::vector<int32_t> gi0;
              std::vector<double>  gw;
              interp_table(nf, nc_g, gi0, gw);
              for (int k = own_lo; k < own_hi; ++k)
                {
                  const int c0 = gi0[size_t(k + zoff)] - coff;
                  if (c0 < 0 || c0 + (gw[size_t(k + zoff)] != 0.0 ? 1 : 0) >= nc)
                    return false;
                }
              interp_table(nc_g, nf, gi0, gw);
              for (int t = 0; t < nc; ++t)
                {
                  const int kl = gi0[size_t(t + coff)] - zoff;
                  if (kl >= own_lo && kl < own_hi && kl + 1 >= nf_loc && gw[size_t(t + coff)] != 0.0)
                    return false;
                }
            }
        }
      catch (const std::exception &)
        {
          return false;
        }
      return true;
    }

    // Transfer tables between a fine context and a coarse GLOBAL context.  The fine context may be a slab: along the
    // decomposed (last) direction its local lattice index k corresponds to the global index k + zoff, and it owns
    // the local planes [own_lo, own_hi).  Ownership keeps sums over slabs free of double counting:
    //   restriction lists only contain owned fine planes; the state transfer serves a coarse node only on the slab
    //   that owns its left source plane; prolongation fills every local fine plane (ghosts included).
    // coarse_is_slab (with fine_is_slab): the coarse context is a slab of a distributed level whose cuts the fine level's
    // cuts induce -- coarse indices along the cut direction are local to ITS box (global index minus its plane offset), a
    // fine ghost plane whose prolongation would need a coarse plane outside that box is left alone (source index -1: the
    // halo exchange that follows brings its value), restriction and state lists keep the ownership rule: what lands on
    // a coarse GHOST plane is a partial result that team_halo_accumulate takes to its owner.
    int build_transfer(mi_ctx *fine, mi_ctx *coarse, MgTransfer &t, bool fine_is_slab, bool coarse_is_slab = false)
    {
      const int dim = fine->dim, zd = dim - 1;
      int       rc;
      t.prolong.n_tgt   = fine->mesh.nnodes;
      t.restrict_.n_tgt = coarse->mesh.nnodes;
      t.state.n_tgt     = coarse->mesh.nnodes;
      for (int d = 0; d < 3; ++d)
        {
          const int nf_loc = d < dim ? fine->mesh.nn[d] : 1, nc = d < dim ? coarse->mesh.nn[d] : 1;
          t.prolong.nt[d]   = nf_loc;
          t.prolong.ns[d]   = nc;
          t.restrict_.nt[d] = nc;
          t.restrict_.ns[d] = nf_loc;
          t.state.nt[d]     = nc;
          t.state.ns[d]     = nf_loc;
          if (d >= dim)
            continue;
          const bool cut  = fine_is_slab && d == zd;
          const int  zoff = cut ? int(fine->slab.node_offset / fine->slab.plane_nodes) : 0;
          const int  nf   = cut ? int(fine->slab.nnodes_global / fine->slab.plane_nodes) : nf_loc; // global extent
          const int  own_lo = cut ? int(fine->slab.own_begin / fine->slab.plane_nodes) : 0;
          const int  own_hi = cut ? int(fine->slab.own_end / fine->slab.plane_nodes) : nf_loc; // local, exclusive
          // the coarse side: nc_g planes globally, this context holds planes [coff, coff + nc)
          const bool ccut = cut && coarse_is_slab;
          const int  coff = ccut ? int(coarse->slab.node_offset / coarse->slab.plane_nodes) : 0;
          const int  nc_g = ccut ? int(coarse->slab.nnodes_global / coarse->slab.plane_nodes) : nc;
          std::vector<int32_t> gi0;
          std::vector<double>  gw;
          interp_table(nf, nc_g, gi0, gw); // fine (global index) <- coarse (global index)
          // prolongation: every local fine index
          std::vector<int32_t> i0((size_t)nf_loc);
          std::vector<double>  w((size_t)nf_loc);
          for (int k = 0; k < nf_loc; ++k)
            {
              i0[size_t(k)] = gi0[size_t(k + zoff)] - coff;
              w[size_t(k)]  = gw[size_t(k + zoff)];
              if (ccut && (i0[size_t(k)] < 0 || i0[size_t(k)] + (w[size_t(k)] != 0.0 ? 1 : 0) >= nc))
                {
                  if (k >= own_lo && k < own_hi)
                    return fail(fine, MI_EINVAL, "multigrid: prolongation of an owned plane needs a coarse plane outside the slab");
                  i0[size_t(k)] = -1; // a ghost plane: left to the halo exchange
                  w[size_t(k)]  = 0.0;
                }
            }
          if ((rc = to_device(fine, t, i0, &t.prolong.i0[d])) || (rc = to_device(fine, t, w, &t.prolong.w[d])))
            return rc;
          // restriction = transpose over the OWNED fine planes
          std::vector<std::vector<std::pair<int32_t, double>>> lists((size_t)nc);
          for (int k = own_lo; k < own_hi; ++k)
            {
              const int    c0 = gi0[size_t(k + zoff)] - coff;
              const double w1 = gw[size_t(k + zoff)];
              if (c0 < 0 || c0 + (w1 != 0.0 ? 1 : 0) >= nc)
                return fail(fine, MI_EINVAL, "multigrid: restriction of an owned plane lands outside the coarse slab");
              if (1.0 - w1 != 0.0)
                lists[size_t(c0)].push_back({k, 1.0 - w1});
              if (w1 != 0.0)
                lists[size_t(c0 + 1)].push_back({k, w1});
            }
          std::vector<int32_t> rs(1, 0), ri;
          std::vector<double>  rw;
          for (const auto &l : lists)
            {
              t.restrict_.rmax = std::max(t.restrict_.rmax, int32_t(l.size()));
              for (const auto &e : l)
                {
                  ri.push_back(e.first);
                  rw.push_back(e.second);
                }
              rs.push_back(int32_t(ri.size()));
            }
          if ((rc = to_device(fine, t, rs, &t.restrict_.rstart[d])) || (rc = to_device(fine, t, ri, &t.restrict_.ri[d])) ||
              (rc = to_device(fine, t, rw, &t.restrict_.rw[d])))
            return rc;
          // state transfer: coarse target <- fine source, served by the slab that owns the left source plane
          interp_table(nc_g, nf, gi0, gw);
          std::vector<int32_t> si0((size_t)nc);
          std::vector<double>  sw((size_t)nc);
          for (int tI = 0; tI < nc; ++tI)
            {
              const int kl = gi0[size_t(tI + coff)] - zoff; // local index of the left source plane
              if (kl >= own_lo && kl < own_hi)
                {
                  si0[size_t(tI)] = kl;
                  sw[size_t(tI)]  = gw[size_t(tI + coff)];
                  if (kl + 1 >= nf_loc && gw[size_t(tI + coff)] != 0.0)
                    return fail(fine, MI_EINVAL, "multigrid: state transfer needs a plane outside the slab");
                }
              else
                {
                  si0[size_t(tI)] = -1;
                  sw[size_t(tI)]  = 0.0;
                }
            }
          if ((rc = to_device(fine, t, si0, &t.state.i0[d])) || (rc = to_device(fine, t, sw, &t.state.w[d])))
            return rc;
        }
      return MI_OK;
    }

    // q = A_l x on level l of every slab: distributed levels exchange the ghost planes of x (overlapped with the
    // interior rows), the coarser levels are replicated
    // ghosts_current: the ghost planes of x are up to date on every slab -- no exchange (see vcycle)
    int level_spmv(Team &T, size_t l, const std::function<double *(mi_ctx *)> &x_of, const ChebFusion *cheb = nullptr,
                   bool ghosts_current = false)
    {
      const int t  = (l == 0) ? tic(T.members[0], MI_T_SPMV_PRECOND) : -1; // fine-level products, for the byte accounting
      int       rc = MI_OK;
      if (is_dist(T, l))
        rc = team_spmv(
          T, [l](mi_ctx *m) { return m->mg->levels[l].ctx; }, x_of, [l](mi_ctx *m) { return m->mg->levels[l].q(); },
          nullptr, true, cheb, ghosts_current);
      else
        for (size_t k = 0; k < T.members.size(); ++k)
          {
            mi_ctx  *m = T.members[k];
            MgLevel &L = m->mg->levels[l];
            enqueue_spmv(L.ctx, x_of(m), L.q(), nullptr, nullptr, nullptr, 0, true, cheb ? &cheb[k] : nullptr);
          }
      toc(T.members[0], t);
      return rc;
    }

    // largest eigenvalue of the symmetric tridiagonal matrix (diag, off): bisection on the Sturm count below the Gershgorin bound
    double tridiag_max_eig(const std::vector<double> &dg, const std::vector<double> &of)
    {
      const size_t m = dg.size();
      if (m == 0)
        return 0.0;
      double lo = dg[0], hi = dg[0];
      for (size_t i = 0; i < m; ++i)
        {
          const double rad = (i ? std::fabs(of[i - 1]) : 0.0) + (i + 1 < m ? std::fabs(of[i]) : 0.0);
          lo               = std::min(lo, dg[i] - rad);
          hi               = std::max(hi, dg[i] + rad);
        }
      for (int it = 0; it < 200 && hi - lo > 1e-13 * std::max(1.0, std::fabs(hi)); ++it)
        {
          const double x = 0.5 * (lo + hi);
          size_t       below = 0; // eigenvalues < x = negative pivots of T - x I
          double       d = 1.0;
          for (size_t i = 0; i < m; ++i)
            {
              d = dg[i] - x - (i ? of[i - 1] * of[i - 1] / d : 0.0);
              if (d == 0.0)
                d = 1e-300;
              below += d < 0.0;
            }
          (below == m ? hi : lo) = x;
        }
      return 0.5 * (lo + hi);
    }

    // Krylov estimate of lambda_max(D^-1 A) on level l: `steps` iterations of the conjugate-gradient recurrence for A with
    // D^-1 as its preconditioner from the start vector in ev(); their coefficients are the Lanczos tridiagonal matrix of
    // D^-1/2 A D^-1/2, whose largest eigenvalue (a Ritz value: never above lambda_max) closes in on the top of the spectrum
    // in a few dozen products where the power iteration can rest on a plateau for as many (a start vector holds next to
    // nothing of eigenvectors that live in the corners of the mesh).  Uses the level's r, d, q and second x buffers.
    int krylov_lmax(Team &T, size_t l, int steps, double *ritz_max)
    {
      mi_ctx    *c0   = T.members[0];
      Multigrid &mg0  = *c0->mg;
      const bool dist = is_dist(T, l);
      int        rc;
      auto vec_r = [l](mi_ctx *m) { return m->mg->levels[l].r(); };
      auto vec_p = [l](mi_ctx *m) { MgLevel &L = m->mg->levels[l]; return L.ws + 6 * L.ctx->n; };
      auto precondition = [&]() { // z (in d) = D^-1 r
        for (mi_ctx *m : T.members)
          {
            MgLevel &L = m->mg->levels[l];
            mi_ctx  *c = L.ctx;
            if (mg0.block)
              mi::launch_blk_apply(c->dim, L.d(), L.r(), c->d_dinv_blk, c->mesh.nnodes, c->stream);
            else
              mi::launch_vec_scale_mul(L.d(), L.r(), c->work(W_DINV), 1.0, c->n, c->stream);
          }
      };
      auto dot = [&](const std::function<double *(mi_ctx *)> &a, const std::function<double *(mi_ctx *)> &b, double *out) -> int {
        for (mi_ctx *m : T.members)
          {
            mi_ctx *c = m->mg->levels[l].ctx;
            mi::launch_dot_partials(a(m) + c->own0, b(m) + c->own0, c->own_n, m->part(5), m->grid_vec, c->stream);
            mi::launch_finish_sum(m->part(5), m->grid_vec, m->d_sc + 14, c->stream);
          }
        int e;
        if (dist && (e = team_allreduce(T, 14, 1)))
          return e;
        HIPCHK(c0, hipMemcpyAsync(c0->h_pinned, c0->d_sc + 14, sizeof(double), hipMemcpyDeviceToHost, c0->stream));
        HIPCHK(c0, hipStreamSynchronize(c0->stream));
        *out = c0->h_pinned[0];
        return MI_OK;
      };
      auto vec_z = [l](mi_ctx *m) { return m->mg->levels[l].d(); };
      auto vec_q = [l](mi_ctx *m) { return m->mg->levels[l].q(); };
      for (mi_ctx *m : T.members) // r = the start vector, p = z = D^-1 r
        {
          MgLevel &L = m->mg->levels[l];
          HIPCHK(m, hipMemcpyAsync(L.r(), L.ev(), size_t(L.ctx->n) * sizeof(double), hipMemcpyDeviceToDevice, L.ctx->stream));
        }
      precondition();
      for (mi_ctx *m : T.members)
        {
          MgLevel &L = m->mg->levels[l];
          HIPCHK(m, hipMemcpyAsync(vec_p(m), L.d(), size_t(L.ctx->n) * sizeof(double), hipMemcpyDeviceToDevice, L.ctx->stream));
        }
      double rz = 0.0;
      if ((rc = dot(vec_r, vec_z, &rz)))
        return rc;
      std::vector<double> dg, of;
      double              alpha_prev = 0.0, beta_prev = 0.0;
      *ritz_max = 0.0;
      for (int it = 0; it < steps && rz > 0.0 && std::isfinite(rz); ++it)
        {
          if ((rc = level_spmv(T, l, vec_p))) // q = A p (the ghost planes of p are exchanged on distributed levels)
            return rc;
          double pq = 0.0;
          if ((rc = dot(vec_p, vec_q, &pq)))
            return rc;
          if (!(pq > 0.0) || !std::isfinite(pq))
            break;
          const double alpha = rz / pq;
          for (mi_ctx *m : T.members)
            {
              MgLevel &L = m->mg->levels[l];
              mi::launch_vec_lincomb2(L.r(), 1.0, L.r(), -alpha, L.q(), L.ctx->n, L.ctx->stream); // r -= alpha q
            }
          precondition();
          double rz_new = 0.0;
          if ((rc = dot(vec_r, vec_z, &rz_new)))
            return rc;
          dg.push_back(1.0 / alpha + (it ? beta_prev / alpha_prev : 0.0));
          if (it)
            of.push_back(std::sqrt(beta_prev) / alpha_prev);
          const double beta = rz_new / rz;
          if (!(beta >= 0.0) || !std::isfinite(beta))
            break;
          for (mi_ctx *m : T.members)
            {
              MgLevel &L = m->mg->levels[l];
              mi::launch_vec_lincomb2(vec_p(m), 1.0, L.d(), beta, vec_p(m), L.ctx->n, L.ctx->stream); // p = z + beta p
            }
          alpha_prev = alpha, beta_prev = beta, rz = rz_new;
          if (rz_new <= 1e-28 * std::fabs(dg[0])) // the Krylov space is exhausted (tiny levels)
            break;
        }
      if (of.size() + 1 > dg.size() && !dg.empty())
        of.resize(dg.size() - 1);
      *ritz_max = tridiag_max_eig(dg, of);
      HIPCHK(c0, hipGetLastError());
      return MI_OK;
    }

    // power iteration for lambda_max(D^-1 A); level 0 is distributed over the team, the others are replicated
    int estimate_lmax(Team &T, size_t l)
    {
      mi_ctx    *c0  = T.members[0];
      Multigrid &mg0 = *c0->mg;
      const bool dist = is_dist(T, l);
      int        its  = mg0.power_its_update;
      for (mi_ctx *m : T.members)
        {
          MgLevel      &L = m->mg->levels[l];
          mi_ctx       *c = L.ctx;
          const int64_t n = c->n;
          if (!L.ev_ready)
            {
              // deterministic start vector with all frequencies (a function of the GLOBAL dof index, so that slabs
              // agree on their ghost copies): v_i = 1 + 0.5 sin(i) on unconstrained dofs
              std::vector<double> h((size_t)n, 0.0);
              const int64_t       g0 = c->slab.node_offset * c->dim;
              for (int64_t i = 0; i < n; ++i)
                h[size_t(i)] =
                  ((c->mesh.cmask[size_t(i / c->dim)] >> int(i % c->dim)) & 1) ? 0.0 : 1.0 + 0.5 * std::sin(double(i + g0));
              HIPCHK(c, hipMemcpyAsync(L.ev(), h.data(), size_t(n) * sizeof(double), hipMemcpyHostToDevice, c->stream));
              HIPCHK(c, hipStreamSynchronize(c->stream)); // h goes out of scope
              L.ev_ready = true;
              its        = -1; // first estimate: iterate until the estimate has settled (below)
            }
        }
      double lam = 0.0;
      int    rc;
      auto   ev_of = [l](mi_ctx *m) { return m->mg->levels[l].ev(); };
      // The estimate |D^-1 A v| of a normalised v grows monotonically towards lambda_max.  A FIRST estimate runs until
      // three consecutive iterations each add less than 0.1 % (at least power_its, at most 300): a fixed count of 15 was
      // enough up to 28 M dofs but stopped 20 % short at 42 M (the top of the spectrum is a cluster that the iteration
      // enters late), and a Chebyshev smoother built on an interval that ends below lambda_max amplifies the modes
      // above it -- the preconditioned CG then crawls (found in round 2 on the 120^3 mesh).  Refreshes continue from
      // the previous eigenvector with power_its_update iterations.
      const bool first = its < 0;
      double     ritz  = 0.0;
      if (first)
        {
          its = 300;
          // (round 6) the first estimate also asks the Krylov space of the same start vector: the power iteration below can
          // settle on a plateau under lambda_max -- 2.57 against 3.13 on a 32 x 32 x 28 mesh cut into two slabs, found by a
          // sweep over random meshes: a Chebyshev interval that ends below lambda_max makes the V-cycle indefinite, the PCG
          // took 46-136 instead of 11 iterations
          if ((rc = krylov_lmax(T, l, mg0.krylov_its, &ritz)))
            return rc;
          if (mi::exp_env("MI_MG_VERBOSE"))
            fprintf(stderr, "mg level %d Krylov estimate: %.6f\n", int(l), ritz);
        }
      double prev = 0.0;
      int    calm = 0;
      for (int it = 0; it < its; ++it)
        {
          if ((rc = level_spmv(T, l, ev_of)))
            return rc;
          for (mi_ctx *m : T.members)
            {
              MgLevel &L = m->mg->levels[l];
              mi_ctx  *c = L.ctx;
              if (mg0.block)
                mi::launch_blk_apply(c->dim, L.d(), L.q(), c->d_dinv_blk, c->mesh.nnodes, c->stream);
              else
                mi::launch_vec_scale_mul(L.d(), L.q(), c->work(W_DINV), 1.0, c->n, c->stream); // w = D^-1 A v
              // |w|^2 over the owned dofs of the level -> scalar slot 14 of the slab
              mi::launch_masked_norm(c->dim, L.d() + c->own0, c->d_cmask + c->slab.own_begin, c->own_n, m->part(5),
                                     m->grid_vec, m->d_sc + 14, c->stream);
            }
          if (dist && (rc = team_allreduce(T, 14, 1)))
            return rc;
          HIPCHK(c0, hipMemcpyAsync(c0->h_pinned, c0->d_sc + 14, sizeof(double), hipMemcpyDeviceToHost, c0->stream));
          HIPCHK(c0, hipStreamSynchronize(c0->stream));
          const double nw = std::sqrt(c0->h_pinned[0]); // |D^-1 A v|, equal to lambda once |v| = 1
          if (nw == 0.0)
            {
              // no unconstrained dof on this level (e.g. a one-cell coarse mesh between two clamped sides): the
              // operator is its own diagonal there, D^-1 A = I
              lam = 1.0;
              break;
            }
          if (!(nw > 0.0) || !std::isfinite(nw))
            return fail(c0, MI_EINVAL, "multigrid: power iteration broke down on level %d", int(l));
          lam = nw;
          if (mi::exp_env("MI_MG_VERBOSE"))
            fprintf(stderr, "mg level %d power iteration %d: |D^-1 A v| = %.6f\n", int(l), it, nw);
          if (first && it >= 1) // iteration 0 only normalises the start vector
            {
              calm = (nw <= prev * 1.001) ? calm + 1 : 0;
              if (calm >= 3 && it >= mg0.power_its)
                its = it + 1; // settled: this was the last iteration (v is still normalised below)
            }
          prev = nw;
          for (mi_ctx *m : T.members)
            {
              MgLevel &L = m->mg->levels[l];
              mi::launch_vec_scale_mul(L.ev(), L.d(), nullptr, 1.0 / nw, L.ctx->n, L.ctx->stream);
            }
        }
      HIPCHK(c0, hipGetLastError());
      // what the Krylov estimate added to the power iteration's value is kept, as a factor, over the refreshes (two iterations
      // from the previous eigenvector each).  It is never given back: a power estimate that grows between two refreshes may
      // have left its plateau or may follow a tangent that stiffens -- the two cannot be told apart, and an interval a
      // little too long costs an iteration where one too short costs the positive definiteness of the cycle
      MgLevel     &Lc    = c0->mg->levels[l];
      const double boost = first ? std::max(1.0, lam > 0.0 ? ritz / lam : 1.0) : std::max(1.0, Lc.lam_boost);
      for (mi_ctx *m : T.members)
        {
          MgLevel &L  = m->mg->levels[l];
          L.lam_power = lam;
          L.lam_boost = boost;
          L.lmax      = lam * boost * mg0.lmax_safety;
        }
      return MI_OK;
    }
  } // namespace

  // forget the eigenvalue estimates of every level: the next operator update estimates them from scratch
  void mg_reset_estimates(Team &T)
  {
    for (mi_ctx *m : T.members)
      if (m->mg)
        {
          for (MgLevel &L : m->mg->levels)
            L.ev_ready = false;
          m->mg_stale = m->mg_force = true;
        }
  }

  // tests: scale the current estimates (a factor < 1 reproduces a smoother interval that ends below lambda_max)
  void mg_scale_estimates(Team &T, double f)
  {
    for (mi_ctx *m : T.members)
      if (m->mg)
        for (MgLevel &L : m->mg->levels)
          L.lmax *= f;
  }

  int mg_set_storage(mi_ctx *c, int bits)
  {
    if (!c->mg)
      return MI_OK;
    for (size_t l = 1; l < c->mg->levels.size(); ++l)
      {
        const int rc = set_precond_storage(c->mg->levels[l].ctx, bits);
        if (rc)
          return rc;
      }
    return MI_OK;
  }

  // 0 never fuse the smoother update into the product, 1 on the latency-bound levels (default), 2 on every level
  int mg_set_fuse(mi_ctx *c, int fuse)
  {
    if (!c->mg)
      return MI_EINVAL;
    c->mg->fuse = fuse;
    return MI_OK;
  }

  // 1 (default): the restriction takes the first smoother step of the coarse level; 0: a launch of its own (A/B, same bits)
  int mg_set_restrict_fuse(mi_ctx *c, int on)
  {
    if (!c->mg)
      return MI_EINVAL;
    c->mg->restrict_fuse = on;
    return MI_OK;
  }

  void mg_destroy(mi_ctx *c)
  {
    if (!c->mg)
      return;
    for (size_t l = 0; l < c->mg->levels.size(); ++l)
      {
        MgLevel &L = c->mg->levels[l];
        for (void *p : L.to_coarse.dev)
          hipFree(p);
        if (L.ws)
          hipFree(L.ws);
        if (L.dense_inv)
          hipFree(L.dense_inv);
        if (l > 0 && L.team)
          destroy_team(L.team); // destroys the level context; the stream is shared and stays
      }
    delete c->mg;
    c->mg = nullptr;
  }

  int mg_setup(mi_ctx *c)
  {
    mg_destroy(c);
    Multigrid *mg = new Multigrid;
    c->mg         = mg;
    if (const char *e = mi::exp_env("MI_MG_NU"))
      mg->nu = std::max(1, atoi(e));
    if (const char *e = mi::exp_env("MI_MG_NU_COARSE"))
      mg->nu_coarse = std::max(1, atoi(e));
    if (const char *e = mi::exp_env("MI_MG_NU_L1"))
      mg->nu_level1 = std::max(0, atoi(e));
    if (const char *e = mi::exp_env("MI_MG_FUSE"))
      mg->fuse = std::min(2, std::max(0, atoi(e)));
    if (const char *e = mi::exp_env("MI_MG_FUSE_MAX_NODES"))
      mg->fuse_max_nodes = std::max(0, atoi(e));
    if (const char *e = mi::exp_env("MI_MG_THREE_TERM"))
      mg->three_term = atoi(e) != 0;
    if (const char *e = mi::exp_env("MI_MG_BLOCK"))
      mg->block = atoi(e) != 0;
    if (const char *e = mi::exp_env("MI_MG_KIND"))
      mg->kind = atoi(e) == 4 ? 4 : 1;
    if (const char *e = mi::exp_env("MI_MG_RATIO"))
      mg->smooth_ratio = std::max(2.0, atof(e));
    if (const char *e = mi::exp_env("MI_MG_COARSEST"))
      mg->coarsest_reps = std::max(1, atoi(e));
    if (const char *e = mi::exp_env("MI_MG_DENSE"))
      mg->dense = atoi(e) != 0;
    if (const char *e = mi::exp_env("MI_MG_SAFETY"))
      mg->lmax_safety = std::max(1.0, atof(e));
    if (const char *e = mi::exp_env("MI_MG_POWER_ITS"))
      mg->power_its_update = std::max(1, atoi(e));
    if (const char *e = mi::exp_env("MI_MG_FACTOR"))
      mg->coarsen_factor = std::max(2, atoi(e));
    if (const char *e = mi::exp_env("MI_MG_COARSE_DEGREE"))
      mg->coarse_degree = std::max(1, atoi(e));
    if (const char *e = mi::exp_env("MI_MG_COARSE_RATIO"))
      mg->coarse_ratio = std::max(2.0, atof(e));
    if (const char *e = mi::exp_env("MI_MG_RESTRICT_FUSE"))
      mg->restrict_fuse = atoi(e) != 0;
    if (const char *e = mi::exp_env("MI_MG_DIST_NODES"))
      mg->dist_nodes = std::max<int64_t>(0, atoll(e));
    if (c->mg_dist_nodes >= 0)
      mg->dist_nodes = c->mg_dist_nodes;
    if (c->mg_coarsest >= 1)
      mg->coarsest_reps = int(c->mg_coarsest);
    if (c->mg_dense >= 0)
      mg->dense = int(c->mg_dense);
    MgLevel L0;
    L0.ctx = c;
    mg->levels.push_back(L0);
    // level hierarchy over the UNDECOMPOSED box
    const mi_mesh_desc &g = c->team->md;
    int       p = c->degree, reps[3] = {g.reps[0], g.reps[1], g.reps[2]};
    const int dim = c->dim;
    for (int guard = 0; guard < 24; ++guard)
      {
        const int prev_layers = reps[dim - 1];
        const int prev_reps[3] = {reps[0], reps[1], reps[2]};
        bool      coarsened   = false;
        if (p > 1)
          p = 1;
        else
          {
            coarsened = true;
            // the hierarchy ends once NO direction has more than coarsest_reps cells; until then every direction with more
            // than floor_reps cells is coarsened (a thin direction keeps coarsening beside the long ones: the cells of a
            // plate-like mesh stay as isotropic as the coarsening factor allows)
            bool more = false;
            for (int d = 0; d < dim; ++d)
              more = more || reps[d] > mg->coarsest_reps;
            if (!more)
              break;
            const int floor_reps = std::min(2, mg->coarsest_reps);
            for (int d = 0; d < dim; ++d)
              if (reps[d] > floor_reps)
                reps[d] = std::max(floor_reps, (reps[d] + mg->coarsen_factor - 1) / mg->coarsen_factor);
          }
        mi_mesh_desc md = g;
        md.degree       = p;
        for (int d = 0; d < 3; ++d)
          md.reps[d] = d < dim ? reps[d] : 1;
        md.vertex_perturbation = nullptr;
        // the Q1 level on the same cells takes over the slab decomposition of the fine level
        const bool slab_level = c->team->size > 1 && mg->levels.size() == 1 && c->degree > 1;
        // The first COARSENED level of a team (round 5): replicated on every slab while it is small -- a latency-bound level
        // gains nothing from being cut --, distributed once it has dist_nodes nodes (weak scaling: the levels are levels of
        // the GLOBAL box and grow with the team).  Its cuts are the ones the finer level's cuts induce: layer boundary
        // floor(F[r] Lc / Lf) for the finer boundary F[r], so that every owned finer plane restricts into the slab's own box
        // and every owned finer plane is prolongated from it (build_transfer, coarse_is_slab).
        std::vector<int> induced;
        if (c->team->size > 1 && coarsened && mg->levels.size() == mg->n_aligned && mg->n_dist == mg->n_aligned)
          {
            int64_t nodes = 1;
            for (int d = 0; d < dim; ++d)
              nodes *= int64_t(p) * reps[d] + 1;
            const int size = c->team->size, Lf = prev_layers, Lc = reps[dim - 1];
            bool      ok   = nodes >= mg->dist_nodes;
            // (never the coarsest level: a level cut into slabs has no matrix to invert on any one slab and would end the
            // hierarchy with a polynomial instead of the exact solve)
            bool more_after = false;
            for (int d = 0; d < dim; ++d)
              more_after = more_after || reps[d] > mg->coarsest_reps;
            ok = ok && more_after;
            for (int r = 0; r <= size && ok; ++r)
              {
                const int F = c->team->cuts.empty() ? int((int64_t(Lf) * r) / size) : c->team->cuts[size_t(r)];
                induced.push_back(r == size ? Lc : int((int64_t(F) * Lc) / Lf));
                ok = r == 0 || induced[size_t(r)] > induced[size_t(r - 1)];
              }
            if (ok) // (the same answer on every rank: from the cuts alone, for all ranks)
              ok = induced_cuts_cover(dim, g, prev_reps, c->team->cuts.empty() ? nullptr : c->team->cuts.data(), reps, induced, size);
            if (!ok)
              induced.clear();
          }
        const bool dist_level = slab_level || !induced.empty();
        Team *T         = new Team;
        T->size         = dist_level ? c->team->size : 1;
        T->cuts         = induced;
        T->device       = c->device;
        T->dim          = dim;
        T->stream       = c->stream;
        T->owns_stream  = false;
        T->md           = md;
        T->amap         = c->team->amap; // the levels' lattices lie over the box like the fine one (geometry only: their
                                         // node ids never leave the library, so no permutation tables)
        T->iface_global = mi::global_interface_nodes(dim, p, md.reps, md.face_role);
        mi_ctx   *lc    = nullptr;
        const int rc    = create_member(*T, &md, &c->mat, &c->nm, dist_level ? c->slab.rank : 0, &lc);
        if (slab_level)
          mg->n_dist = mg->n_aligned = 2;
        else if (!induced.empty())
          mg->n_dist = mg->n_aligned + 1;
        T->members.push_back(lc);
        if (rc != MI_OK)
          {
            c->err = lc ? lc->err : "multigrid level creation failed";
            destroy_team(T);
            return rc;
          }
        lc->precond = 0; // levels are never preconditioned themselves
        MgLevel L;
        L.ctx  = lc;
        L.team = T;
        mg->levels.push_back(L);
      }
    for (size_t l = 0; l < mg->levels.size(); ++l)
      {
        MgLevel &L = mg->levels[l];
        HIPCHK(c, hipMalloc((void **)&L.ws, size_t(7) * size_t(L.ctx->n) * sizeof(double)));
        HIPCHK(c, hipMemsetAsync(L.ws, 0, size_t(7) * size_t(L.ctx->n) * sizeof(double), c->stream));
        if (l + 1 < mg->levels.size())
          {
            // slab -> replicated level and slab -> slab with induced cuts: ownership-aware tables; slab -> slab on the same
            // cells and box -> box: plain local tables
            const bool team  = c->team->size > 1;
            const bool slabs = team && l + 1 >= mg->n_aligned && l + 1 < mg->n_dist;
            const bool cut   = team && (l + 1 == mg->n_dist || slabs);
            const int  rc    = build_transfer(L.ctx, mg->levels[l + 1].ctx, L.to_coarse, cut, slabs);
            if (rc)
              return rc;
          }
      }
    // (a coarsest level that is cut into slabs -- a mesh of at most coarsest_reps cells per direction on a team -- has no
    // matrix to invert on any one slab: it keeps the polynomial)
    const bool coarsest_is_cut = c->team->size > 1 && mg->levels.size() - 1 < mg->n_dist;
    if (mg->dense && mg->levels.size() > 1 && !coarsest_is_cut && mg->levels.back().ctx->n <= 384 &&
        mg->levels.back().ctx->spmv_variant == 3)
      {
        MgLevel &L = mg->levels.back();
        HIPCHK(c, hipMalloc((void **)&L.dense_inv, size_t(L.ctx->n) * size_t(L.ctx->n) * sizeof(double)));
      }
    if (mg->block)
      for (MgLevel &L : mg->levels)
        L.ctx->want_dinv_blk = true; // filled by the next assembly of the level
    c->mg_stale = true;
    return MI_OK;
  }

  // coarse operators for the current state: u_total interpolated down the hierarchy (level 1 summed over the
  // slabs), every level re-assembled (replicated), eigenvalue estimates refreshed
  int mg_update(Team &T)
  {
    mi_ctx *c0 = T.members[0];
    if (!c0->mg || c0->mg->levels.size() < 2)
      return MI_OK;
    const size_t nl = c0->mg->levels.size();
    int          rc;
    for (mi_ctx *m : T.members)
      if (m->mg->block && !m->d_dinv_blk && !m->mf_fine) // switched on after the fine tangent was assembled
        {
          HIPCHK(m, hipMalloc((void **)&m->d_dinv_blk, size_t(m->mesh.nnodes) * m->dim * m->dim * sizeof(double)));
          if (!m->d_dinv_sym6 && m->dim == 3)
            HIPCHK(m, hipMalloc((void **)&m->d_dinv_sym6, size_t(m->mesh.nnodes) * 6 * sizeof(double)));
          mi::launch_extract_dinv_blk(m->dim, m->d_vals, m->d_diagpos, m->d_dinv_blk, m->d_dinv_sym6, m->mesh.nnodes, m->stream);
        }
    for (mi_ctx *m : T.members)
      {
        // u_total = u + du of the slab (all local nodes: the ghost copies are kept consistent)
        MgLevel &L0 = m->mg->levels[0];
        HIPCHK(m, hipMemcpyAsync(L0.x(), m->vec(MI_V_TOTAL_DISPLACEMENT), size_t(m->n) * sizeof(double),
                                 hipMemcpyDeviceToDevice, m->stream));
        mi::launch_vec_add(L0.x(), m->vec(MI_V_SOLUTION_DELTA), m->n, m->stream);
      }
    for (size_t l = 1; l < nl; ++l)
      {
        // state of level l from level l-1.  slab -> slab: plain interpolation inside the local box (ghosts
        // included); slab -> replicated: ownership-masked interpolation, summed over the slabs
        for (mi_ctx *m : T.members)
          {
            Multigrid    &mg  = *m->mg;
            MgLevel      &F   = mg.levels[l - 1], &C = mg.levels[l];
            const double *src = (l == 1) ? F.x() : F.ctx->vec(MI_V_TOTAL_DISPLACEMENT);
            mi::launch_lattice_interp(m->dim, false, F.to_coarse.state, C.ctx->vec(MI_V_TOTAL_DISPLACEMENT), src,
                                      C.ctx->d_cmask, m->stream);
          }
        if (is_dist(T, l - 1) && !is_dist(T, l) &&
            (rc = team_allreduce_vectors(
               T, [l](mi_ctx *m) { return m->mg->levels[l].ctx->vec(MI_V_TOTAL_DISPLACEMENT); },
               size_t(c0->mg->levels[l].ctx->n))))
          return rc;
        if (is_dist(T, l) && l >= c0->mg->n_aligned)
          {
            // induced cuts: a node was served by the slab that owns its left source plane -- possibly on one of that slab's
            // ghost planes; to the owner, then to every ghost copy
            auto u_of   = [l](mi_ctx *m) { return m->mg->levels[l].ctx->vec(MI_V_TOTAL_DISPLACEMENT); };
            auto ctx_of = [l](mi_ctx *m) { return m->mg->levels[l].ctx; };
            if ((rc = team_halo_accumulate(T, u_of, ctx_of)) || (rc = team_halo(T, u_of, ctx_of)))
              return rc;
          }
        for (mi_ctx *m : T.members)
          {
            MgLevel &C = m->mg->levels[l];
            if ((rc = enqueue_assembly(C.ctx)))
              return fail(c0, rc, "multigrid level %d: %s", int(l), C.ctx->err.c_str());
          }
      }
    HIPCHK(c0, hipGetLastError());
    for (size_t l = 0; l < nl; ++l)
      {
        if (l + 1 == nl && c0->mg->levels[l].dense_inv) // exact coarse solve: invert the level matrix instead
          {
            for (mi_ctx *m : T.members)
              {
                MgLevel &L = m->mg->levels[l];
                mi::SellParams sp = sell_params(L.ctx, nullptr, nullptr, nullptr, nullptr, nullptr);
                if (mi::launch_dense_inverse_from_sell(L.ctx->dim, sp, int(L.ctx->n), L.dense_inv, L.ctx->stream))
                  return fail(c0, MI_EINVAL, "multigrid: coarsest level too large for the dense solve");
              }
            continue;
          }
        if ((rc = estimate_lmax(T, l)))
          return rc;
      }
    for (mi_ctx *m : T.members)
      m->mg_stale = false;
    return MI_OK;
  }

  namespace
  {
    bool L_restrict_lists_fit(const mi::LatticeParams &p)
    {
      return p.rmax >= 1 && p.rmax <= 4 && !(p.rmax == 4 && p.n_tgt > 100000);
    }

    // whether level l runs its smoother steps and residual as fused products (see Multigrid::fuse)
    bool fuse_level(Team &T, size_t l)
    {
      const Multigrid &mg0 = *T.members[0]->mg;
      if (mg0.fuse == 0)
        return false;
      for (mi_ctx *m : T.members)
        {
          const mi_ctx *lc = m->mg->levels[l].ctx;
          if (lc->mf_fine) // no assembled rows to fuse an epilogue into (the matrix-free gather takes the step instead)
            return false;
          if (lc->spmv_variant != 3 || (mg0.fuse == 1 && lc->mesh.nnodes > mg0.fuse_max_nodes))
            return false;
        }
      return true;
    }

    // k Chebyshev-Jacobi steps on level l for A x = b over [lmax/ratio, lmax]; zero_start: x = 0 on entry.
    // Level 0 runs on all slabs in lockstep (halo exchange of x before every SpMV, update on the owned dofs).
    // ghosts_current: the ghost planes of x are up to date, the first product needs no exchange
    // first_done: the first step of a zero start (x = d = c2 D^-1 b) was taken by the restriction that produced b
    int chebyshev(Team &T, size_t l, int k, double ratio, bool zero_start, bool ghosts_current = false, bool first_done = false)
    {
      const double b = T.members[0]->mg->levels[l].lmax, a = b / ratio;
      const double theta = 0.5 * (b + a), delta = 0.5 * (b - a), sigma = theta / delta;
      double       rho_old = 1.0 / sigma;
      auto         x_of    = [l](mi_ctx *m) { return m->mg->levels[l].x(); };
      int          rc;
      for (int j = 0; j < k; ++j)
        {
          const bool first = (j == 0), skip_spmv = first && zero_start;
          if (skip_spmv && first_done)
            continue; // (the coefficients of step 0 leave no state behind)
          double c1, c2;
          if (first)
            {
              c1 = 0.0;
              c2 = 1.0 / theta;
            }
          else
            {
              const double rho = 1.0 / (2.0 * sigma - rho_old);
              c1               = rho * rho_old;
              c2               = 2.0 * rho / delta;
              rho_old          = rho;
            }
          // level 0 with the single-launch matrix-free product: the gather of the product applies this step itself
          bool mf_fused = !skip_spmv && l == 0 && T.members[0]->mg->block;
          for (mi_ctx *m : T.members)
            mf_fused = mf_fused && mf_gather_fusable(m);
          if (mf_fused)
            {
              // three-term form where the symmetric halves of D^-1 exist (3D): x'' over the buffer of the iterate before
              // (zero after the first step of a zero start), then the two x buffers change roles
              bool three = T.members[0]->mg->three_term;
              for (mi_ctx *m : T.members)
                three = three && m->mg->levels[l].ctx->d_dinv_sym6;
              std::vector<ChebFusion> cf;
              for (mi_ctx *m : T.members)
                {
                  MgLevel   &L = m->mg->levels[l];
                  ChebFusion f{L.b(), L.ctx->d_dinv_blk, L.d(), nullptr, c1, c2, 1, 1};
                  if (three)
                    {
                      f.dinv6 = L.ctx->d_dinv_sym6;
                      f.xprev = (j == 1 && zero_start) ? nullptr : L.x_other();
                      f.xnext = L.x_other();
                    }
                  cf.push_back(f);
                }
              if ((rc = level_spmv(T, l, x_of, cf.data(), first && ghosts_current)))
                return rc;
              if (three)
                for (mi_ctx *m : T.members)
                  m->mg->levels[l].x_swapped = !m->mg->levels[l].x_swapped;
              continue;
            }
          bool fused = !skip_spmv && fuse_level(T, l);
          if (fused)
            {
              // one launch: q = A x, d = c1 d + c2 D^-1 (b - q), x' = x + d written to the other x buffer
              std::vector<ChebFusion> cf;
              for (mi_ctx *m : T.members)
                {
                  MgLevel &L = m->mg->levels[l];
                  if (m->mg->block)
                    cf.push_back(ChebFusion{L.b(), L.ctx->d_dinv_blk, L.d(), L.x_other(), c1, c2, 1});
                  else
                    cf.push_back(ChebFusion{L.b(), L.ctx->work(W_DINV), L.d(), L.x_other(), c1, c2, 0});
                }
              if ((rc = level_spmv(T, l, x_of, cf.data(), first && ghosts_current)))
                return rc;
              for (mi_ctx *m : T.members)
                m->mg->levels[l].x_swapped = !m->mg->levels[l].x_swapped;
              continue;
            }
          if (!skip_spmv && (rc = level_spmv(T, l, x_of, nullptr, first && ghosts_current)))
            return rc;
          for (mi_ctx *m : T.members)
            {
              MgLevel      &L  = m->mg->levels[l];
              const int64_t o0 = L.ctx->own0, on = L.ctx->own_n;
              if (m->mg->block)
                mi::launch_cheb_step_blk(L.ctx->dim, L.x(), L.d(), L.b(), skip_spmv ? nullptr : L.q(), L.ctx->d_dinv_blk,
                                         c1, c2, o0 / L.ctx->dim, on / L.ctx->dim, L.ctx->stream);
              else
                mi::launch_cheb_step(L.x() + o0, L.d() + o0, L.b() + o0, skip_spmv ? nullptr : L.q() + o0,
                                     L.ctx->work(W_DINV) + o0, c1, c2, on, L.ctx->stream);
            }
        }
      return MI_OK;
    }

    // optimised 4th-kind weights beta_1..beta_k (Lottes 2022, table 1), k <= 6
    const double *cheb4_betas(int k)
    {
      static const double b1[] = {1.12500000000000};
      static const double b2[] = {1.02387287570313, 1.26408905371085};
      static const double b3[] = {1.00842544782028, 1.08867839208730, 1.33753125909618};
      static const double b4[] = {1.00391310427285, 1.04035811188593, 1.14863498546254, 1.38268869241000};
      static const double b5[] = {1.00212930146164, 1.02173711549260, 1.07872433192603, 1.19810065292663,
                                  1.41322542791682};
      static const double b6[] = {1.00128517255940, 1.01304293035233, 1.04678215124113, 1.11616489419675,
                                  1.23829020218444, 1.43524297106744};
      static const double *const tab[] = {b1, b2, b3, b4, b5, b6};
      return tab[k - 1];
    }

    // k steps of the 4th-kind Chebyshev smoother on level l (see cheb4_start / cheb4_step): k-1 SpMVs from x = 0,
    // k otherwise -- the same count as the 1st-kind recurrence above
    int chebyshev4(Team &T, size_t l, int k, bool zero_start)
    {
      k = std::min(k, 6);
      const double  rho  = T.members[0]->mg->levels[l].lmax;
      const double *beta = cheb4_betas(k);
      auto          x_of = [l](mi_ctx *m) { return m->mg->levels[l].x(); };
      auto          d_of = [l](mi_ctx *m) { return m->mg->levels[l].d(); };
      int           rc;
      if (!zero_start && (rc = level_spmv(T, l, x_of)))
        return rc;
      for (mi_ctx *m : T.members)
        {
          MgLevel      &L  = m->mg->levels[l];
          const int64_t o0 = L.ctx->own0, on = L.ctx->own_n;
          mi::launch_cheb4_start(L.x() + o0, L.d() + o0, L.r() + o0, L.b() + o0, zero_start ? nullptr : L.q() + o0,
                                 L.ctx->work(W_DINV) + o0, 4.0 / (3.0 * rho), on, L.ctx->stream);
        }
      for (int i = 1; i <= k; ++i)
        {
          const bool last = (i == k);
          if (!last && (rc = level_spmv(T, l, d_of)))
            return rc;
          const double ca = (2.0 * i - 1.0) / (2.0 * i + 3.0), cb = (8.0 * i + 4.0) / ((2.0 * i + 3.0) * rho);
          for (mi_ctx *m : T.members)
            {
              MgLevel      &L  = m->mg->levels[l];
              const int64_t o0 = L.ctx->own0, on = L.ctx->own_n;
              mi::launch_cheb4_step(L.x() + o0, L.d() + o0, L.r() + o0, last ? nullptr : L.q() + o0,
                                    L.ctx->work(W_DINV) + o0, beta[i - 1], ca, cb, on, L.ctx->stream);
            }
        }
      return MI_OK;
    }

    int smooth(Team &T, size_t l, int k, bool zero_start, bool ghosts_current = false, bool first_done = false)
    {
      Multigrid &mg0 = *T.members[0]->mg;
      return mg0.kind == 4 ? chebyshev4(T, l, k, zero_start) :
                             chebyshev(T, l, k, mg0.smooth_ratio, zero_start, ghosts_current, first_done);
    }

    // first_done: see chebyshev
    int vcycle(Team &T, size_t l, bool first_done = false)
    {
      mi_ctx      *c0  = T.members[0];
      Multigrid   &mg0 = *c0->mg;
      const size_t nl  = mg0.levels.size();
      int          rc;
      if (l + 1 == nl)
        {
          if (!mg0.levels[l].dense_inv)
            return chebyshev(T, l, mg0.coarse_degree, mg0.coarse_ratio, true);
          for (mi_ctx *m : T.members)
            {
              MgLevel &L = m->mg->levels[l];
              mi::launch_dense_apply(L.dense_inv, L.b(), L.x(), int(L.ctx->n), L.ctx->stream);
            }
          return MI_OK;
        }
      const bool dist_l = is_dist(T, l), dist_c = is_dist(T, l + 1);
      const bool induced_c = dist_c && l + 1 >= mg0.n_aligned; // the coarser level has cuts of its own (see mg_setup)
      const int  nu     = (l == 0) ? mg0.nu : ((l == 1 && mg0.nu_level1 > 0) ? mg0.nu_level1 : mg0.nu_coarse);
      if ((rc = smooth(T, l, nu, true, false, first_done)))
        return rc;
      auto x_of   = [l](mi_ctx *m) { return m->mg->levels[l].x(); };
      auto q_of   = [l](mi_ctx *m) { return m->mg->levels[l].q(); };
      auto ctx_l  = [l](mi_ctx *m) { return m->mg->levels[l].ctx; };
      auto ctx_c  = [l](mi_ctx *m) { return m->mg->levels[l + 1].ctx; };
      auto xc_of  = [l](mi_ctx *m) { return m->mg->levels[l + 1].x(); };
      bool mf_fused = l == 0;
      for (mi_ctx *m : T.members)
        mf_fused = mf_fused && mf_gather_fusable(m);
      if (mf_fused)
        {
          std::vector<ChebFusion> cf; // the gather of the matrix-free product writes q = b - A x on the owned rows
          for (mi_ctx *m : T.members)
            cf.push_back(ChebFusion{m->mg->levels[l].b(), nullptr, nullptr, nullptr, 0.0, 0.0, 0, 1});
          if ((rc = level_spmv(T, l, x_of, cf.data())))
            return rc;
        }
      else if (fuse_level(T, l))
        {
          std::vector<ChebFusion> cf; // residual mode of the fused epilogue: q = b - A x on the owned rows
          for (mi_ctx *m : T.members)
            cf.push_back(ChebFusion{m->mg->levels[l].b(), nullptr, nullptr, nullptr, 0.0, 0.0});
          if ((rc = level_spmv(T, l, x_of, cf.data())))
            return rc;
        }
      else
        {
          if ((rc = level_spmv(T, l, x_of)))
            return rc;
          for (mi_ctx *m : T.members)
            {
              MgLevel &L = m->mg->levels[l];
              mi::launch_vec_residual(L.q() + L.ctx->own0, L.b() + L.ctx->own0, L.q() + L.ctx->own0, L.ctx->own_n,
                                      L.ctx->stream); // q = b - A x (owned)
            }
        }
      // restriction.  distributed -> distributed (same slabs): the ghost planes of the residual come from the
      // neighbours, every owned coarse node then sums its complete fine neighbourhood.  distributed -> replicated:
      // every slab sums over its owned fine planes only and the partial sums are all-reduced.
      if (dist_l && dist_c && !induced_c && (rc = team_halo(T, q_of, ctx_l)))
        return rc;
      // Where no collective completes the restricted residual afterwards, the restriction also takes the first step of the
      // coarse level's smoother (x = d = c2 D^-1 b from a zero start: one launch per level fewer; "mg_restrict_fuse" 0: never)
      bool fuse_first = mg0.restrict_fuse && mg0.kind == 1 && mg0.block && l + 2 < nl && !(dist_l && !dist_c) && !induced_c;
      for (mi_ctx *m : T.members)
        {
          const MgLevel &C = m->mg->levels[l + 1];
          fuse_first = fuse_first && C.ctx->dim == 3 && C.ctx->d_dinv_blk && C.lmax > 0.0 &&
                       L_restrict_lists_fit(m->mg->levels[l].to_coarse.restrict_);
        }
      for (mi_ctx *m : T.members)
        {
          MgLevel &L = m->mg->levels[l], &C = m->mg->levels[l + 1];
          if (fuse_first)
            {
              const double bb = C.lmax, aa = bb / mg0.smooth_ratio, theta = 0.5 * (bb + aa);
              mi::launch_lattice_restrict_first_step(L.ctx->dim, L.to_coarse.restrict_, C.b(), L.q(), C.ctx->d_cmask, C.x(), C.d(),
                                                     C.ctx->d_dinv_blk, 1.0 / theta, C.ctx->own0 / C.ctx->dim,
                                                     C.ctx->own_n / C.ctx->dim, L.ctx->stream);
            }
          else
            mi::launch_lattice_restrict(L.ctx->dim, L.to_coarse.restrict_, C.b(), L.q(), C.ctx->d_cmask, L.ctx->stream);
        }
      if (dist_l && !dist_c &&
          (rc = team_allreduce_vectors(
             T, [l](mi_ctx *m) { return m->mg->levels[l + 1].b(); }, size_t(mg0.levels[l + 1].ctx->n))))
        return rc;
      // induced cuts: every slab has restricted its OWNED planes; what landed on a coarse ghost plane belongs to a neighbour
      if (induced_c && (rc = team_halo_accumulate(T, [l](mi_ctx *m) { return m->mg->levels[l + 1].b(); }, ctx_c)))
        return rc;
      if ((rc = vcycle(T, l + 1, fuse_first)))
        return rc;
      if (dist_c && (rc = team_halo(T, xc_of, ctx_c))) // prolongation reads the coarse ghost planes
        return rc;
      for (mi_ctx *m : T.members)
        {
          MgLevel &L = m->mg->levels[l], &C = m->mg->levels[l + 1];
          mi::launch_lattice_interp(L.ctx->dim, true, L.to_coarse.prolong, L.x(), C.x(), L.ctx->d_cmask, L.ctx->stream);
        }
      // The first product of the post-smoother needs no halo exchange (round 4): the residual product above exchanged the
      // ghost planes of x, nothing has written x since, and the prolongation has just added the coarse correction on ALL
      // local planes -- from coarse values that are the same on both sides of a cut (replicated level, or ghost planes
      // exchanged two statements up) through tables that are functions of the global lattice index.  A ghost value is
      // therefore what its owner holds, bit by bit ("halo_skip" 0 exchanges anyway: test_halo_skip_is_bitwise_neutral).
      // (induced cuts: a fine ghost plane may lie beyond the coarse slab's reach and was left alone -- exchange)
      return smooth(T, l, nu, false, dist_l && !induced_c);
    }
  } // namespace

  // W_Z = V-cycle(W_R) on every slab (owned residual in, correction out)
  int mg_apply(Team &T)
  {
    for (mi_ctx *m : T.members)
      {
        MgLevel &L0 = m->mg->levels[0];
        L0.b_ext    = m->work(W_R);
        L0.x_ext    = m->work(W_Z);
      }
    {
      // the three-term smoother steps change the roles of level 0's two x buffers 2 nu - 1 times per cycle: start in the
      // buffer from which the last step lands in W_Z
      Multigrid &mg0   = *T.members[0]->mg;
      bool       three = mg0.three_term && mg0.block && mg0.kind == 1;
      for (mi_ctx *m : T.members)
        three = three && mf_gather_fusable(m) && m->d_dinv_sym6;
      if (three)
        for (mi_ctx *m : T.members)
          m->mg->levels[0].x_swapped = ((2 * mg0.nu - 1) & 1) != 0;
    }
    int rc = vcycle(T, 0);
    if (rc)
      return rc;
    for (mi_ctx *m : T.members)
      if (m->mg->levels[0].x() != m->work(W_Z))
        HIPCHK(m, hipMemcpyAsync(m->work(W_Z), m->mg->levels[0].x(), size_t(m->n) * sizeof(double), hipMemcpyDeviceToDevice,
                                 m->stream));
    HIPCHK(T.members[0], hipGetLastError());
    return MI_OK;
  }
} // namespace mi_detail
