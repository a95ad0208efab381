// mi_mesh.hpp -- host-side mesh / DoF / sparsity / colouring setup for the device solver.
//
// Replaces, for the hot path, what the reference gets from deal.II in make_grid
// (nonlinear_elasticity.cc:171-301) and system_setup (:305-380): a subdivided box with Q1 geometry,
// FE_Q(p)^dim DoFs, the all-components-couple sparsity (:339-345) and the Dirichlet index set of
// make_constraints (:1094-1150).  Layout is chosen for the GPU, not translated from deal.II:
//   * node-major DoF numbering  dof = dim*node + comp,  nodes lexicographic on the (p*reps+1)^dim lattice
//   * tangent stored as block rows with dim x dim blocks over nodes (every node pair couples in all components), in
//     ONE layout shared by the element scatter and the SpMV: "x-line-interleaved block rows" -- rows grouped by shape
//     into slices of 64; a block is dim*dim contiguous doubles, the blocks of one x-line of a row's column box stay
//     together (what a cell writes in one run), and one x-line of all 64 rows of a slice is one contiguous chunk
//     (what a wave of the SpMV streams in one go); see HostMesh::rowinfo
//   * cells grouped by parity colour (2^dim colours): two cells of one colour share no node, so a colour
//     scatters into the matrix with plain read-modify-write
//   * all per-cell arrays are stored in colour-sorted order so a colour's launch reads them contiguously
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

namespace mi
{
  // ------------------------------------------------------------------ 1D rules and nodal bases
  // n-point Gauss-Legendre on [0,1] (deal.II QGauss<1>(n))
  inline void gauss_legendre_unit(int n, std::vector<double> &x, std::vector<double> &w)
  {
    x.assign(n, 0.0);
    w.assign(n, 0.0);
    const int half = (n + 1) / 2;
    for (int i = 0; i < half; ++i)
      {
        long double t = std::cos(3.14159265358979323846L * (4 * i + 3) / (4 * n + 2)); // root i, descending
        long double dp = 0;
        for (int it = 0; it < 64; ++it)
          {
            long double pm = 1, pc = t; // P_0, P_1
            for (int k = 2; k <= n; ++k)
              {
                const long double pn = ((2 * k - 1) * t * pc - (k - 1) * pm) / k;
                pm                   = pc;
                pc                   = pn;
              }
            if (n == 1)
              {
                pm = 1;
                pc = t;
              }
            dp                   = n * (pm - t * pc) / (1 - t * t);
            const long double dt = pc / dp;
            t -= dt;
            if (std::fabs((double)dt) < 1e-18)
              break;
          }
        const long double wt = 2 / ((1 - t * t) * dp * dp);
        x[n - 1 - i]         = double(0.5L * (1 + t));
        x[i]                 = double(0.5L * (1 - t));
        w[i] = w[n - 1 - i] = double(0.5L * wt);
      }
  }

  // FE_Q(p) support points on [0,1]: equidistant for p <= 2, Gauss-Lobatto for p >= 3
  inline std::vector<double> feq_nodes_unit(int p)
  {
    std::vector<double> x(p + 1);
    if (p <= 2)
      {
        for (int i = 0; i <= p; ++i)
          x[i] = double(i) / p;
        return x;
      }
    const int m = p; // interior points are the roots of P'_m
    x[0]        = 0.0;
    x[p]        = 1.0;
    for (int i = 1; i < p; ++i)
      {
        long double t = -std::cos(3.14159265358979323846L * i / p);
        for (int it = 0; it < 64; ++it)
          {
            long double pm = 1, pc = t;
            for (int k = 2; k <= m; ++k)
              {
                const long double pn = ((2 * k - 1) * t * pc - (k - 1) * pm) / k;
                pm                   = pc;
                pc                   = pn;
              }
            const long double d1 = m * (pm - t * pc) / (1 - t * t);         // P'_m
            const long double d2 = (2 * t * d1 - m * (m + 1) * pc) / (1 - t * t); // P''_m
            const long double dt = d1 / d2;
            t -= dt;
            if (std::fabs((double)dt) < 1e-18)
              break;
          }
        x[i] = double(0.5L * (1 + t));
      }
    for (int i = 0; i < (p + 1) / 2; ++i) // enforce symmetry
      {
        const double s = 0.5 * (x[i] + 1.0 - x[p - i]);
        x[i]           = s;
        x[p - i]       = 1.0 - s;
      }
    if (p % 2 == 0)
      x[p / 2] = 0.5;
    return x;
  }

  // barycentric-free Lagrange value/derivative of basis a at x
  inline void lagrange_eval(const std::vector<double> &nodes, double x, double *val, double *der)
  {
    const int n = int(nodes.size());
    for (int a = 0; a < n; ++a)
      {
        double denom = 1.0;
        for (int m = 0; m < n; ++m)
          if (m != a)
            denom *= nodes[a] - nodes[m];
        double v = 1.0, d = 0.0;
        for (int m = 0; m < n; ++m)
          if (m != a)
            v *= x - nodes[m];
        for (int k = 0; k < n; ++k)
          if (k != a)
            {
              double t = 1.0;
              for (int m = 0; m < n; ++m)
                if (m != a && m != k)
                  t *= x - nodes[m];
              d += t;
            }
        val[a] = v / denom;
        der[a] = d / denom;
      }
  }

  // 1D tables handed to the kernels: N1[nq1][np1], dN1[nq1][np1], qw[nq1], qx[nq1], then dN_end[2][np1] = the basis
  // derivatives at the two ends of the unit interval (the face term with F evaluated ON the face, "correct_face_F")
  struct Tables1D
  {
    int                 p, np1, nq1;
    std::vector<double> nodes, qx, qw, N, dN, dN_end;
    void                build(int p_, int nq1_)
    {
      p   = p_;
      np1 = p + 1;
      nq1 = nq1_;
      nodes = feq_nodes_unit(p);
      gauss_legendre_unit(nq1, qx, qw);
      N.resize(size_t(nq1) * np1);
      dN.resize(size_t(nq1) * np1);
      for (int q = 0; q < nq1; ++q)
        lagrange_eval(nodes, qx[q], &N[size_t(q) * np1], &dN[size_t(q) * np1]);
      dN_end.resize(size_t(2) * np1);
      std::vector<double> val(static_cast<size_t>(np1), 0.0);
      lagrange_eval(nodes, 0.0, val.data(), &dN_end[0]);
      lagrange_eval(nodes, 1.0, val.data(), &dN_end[size_t(np1)]);
    }
    std::vector<double> packed() const
    {
      std::vector<double> t;
      t.insert(t.end(), N.begin(), N.end());
      t.insert(t.end(), dN.begin(), dN.end());
      t.insert(t.end(), qw.begin(), qw.end());
      t.insert(t.end(), qx.begin(), qx.end());
      t.insert(t.end(), dN_end.begin(), dN_end.end());
      return t;
    }
  };

  // ------------------------------------------------------------------ which direction a decomposition cuts (host only)
  // Slabs are cut along the LAST lattice direction (the slowest in the node numbering: a slab then owns a contiguous
  // range of node planes).  To cut a box along another direction -- the reference's flap is 18 x 3 x 1 cells
  // (nonlinear_elasticity.cc:189-205): only x can be cut at all -- the lattice is laid over the box ROTATED: internal
  // lattice direction d runs along physical coordinate ext_axis[d], forwards or backwards (dir[d]), with the wanted
  // direction last.  3D: a cyclic shift of (x, y, z); 2D cut along x: (y, -x).  Both keep the cells right handed.  Inside
  // the library nothing but the vertex coordinates knows about it (the element kernels work on general Q1 cells);
  // at the C-ABI the global arrays are permuted back to the reference's node order (x fastest).
  struct AxisMap
  {
    int  ext_axis[3] = {0, 1, 2};
    int  dir[3]      = {1, 1, 1};
    bool identity    = true;
  };
  // cut = 0, 1, 2: the physical direction to cut; -1: the one with most cell layers (ties: the last)
  inline AxisMap make_axis_map(int dim, const int *reps, int cut)
  {
    AxisMap m;
    if (cut < 0)
      {
        cut = dim - 1;
        for (int d = dim - 2; d >= 0; --d)
          if (reps[d] > reps[cut])
            cut = d;
      }
    if (cut >= dim)
      throw std::invalid_argument("cut direction outside the dimension of the mesh");
    if (cut == dim - 1)
      return m;
    m.identity = false;
    if (dim == 3)
      for (int d = 0; d < 3; ++d)
        m.ext_axis[d] = (cut + 1 + d) % 3; // (cut+1, cut+2, cut): cyclic, right handed
    else
      {
        m.ext_axis[0] = 1;
        m.ext_axis[1] = 0;
        m.dir[1]      = -1; // (y, -x): right handed
      }
    return m;
  }
  // the box as the rotated lattice sees it
  inline void rotate_box(const AxisMap &m, int dim, const int *reps, const double *lo, const double *hi, const int *face_role,
                         int *reps_i, double *lo_i, double *hi_i, int *role_i)
  {
    for (int d = 0; d < 3; ++d)
      {
        reps_i[d] = 1;
        lo_i[d] = hi_i[d] = 0.0;
        role_i[2 * d] = role_i[2 * d + 1] = 0;
      }
    for (int d = 0; d < dim; ++d)
      {
        const int e = m.ext_axis[d];
        reps_i[d]   = reps[e];
        lo_i[d]     = m.dir[d] > 0 ? lo[e] : hi[e];
        hi_i[d]     = m.dir[d] > 0 ? hi[e] : lo[e];
        for (int side = 0; side < 2; ++side)
          role_i[2 * d + side] = face_role[2 * e + (m.dir[d] > 0 ? side : 1 - side)];
      }
  }
  // external (reference order: x fastest) lattice point -> internal lattice point, for lattices with n_ext[] points
  inline int64_t ext_to_int_point(const AxisMap &m, int dim, const int *n_ext, int64_t id)
  {
    int e[3] = {0, 0, 0};
    for (int d = 0; d < dim; ++d)
      {
        e[d] = int(id % n_ext[d]);
        id /= n_ext[d];
      }
    int64_t out = 0, stride = 1;
    for (int d = 0; d < dim; ++d)
      {
        const int a = m.ext_axis[d], idx = m.dir[d] > 0 ? e[a] : n_ext[a] - 1 - e[a];
        out += stride * idx;
        stride *= n_ext[a];
      }
    return out;
  }

  // ------------------------------------------------------------------ slab decomposition (host only)
  // The last lattice direction (z in 3D, y in 2D) is the slowest in the node numbering, so a slab of cell layers
  // owns a contiguous range of node planes.  Rank r works on the box of its own layers [z0, z1) plus one ghost
  // layer above (if any); it owns the node planes (p*z0, p*z1] (rank 0 also owns plane 0).
  struct SlabPartition
  {
    int     rank = 0, size = 1, dim = 0, p = 0;
    int     z0 = 0, z1 = 0;          // owned cell layers [z0, z1)
    int     local_layers = 0;        // z1 - z0 (+1 ghost layer if rank < size-1)
    int64_t plane_nodes = 0;         // nodes per lattice plane
    int64_t node_offset = 0;         // global id of local node 0
    int64_t nnodes_global = 0, nnodes_local = 0;
    int64_t own_begin = 0, own_end = 0; // owned LOCAL node range
    // halo ranges in LOCAL nodes (count 0 = no neighbour)
    int64_t up_send = 0, up_send_n = 0;     // my top owned plane           -> rank+1 (its plane 0)
    int64_t up_recv = 0, up_recv_n = 0;     // ghost planes above           <- rank+1 (its first p owned planes)
    int64_t down_send = 0, down_send_n = 0; // my first p owned planes      -> rank-1
    int64_t down_recv = 0, down_recv_n = 0; // plane 0 (owned by rank-1)    <- rank-1
    int     local_reps[3] = {1, 1, 1};
    double  local_lo[3] = {0, 0, 0}, local_hi[3] = {0, 0, 0};
    int     local_face_role[6] = {0, 0, 0, 0, 0, 0};
    int64_t vertex_offset = 0; // global vertex id of local vertex 0 (for the perturbation array)
  };

  // cuts (optional, size + 1 entries, cuts[0] = 0, cuts[size] = reps[dim-1], strictly increasing): the cell layers of rank r
  // are [cuts[r], cuts[r+1]) instead of the balanced split -- the first coarsened multigrid level of a team takes the cuts
  // its finer level's cuts induce (mi_mg.cpp)
  inline SlabPartition make_slab_partition(int dim, int p, const int *reps, const double *lo, const double *hi,
                                           const int *face_role, int rank, int size, const int *cuts = nullptr)
  {
    if (size < 1 || rank < 0 || rank >= size)
      throw std::invalid_argument("bad rank/size");
    const int zd = dim - 1;
    if (size > reps[zd])
      throw std::invalid_argument("more ranks than cell layers in the decomposed direction");
    SlabPartition s;
    s.rank = rank;
    s.size = size;
    s.dim  = dim;
    s.p    = p;
    // balanced split of the layers
    s.z0 = int((int64_t(reps[zd]) * rank) / size);
    s.z1 = int((int64_t(reps[zd]) * (rank + 1)) / size);
    if (cuts)
      {
        if (cuts[0] != 0 || cuts[size] != reps[zd])
          throw std::invalid_argument("slab cuts do not span the cell layers");
        for (int r = 0; r < size; ++r)
          if (cuts[r + 1] <= cuts[r])
            throw std::invalid_argument("a slab without cell layers");
        s.z0 = cuts[rank];
        s.z1 = cuts[rank + 1];
      }
    const bool ghost_above = rank < size - 1;
    s.local_layers         = s.z1 - s.z0 + (ghost_above ? 1 : 0);
    s.plane_nodes          = 1;
    int64_t plane_verts    = 1;
    for (int d = 0; d < zd; ++d)
      {
        s.plane_nodes *= int64_t(p) * reps[d] + 1;
        plane_verts *= reps[d] + 1;
      }
    s.nnodes_global = s.plane_nodes * (int64_t(p) * reps[zd] + 1);
    s.nnodes_local  = s.plane_nodes * (int64_t(p) * s.local_layers + 1);
    s.node_offset   = s.plane_nodes * int64_t(p) * s.z0;
    s.vertex_offset = plane_verts * s.z0;
    const int64_t own_lo_plane = rank == 0 ? 0 : 1, own_hi_plane = int64_t(p) * (s.z1 - s.z0);
    s.own_begin = own_lo_plane * s.plane_nodes;
    s.own_end   = (own_hi_plane + 1) * s.plane_nodes;
    if (ghost_above)
      {
        s.up_send   = own_hi_plane * s.plane_nodes;
        s.up_send_n = s.plane_nodes;
        s.up_recv   = (own_hi_plane + 1) * s.plane_nodes;
        s.up_recv_n = int64_t(p) * s.plane_nodes;
      }
    if (rank > 0)
      {
        s.down_send   = s.plane_nodes;
        s.down_send_n = int64_t(p) * s.plane_nodes;
        s.down_recv   = 0;
        s.down_recv_n = s.plane_nodes;
      }
    for (int d = 0; d < 3; ++d)
      {
        s.local_reps[d] = d < dim ? reps[d] : 1;
        s.local_lo[d]   = lo[d];
        s.local_hi[d]   = hi[d];
      }
    s.local_reps[zd] = s.local_layers;
    const double h   = (hi[zd] - lo[zd]) / reps[zd];
    s.local_lo[zd]   = lo[zd] + h * s.z0;
    s.local_hi[zd]   = lo[zd] + h * (s.z0 + s.local_layers);
    for (int f = 0; f < 6; ++f)
      s.local_face_role[f] = face_role[f];
    if (rank > 0)
      s.local_face_role[2 * zd] = 0; // interior cut, not a boundary
    // the top of the local box is an interior cut unless the ghost layer is the topmost cell layer of the mesh:
    // then its far plane carries the real boundary role (its Dirichlet bits decide which columns the owned rows drop)
    if (ghost_above && s.z1 + 1 < reps[zd])
      s.local_face_role[2 * zd + 1] = 0;
    return s;
  }

  // global interface nodes (ascending) of the undecomposed box: lattice nodes on sides with role 7
  inline std::vector<int64_t> global_interface_nodes(int dim, int p, const int *reps, const int *face_role)
  {
    int64_t nn[3] = {1, 1, 1}, total = 1;
    for (int d = 0; d < dim; ++d)
      {
        nn[d] = int64_t(p) * reps[d] + 1;
        total *= nn[d];
      }
    std::vector<int64_t> out;
    for (int64_t n = 0; n < total; ++n)
      {
        const int64_t i[3] = {n % nn[0], (n / nn[0]) % nn[1], n / (nn[0] * nn[1])};
        bool          on   = false;
        for (int d = 0; d < dim && !on; ++d)
          on = (i[d] == 0 && face_role[2 * d] == 7) || (i[d] == nn[d] - 1 && face_role[2 * d + 1] == 7);
        if (on)
          out.push_back(n);
      }
    return out;
  }

  struct InterfaceFace
  {
    int32_t cell; // colour-sorted cell position
    int32_t face; // bit f set: side f (x-,x+,y-,y+,z-,z+) of the cell is an interface face
  };

  struct HostMesh
  {
    int dim = 0, p = 0, np1 = 0, npc = 0, nv = 0;
    int reps[3] = {1, 1, 1}, nn[3] = {1, 1, 1}, nvx[3] = {1, 1, 1};
    int64_t ncells = 0, nnodes = 0, nverts = 0, ndofs = 0, nnzb = 0;
    int     ncolours = 0;

    std::vector<double>   node_xyz;     // [nnodes][dim]
    std::vector<int32_t>  conn;         // [ncells][npc], colour-sorted cells
    std::vector<double>   cverts;       // [ncells][nv][dim], colour-sorted cells
    std::vector<int32_t>  cell_orig;    // colour-sorted position -> lexicographic cell id
    std::vector<int64_t>  colour_begin; // [ncolours+1] into the colour-sorted order
    std::vector<int32_t>  rowptr;       // [nnodes+1] block rows
    std::vector<int32_t>  colidx;       // [nnzb]
    std::vector<int32_t>  diagpos;      // [nnodes] position of block (node,node) in the value array (in blocks; see rowbase),
                                        // -1: the node has no row here
    std::vector<int32_t>  diagk;        // [nnodes] slot of (node,node) within its row
    std::vector<int32_t>  rowinfo;      // [nnodes][2] = {base, gstride}: slot k = (g, kx) of the node's row sits at block
                                        // base + g * gstride + kx of the value array; base -1: no row here (ghost node of
                                        // a slab: the row is complete on the neighbouring slab only)
    std::vector<uint8_t>  rowwx;        // [nnodes] width of the row's column box along x
    // storage order of the tangent, "x-line-interleaved block rows": rows grouped by (length, x-width wx of the column
    // box) into slices of 64, lane = row (see build_sell); the wx blocks of one x-line of a row's column box stay
    // together: block (g, kx) of lane l at off * 64 + g * 64 wx + l * wx + kx, i.e. gstride = 64 wx.  A cell writes runs
    // of 3 blocks = 216 contiguous bytes (as it would into block-CSR rows), and for one x-line g the 64 rows of a slice
    // are one contiguous chunk of 64 wx blocks for the SpMV.
    std::vector<uint16_t> off;          // [ncells][npc][npc]: where block (a, b) sits in the row of node a: bits 0-3 kx,
                                        // bits 4-14 g (see rowinfo); bit 15: this cell is the FIRST (in processing order)
                                        // to touch that block
    std::vector<uint32_t> node_first;   // [ncells] bit a set: this cell is the FIRST (in processing order) that contains
                                        // its local node a -> a cell-by-cell product may store instead of add (no memset)
    std::vector<uint8_t>  cmask;        // [nnodes] bit c set: dof (node,c) is Dirichlet-constrained
    std::vector<int32_t>  iface_nodes;  // ascending
    std::vector<InterfaceFace> iface_faces;         // sorted by colour
    std::vector<int64_t>       iface_colour_begin;  // [ncolours+1]

    // sliced-ELL view of the block pattern = THE storage order of the tangent: rows grouped by length (a box mesh has
    // dim+1 distinct lengths), 64 rows per slice, lane = row, no padding inside a slice
    int64_t              sell_nslices = 0, sell_nblk64 = 0; // sum over slices of their length (units of 64 blocks)
    int64_t              sell_nslices_interior = 0;         // slices [0, this) hold rows without ghost columns
    std::vector<int32_t> sell_perm;                         // [nslices*64] node of a slot, -1 = padding row
    std::vector<int32_t> sell_len;                          // [nslices] blocks per row in the slice
    std::vector<int32_t> sell_wx;                           // [nslices] x-width of the rows' column boxes (one per slice)
    std::vector<int64_t> sell_off;                          // [nslices+1] prefix sum of sell_len
    std::vector<int32_t> sell_box;                          // [nslices*64][2] first column, widths (wx | wy << 8) of the
                                                            // row's column box: columns need not be read from memory

    static void split(int64_t id, const int *ext, int dim, int *out)
    {
      for (int d = 0; d < 3; ++d)
        out[d] = 0;
      for (int d = 0; d < dim; ++d)
        {
          out[d] = int(id % ext[d]);
          id /= ext[d];
        }
    }

    // range of lattice nodes coupled to lattice index i along one direction
    void couple_range(int d, int i, int &lo, int &hi) const
    {
      if (i % p == 0)
        {
          lo = std::max(0, i - p);
          hi = std::min(nn[d] - 1, i + p);
        }
      else
        {
          lo = (i / p) * p;
          hi = lo + p;
        }
    }

    // zoff / zreps_global / own range: slab of a decomposed box (see SlabPartition): `reps_` are the LOCAL
    // repetitions, lo/hi the GLOBAL box, the last direction starts at global cell layer zoff
    // coord_of_axis (optional): lattice direction d runs along physical coordinate coord_of_axis[d] (see AxisMap; lo / hi
    // are given per LATTICE direction, hi < lo for a direction that runs backwards); perturb: offsets of the local
    // vertices, physical components
    void build(int dim_, int p_, const int *reps_, const double *lo, const double *hi, const int *face_role,
               const double *perturb, int zoff = 0, int zreps_global = 0, int64_t own_begin = 0, int64_t own_end = -1,
               const int *coord_of_axis = nullptr)
    {
      dim = dim_;
      p   = p_;
      if (dim != 2 && dim != 3)
        throw std::invalid_argument("dim must be 2 or 3");
      if (p < 1 || p > 4)
        throw std::invalid_argument("polynomial degree must be in 1..4");
      np1 = p + 1;
      npc = 1;
      nv  = 1 << dim;
      ncells = nnodes = nverts = 1;
      for (int d = 0; d < dim; ++d)
        {
          if (reps_[d] < 1)
            throw std::invalid_argument("repetitions must be >= 1");
          reps[d] = reps_[d];
          nn[d]   = p * reps[d] + 1;
          nvx[d]  = reps[d] + 1;
          npc *= np1;
          ncells *= reps[d];
          nnodes *= nn[d];
          nverts *= nvx[d];
        }
      ndofs = nnodes * dim;
      const std::vector<double> nodes1 = feq_nodes_unit(p);

      // vertices
      std::vector<double> vx(size_t(nverts) * dim);
      for (int64_t v = 0; v < nverts; ++v)
        {
          int vi[3];
          split(v, nvx, dim, vi);
          for (int d = 0; d < dim; ++d)
            {
              const bool cut = (d == dim - 1) && zreps_global > 0;
              const int  c   = coord_of_axis ? coord_of_axis[d] : d;
              vx[size_t(v) * dim + c] = lo[d] + (hi[d] - lo[d]) * (vi[d] + (cut ? zoff : 0)) / (cut ? zreps_global : reps[d]) +
                                        (perturb ? perturb[size_t(v) * dim + c] : 0.0);
            }
        }

      // parity colouring, colour-sorted cell order
      ncolours = 1 << dim;
      std::vector<std::vector<int32_t>> by_colour(ncolours);
      for (int64_t c = 0; c < ncells; ++c)
        {
          int ci[3];
          split(c, reps, dim, ci);
          const int col = (ci[0] & 1) | ((ci[1] & 1) << 1) | ((ci[2] & 1) << 2);
          by_colour[col].push_back(int32_t(c));
        }
      cell_orig.clear();
      colour_begin.assign(1, 0);
      {
        std::vector<std::vector<int32_t>> kept;
        for (auto &v : by_colour)
          if (!v.empty())
            kept.push_back(std::move(v));
        ncolours = int(kept.size());
        for (auto &v : kept)
          {
            cell_orig.insert(cell_orig.end(), v.begin(), v.end());
            colour_begin.push_back(int64_t(cell_orig.size()));
          }
      }

      // block-CSR pattern: the coupled set of a lattice node is a tensor-product box
      rowptr.assign(size_t(nnodes) + 1, 0);
      for (int64_t n = 0; n < nnodes; ++n)
        {
          int ni[3];
          split(n, nn, dim, ni);
          int64_t cnt = 1;
          for (int d = 0; d < dim; ++d)
            {
              int a, b;
              couple_range(d, ni[d], a, b);
              cnt *= (b - a + 1);
            }
          if (cnt > 32767)
            throw std::invalid_argument("row too long for 15-bit scatter offsets");
          rowptr[size_t(n) + 1] = int32_t(cnt);
        }
      {
        int64_t run = 0;
        for (int64_t n = 0; n < nnodes; ++n)
          {
            run += rowptr[size_t(n) + 1];
            if (run > INT32_MAX)
              throw std::invalid_argument("more than 2^31 blocks: partition the mesh over more GPUs");
            rowptr[size_t(n) + 1] = int32_t(run);
          }
        nnzb = run;
      }
      colidx.resize(size_t(nnzb));
      diagk.resize(size_t(nnodes));
      for (int64_t n = 0; n < nnodes; ++n)
        {
          int ni[3], a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
          split(n, nn, dim, ni);
          for (int d = 0; d < dim; ++d)
            couple_range(d, ni[d], a[d], b[d]);
          int64_t k = rowptr[size_t(n)];
          for (int z = a[2]; z <= b[2]; ++z)
            for (int y = a[1]; y <= b[1]; ++y)
              for (int x = a[0]; x <= b[0]; ++x)
                {
                  const int64_t m = x + int64_t(nn[0]) * (y + int64_t(nn[1]) * z);
                  if (m == n)
                    diagk[size_t(n)] = int32_t(k - rowptr[size_t(n)]);
                  colidx[size_t(k++)] = int32_t(m);
                }
        }

      // per-cell data in colour-sorted order
      conn.resize(size_t(ncells) * npc);
      cverts.resize(size_t(ncells) * nv * dim);
      off.resize(size_t(ncells) * npc * npc);
      node_xyz.assign(size_t(nnodes) * dim, 0.0);
      cmask.assign(size_t(nnodes), 0);
      std::vector<uint8_t> on_iface((size_t)nnodes, 0);
      std::vector<std::vector<InterfaceFace>> faces_by_colour(ncolours);
      std::vector<int> colour_of_pos((size_t)ncells, 0);
      for (int c = 0; c < ncolours; ++c)
        for (int64_t k = colour_begin[c]; k < colour_begin[c + 1]; ++k)
          colour_of_pos[size_t(k)] = c;

      for (int64_t pos = 0; pos < ncells; ++pos)
        {
          int ci[3];
          split(cell_orig[size_t(pos)], reps, dim, ci);
          double *cv = &cverts[size_t(pos) * nv * dim];
          for (int v = 0; v < nv; ++v)
            {
              const int     vi[3] = {ci[0] + (v & 1), ci[1] + ((v >> 1) & 1), ci[2] + ((v >> 2) & 1)};
              const int64_t id    = vi[0] + int64_t(nvx[0]) * (vi[1] + int64_t(nvx[1]) * vi[2]);
              for (int d = 0; d < dim; ++d)
                cv[v * dim + d] = vx[size_t(id) * dim + d];
            }
          int32_t *cn = &conn[size_t(pos) * npc];
          int      lat[125][3]; // up to 3D Q4
          for (int a = 0; a < npc; ++a)
            {
              int ai[3];
              const int ext[3] = {np1, np1, np1};
              split(a, ext, dim, ai);
              for (int d = 0; d < 3; ++d)
                lat[a][d] = d < dim ? ci[d] * p + ai[d] : 0;
              const int64_t node = lat[a][0] + int64_t(nn[0]) * (lat[a][1] + int64_t(nn[1]) * lat[a][2]);
              cn[a]              = int32_t(node);
              // support point under the d-linear (Q1) map of the cell
              double X[3] = {0, 0, 0};
              for (int v = 0; v < nv; ++v)
                {
                  double w = 1.0;
                  for (int d = 0; d < dim; ++d)
                    {
                      const double xi = nodes1[ai[d]];
                      w *= ((v >> d) & 1) ? xi : 1.0 - xi;
                    }
                  for (int d = 0; d < dim; ++d)
                    X[d] += w * cv[v * dim + d];
                }
              for (int d = 0; d < dim; ++d)
                node_xyz[size_t(node) * dim + d] = X[d];
            }
          // scatter slots
          uint16_t *co = &off[size_t(pos) * npc * npc];
          for (int a = 0; a < npc; ++a)
            {
              int rlo[3] = {0, 0, 0}, rhi[3] = {0, 0, 0};
              for (int d = 0; d < dim; ++d)
                couple_range(d, lat[a][d], rlo[d], rhi[d]);
              const int lx = rhi[0] - rlo[0] + 1, ly = rhi[1] - rlo[1] + 1;
              for (int b = 0; b < npc; ++b)
                co[a * npc + b] =
                  uint16_t((lat[b][0] - rlo[0]) + lx * ((lat[b][1] - rlo[1]) + ly * (lat[b][2] - rlo[2])));
            }
          // boundary faces of this cell
          int iface_mask = 0;
          for (int f = 0; f < 2 * dim; ++f)
            {
              const int  d    = f / 2;
              const bool side = f & 1;
              if (side ? (ci[d] != reps[d] - 1) : (ci[d] != 0))
                continue;
              const int role = face_role[f];
              if (role == 0)
                continue;
              for (int a = 0; a < npc; ++a)
                {
                  int ai[3];
                  const int ext[3] = {np1, np1, np1};
                  split(a, ext, dim, ai);
                  if (ai[d] != (side ? p : 0))
                    continue;
                  const int32_t node = cn[a];
                  if (role == 1) // clamped: all components (:1107-1124)
                    cmask[size_t(node)] |= uint8_t((1 << dim) - 1);
                  else if (role == 8 && dim == 3) // out-of-plane: z only (:1126-1147)
                    cmask[size_t(node)] |= 4;
                  else if (role == 7)
                    on_iface[size_t(node)] = 1;
                }
              if (role == 7)
                iface_mask |= 1 << f;
            }
          if (iface_mask)
            faces_by_colour[colour_of_pos[size_t(pos)]].push_back({int32_t(pos), int32_t(iface_mask)});
        }
      // first-touch flags: cells are processed colour by colour in this (colour-sorted) order and two cells of one
      // colour never share a block, so the first cell in this order that contains a block may WRITE it; all later
      // ones add to it.  Every pattern block is touched at least once, so the value array needs no zero fill.
      {
        std::vector<uint8_t> touched((size_t)nnzb, 0);
        for (int64_t pos = 0; pos < ncells; ++pos)
          {
            const int32_t *cn = &conn[size_t(pos) * npc];
            uint16_t      *co = &off[size_t(pos) * npc * npc];
            for (int a = 0; a < npc; ++a)
              for (int b = 0; b < npc; ++b)
                {
                  const size_t blk = size_t(rowptr[size_t(cn[a])]) + co[a * npc + b];
                  if (!touched[blk])
                    {
                      touched[blk] = 1;
                      co[a * npc + b] |= uint16_t(0x8000);
                    }
                }
          }
      }
      // plain slot k -> (g, kx) of the storage order
      rowwx.assign(size_t(nnodes), 0);
      for (int64_t n = 0; n < nnodes; ++n)
        {
          int ni[3], a0, b0;
          split(n, nn, dim, ni);
          couple_range(0, ni[0], a0, b0);
          rowwx[size_t(n)] = uint8_t(b0 - a0 + 1);
        }
      for (int64_t pos = 0; pos < ncells; ++pos)
        {
          const int32_t *cn = &conn[size_t(pos) * npc];
          uint16_t      *co = &off[size_t(pos) * npc * npc];
          for (int a = 0; a < npc; ++a)
            {
              const int wx = rowwx[size_t(cn[a])];
              for (int b = 0; b < npc; ++b)
                {
                  const int k = co[a * npc + b] & 0x7fff, g = k / wx, kx = k % wx;
                  if (g > 2047 || kx > 15)
                    throw std::invalid_argument("row too long for the 11+4-bit scatter offsets");
                  co[a * npc + b] = uint16_t((co[a * npc + b] & 0x8000) | (g << 4) | kx);
                }
            }
        }
      // the same for nodes (cells with up to 32 nodes: every element of the 3D Q2 product; empty otherwise)
      node_first.clear();
      if (npc <= 32)
        {
          node_first.assign(size_t(ncells), 0u);
          std::vector<uint8_t> seen((size_t)nnodes, 0);
          for (int64_t pos = 0; pos < ncells; ++pos)
            for (int a = 0; a < npc; ++a)
              {
                const size_t nd = size_t(conn[size_t(pos) * npc + a]);
                if (!seen[nd])
                  {
                    seen[nd] = 1;
                    node_first[size_t(pos)] |= 1u << a;
                  }
              }
        }
      iface_nodes.clear();
      for (int64_t n = 0; n < nnodes; ++n)
        if (on_iface[size_t(n)])
          iface_nodes.push_back(int32_t(n));
      iface_faces.clear();
      iface_colour_begin.assign(1, 0);
      for (int c = 0; c < ncolours; ++c)
        {
          iface_faces.insert(iface_faces.end(), faces_by_colour[c].begin(), faces_by_colour[c].end());
          iface_colour_begin.push_back(int64_t(iface_faces.size()));
        }
      build_sell(own_begin, own_end < 0 ? nnodes : own_end);
    }

    // SpMV rows = the OWNED rows only: ghost rows are incomplete here and complete on the neighbouring slab, their
    // values arrive by the halo exchange
    void build_sell(int64_t own_begin, int64_t own_end)
    {
      std::vector<int32_t> lens;
      for (int64_t n = 0; n < nnodes; ++n)
        lens.push_back(rowptr[size_t(n) + 1] - rowptr[size_t(n)]);
      // a class = rows of one length and one x-width of the column box
      auto cls = [&](int64_t n) { return int32_t(lens[size_t(n)] * 16 + rowwx[size_t(n)]); };
      std::vector<int32_t> classes;
      for (int64_t n = 0; n < nnodes; ++n)
        classes.push_back(cls(n));
      std::sort(classes.begin(), classes.end());
      classes.erase(std::unique(classes.begin(), classes.end()), classes.end());
      sell_perm.clear();
      sell_len.clear();
      sell_wx.clear();
      // interior rows (all columns owned) first, then the boundary rows, which read ghost columns and have to wait
      // for the halo exchange; the columns of a row ascend, so its first and last one decide
      auto boundary = [&](int64_t n) {
        return colidx[size_t(rowptr[size_t(n)])] < own_begin || colidx[size_t(rowptr[size_t(n) + 1]) - 1] >= own_end;
      };
      for (int bnd = 0; bnd < 2; ++bnd)
        {
          for (int32_t L : classes)
            {
              int64_t cnt = 0;
              for (int64_t n = own_begin; n < own_end; ++n)
                if (cls(n) == L && int(boundary(n)) == bnd)
                  {
                    sell_perm.push_back(int32_t(n));
                    ++cnt;
                  }
              while (cnt % 64)
                {
                  sell_perm.push_back(-1);
                  ++cnt;
                }
              for (int64_t sl = 0; sl < cnt / 64; ++sl)
                {
                  sell_len.push_back(L / 16);
                  sell_wx.push_back(L % 16);
                }
            }
          if (bnd == 0)
            sell_nslices_interior = int64_t(sell_len.size());
        }
      sell_nslices = int64_t(sell_len.size());
      sell_box.assign(sell_perm.size() * 2, 0);
      for (size_t slot = 0; slot < sell_perm.size(); ++slot)
        {
          const int32_t n = sell_perm[slot];
          if (n < 0)
            {
              sell_box[2 * slot + 1] = 255 | (255 << 8); // padding row: columns 0, 1, 2, ... (values are zero)
              continue;
            }
          int ni[3], a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
          split(n, nn, dim, ni);
          for (int d = 0; d < dim; ++d)
            couple_range(d, ni[d], a[d], b[d]);
          sell_box[2 * slot]     = colidx[size_t(rowptr[size_t(n)])];
          sell_box[2 * slot + 1] = (b[0] - a[0] + 1) | ((b[1] - a[1] + 1) << 8);
        }
      sell_off.assign(size_t(sell_nslices) + 1, 0);
      for (int64_t sl = 0; sl < sell_nslices; ++sl)
        sell_off[size_t(sl) + 1] = sell_off[size_t(sl)] + sell_len[size_t(sl)];
      sell_nblk64 = sell_off[size_t(sell_nslices)];
      if (sell_nblk64 * 64 > INT32_MAX)
        throw std::invalid_argument("more than 2^31 stored blocks: partition the mesh over more GPUs");
      // where the element scatter finds the rows
      rowinfo.assign(size_t(nnodes) * 2, -1);
      diagpos.assign(size_t(nnodes), -1);
      for (size_t slot = 0; slot < sell_perm.size(); ++slot)
        {
          const int32_t n = sell_perm[slot];
          if (n < 0)
            continue;
          const int wx               = rowwx[size_t(n)];
          rowinfo[2 * size_t(n)]     = int32_t(sell_off[slot / 64] * 64 + int64_t(slot % 64) * wx);
          rowinfo[2 * size_t(n) + 1] = 64 * wx;
          diagpos[size_t(n)]         = int32_t(valpos(n, diagk[size_t(n)]));
        }
    }

    // node numbering for the banded direct solver: the lattice directions sorted by extent, the SHORTEST running
    // fastest, which minimises the bandwidth of the box's pattern.  Returns the half bandwidth in DOFS.
    int band_perm(std::vector<int32_t> &perm) const
    {
      int ord[3] = {0, 1, 2};
      std::sort(ord, ord + dim, [&](int a, int b) { return nn[a] < nn[b] || (nn[a] == nn[b] && a < b); });
      int64_t stride[3] = {0, 0, 0}, run = 1;
      for (int k = 0; k < dim; ++k)
        {
          stride[ord[k]] = run;
          run *= nn[ord[k]];
        }
      perm.resize(size_t(nnodes));
      for (int64_t n = 0; n < nnodes; ++n)
        {
          int ni[3];
          split(n, nn, dim, ni);
          int64_t q = 0;
          for (int d = 0; d < dim; ++d)
            q += stride[d] * ni[d];
          perm[size_t(n)] = int32_t(q);
        }
      int64_t hb = 0; // coupled nodes differ by at most p lattice steps in every direction
      for (int d = 0; d < dim; ++d)
        hb += stride[d] * std::min(p, nn[d] - 1);
      return int(hb * dim + (dim - 1));
    }

    // number of dim x dim blocks of the value array (padding rows of the last slice of a length class included)
    int64_t nvalblocks() const { return sell_nblk64 * 64; }
    // position (in blocks) of slot k of the row of node n, -1 if the node has no row here
    int64_t valpos(int64_t n, int k) const
    {
      if (rowinfo[2 * size_t(n)] < 0)
        return -1;
      const int wx = rowwx[size_t(n)];
      return int64_t(rowinfo[2 * size_t(n)]) + int64_t(k / wx) * rowinfo[2 * size_t(n) + 1] + k % wx;
    }
  };
} // namespace mi
