// Legacy-VTK output in the spirit of DataOut + Postprocessor + MappingQEulerian of the reference
// (nonlinear_elasticity.cc:1215-1254, postprocessor.h:46-111): one patch per cell with (p+1)^dim points on the
// DISPLACED mesh; point data "displacement" (vector) and the dim*dim scalars strain_xx, strain_xy, ... = sym(grad u),
// with the gradient taken in the mapping used for output (the displaced configuration, as DataOut does when it is
// given the Eulerian mapping).  Points are duplicated per cell, so the strain is cell-wise discontinuous exactly as in
// deal.II patches.
// Cells (round 6): ONE higher-order cell per patch, VTK_LAGRANGE_QUADRILATERAL (70) / VTK_LAGRANGE_HEXAHEDRON (72) with
// the patch's (p+1)^dim points listed in VTK's Lagrange order, as the reference's `flags.write_higher_order_cells = true`
// makes DataOut write them (nonlinear_elasticity.cc:1222-1225, linear_elasticity.cc:599; needs ParaView >= 5.5, :1221).
// higher_order = false: the patch split into p^dim linear sub-cells (types 9 / 12; rounds 1-5), for older readers.
#pragma once
#include <cmath>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "device_vector.h"
#include "vtk_lagrange.h"

namespace mi
{
  namespace vtk_detail
  {
    // FE_Q(p) support points on [0,1]: equidistant for p <= 2, Gauss-Lobatto for p >= 3
    inline std::vector<double> support_points(int p)
    {
      std::vector<double> x(p + 1);
      for (int i = 0; i <= p; ++i)
        x[i] = double(i) / p;
      if (p <= 2)
        return x;
      for (int i = 1; i < p; ++i) // roots of P'_p by Newton, started at the Chebyshev-Lobatto points
        {
          double t = -std::cos(3.14159265358979323846 * i / p);
          for (int it = 0; it < 50; ++it)
            {
              double pm = 1, pc = t;
              for (int k = 2; k <= p; ++k)
                {
                  const double pn = ((2 * k - 1) * t * pc - (k - 1) * pm) / k;
                  pm              = pc;
                  pc              = pn;
                }
              const double d1 = p * (pm - t * pc) / (1 - t * t), d2 = (2 * t * d1 - p * (p + 1) * pc) / (1 - t * t);
              const double dt = d1 / d2;
              t -= dt;
              if (std::fabs(dt) < 1e-15)
                break;
            }
          x[i] = 0.5 * (1 + t);
        }
      return x;
    }
    // D[a][b] = dN_b/dxi at support point a
    inline std::vector<double> derivative_matrix(const std::vector<double> &x)
    {
      const int           n = int(x.size());
      std::vector<double> D(size_t(n) * n, 0.0);
      for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b)
          {
            double s = 0;
            for (int k = 0; k < n; ++k)
              if (k != b)
                {
                  double t = 1.0 / (x[b] - x[k]);
                  for (int m = 0; m < n; ++m)
                    if (m != b && m != k)
                      t *= (x[a] - x[m]) / (x[b] - x[m]);
                  s += t;
                }
            D[size_t(a) * n + b] = s;
          }
      return D;
    }
    inline bool invert(int dim, const double A[3][3], double B[3][3])
    {
      if (dim == 2)
        {
          const double d = A[0][0] * A[1][1] - A[0][1] * A[1][0];
          if (d == 0)
            return false;
          B[0][0] = A[1][1] / d;
          B[0][1] = -A[0][1] / d;
          B[1][0] = -A[1][0] / d;
          B[1][1] = A[0][0] / d;
          return true;
        }
      const double d = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                       A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
      if (d == 0)
        return false;
      B[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / d;
      B[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / d;
      B[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / d;
      B[1][0] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / d;
      B[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / d;
      B[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / d;
      B[2][0] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / d;
      B[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / d;
      B[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / d;
      return true;
    }
  } // namespace vtk_detail

  // EVERY rank of a decomposed run has to call this: mi_get_node_coords and mi_vec_get assemble global views through
  // team collectives (ncclAllReduce under RCCL), which hang or pair up with the wrong collective when only one rank
  // enters them.  write = false (ranks > 0): take part in the gathers, write nothing.
  inline void write_vtk(const Device &dev, int dim, int p, const int reps[3], const std::string &path, bool write = true,
                        bool higher_order = true)
  {
    const int64_t       nn = mi_n_nodes(dev.ctx()), n = mi_n_dofs(dev.ctx());
    std::vector<double> xyz(size_t(nn) * dim), u(size_t(n), 0.0);
    dev.check(mi_get_node_coords(dev.ctx(), xyz.data()), "mi_get_node_coords");
    dev.check(mi_vec_get(dev.ctx(), MI_V_TOTAL_DISPLACEMENT, u.data(), n), "mi_vec_get");
    if (!write)
      return;
    std::ofstream out(path);
    if (!out)
      throw std::runtime_error("Cannot open output file <" + path + ">");
    out.precision(12);

    const int np1 = p + 1;
    int       lat[3] = {1, 1, 1}, rr[3] = {1, 1, 1}, npc = 1;
    for (int d = 0; d < dim; ++d)
      {
        lat[d] = p * reps[d] + 1;
        rr[d]  = reps[d];
        npc *= np1;
      }
    const int64_t             ncells = int64_t(rr[0]) * rr[1] * rr[2];
    const std::vector<double> sp = vtk_detail::support_points(p), D = vtk_detail::derivative_matrix(sp);

    std::vector<double> pts(size_t(ncells) * npc * 3, 0.0), disp(size_t(ncells) * npc * 3, 0.0),
      strain(size_t(ncells) * npc * dim * dim, 0.0);
    std::vector<int64_t> node((size_t)npc, 0);
    for (int64_t c = 0; c < ncells; ++c)
      {
        const int ci[3] = {int(c % rr[0]), int((c / rr[0]) % rr[1]), int(c / (int64_t(rr[0]) * rr[1]))};
        int       ai[125][3]; // up to 3D Q4
        for (int a = 0; a < npc; ++a)
          {
            ai[a][0] = a % np1;
            ai[a][1] = (a / np1) % np1;
            ai[a][2] = dim == 3 ? a / (np1 * np1) : 0;
            node[size_t(a)] = (ci[0] * p + ai[a][0]) +
                              int64_t(lat[0]) * ((ci[1] * p + ai[a][1]) + int64_t(lat[1]) * (dim == 3 ? ci[2] * p + ai[a][2] : 0));
          }
        for (int a = 0; a < npc; ++a)
          {
            // gradients of all cell shape functions at support point a (unit cell), then the d-linear geometry map
            double Jm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, gu[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            std::vector<double> dN(size_t(npc) * 3, 0.0);
            for (int b = 0; b < npc; ++b)
              for (int k = 0; k < dim; ++k)
                {
                  double v = 1.0;
                  for (int d = 0; d < dim; ++d)
                    v *= (d == k) ? D[size_t(ai[a][d]) * np1 + ai[b][d]] : (ai[a][d] == ai[b][d] ? 1.0 : 0.0);
                  dN[size_t(b) * 3 + k] = v;
                }
            const double xi[3] = {sp[ai[a][0]], sp[ai[a][1]], dim == 3 ? sp[ai[a][2]] : 0.0};
            for (int v = 0; v < (1 << dim); ++v)
              {
                // corner v of the cell is the lattice node with local index p * bit
                int cb = 0, stride = 1;
                for (int d = 0; d < dim; ++d)
                  {
                    cb += (((v >> d) & 1) ? p : 0) * stride;
                    stride *= np1;
                  }
                for (int j = 0; j < dim; ++j)
                  {
                    double g = ((v >> j) & 1) ? 1.0 : -1.0;
                    for (int d = 0; d < dim; ++d)
                      if (d != j)
                        g *= ((v >> d) & 1) ? xi[d] : 1.0 - xi[d];
                    for (int i = 0; i < dim; ++i)
                      Jm[i][j] += xyz[size_t(node[size_t(cb)]) * dim + i] * g;
                  }
              }
            double Ji[3][3];
            if (!vtk_detail::invert(dim, Jm, Ji))
              throw std::runtime_error("degenerate cell in VTK output");
            for (int b = 0; b < npc; ++b)
              for (int j = 0; j < dim; ++j)
                {
                  double G = 0;
                  for (int k = 0; k < dim; ++k)
                    G += dN[size_t(b) * 3 + k] * Ji[k][j];
                  for (int i = 0; i < dim; ++i)
                    gu[i][j] += u[size_t(node[size_t(b)]) * dim + i] * G; // Grad_X u
                }
            double F[3][3], Fi[3][3], gx[3][3];
            for (int i = 0; i < dim; ++i)
              for (int j = 0; j < dim; ++j)
                F[i][j] = (i == j) + gu[i][j];
            if (!vtk_detail::invert(dim, F, Fi))
              throw std::runtime_error("det F = 0 in VTK output");
            for (int i = 0; i < dim; ++i)
              for (int j = 0; j < dim; ++j)
                {
                  gx[i][j] = 0;
                  for (int k = 0; k < dim; ++k)
                    gx[i][j] += gu[i][k] * Fi[k][j]; // grad_x u: gradient in the (displaced) output mapping
                }
            const size_t q = size_t(c) * npc + a;
            for (int d = 0; d < dim; ++d)
              {
                disp[q * 3 + d] = u[size_t(node[size_t(a)]) * dim + d];
                pts[q * 3 + d]  = xyz[size_t(node[size_t(a)]) * dim + d] + disp[q * 3 + d];
                for (int e = 0; e < dim; ++e)
                  strain[q * dim * dim + d * dim + e] = 0.5 * (gx[d][e] + gx[e][d]); // postprocessor.h:62-72
              }
          }
      }

    const int64_t npts = ncells * npc;
    int64_t       nsub = ncells;
    for (int d = 0; d < dim; ++d)
      nsub *= p;
    const int nv = 1 << dim;
    out << "# vtk DataFile Version 3.0\nsolid solver output (displaced configuration)\nASCII\nDATASET UNSTRUCTURED_GRID\n";
    out << "POINTS " << npts << " double\n";
    for (int64_t i = 0; i < npts; ++i)
      out << pts[size_t(i) * 3] << ' ' << pts[size_t(i) * 3 + 1] << ' ' << pts[size_t(i) * 3 + 2] << '\n';
    if (higher_order)
      {
        // one Lagrange cell per patch: the patch's points (stored lexicographically, x fastest) in VTK's order
        std::vector<int> perm((size_t)npc, 0); // perm[vtk position] = lexicographic point of the patch
        for (int a = 0; a < npc; ++a)
          perm[size_t(vtk_detail::lagrange_index(dim, p, a % np1, (a / np1) % np1, dim == 3 ? a / (np1 * np1) : 0))] = a;
        out << "CELLS " << ncells << ' ' << ncells * (npc + 1) << '\n';
        for (int64_t c = 0; c < ncells; ++c)
          {
            out << npc;
            for (int a = 0; a < npc; ++a)
              out << ' ' << c * npc + perm[size_t(a)];
            out << '\n';
          }
        out << "CELL_TYPES " << ncells << '\n';
        for (int64_t c = 0; c < ncells; ++c)
          out << (dim == 2 ? 70 : 72) << '\n'; // VTK_LAGRANGE_QUADRILATERAL / VTK_LAGRANGE_HEXAHEDRON
      }
    else
      {
    out << "CELLS " << nsub << ' ' << nsub * (nv + 1) << '\n';
    for (int64_t c = 0; c < ncells; ++c)
      for (int k = 0; k < (dim == 3 ? p : 1); ++k)
        for (int j = 0; j < p; ++j)
          for (int i = 0; i < p; ++i)
            {
              auto id = [&](int a, int b, int cc) { return c * npc + (i + a) + np1 * ((j + b) + np1 * (k + cc)); };
              if (dim == 2)
                out << "4 " << id(0, 0, 0) << ' ' << id(1, 0, 0) << ' ' << id(1, 1, 0) << ' ' << id(0, 1, 0) << '\n';
              else
                out << "8 " << id(0, 0, 0) << ' ' << id(1, 0, 0) << ' ' << id(1, 1, 0) << ' ' << id(0, 1, 0) << ' '
                    << id(0, 0, 1) << ' ' << id(1, 0, 1) << ' ' << id(1, 1, 1) << ' ' << id(0, 1, 1) << '\n';
            }
    out << "CELL_TYPES " << nsub << '\n';
    for (int64_t c = 0; c < nsub; ++c)
      out << (dim == 2 ? 9 : 12) << '\n';
      }
    out << "POINT_DATA " << npts << "\nVECTORS displacement double\n";
    for (int64_t i = 0; i < npts; ++i)
      out << disp[size_t(i) * 3] << ' ' << disp[size_t(i) * 3 + 1] << ' ' << disp[size_t(i) * 3 + 2] << '\n';
    static const char suffix[] = {'x', 'y', 'z'};
    for (int d = 0; d < dim; ++d)
      for (int e = 0; e < dim; ++e)
        {
          out << "SCALARS strain_" << suffix[d] << suffix[e] << " double 1\nLOOKUP_TABLE default\n";
          for (int64_t i = 0; i < npts; ++i)
            out << strain[size_t(i) * dim * dim + d * dim + e] << '\n';
        }
  }
} // namespace mi
