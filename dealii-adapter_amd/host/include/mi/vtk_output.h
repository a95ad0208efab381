// Legacy-VTK output of the displaced mesh: nodes of the FE_Q(p) lattice, every cell split into p^dim linear
// sub-cells, point data "displacement".  Stands in for DataOut + MappingQEulerian of
// nonlinear_elasticity.cc:1215-1254 (the strain post-processor fields of postprocessor.h are not written).
#pragma once
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "device_vector.h"

namespace mi
{
  inline void write_vtk(const Device &dev, int dim, int p, const int reps[3], const std::string &path)
  {
    const int64_t       nn = mi_n_nodes(dev.ctx()), n = mi_n_dofs(dev.ctx());
    std::vector<double> xyz(size_t(nn) * dim), u(size_t(n), 0.0);
    dev.check(mi_get_node_coords(dev.ctx(), xyz.data()), "mi_get_node_coords");
    dev.check(mi_vec_get(dev.ctx(), MI_V_TOTAL_DISPLACEMENT, u.data(), n), "mi_vec_get");
    std::ofstream out(path);
    if (!out)
      throw std::runtime_error("Cannot open output file <" + path + ">");
    int lat[3] = {1, 1, 1};
    for (int d = 0; d < dim; ++d)
      lat[d] = p * reps[d] + 1;
    out << "# vtk DataFile Version 3.0\nsolid displacement (displaced configuration)\nASCII\nDATASET UNSTRUCTURED_GRID\n";
    out << "POINTS " << nn << " double\n";
    for (int64_t i = 0; i < nn; ++i)
      {
        for (int d = 0; d < 3; ++d)
          out << (d < dim ? xyz[size_t(i) * dim + d] + u[size_t(i) * dim + d] : 0.0) << (d < 2 ? ' ' : '\n');
      }
    int64_t ncell = 1;
    for (int d = 0; d < dim; ++d)
      ncell *= lat[d] - 1;
    const int nv = 1 << dim;
    out << "CELLS " << ncell << ' ' << ncell * (nv + 1) << '\n';
    for (int k = 0; k < (dim == 3 ? lat[2] - 1 : 1); ++k)
      for (int j = 0; j < lat[1] - 1; ++j)
        for (int i = 0; i < lat[0] - 1; ++i)
          {
            auto id = [&](int a, int b, int c) { return int64_t(i + a) + int64_t(lat[0]) * ((j + b) + int64_t(lat[1]) * (k + c)); };
            if (dim == 2)
              out << "4 " << id(0, 0, 0) << ' ' << id(1, 0, 0) << ' ' << id(1, 1, 0) << ' ' << id(0, 1, 0) << '\n';
            else
              out << "8 " << id(0, 0, 0) << ' ' << id(1, 0, 0) << ' ' << id(1, 1, 0) << ' ' << id(0, 1, 0) << ' '
                  << id(0, 0, 1) << ' ' << id(1, 0, 1) << ' ' << id(1, 1, 1) << ' ' << id(0, 1, 1) << '\n';
          }
    out << "CELL_TYPES " << ncell << '\n';
    for (int64_t c = 0; c < ncell; ++c)
      out << (dim == 2 ? 9 : 12) << '\n';
    out << "POINT_DATA " << nn << "\nVECTORS displacement double\n";
    for (int64_t i = 0; i < nn; ++i)
      for (int d = 0; d < 3; ++d)
        out << (d < dim ? u[size_t(i) * dim + d] : 0.0) << (d < 2 ? ' ' : '\n');
  }
} // namespace mi
