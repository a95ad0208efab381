// Node order of VTK's Lagrange cells (pure host arithmetic, no device): shared by the VTK writer and its unit test
#pragma once

namespace mi
{
  namespace vtk_detail
  {
    // Position of lattice point (i, j, k) of a patch of `order` intervals per direction in VTK's node order of Lagrange cells
    // [vtkLagrangeQuadrilateral / vtkLagrangeHexahedron::PointIndexFromIJK as deal.II >= 9.1 restates it in DataOutBase for
    // write_higher_order_cells]: first the 2^dim corners in the order of the linear cell, then the edges' interior points
    // (the four edges of the bottom face counter-clockwise from the x-edge at y = 0: x y x y; the same four of the top face;
    // the four vertical edges in the order (x0,y0) (x1,y0) (x0,y1) (x1,y1) -- the VTK 8 order deal.II 9.2-9.5 write; VTK 9
    // readers accept both by file version), then the faces' interior points (x-, x+, y-, y+, z-, z+; lexicographic inside a
    // face), last the interior of the body, lexicographic.
    inline int lagrange_index(int dim, int order, int i, int j, int k)
    {
      const int  m    = order - 1; // interior points per edge
      const bool ib = i == 0 || i == order, jb = j == 0 || j == order, kb = dim == 3 ? (k == 0 || k == order) : true;
      const int  nb   = int(ib) + int(jb) + (dim == 3 ? int(kb) : 0);
      const int  corner = (i ? (j ? 2 : 1) : (j ? 3 : 0));
      if (dim == 2)
        {
          if (nb == 2)
            return corner;
          if (nb == 1 && !ib) // bottom / top edge
            return 4 + (i - 1) + (j ? 2 * m : 0);
          if (nb == 1) // right / left edge
            return 4 + (j - 1) + (i ? m : 3 * m);
          return 4 + 4 * m + (i - 1) + m * (j - 1);
        }
      if (nb == 3)
        return corner + (k ? 4 : 0);
      int off = 8;
      if (nb == 2)
        {
          if (!ib)
            return off + (i - 1) + (j ? 2 * m : 0) + (k ? 4 * m : 0);
          if (!jb)
            return off + (j - 1) + (i ? m : 3 * m) + (k ? 4 * m : 0);
          return off + 8 * m + (k - 1) + m * (i ? (j ? 3 : 1) : (j ? 2 : 0));
        }
      off += 12 * m;
      if (nb == 1)
        {
          if (ib)
            return off + (j - 1) + m * (k - 1) + (i ? m * m : 0);
          off += 2 * m * m;
          if (jb)
            return off + (i - 1) + m * (k - 1) + (j ? m * m : 0);
          off += 2 * m * m;
          return off + (i - 1) + m * (j - 1) + (k ? m * m : 0);
        }
      off += 6 * m * m;
      return off + (i - 1) + m * ((j - 1) + m * (k - 1));
    }
  } // namespace vtk_detail
} // namespace mi
