// Sanitizer driver (CPU, test infrastructure): the context-free entry points of the C-ABI -- mi_partition_describe and
// mi_partition_spmv_rows (include/mi_elasticity.h), i.e. mi::SlabPartition and mi::HostMesh::build of the product -- over a
// sweep of meshes and rank counts, built with -fsanitize=address,undefined.  Checks the tiling invariants the Python tests
// check (tests/test_partition_cpu.py) so that the sweep is not dead code to the optimiser, and that errors come back as
// codes with a message.
#include <cstdio>
#include <cstring>
#include <vector>

#include "mi_elasticity.h"

static int g_fail = 0;
#define CHECK(x)                                                          \
  do                                                                      \
    {                                                                     \
      if (!(x))                                                           \
        {                                                                 \
          std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #x);      \
          ++g_fail;                                                       \
        }                                                                 \
    }                                                                     \
  while (0)

int main()
{
  struct Case
  {
    int dim, p, reps[3], size;
  };
  const Case cases[] = {{3, 2, {3, 2, 7}, 3}, {3, 1, {4, 4, 19}, 8}, {2, 3, {5, 9, 1}, 4}, {3, 2, {5, 5, 5}, 1},
                        {2, 1, {7, 3, 1}, 2}, {3, 3, {2, 2, 4}, 2}, {2, 4, {2, 6, 1}, 3}, {3, 2, {6, 6, 6}, 5}};
  for (const Case &c : cases)
    {
      mi_mesh_desc md;
      std::memset(&md, 0, sizeof(md));
      md.dim    = c.dim;
      md.degree = c.p;
      for (int d = 0; d < 3; ++d)
        {
          md.reps[d] = c.reps[d];
          md.lo[d]   = 0.0;
          md.hi[d]   = 0.1 * c.reps[d];
        }
      const int roles[6] = {MI_FACE_CLAMPED, MI_FACE_INTERFACE, MI_FACE_INTERFACE, MI_FACE_INTERFACE, MI_FACE_ZCLAMP, MI_FACE_INTERFACE};
      for (int f = 0; f < 6; ++f)
        md.face_role[f] = roles[f];
      long long covered = 0;
      mi_partition_info prev;
      std::memset(&prev, 0, sizeof(prev));
      for (int r = 0; r < c.size; ++r)
        {
          mi_partition_info s;
          CHECK(mi_partition_describe(&md, r, c.size, &s) == MI_OK);
          CHECK(s.z1 > s.z0 && (r == 0 || s.z0 == prev.z1));
          CHECK(s.node_offset + s.own_begin == covered);
          covered = s.node_offset + s.own_end;
          if (r > 0)
            {
              CHECK(prev.up_send_n == s.down_recv_n && prev.up_recv_n == s.down_send_n);
              CHECK(prev.node_offset + prev.up_send == s.node_offset + s.down_recv);
              CHECK(prev.node_offset + prev.up_recv == s.node_offset + s.down_send);
            }
          // the SpMV row order of the slab: every owned node exactly once, interior rows first
          int64_t n_slices = 0, n_interior = 0;
          CHECK(mi_partition_spmv_rows(&md, r, c.size, &n_slices, &n_interior, nullptr, 0) == MI_OK);
          CHECK(n_slices > 0 && n_interior >= 0 && n_interior <= n_slices);
          std::vector<int32_t> rows(size_t(n_slices) * 64, -2);
          CHECK(mi_partition_spmv_rows(&md, r, c.size, &n_slices, &n_interior, rows.data(), int64_t(rows.size())) == MI_OK);
          std::vector<char> seen(size_t(s.nnodes_local), 0);
          long long         owned = 0;
          for (int32_t v : rows)
            if (v >= 0)
              {
                CHECK(v < s.nnodes_local && v >= s.own_begin && v < s.own_end && !seen[size_t(v)]);
                seen[size_t(v)] = 1;
                ++owned;
              }
          CHECK(owned == s.own_end - s.own_begin);
          CHECK(mi_partition_spmv_rows(&md, r, c.size, &n_slices, &n_interior, rows.data(), 3) == MI_EINVAL); // too small
          prev = s;
        }
      CHECK(covered == prev.nnodes_global);
      mi_partition_info s;
      CHECK(mi_partition_describe(&md, 0, c.reps[c.dim - 1] + 1, &s) == MI_EINVAL); // more slabs than cell layers
      CHECK(std::strlen(mi_last_error(nullptr)) > 0);
    }
  CHECK(mi_partition_describe(nullptr, 0, 1, nullptr) == MI_EINVAL);
  std::printf(g_fail ? "PARTITION SANITIZER RUN FAILED (%d)\n" : "PARTITION SANITIZER RUN OK\n", g_fail);
  return g_fail ? 1 : 0;
}
