// Context-free entry points of the C-ABI: the slab partition of a mesh (include/mi_elasticity.h).  Pure host code -- no device,
// no HIP call -- so that the CPU tests of the multi-GPU host logic can run it anywhere, and so that it can be built on its
// own with the host compiler's sanitizers (tests/asan: -DMI_PARTITION_STANDALONE supplies the two symbols it takes from
// mi_ctx.cpp otherwise).
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <string>

#include "mi_elasticity.h"
#include "mi_mesh.hpp"

namespace mi_detail
{
  void set_create_error(const char *msg);
}

#ifdef MI_PARTITION_STANDALONE
namespace
{
  std::string g_standalone_error;
}
namespace mi_detail
{
  void set_create_error(const char *msg) { g_standalone_error = msg; }
} // namespace mi_detail
extern "C" const char *mi_last_error(const mi_ctx *) { return g_standalone_error.c_str(); }
#endif

namespace
{
  int fail_noctx(int code, const char *fmt, ...)
  {
    char    buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    mi_detail::set_create_error(buf);
    return code;
  }
} // namespace

extern "C" {

// host-only description of slab `rank` of `size` (no device needed): z-range, owned node range, halo ranges
int mi_partition_describe(const mi_mesh_desc *md, int rank, int size, mi_partition_info *out)
{
  if (!md || !out)
    return fail_noctx(MI_EINVAL, "null argument");
  try
    {
      const mi::SlabPartition s =
        mi::make_slab_partition(md->dim, md->degree, md->reps, md->lo, md->hi, md->face_role, rank, size);
      out->z0            = s.z0;
      out->z1            = s.z1;
      out->local_layers  = s.local_layers;
      out->plane_nodes   = s.plane_nodes;
      out->node_offset   = s.node_offset;
      out->nnodes_global = s.nnodes_global;
      out->nnodes_local  = s.nnodes_local;
      out->own_begin     = s.own_begin;
      out->own_end       = s.own_end;
      out->up_send       = s.up_send;
      out->up_send_n     = s.up_send_n;
      out->up_recv       = s.up_recv;
      out->up_recv_n     = s.up_recv_n;
      out->down_send     = s.down_send;
      out->down_send_n   = s.down_send_n;
      out->down_recv     = s.down_recv;
      out->down_recv_n   = s.down_recv_n;
      for (int d = 0; d < 3; ++d)
        {
          out->local_reps[d] = s.local_reps[d];
          out->local_lo[d]   = s.local_lo[d];
          out->local_hi[d]   = s.local_hi[d];
        }
      for (int f = 0; f < 6; ++f)
        out->local_face_role[f] = s.local_face_role[f];
    }
  catch (const std::exception &e)
    {
      return fail_noctx(MI_EINVAL, "%s", e.what());
    }
  return MI_OK;
}

int mi_partition_spmv_rows(const mi_mesh_desc *md, int rank, int size, int64_t *n_slices, int64_t *n_interior_slices,
                           int32_t *rows, int64_t capacity)
{
  if (!md || !n_slices || !n_interior_slices)
    return fail_noctx(MI_EINVAL, "null argument");
  try
    {
      const mi::SlabPartition s =
        mi::make_slab_partition(md->dim, md->degree, md->reps, md->lo, md->hi, md->face_role, rank, size);
      mi::HostMesh m;
      if (size == 1)
        m.build(md->dim, md->degree, md->reps, md->lo, md->hi, md->face_role, nullptr);
      else
        m.build(md->dim, md->degree, s.local_reps, md->lo, md->hi, s.local_face_role, nullptr, s.z0,
                md->reps[md->dim - 1], s.own_begin, s.own_end);
      *n_slices          = m.sell_nslices;
      *n_interior_slices = m.sell_nslices_interior;
      if (rows)
        {
          if (capacity < int64_t(m.sell_perm.size()))
            return fail_noctx(MI_EINVAL, "rows[] holds %lld entries, %lld needed", (long long)capacity,
                        (long long)m.sell_perm.size());
          std::copy(m.sell_perm.begin(), m.sell_perm.end(), rows);
        }
    }
  catch (const std::exception &e)
    {
      return fail_noctx(MI_EINVAL, "%s", e.what());
    }
  return MI_OK;
}

} // extern "C"
