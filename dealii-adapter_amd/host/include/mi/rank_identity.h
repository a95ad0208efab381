// mi/rank_identity.h -- which rank of a decomposed run this process (or rank thread) is.
//
// The reference is single-rank (include/adapter/adapter.h:152-154).  Here the executables may run as
//   MI_SLABS=N                                   N slabs inside this process on one GPU (emulation; tests)
//   MI_WORLD_SIZE=N MI_RANK=r MI_UID_FILE=path   one process per GPU over RCCL (tools/launch_elasticity.py sets them
//   [MI_LOCAL_RANK=d]                            and starts the N processes); rank 0 creates the RCCL id and leaves it
//                                                in the file, the others wait for it; device = local rank
// or, for tests on a single-GPU box, as rank THREADS of one process (tests/fake_rccl/elasticity_ranks.cc): a thread
// announces its rank through thread_identity() before it builds its solver, and the environment is not consulted.
#pragma once
#include <algorithm>
#include <cstdlib>

namespace mi
{
  struct RankIdentity
  {
    int                  rank = -1, world = 0; // world == 0: not set, use the environment
    const unsigned char *uid = nullptr;        // 128-byte RCCL id shared by the rank threads
    int                  device = 0;
  };
  inline RankIdentity &thread_identity()
  {
    static thread_local RankIdentity id;
    return id;
  }
  inline int host_world_size()
  {
    if (thread_identity().world > 0)
      return thread_identity().world;
    const char *e = std::getenv("MI_WORLD_SIZE");
    return e ? std::max(1, std::atoi(e)) : 1;
  }
  inline int host_rank()
  {
    if (thread_identity().world > 0)
      return thread_identity().rank;
    const char *e = std::getenv("MI_RANK");
    return (e && host_world_size() > 1) ? std::atoi(e) : 0;
  }
} // namespace mi
