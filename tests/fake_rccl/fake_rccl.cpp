// TEST DOUBLE of RCCL for boxes with ONE GPU (tests/test_gpu_fake_rccl.py).
//
// RCCL refuses two ranks on one device, so the multi-rank branch of the library (ncclSend/ncclRecv halo exchange in
// groups, ncclAllReduce of scalars and vectors, the communication stream and its events, the order of collective
// calls on every rank) cannot run on the single-GPU test box.  This file implements the handful of RCCL entry
// points the library uses -- with the real <rccl/rccl.h> prototypes -- for ranks that are THREADS of one process
// sharing one device.  Every call is synchronous: it waits for the stream it was given, meets the other ranks at a
// barrier (60 s timeout -> ncclSystemError instead of a hang, which is how a mismatched call order shows), moves
// the data with plain device copies, and returns.  Stream order is therefore trivially preserved.
// tests/fake_rccl/Makefile links the library's own objects against this file instead of librccl.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace
{
  struct Op
  {
    bool        send;
    int         peer;
    const void *src;
    void       *dst;
    size_t      bytes;
    hipStream_t stream;
    bool        matched = false;
  };

  struct Group
  {
    int                                n;
    std::mutex                         m;
    std::condition_variable            cv;
    int                                arrived = 0;
    long                               gen     = 0;
    std::vector<std::vector<Op>>       pending;
    std::vector<std::vector<double>>   host;
    explicit Group(int n_)
      : n(n_)
      , pending(n_)
      , host(n_)
    {}
    bool barrier()
    {
      std::unique_lock<std::mutex> lk(m);
      const long                   g = gen;
      if (++arrived == n)
        {
          arrived = 0;
          ++gen;
          cv.notify_all();
          return true;
        }
      return cv.wait_for(lk, std::chrono::seconds(60), [&] { return gen != g; });
    }
  };

  std::mutex                                     g_reg_mutex;
  std::map<std::string, std::shared_ptr<Group>>  g_registry;
  long                                           g_id_counter = 0;

  thread_local int             t_depth = 0;
  thread_local std::vector<Op> t_ops;
} // namespace

struct ncclComm
{
  std::shared_ptr<Group> g;
  int                    rank;
};

static ncclResult_t flush(ncclComm *c)
{
  Group &G = *c->g;
  for (const Op &o : t_ops)
    if (hipStreamSynchronize(o.stream) != hipSuccess)
      return ncclUnhandledCudaError;
  {
    std::lock_guard<std::mutex> lk(G.m);
    G.pending[c->rank] = t_ops;
  }
  if (!G.barrier())
    return ncclSystemError;
  ncclResult_t rc = ncclSuccess;
  for (const Op &r : t_ops)
    {
      if (r.send)
        continue;
      bool found = false;
      {
        std::lock_guard<std::mutex> lk(G.m);
        for (Op &s : G.pending[r.peer])
          if (s.send && s.peer == c->rank && !s.matched && s.bytes == r.bytes)
            {
              s.matched = true;
              found     = true;
              if (hipMemcpy(r.dst, s.src, r.bytes, hipMemcpyDeviceToDevice) != hipSuccess)
                rc = ncclUnhandledCudaError;
              break;
            }
      }
      if (!found)
        rc = ncclInvalidUsage; // a receive without a matching send of the same size
    }
  if (!G.barrier())
    return ncclSystemError;
  bool unmatched = false;
  {
    std::lock_guard<std::mutex> lk(G.m);
    for (const Op &s : G.pending[c->rank])
      if (s.send && !s.matched)
        unmatched = true;
    G.pending[c->rank].clear();
  }
  t_ops.clear();
  if (!G.barrier())
    return ncclSystemError;
  return unmatched ? ncclInvalidUsage : rc;
}

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
  std::lock_guard<std::mutex> lk(g_reg_mutex);
  std::memset(id, 0, sizeof(*id));
  const long k = ++g_id_counter;
  std::memcpy(id->internal, "FAKE-RCCL", 9);
  std::memcpy(id->internal + 16, &k, sizeof(k));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
  if (nranks < 1 || rank < 0 || rank >= nranks)
    return ncclInvalidArgument;
  std::shared_ptr<Group> g;
  {
    std::lock_guard<std::mutex> lk(g_reg_mutex);
    const std::string           key(id.internal, sizeof(id.internal));
    auto                        it = g_registry.find(key);
    if (it == g_registry.end())
      it = g_registry.emplace(key, std::make_shared<Group>(nranks)).first;
    g = it->second;
  }
  if (g->n != nranks)
    return ncclInvalidArgument;
  *comm = new ncclComm{g, rank};
  return g->barrier() ? ncclSuccess : ncclSystemError;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count)
{
  if (!comm || !count)
    return ncclInvalidArgument;
  *count = comm->g->n;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
  delete comm;
  return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
  switch (r)
    {
      case ncclSuccess: return "no error";
      case ncclUnhandledCudaError: return "fake rccl: HIP error";
      case ncclSystemError: return "fake rccl: barrier timed out (ranks issued different collectives?)";
      case ncclInvalidUsage: return "fake rccl: unmatched send/recv";
      default: return "fake rccl: error";
    }
}

ncclResult_t ncclGroupStart()
{
  ++t_depth;
  return ncclSuccess;
}

static thread_local ncclComm *t_group_comm = nullptr;

ncclResult_t ncclGroupEnd()
{
  if (--t_depth > 0)
    return ncclSuccess;
  ncclComm *c  = t_group_comm;
  t_group_comm = nullptr;
  if (!c) // an empty group is still collective in the library's usage?  no: nothing to do
    return ncclSuccess;
  return flush(c);
}

static ncclResult_t post(ncclComm *c, Op o)
{
  t_ops.push_back(o);
  t_group_comm = c;
  if (t_depth == 0)
    {
      t_group_comm = nullptr;
      return flush(c);
    }
  return ncclSuccess;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s)
{
  if (dt != ncclDouble)
    return ncclInvalidArgument;
  return post(comm, Op{true, peer, buf, nullptr, count * sizeof(double), s});
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s)
{
  if (dt != ncclDouble)
    return ncclInvalidArgument;
  return post(comm, Op{false, peer, nullptr, buf, count * sizeof(double), s});
}

ncclResult_t ncclAllReduce(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t dt, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t s)
{
  if (dt != ncclDouble || op != ncclSum)
    return ncclInvalidArgument;
  Group &G = *comm->g;
  if (hipStreamSynchronize(s) != hipSuccess)
    return ncclUnhandledCudaError;
  std::vector<double> mine(count);
  if (hipMemcpy(mine.data(), sendbuf, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
    return ncclUnhandledCudaError;
  {
    std::lock_guard<std::mutex> lk(G.m);
    G.host[comm->rank] = mine;
  }
  if (!G.barrier())
    return ncclSystemError;
  std::vector<double> sum(count, 0.0);
  {
    std::lock_guard<std::mutex> lk(G.m);
    for (int r = 0; r < G.n; ++r)
      {
        if (G.host[r].size() != count)
          return ncclInvalidUsage; // ranks disagree on the collective
        for (size_t i = 0; i < count; ++i)
          sum[i] += G.host[r][i];
      }
  }
  if (!G.barrier())
    return ncclSystemError;
  if (hipMemcpy(recvbuf, sum.data(), count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
    return ncclUnhandledCudaError;
  return ncclSuccess;
}

} // extern "C"
