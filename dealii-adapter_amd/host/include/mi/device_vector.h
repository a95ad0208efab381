// mi::Device / mi::Vector -- thin C++ RAII layer over the C-ABI (include/mi_elasticity.h) for the host solvers.
//
// mi::Vector plays the role of the reference's VectorType (BlockVector<double>): a *handle* to a
// device-resident global vector.  Copy-assignment copies device-to-device, so the Adapter's checkpoint code
// (`old_state_data[i] = *state_variables[i]`, adapter.h:457-460) keeps its shape while the data never leaves HBM.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "mi_elasticity.h"
#include "rank_identity.h"

namespace mi
{
  // (decomposition of the executables: mi/rank_identity.h)
  struct Error : std::runtime_error
  {
    int code;
    Error(int c, const std::string &what)
      : std::runtime_error(what)
      , code(c)
    {}
  };

  class Device
  {
  public:
    Device(const mi_mesh_desc &mesh, const mi_material_desc &mat, const mi_newmark_desc &nm, int device_id = 0)
    {
      mi_comm_desc  comm;
      mi_comm_desc *use = nullptr;
      std::memset(&comm, 0, sizeof(comm));
      const int world = host_world_size();
      if (world > 1)
        {
          comm.rank = host_rank();
          comm.size = world;
          if (comm.rank < 0 || comm.rank >= world)
            throw Error(MI_EINVAL, "MI_RANK outside [0, MI_WORLD_SIZE)");
          if (thread_identity().world > 0) // rank threads of one process (tests): id and device come with the identity
            {
              if (!thread_identity().uid)
                throw Error(MI_EINVAL, "rank thread without an RCCL id");
              std::memcpy(uid_, thread_identity().uid, 128);
              device_id = thread_identity().device;
            }
          else
            {
              const char *file = std::getenv("MI_UID_FILE");
              if (!file)
                throw Error(MI_EINVAL, "MI_WORLD_SIZE > 1 needs MI_UID_FILE (see tools/launch_elasticity.py)");
              exchange_unique_id(file, comm.rank, uid_);
              if (const char *e = std::getenv("MI_LOCAL_RANK"))
                device_id = std::atoi(e);
              else
                device_id = comm.rank;
            }
          comm.nccl_unique_id = uid_;
          use                 = &comm;
        }
      else if (const char *e = std::getenv("MI_SLABS"))
        {
          if (std::atoi(e) > 1)
            {
              comm.rank = -1; // all slabs in this process
              comm.size = std::atoi(e);
              use       = &comm;
            }
        }
      const int rc = mi_ctx_create(&mesh, &mat, &nm, device_id, use, &ctx_);
      if (rc != MI_OK)
        throw Error(rc, std::string("mi_ctx_create: ") + mi_last_error(nullptr));
    }
    ~Device() { mi_ctx_destroy(ctx_); }
    Device(const Device &)            = delete;
    Device &operator=(const Device &) = delete;

    mi_ctx *ctx() const { return ctx_; }
    void    check(int rc, const char *what) const
    {
      if (rc != MI_OK)
        throw Error(rc, std::string(what) + ": " + mi_last_error(ctx_));
    }

  private:
    // rank 0 writes the 128-byte RCCL id (to a temporary name, then renamed: never seen half written), the others
    // poll for it for up to two minutes
    static void exchange_unique_id(const std::string &file, int rank, unsigned char *uid)
    {
      const size_t nb = 128;
      if (rank == 0)
        {
          if (mi_comm_unique_id(uid) != MI_OK)
            throw Error(MI_EINVAL, std::string("mi_comm_unique_id: ") + mi_last_error(nullptr));
          const std::string tmp = file + ".tmp";
          std::FILE        *f   = std::fopen(tmp.c_str(), "wb");
          if (!f || std::fwrite(uid, 1, nb, f) != nb)
            throw Error(MI_EINVAL, "cannot write " + tmp);
          std::fclose(f);
          if (std::rename(tmp.c_str(), file.c_str()))
            throw Error(MI_EINVAL, "cannot rename " + tmp);
          return;
        }
      for (int tries = 0; tries < 2400; ++tries)
        {
          if (std::FILE *f = std::fopen(file.c_str(), "rb"))
            {
              const size_t got = std::fread(uid, 1, nb, f);
              std::fclose(f);
              if (got == nb)
                return;
            }
          std::this_thread::sleep_for(std::chrono::milliseconds(50));
        }
      throw Error(MI_EINVAL, "no RCCL id in " + file + " after two minutes (is rank 0 running?)");
    }
    mi_ctx       *ctx_ = nullptr;
    unsigned char uid_[128] = {0};
  };

  class Vector
  {
  public:
    Vector() = default; // empty snapshot, storage created on first assignment
    Vector(const Device &dev, int which)
      : dev_(&dev)
      , which_(which)
    {}
    Vector(const Vector &o) { *this = o; }
    Vector(Vector &&o) noexcept
      : dev_(o.dev_)
      , which_(o.which_)
      , snap_(o.snap_)
    {
      o.snap_ = nullptr;
    }
    Vector &operator=(Vector &&o) noexcept
    {
      if (this != &o)
        {
          release();
          dev_    = o.dev_;
          which_  = o.which_;
          snap_   = o.snap_;
          o.snap_ = nullptr;
        }
      return *this;
    }
    ~Vector() { release(); }

    // make this handle refer to state vector `which` of the device context (no data is copied)
    void bind(const Device &dev, int which)
    {
      release();
      dev_   = &dev;
      which_ = which;
    }

    Vector &operator=(const Vector &o)
    {
      if (this == &o || !o.dev_)
        return *this;
      if (!dev_)
        dev_ = o.dev_;
      if (which_ < 0 && !snap_)
        dev_->check(mi_snapshot_create(dev_->ctx(), &snap_), "mi_snapshot_create");
      if (which_ >= 0 && o.which_ < 0) // state vector := snapshot
        dev_->check(mi_snapshot_load(dev_->ctx(), o.snap_, which_), "mi_snapshot_load");
      else if (which_ < 0 && o.which_ >= 0) // snapshot := state vector
        dev_->check(mi_snapshot_store(dev_->ctx(), snap_, o.which_), "mi_snapshot_store");
      else
        throw std::logic_error("mi::Vector: only state <-> snapshot assignments are supported");
      return *this;
    }

    bool is_state() const { return which_ >= 0; }
    int  id() const { return which_; }
    const Device &device() const { return *dev_; }

    // interface-sized transfers (format_deal_to_precice / format_precice_to_deal, adapter.h:389-443)
    void gather_interface(std::vector<double> &out, int n_nodes) const
    {
      if (which_ != MI_V_TOTAL_DISPLACEMENT)
        throw std::logic_error("only the total displacement is written to the coupling interface");
      dev_->check(mi_get_interface_displacement(dev_->ctx(), n_nodes, out.data()), "mi_get_interface_displacement");
    }
    void scatter_interface(const std::vector<double> &in, int n_nodes)
    {
      if (which_ != MI_V_EXTERNAL_STRESS)
        throw std::logic_error("coupling data is read into the external stress vector");
      dev_->check(mi_set_interface_traction(dev_->ctx(), n_nodes, in.data()), "mi_set_interface_traction");
    }

  private:
    void release()
    {
      if (snap_ && dev_)
        mi_snapshot_destroy(dev_->ctx(), snap_);
      snap_ = nullptr;
    }
    const Device *dev_   = nullptr;
    int           which_ = -1;
    mi_snapshot  *snap_  = nullptr;
  };
} // namespace mi
