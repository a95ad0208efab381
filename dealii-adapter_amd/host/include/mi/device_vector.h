// mi::Device / mi::Vector -- thin C++ RAII layer over the C-ABI (include/mi_elasticity.h) for the host solvers.
//
// mi::Vector plays the role of the reference's VectorType (BlockVector<double>): a *handle* to a
// device-resident global vector.  Copy-assignment copies device-to-device, so the Adapter's checkpoint code
// (`old_state_data[i] = *state_variables[i]`, adapter.h:457-460) keeps its shape while the data never leaves HBM.
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "mi_elasticity.h"

namespace mi
{
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
      const int rc = mi_ctx_create(&mesh, &mat, &nm, device_id, nullptr, &ctx_);
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
    mi_ctx *ctx_ = nullptr;
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
