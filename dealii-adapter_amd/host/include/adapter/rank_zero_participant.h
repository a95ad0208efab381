// RankZeroParticipant -- the ONE preCICE-facing process of a run on several GPUs.
//
// The reference constructs precice::Participant(name, config, 0, 1) (include/adapter/adapter.h:213-225: "this_mpi_process
// = 0, n_mpi_processes = 1", :152-154) and makes the 14 calls of SURVEY.md section 8b on it.  With the mesh spread over N
// ranks that object exists on rank 0 ONLY: N processes that each registered as the single rank of participant "Solid"
// would be N participants to the coupling library.  This class has the surface the Adapter and the solvers use
// (`adapter.precice.isCouplingOngoing()`, ...) and
//   * forwards every call to the real participant on rank 0;
//   * hands what rank 0 learnt to the other ranks -- mesh dimension, requiresInitialData, read data (:346-361),
//     isCouplingOngoing, getMaxTimeStepSize, isTimeWindowComplete, the checkpoint requests (:447-489) -- through a
//     broadcast the solver binds (mi_comm_broadcast: the library's own RCCL communicator, include/mi_elasticity.h);
//   * makes the write-only calls (setMeshVertices, writeData, initialize, advance, finalize) no-ops on ranks > 0: every
//     rank holds the whole interface (the library gathers it), so rank 0 has all there is to write.
// The ranks stay in step because every rank makes the same sequence of calls and each value-returning call is a
// collective -- which also carries rank 0's status, so that an error of the coupling library ends EVERY rank (below).  One rank (the reference's situation, and MI_SLABS emulation): a plain forwarder.
#pragma once
#include <algorithm>
#include <cstddef>
#include <exception>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "precice_participant.h"

namespace Adapter
{
  class RankZeroParticipant
  {
  public:
    using Broadcast = std::function<void(double *, int)>; // values of rank 0 to all ranks (collective)

    RankZeroParticipant(const std::string &participant_name, const std::string &config_file, int rank, int size)
      : rank_(rank)
      , size_(size)
    {
      if (rank_ == 0)
        impl_ = std::make_unique<precice::Participant>(participant_name, config_file, 0, 1); // adapter.h:217-220
    }

    // before the first value-returning call of a run with several ranks (Adapter::initialize does it)
    void bind(Broadcast b) { bcast_ = std::move(b); }
    int  rank() const { return rank_; }
    int  size() const { return size_; }

    int getMeshDimensions(const std::string &mesh) const { return int(shared([&] { return double(impl_->getMeshDimensions(mesh)); })); }
    bool requiresInitialData() { return shared([&] { return double(impl_->requiresInitialData()); }) != 0.0; }
    bool requiresWritingCheckpoint() { return shared([&] { return double(impl_->requiresWritingCheckpoint()); }) != 0.0; }
    bool requiresReadingCheckpoint() { return shared([&] { return double(impl_->requiresReadingCheckpoint()); }) != 0.0; }
    bool isCouplingOngoing() const { return shared([&] { return double(impl_->isCouplingOngoing()); }) != 0.0; }
    bool isTimeWindowComplete() const { return shared([&] { return double(impl_->isTimeWindowComplete()); }) != 0.0; }
    double getMaxTimeStepSize() const { return shared([&] { return impl_->getMaxTimeStepSize(); }); }

    void setMeshVertices(const std::string &mesh, const std::vector<double> &positions, std::vector<int> &ids)
    {
      if (rank_ == 0)
        guarded([&] { impl_->setMeshVertices(mesh, positions, ids); });
      else
        for (std::size_t i = 0; i < ids.size(); ++i)
          ids[i] = int(i); // never handed to a participant
    }
    void writeData(const std::string &mesh, const std::string &data, const std::vector<int> &ids, const std::vector<double> &values)
    {
      if (rank_ == 0)
        guarded([&] { impl_->writeData(mesh, data, ids, values); });
    }
    void readData(const std::string &mesh, const std::string &data, const std::vector<int> &ids, double relative_read_time,
                  std::vector<double> &values) const
    {
      if (rank_ == 0)
        guarded([&] { impl_->readData(mesh, data, ids, relative_read_time, values); });
      if (size_ > 1)
        {
          // the values and, behind them, the status word of rank 0
          std::vector<double> buf(values.size() + 1);
          std::copy(values.begin(), values.end(), buf.begin());
          buf.back() = pending_ ? 1.0 : 0.0;
          share(buf.data(), int(buf.size()));
          std::copy(buf.begin(), buf.end() - 1, values.begin());
          leave_if_failed(buf.back() != 0.0);
        }
    }
    void initialize()
    {
      if (rank_ == 0)
        guarded([&] { impl_->initialize(); });
    }
    void advance(double dt)
    {
      if (rank_ == 0)
        guarded([&] { impl_->advance(dt); });
    }
    void finalize()
    {
      // the last call of a run: no collective follows, so an error of rank 0 (also one still pending) surfaces here
      if (rank_ == 0)
        {
          guarded([&] { impl_->finalize(); });
          if (pending_)
            leave_if_failed(true);
        }
    }

  private:
    // Failure path (several ranks).  An exception of the coupling library on rank 0 must not leave the other ranks
    // blocked in the next collective: rank 0 catches it, makes every later forwarding call a no-op, and the next
    // value-returning call -- a collective every rank takes part in -- carries a status word next to its value.  Every
    // rank then leaves through an exception: rank 0 with the original one, the others with a message that names it, so
    // that each process ends in main's "Exception on processing" block with exit code 1 (elasticity.cc:101-126).
    // One rank: a plain forwarder, exceptions pass through untouched.
    template <class F>
    void guarded(F &&f) const
    {
      if (size_ == 1)
        {
          f();
          return;
        }
      if (pending_)
        return;
      try
        {
          f();
        }
      catch (...)
        {
          pending_ = std::current_exception();
        }
    }
    void leave_if_failed(const bool failed) const
    {
      if (!failed)
        return;
      if (rank_ == 0 && pending_)
        {
          const std::exception_ptr e = pending_;
          pending_                   = nullptr;
          std::rethrow_exception(e);
        }
      throw std::runtime_error("the preCICE-facing rank (rank 0) stopped with an error in the coupling library; rank " +
                               std::to_string(rank_) + " stops with it");
    }
    void share(double *v, int n) const
    {
      if (!bcast_)
        throw std::logic_error("RankZeroParticipant: several ranks but no broadcast bound (Adapter::initialize binds it)");
      bcast_(v, n);
    }
    template <class F>
    double shared(F &&f) const
    {
      double buf[2] = {0.0, 0.0}; // value, status of rank 0
      if (rank_ == 0)
        guarded([&] { buf[0] = f(); });
      if (size_ > 1)
        {
          buf[1] = pending_ ? 1.0 : 0.0;
          share(buf, 2);
          leave_if_failed(buf[1] != 0.0);
        }
      return buf[0];
    }
    const int                             rank_, size_;
    std::unique_ptr<precice::Participant> impl_;
    Broadcast                             bcast_;
    mutable std::exception_ptr            pending_; // rank 0: what the coupling library threw, until every rank has been told
  };
} // namespace Adapter
