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
// collective.  One rank (the reference's situation, and MI_SLABS emulation): a plain forwarder.
#pragma once
#include <cstddef>
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

    int getMeshDimensions(const std::string &mesh) const { return int(shared(rank_ == 0 ? double(impl_->getMeshDimensions(mesh)) : 0.0)); }
    bool requiresInitialData() { return shared(rank_ == 0 ? double(impl_->requiresInitialData()) : 0.0) != 0.0; }
    bool requiresWritingCheckpoint() { return shared(rank_ == 0 ? double(impl_->requiresWritingCheckpoint()) : 0.0) != 0.0; }
    bool requiresReadingCheckpoint() { return shared(rank_ == 0 ? double(impl_->requiresReadingCheckpoint()) : 0.0) != 0.0; }
    bool isCouplingOngoing() const { return shared(rank_ == 0 ? double(impl_->isCouplingOngoing()) : 0.0) != 0.0; }
    bool isTimeWindowComplete() const { return shared(rank_ == 0 ? double(impl_->isTimeWindowComplete()) : 0.0) != 0.0; }
    double getMaxTimeStepSize() const { return shared(rank_ == 0 ? impl_->getMaxTimeStepSize() : 0.0); }

    void setMeshVertices(const std::string &mesh, const std::vector<double> &positions, std::vector<int> &ids)
    {
      if (rank_ == 0)
        impl_->setMeshVertices(mesh, positions, ids);
      else
        for (std::size_t i = 0; i < ids.size(); ++i)
          ids[i] = int(i); // never handed to a participant
    }
    void writeData(const std::string &mesh, const std::string &data, const std::vector<int> &ids, const std::vector<double> &values)
    {
      if (rank_ == 0)
        impl_->writeData(mesh, data, ids, values);
    }
    void readData(const std::string &mesh, const std::string &data, const std::vector<int> &ids, double relative_read_time,
                  std::vector<double> &values) const
    {
      if (rank_ == 0)
        impl_->readData(mesh, data, ids, relative_read_time, values);
      if (size_ > 1 && !values.empty())
        share(values.data(), int(values.size()));
    }
    void initialize()
    {
      if (rank_ == 0)
        impl_->initialize();
    }
    void advance(double dt)
    {
      if (rank_ == 0)
        impl_->advance(dt);
    }
    void finalize()
    {
      if (rank_ == 0)
        impl_->finalize();
    }

  private:
    void share(double *v, int n) const
    {
      if (!bcast_)
        throw std::logic_error("RankZeroParticipant: several ranks but no broadcast bound (Adapter::initialize binds it)");
      bcast_(v, n);
    }
    double shared(double v) const
    {
      if (size_ > 1)
        share(&v, 1);
      return v;
    }
    const int                             rank_, size_;
    std::unique_ptr<precice::Participant> impl_;
    Broadcast                             bcast_;
  };
} // namespace Adapter
