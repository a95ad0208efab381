// Test double of libprecice (v3 C++ API, the 14 calls the reference makes) -- TEST INFRASTRUCTURE.
//
// libprecice is not available in this environment, so the host's `-DMI_WITH_PRECICE` build (which includes
// <precice/precice.hpp> instead of the replay participant) could only be type-checked.  This library implements the
// declarations of tests/precice_api/precice/precice.hpp -- the real library's signatures: string_view =
// span<const char>, span<const double>, span<VertexID>, const-qualification -- by forwarding to the replay participant,
// so that the MI_WITH_PRECICE build can be LINKED and RUN: same directives in the configuration file, same coupling
// behaviour, and therefore bit-identical results with the default build (tests/test_host_gpu.py).
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <regex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

// the replay participant under another namespace name (its header is written for namespace precice)
#define precice replay_precice
#include "../../dealii-adapter_amd/host/include/adapter/precice_participant.h"
#undef precice

#include <precice/precice.hpp>

namespace
{
  std::mutex                                                                  g_mutex;
  std::map<const precice::Participant *, std::unique_ptr<replay_precice::Participant>> g_impl;
  replay_precice::Participant &impl(const precice::Participant *p)
  {
    std::lock_guard<std::mutex> lk(g_mutex);
    return *g_impl.at(p);
  }
  std::string str(precice::string_view s)
  {
    return std::string(s.data(), s.size());
  }
} // namespace

namespace precice
{
  Participant::Participant(string_view participantName, string_view configurationFileName, int solverProcessIndex,
                           int solverProcessSize)
  {
    auto                        obj = std::make_unique<replay_precice::Participant>(str(participantName), str(configurationFileName),
                                                             solverProcessIndex, solverProcessSize);
    std::lock_guard<std::mutex> lk(g_mutex);
    g_impl[this] = std::move(obj);
    // tests: how many participants this PROCESS has constructed so far (a multi-rank run must construct exactly one)
    static int constructed = 0;
    ++constructed;
    if (const char *f = std::getenv("MI_FAKE_PRECICE_COUNT"))
      std::ofstream(f) << constructed << "\n";
  }
  Participant::~Participant()
  {
    std::lock_guard<std::mutex> lk(g_mutex);
    g_impl.erase(this);
  }
  void Participant::initialize()
  {
    impl(this).initialize();
  }
  void Participant::advance(double dt)
  {
    impl(this).advance(dt);
  }
  void Participant::finalize()
  {
    impl(this).finalize();
  }
  int Participant::getMeshDimensions(string_view meshName) const
  {
    return impl(this).getMeshDimensions(str(meshName));
  }
  bool Participant::isCouplingOngoing() const
  {
    return impl(this).isCouplingOngoing();
  }
  bool Participant::isTimeWindowComplete() const
  {
    return impl(this).isTimeWindowComplete();
  }
  double Participant::getMaxTimeStepSize() const
  {
    return impl(this).getMaxTimeStepSize();
  }
  bool Participant::requiresInitialData()
  {
    return impl(this).requiresInitialData();
  }
  bool Participant::requiresWritingCheckpoint()
  {
    return impl(this).requiresWritingCheckpoint();
  }
  bool Participant::requiresReadingCheckpoint()
  {
    return impl(this).requiresReadingCheckpoint();
  }
  void Participant::setMeshVertices(string_view meshName, span<const double> coordinates, span<VertexID> ids)
  {
    impl(this).setMeshVertices(str(meshName), replay_precice::span<const double>(coordinates.data(), coordinates.size()),
                               replay_precice::span<int>(ids.data(), ids.size()));
  }
  void Participant::writeData(string_view meshName, string_view dataName, span<const VertexID> ids, span<const double> values)
  {
    impl(this).writeData(str(meshName), str(dataName), replay_precice::span<const int>(ids.data(), ids.size()),
                         replay_precice::span<const double>(values.data(), values.size()));
  }
  void Participant::readData(string_view meshName, string_view dataName, span<const VertexID> ids, double relativeReadTime,
                             span<double> values) const
  {
    impl(this).readData(str(meshName), str(dataName), replay_precice::span<const int>(ids.data(), ids.size()), relativeReadTime,
                        replay_precice::span<double>(values.data(), values.size()));
  }
} // namespace precice
