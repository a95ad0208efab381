// DECLARATIONS ONLY -- test infrastructure for tests/test_host_cpu.py::test_host_compiles_against_precice_v3_api.
//
// The public C++ API of preCICE v3 (precice/Participant.hpp, precice/span.hpp of libprecice 3.0, the version the
// reference pins: CMakeLists.txt:54) restricted to the 14 calls the reference's Adapter and solvers make
// (adapter.h:213-489, nonlinear_elasticity.cc:115-166).  Nothing here is implemented and nothing links against
// it: compiling the host sources with -DMI_WITH_PRECICE against these declarations checks that every call site
// type-checks against the real library's signatures (string_view = span<const char>, span<const double>,
// span<VertexID>, const-qualification), so that switching from the replay participant to libprecice is a link
// flag and not a port.
#pragma once
#include <cstddef>
#include <cstring>
#include <type_traits>

namespace precice
{
  using VertexID = int;

  template <typename T>
  class span
  {
  public:
    using element_type = T;
    constexpr span() noexcept = default;
    constexpr span(T *p, std::size_t n) noexcept
      : p_(p)
      , n_(n)
    {}
    // contiguous containers (std::vector, std::string, std::array): data()/size(), qualification conversions only
    template <typename C, typename = std::enable_if_t<std::is_convertible<
                            std::remove_pointer_t<decltype(std::declval<C &>().data())> (*)[], T (*)[]>::value>>
    constexpr span(C &c) noexcept
      : p_(c.data())
      , n_(c.size())
    {}
    // string literals / C strings for string_view
    template <typename U = T, typename = std::enable_if_t<std::is_same<U, const char>::value>>
    span(const char *s) noexcept
      : p_(s)
      , n_(std::strlen(s))
    {}
    constexpr T          *data() const noexcept { return p_; }
    constexpr std::size_t size() const noexcept { return n_; }
    constexpr T          &operator[](std::size_t i) const { return p_[i]; }

  private:
    T          *p_ = nullptr;
    std::size_t n_ = 0;
  };

  using string_view = span<const char>;

  class Participant
  {
  public:
    Participant(::precice::string_view participantName, ::precice::string_view configurationFileName,
                int solverProcessIndex, int solverProcessSize);
    ~Participant();
    void   initialize();
    void   advance(double computedTimeStepSize);
    void   finalize();
    int    getMeshDimensions(::precice::string_view meshName) const;
    bool   isCouplingOngoing() const;
    bool   isTimeWindowComplete() const;
    double getMaxTimeStepSize() const;
    bool   requiresInitialData();
    bool   requiresWritingCheckpoint();
    bool   requiresReadingCheckpoint();
    void   setMeshVertices(::precice::string_view meshName, ::precice::span<const double> coordinates,
                           ::precice::span<VertexID> ids);
    void   writeData(::precice::string_view meshName, ::precice::string_view dataName,
                     ::precice::span<const VertexID> ids, ::precice::span<const double> values);
    void   readData(::precice::string_view meshName, ::precice::string_view dataName,
                    ::precice::span<const VertexID> ids, double relativeReadTime, ::precice::span<double> values) const;
  };
} // namespace precice
