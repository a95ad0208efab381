// precice::Participant as seen by the Adapter.
//
// The reference links libprecice >= 3.0 (CMakeLists.txt:54) and uses exactly the 14 calls below
// (adapter.h:213-489, nonlinear_elasticity.cc:115-166).  libprecice is not available in this build
// environment, so unless MI_WITH_PRECICE is defined (then <precice/precice.hpp> is used unchanged) this header
// provides a *replay participant* with the same call surface: it plays the role of the other solver from a
// script embedded in the configuration file and records what the solid writes.
//
// Replay directives are XML comments inside the preCICE configuration file, so a real config stays valid:
//   <!-- replay: read-data = ramp 10 0 -2000 0 -->      traction vector, linear ramp over 10 windows
//   <!-- replay: read-data = constant 0 -40 0 -->       constant traction vector
//   <!-- replay: read-data = trace forces.txt -->       rows "t fx fy fz", linear interpolation in t
//   <!-- replay: read-data = vertex-trace forces.txt --> a recorded per-vertex trace (e.g. the fluid forces of a Turek-Hron
//                                                       FSI3 run): rows "t vertex fx fy [fz]" (dimensions values), vertex =
//                                                       the id handed out by setMeshVertices (position order); every time
//                                                       frame lists every vertex; linear interpolation in t per vertex
//   <!-- replay: write-vertices = vertices.txt -->      "vertex x y [z]" of the coupling mesh, for producing such traces
//   <!-- replay: iterations = 3 -->                     coupling iterations per window (implicit schemes)
//   <!-- replay: write-log = solid-displacement.log --> one row per completed window: t, then all written values
// Scheme, dimensions, time-window-size and max-time (or max-time-windows) are read from the usual tags.
// Iteration i < last of an implicit window receives the window's traction scaled by (1 - 2^-(i+1)), the last
// one the unscaled value -- a deterministic stand-in for a converging fixed-point iteration.
#pragma once

#ifdef MI_WITH_PRECICE
#include <precice/precice.hpp>
#else

#include <cmath>
#include <cstdlib>
#include <cstddef>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <regex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace precice
{
  // minimal stand-in for precice::span (std::span needs C++20)
  template <typename T>
  class span
  {
  public:
    span(T *p, std::size_t n)
      : p_(p)
      , n_(n)
    {}
    template <typename V, typename = decltype(std::declval<V &>().data())>
    span(V &v)
      : p_(v.data())
      , n_(v.size())
    {}
    T          *data() const { return p_; }
    std::size_t size() const { return n_; }
    T          &operator[](std::size_t i) const { return p_[i]; }

  private:
    T          *p_;
    std::size_t n_;
  };

  // rank of this process in a decomposed run of the executables (mi/device_vector.h): ranks > 0 write no files
  inline int quiet_rank()
  {
    const char *w = std::getenv("MI_WORLD_SIZE"), *r = std::getenv("MI_RANK");
    return (w && std::atoi(w) > 1 && r) ? std::atoi(r) : 0;
  }

  class Participant
  {
  public:
    Participant(const std::string &participantName, const std::string &configurationFileName,
                int solverProcessIndex, int solverProcessSize)
      : name_(participantName)
    {
      if (solverProcessIndex != 0 || solverProcessSize != 1)
        throw std::runtime_error("replay participant: only one solver process is supported");
      std::ifstream in(configurationFileName);
      if (!in)
        throw std::runtime_error("preCICE configuration file <" + configurationFileName + "> not found");
      std::stringstream ss;
      ss << in.rdbuf();
      const std::string xml = ss.str();
      auto              attr = [&](const std::string &re, double fallback) {
        std::smatch m;
        return std::regex_search(xml, m, std::regex(re)) ? std::stod(m[1]) : fallback;
      };
      dims_        = int(attr("dimensions\\s*=\\s*\"(\\d)\"", 2));
      window_      = attr("<time-window-size\\s+value\\s*=\\s*\"([-+0-9.eE]+)\"", 0.0);
      max_time_    = attr("<max-time\\s+value\\s*=\\s*\"([-+0-9.eE]+)\"", -1.0);
      max_windows_ = long(attr("<max-time-windows\\s+value\\s*=\\s*\"(\\d+)\"", -1.0));
      implicit_    = std::regex_search(xml, std::regex("coupling-scheme:(serial|parallel)-implicit"));
      iterations_  = implicit_ ? int(attr("<max-iterations\\s+value\\s*=\\s*\"(\\d+)\"", 2)) : 1;
      if (!(window_ > 0))
        throw std::runtime_error("replay participant: <time-window-size value=...> missing in " + configurationFileName);
      const std::regex directive("replay:\\s*([a-z-]+)\\s*=\\s*([^\\n>]*?)\\s*(-->|\\n)");
      for (auto it = std::sregex_iterator(xml.begin(), xml.end(), directive); it != std::sregex_iterator(); ++it)
        {
          const std::string key = (*it)[1], val = (*it)[2];
          std::stringstream v(val);
          if (key == "read-data")
            {
              v >> mode_;
              if (mode_ == "constant")
                v >> base_[0] >> base_[1] >> base_[2];
              else if (mode_ == "ramp")
                v >> ramp_ >> base_[0] >> base_[1] >> base_[2];
              else if (mode_ == "vertex-trace")
                {
                  std::string file;
                  v >> file;
                  std::ifstream tf(file);
                  if (!tf)
                    throw std::runtime_error("replay participant: trace file <" + file + "> not found");
                  std::string row;
                  while (std::getline(tf, row))
                    {
                      if (row.empty() || row[0] == '#')
                        continue;
                      std::stringstream r(row);
                      VertexSample      p{};
                      r >> p.t >> p.vertex;
                      for (int d = 0; d < dims_; ++d)
                        r >> p.f[d];
                      if (!r)
                        throw std::runtime_error("replay participant: malformed row in <" + file + ">: " + row);
                      vertex_rows_.push_back(p);
                    }
                  if (vertex_rows_.empty())
                    throw std::runtime_error("replay participant: empty trace file <" + file + ">");
                }
              else if (mode_ == "trace")
                {
                  std::string file;
                  v >> file;
                  std::ifstream tf(file);
                  if (!tf)
                    throw std::runtime_error("replay participant: trace file <" + file + "> not found");
                  std::string row;
                  while (std::getline(tf, row))
                    {
                      if (row.empty() || row[0] == '#')
                        continue;
                      std::stringstream r(row);
                      TracePoint        p{};
                      r >> p.t >> p.f[0] >> p.f[1] >> p.f[2];
                      trace_.push_back(p);
                    }
                  if (trace_.empty())
                    throw std::runtime_error("replay participant: empty trace file <" + file + ">");
                }
              else
                throw std::runtime_error("replay participant: unknown read-data mode <" + mode_ + ">");
            }
          else if (key == "iterations")
            {
              v >> iterations_;
              if (!implicit_)
                iterations_ = 1;
            }
          else if (key == "write-log")
            v >> log_file_;
          else if (key == "write-vertices")
            v >> vertices_file_;
          else
            throw std::runtime_error("replay participant: unknown directive <" + key + ">");
        }
      if (iterations_ < 1)
        iterations_ = 1;
    }

    int getMeshDimensions(const std::string & /*meshName*/) const { return dims_; }

    void setMeshVertices(const std::string & /*meshName*/, span<const double> positions, span<int> ids)
    {
      if (positions.size() != ids.size() * std::size_t(dims_))
        throw std::runtime_error("setMeshVertices: positions/ids size mismatch");
      n_vertices_ = ids.size();
      positions_.assign(positions.data(), positions.data() + positions.size());
      for (std::size_t i = 0; i < ids.size(); ++i)
        ids[i] = int(i);
      if (!vertices_file_.empty() && quiet_rank() == 0)
        {
          std::ofstream vf(vertices_file_);
          vf << "# vertex x y" << (dims_ == 3 ? " z" : "") << "\n" << std::setprecision(17);
          for (std::size_t i = 0; i < ids.size(); ++i)
            {
              vf << i;
              for (int d = 0; d < dims_; ++d)
                vf << ' ' << positions[i * dims_ + d];
              vf << '\n';
            }
        }
      if (mode_ == "vertex-trace")
        build_vertex_frames();
    }

    bool requiresInitialData() { return false; }

    void writeData(const std::string & /*meshName*/, const std::string & /*dataName*/, span<const int> ids,
                   span<const double> values)
    {
      if (values.size() != ids.size() * std::size_t(dims_))
        throw std::runtime_error("writeData: values/ids size mismatch");
      last_written_.assign(values.data(), values.data() + values.size());
    }

    void initialize()
    {
      initialized_   = true;
      need_write_cp_ = implicit_;
      if (!log_file_.empty())
        {
          if (quiet_rank() == 0) // decomposed runs: every rank replays the same partner, rank 0 keeps the log
            log_.open(log_file_);
          log_ << "# replay participant for " << name_ << ": t, then " << n_vertices_ * dims_
               << " written values per completed window\n";
        }
    }

    void readData(const std::string & /*meshName*/, const std::string & /*dataName*/, span<const int> ids,
                  double relativeReadTime, span<double> values) const
    {
      if (values.size() != ids.size() * std::size_t(dims_))
        throw std::runtime_error("readData: values/ids size mismatch");
      const double scale = (implicit_ && iteration_ + 1 < iterations_) ? 1.0 - std::pow(0.5, iteration_ + 1) : 1.0;
      if (mode_ == "vertex-trace")
        {
          // per-vertex linear interpolation between the two recorded frames around the read time
          const double t  = time_ + relativeReadTime;
          std::size_t  hi = 0;
          while (hi < frame_t_.size() && frame_t_[hi] < t)
            ++hi;
          const std::size_t a = hi == 0 ? 0 : hi - 1, b = hi < frame_t_.size() ? hi : frame_t_.size() - 1;
          const double      w = (a == b || hi == 0) ? (hi == 0 ? 1.0 : 0.0) : (t - frame_t_[a]) / (frame_t_[b] - frame_t_[a]);
          const std::size_t stride = n_vertices_ * std::size_t(dims_);
          for (std::size_t i = 0; i < ids.size(); ++i)
            {
              const std::size_t vtx = std::size_t(ids[i]);
              if (vtx >= n_vertices_)
                throw std::runtime_error("readData: vertex id outside the coupling mesh");
              for (int d = 0; d < dims_; ++d)
                values[i * dims_ + d] = scale * ((1 - w) * frames_[a * stride + vtx * dims_ + d] +
                                                 w * frames_[b * stride + vtx * dims_ + d]);
            }
          return;
        }
      double f[3];
      traction_at(time_ + relativeReadTime, f);
      for (std::size_t i = 0; i < ids.size(); ++i)
        for (int d = 0; d < dims_; ++d)
          values[i * dims_ + d] = scale * f[d];
    }

    void advance(double computedTimeStepSize)
    {
      if (!initialized_)
        throw std::runtime_error("advance() called before initialize()");
      if (std::abs(computedTimeStepSize - window_) > 1e-10)
        throw std::runtime_error("replay participant: sub-cycling is not supported (dt != time-window-size)");
      if (implicit_ && iteration_ + 1 < iterations_)
        {
          ++iteration_;
          need_read_cp_    = true;
          window_complete_ = false;
          return;
        }
      iteration_       = 0;
      window_complete_ = true;
      need_write_cp_   = implicit_;
      time_ += window_;
      ++windows_done_;
      if (log_.is_open())
        {
          log_ << std::setprecision(17) << time_;
          for (double v : last_written_)
            log_ << ' ' << v;
          log_ << '\n';
        }
    }

    bool requiresWritingCheckpoint()
    {
      const bool r   = need_write_cp_;
      need_write_cp_ = false;
      return r;
    }
    bool requiresReadingCheckpoint()
    {
      const bool r  = need_read_cp_;
      need_read_cp_ = false;
      return r;
    }

    bool isCouplingOngoing() const
    {
      if (max_windows_ >= 0 && windows_done_ >= max_windows_)
        return false;
      if (max_time_ >= 0 && time_ >= max_time_ - 1e-12 * std::max(1.0, max_time_))
        return false;
      return max_windows_ >= 0 || max_time_ >= 0;
    }
    double getMaxTimeStepSize() const { return window_; }
    bool   isTimeWindowComplete() const { return window_complete_; }
    void   finalize()
    {
      if (log_.is_open())
        log_.close();
    }

    // replay-only accessors (not part of preCICE)
    const std::vector<double> &replay_last_written() const { return last_written_; }
    const std::vector<double> &replay_positions() const { return positions_; }

  private:
    struct TracePoint
    {
      double t, f[3];
    };
    struct VertexSample
    {
      double t;
      long   vertex;
      double f[3];
    };
    // rows of a vertex trace -> dense frames [time][vertex][dim]; every frame must list every vertex exactly once
    void build_vertex_frames()
    {
      frame_t_.clear();
      for (const VertexSample &p : vertex_rows_)
        if (frame_t_.empty() || p.t > frame_t_.back() + 1e-14 * std::max(1.0, std::abs(p.t)))
          frame_t_.push_back(p.t);
        else if (p.t < frame_t_.back() - 1e-14 * std::max(1.0, std::abs(p.t)))
          throw std::runtime_error("replay participant: vertex trace rows must be ordered by time");
      const std::size_t stride = n_vertices_ * std::size_t(dims_);
      frames_.assign(frame_t_.size() * stride, 0.0);
      std::vector<int> seen(frame_t_.size() * n_vertices_, 0);
      std::size_t      k = 0;
      for (const VertexSample &p : vertex_rows_)
        {
          while (p.t > frame_t_[k] + 1e-14 * std::max(1.0, std::abs(p.t)))
            ++k;
          if (p.vertex < 0 || std::size_t(p.vertex) >= n_vertices_)
            throw std::runtime_error("replay participant: vertex trace names a vertex outside the coupling mesh");
          for (int d = 0; d < dims_; ++d)
            frames_[k * stride + std::size_t(p.vertex) * dims_ + d] = p.f[d];
          seen[k * n_vertices_ + std::size_t(p.vertex)]++;
        }
      for (int c : seen)
        if (c != 1)
          throw std::runtime_error("replay participant: every frame of a vertex trace must list every vertex exactly once");
    }
    void traction_at(double t, double f[3]) const
    {
      f[0] = f[1] = f[2] = 0.0;
      if (mode_ == "constant")
        for (int d = 0; d < 3; ++d)
          f[d] = base_[d];
      else if (mode_ == "ramp")
        {
          const double s = std::min(1.0, std::max(0.0, t / (ramp_ * window_)));
          for (int d = 0; d < 3; ++d)
            f[d] = s * base_[d];
        }
      else if (mode_ == "trace")
        {
          if (t <= trace_.front().t)
            for (int d = 0; d < 3; ++d)
              f[d] = trace_.front().f[d];
          else if (t >= trace_.back().t)
            for (int d = 0; d < 3; ++d)
              f[d] = trace_.back().f[d];
          else
            for (std::size_t i = 1; i < trace_.size(); ++i)
              if (t <= trace_[i].t)
                {
                  const double w = (t - trace_[i - 1].t) / (trace_[i].t - trace_[i - 1].t);
                  for (int d = 0; d < 3; ++d)
                    f[d] = (1 - w) * trace_[i - 1].f[d] + w * trace_[i].f[d];
                  break;
                }
        }
    }

    std::string name_, mode_ = "constant", log_file_;
    int         dims_ = 2, iterations_ = 1, iteration_ = 0;
    double      window_ = 0, max_time_ = -1, time_ = 0, ramp_ = 1, base_[3] = {0, 0, 0};
    long        max_windows_ = -1, windows_done_ = 0;
    bool        implicit_ = false, initialized_ = false, need_write_cp_ = false, need_read_cp_ = false,
         window_complete_ = false;
    std::size_t             n_vertices_ = 0;
    std::vector<double>     positions_, last_written_;
    std::vector<TracePoint> trace_;
    std::vector<VertexSample> vertex_rows_;
    std::vector<double>       frame_t_, frames_;
    std::string               vertices_file_;
    std::ofstream           log_;
  };
} // namespace precice
#endif
