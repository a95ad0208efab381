// Adapter::Time -- step counter and absolute time with manual reset for implicit coupling / sub-cycling.
// Same interface and behaviour as the reference's include/adapter/time_handler.h:21-84, including the
// "round at 1e-10, then truncate to unsigned" rule of set_absolute_time (:63-70).
#pragma once
#include <cmath>

namespace Adapter
{
  class Time
  {
  public:
    Time(const double time_end, const double delta_t)
      : timestep_(0)
      , now_(0.0)
      , end_(time_end)
      , dt_(delta_t)
    {}
    virtual ~Time() = default;

    double       current() const { return now_; }
    double       end() const { return end_; }
    double       get_delta_t() const { return dt_; }
    unsigned int get_timestep() const { return timestep_; }

    // used by Adapter::reload_old_state_if_required to rewind after a rejected coupling iteration
    void set_absolute_time(const double new_time)
    {
      const double scale = 1e10;
      timestep_          = static_cast<unsigned int>(std::round((new_time / dt_) * scale) / scale);
      now_               = new_time;
    }

    void increment()
    {
      now_ += dt_;
      ++timestep_;
    }

  private:
    unsigned int timestep_;
    double       now_;
    const double end_;
    const double dt_;
  };
} // namespace Adapter
