// Stand-in for <ros/ros.h> so that the reference's solve-path sources compile into oracle/_ref without ROS
// (TEST INFRASTRUCTURE, see oracle/Makefile.ref).  Nothing here computes: logging macros print to stderr only when
// DFTPAV_REF_VERBOSE is defined, publishers swallow their messages, the clock reads the steady clock.
#pragma once
#include <chrono>
#include <cstdio>
#include <memory>
#include <string>
namespace ros {
struct Duration {
  double s = 0.0;
  double toSec() const { return s; }
};
struct Time {
  double s = 0.0;
  Time() = default;
  explicit Time(double t) : s(t) {}
  static Time now() {
    return Time(std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count());
  }
  double toSec() const { return s; }
  Duration operator-(const Time &o) const { return Duration{s - o.s}; }
};
struct Publisher {
  template <typename M> void publish(const M &) const {}
};
struct NodeHandle {
  NodeHandle() = default;
  explicit NodeHandle(const std::string &) {}
  template <typename M> Publisher advertise(const std::string &, int, bool = false) { return Publisher(); }
  template <typename T> bool getParam(const std::string &, T &) const { return false; }
  template <typename T> void param(const std::string &, T &v, const T &d) const { v = d; }
};
inline void shutdown() {}
inline bool ok() { return true; }
}  // namespace ros
#ifdef DFTPAV_REF_VERBOSE
#define DFTPAV_REF_LOG(tag, ...) do { std::fprintf(stderr, "[ref %s] ", tag); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#else
#define DFTPAV_REF_LOG(tag, ...) do { if (0) std::fprintf(stderr, __VA_ARGS__); } while (0)
#endif
#define ROS_ERROR(...) DFTPAV_REF_LOG("error", __VA_ARGS__)
#define ROS_WARN(...) DFTPAV_REF_LOG("warn", __VA_ARGS__)
#define ROS_INFO(...) DFTPAV_REF_LOG("info", __VA_ARGS__)
#define ROS_DEBUG(...) DFTPAV_REF_LOG("debug", __VA_ARGS__)
#define ROS_ERROR_STREAM(x) do { } while (0)
#define ROS_WARN_STREAM(x) do { } while (0)
#define ROS_INFO_STREAM(x) do { } while (0)
