// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for the sliver of roscpp that legged_estimation's sources touch:
// Time / Duration arithmetic, NodeHandle::subscribe (callbacks are dropped: the golden vectors drive update() directly).
#pragma once
#include <cassert>
#include <deque>
#include <functional>
#include <memory>
#include <string>
namespace ros {
struct Duration {
  double s = 0.0;
  Duration() = default;
  explicit Duration(double v) : s(v) {}
  double toSec() const { return s; }
};
struct Time {
  double t = 0.0;
  Time() = default;
  explicit Time(double v) : t(v) {}
  double toSec() const { return t; }
  static Time now() { return Time(); }
};
inline Time operator+(const Time& a, const Duration& d) { return Time(a.t + d.s); }
inline bool operator<(const Time& a, const Time& b) { return a.t < b.t; }
struct Subscriber {};
struct Publisher { template <class M> void publish(const M&) const {} };
class NodeHandle {
 public:
  NodeHandle() = default;
  explicit NodeHandle(const std::string&) {}
  template <class M, class T> Subscriber subscribe(const std::string&, int, void (T::*)(const typename M::ConstPtr&), T*) { return Subscriber(); }
  template <class M, class F> Subscriber subscribe(const std::string&, int, F) { return Subscriber(); }
  template <class M> Publisher advertise(const std::string&, int, bool = false) { return Publisher(); }
  template <class T> bool getParam(const std::string&, T&) const { return false; }
};
}  // namespace ros
#define ROS_WARN(...) ((void)0)
#define ROS_INFO(...) ((void)0)
#define ROS_ERROR(...) ((void)0)
#define ROS_WARN_STREAM(x) ((void)0)
#define ROS_INFO_STREAM(x) ((void)0)
#define ROS_ERROR_STREAM(x) ((void)0)
