// TEST STAND-IN (tests/test_ros_adapter_syntax.py only): the sliver of roscpp HipLeggedController touches, declarations only.
#pragma once
#include <cstdio>
#include <memory>
#include <string>
namespace ros {
struct Duration {
  Duration() {}
  explicit Duration(double s) : sec_(s) {}
  double toSec() const { return sec_; }
  Duration& fromSec(double s) { sec_ = s; return *this; }
  double sec_ = 0.0;
};
struct Time {
  Time() {}
  double toSec() const { return sec_; }
  Time operator-(const Duration& d) const { Time t; t.sec_ = sec_ - d.sec_; return t; }
  double sec_ = 0.0;
};
struct Subscriber {};
class NodeHandle {
 public:
  template <class T> bool getParam(const std::string&, T&) const { return false; }
  template <class M, class C> Subscriber subscribe(const std::string&, int, void (C::*)(const typename M::ConstPtr&), C*) { return Subscriber(); }
};
}  // namespace ros
#define ROS_ERROR(...) std::fprintf(stderr, __VA_ARGS__)
