// MOCK ros_control layer (tests only): the part of roscpp legged/HipLeggedController touches, functional enough to RUN the
// plugin without ROS — a parameter map behind NodeHandle::getParam, a topic registry that delivers messages to the subscribed
// callbacks, Time / Duration arithmetic.  tests/cpp/plugin_test.cpp drives the controller through it.
#pragma once
#include <cstdio>
#include <functional>
#include <map>
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
  explicit Time(double s) : sec_(s) {}
  double toSec() const { return sec_; }
  Time operator-(const Duration& d) const { return Time(sec_ - d.sec_); }
  Time operator+(const Duration& d) const { return Time(sec_ + d.sec_); }
  double sec_ = 0.0;
};
struct Subscriber {};
namespace mock {
// last message published on every advertised topic (type-erased; the test harness knows the type it advertised)
inline std::map<std::string, std::shared_ptr<const void>>& published() { static std::map<std::string, std::shared_ptr<const void>> p; return p; }
inline std::map<std::string, long>& publishCount() { static std::map<std::string, long> c; return c; }
template <class M>
std::shared_ptr<const M> lastPublished(const std::string& topic) {
  auto it = published().find(topic);
  return it == published().end() ? nullptr : std::static_pointer_cast<const M>(it->second);
}
struct Param { bool is_string = false; std::string s; double d = 0.0; };
inline std::map<std::string, Param>& params() { static std::map<std::string, Param> p; return p; }
inline void setParam(const std::string& k, const std::string& v) { Param p; p.is_string = true; p.s = v; params()[k] = p; }
inline void setParam(const std::string& k, double v) { Param p; p.d = v; params()[k] = p; }
inline std::map<std::string, std::function<void(const std::shared_ptr<const void>&)>>& topics() {
  static std::map<std::string, std::function<void(const std::shared_ptr<const void>&)>> t;
  return t;
}
template <class M>
bool publish(const std::string& topic, const M& msg) {
  auto it = topics().find(topic);
  if (it == topics().end()) return false;
  it->second(std::static_pointer_cast<const void>(std::make_shared<const M>(msg)));
  return true;
}
}  // namespace mock
class Publisher {
 public:
  Publisher() {}
  explicit Publisher(std::string topic) : topic_(std::move(topic)) {}
  template <class M> void publish(const M& msg) const {
    if (topic_.empty()) return;
    mock::published()[topic_] = std::static_pointer_cast<const void>(std::make_shared<const M>(msg));
    ++mock::publishCount()[topic_];
  }
 private:
  std::string topic_;
};
class NodeHandle {
 public:
  NodeHandle() {}
  explicit NodeHandle(const std::string&) {}
  template <class M> Publisher advertise(const std::string& topic, int) { return Publisher(topic); }
  bool getParam(const std::string& k, std::string& v) const {
    auto it = mock::params().find(k);
    if (it == mock::params().end() || !it->second.is_string) return false;
    v = it->second.s;
    return true;
  }
  template <class T> bool getParam(const std::string& k, T& v) const {
    auto it = mock::params().find(k);
    if (it == mock::params().end() || it->second.is_string) return false;
    v = T(it->second.d);
    return true;
  }
  template <class M, class C> Subscriber subscribe(const std::string& topic, int, void (C::*cb)(const typename M::ConstPtr&), C* obj) {
    mock::topics()[topic] = [cb, obj](const std::shared_ptr<const void>& p) { (obj->*cb)(std::static_pointer_cast<const M>(p)); };
    return Subscriber();
  }
};
}  // namespace ros
#define ROS_ERROR(...) (std::fprintf(stderr, __VA_ARGS__), std::fprintf(stderr, "\n"))
