// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for the sliver of roscpp that legged_estimation's sources touch:
// Time / Duration arithmetic, NodeHandle::subscribe (callbacks are dropped: the golden vectors drive update() directly).
#pragma once
#include <cassert>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <string>
namespace ros {
struct Duration {
  double s = 0.0;
  Duration() = default;
  explicit Duration(double v) : s(v) {}
  double toSec() const { return s; }
  Duration& fromSec(double v) { s = v; return *this; }
};
struct Time {
  double t = 0.0;
  Time() = default;
  explicit Time(double v) : t(v) {}
  double toSec() const { return t; }
  static Time now() { return Time(); }
};
inline Time operator+(const Time& a, const Duration& d) { return Time(a.t + d.s); }
inline Time operator-(const Time& a, const Duration& d) { return Time(a.t - d.s); }
inline bool operator<(const Time& a, const Time& b) { return a.t < b.t; }
struct Subscriber {};
struct Publisher { template <class M> void publish(const M&) const {} };
// topic registry: the generators deliver "messages" to the callbacks the reference code subscribed (lambda subscriptions only;
// member-function subscriptions of the estimator are dropped, its golden vectors drive update() directly)
namespace ref_shim {
inline std::map<std::string, std::function<void(const std::shared_ptr<const void>&)>>& callbacks() {
  static std::map<std::string, std::function<void(const std::shared_ptr<const void>&)>> m;
  return m;
}
template <class M>
void deliver(const std::string& topic, const M& msg) { callbacks().at(topic)(std::static_pointer_cast<const void>(std::make_shared<const M>(msg))); }
inline std::map<std::string, std::string>& string_params() { static std::map<std::string, std::string> p; return p; }
}  // namespace ref_shim
class NodeHandle {
 public:
  NodeHandle() = default;
  explicit NodeHandle(const std::string&) {}
  template <class M, class T> Subscriber subscribe(const std::string& topic, int, void (T::*m)(const typename M::ConstPtr&), T* obj) {
    ref_shim::callbacks()[topic] = [m, obj](const std::shared_ptr<const void>& p) { (obj->*m)(std::static_pointer_cast<const M>(p)); };
    return Subscriber();
  }
  template <class M, class F> Subscriber subscribe(const std::string& topic, int, F cb) {
    ref_shim::callbacks()[topic] = [cb](const std::shared_ptr<const void>& p) { cb(std::static_pointer_cast<const M>(p)); };
    return Subscriber();
  }
  template <class M> Publisher advertise(const std::string&, int, bool = false) { return Publisher(); }
  bool getParam(const std::string& k, std::string& v) const {
    auto it = ref_shim::string_params().find(k);
    if (it == ref_shim::string_params().end()) return false;
    v = it->second;
    return true;
  }
  template <class T> bool getParam(const std::string&, T&) const { return false; }
};
}  // namespace ros
#define ROS_WARN(...) ((void)0)
#define ROS_INFO(...) ((void)0)
#define ROS_ERROR(...) ((void)0)
#define ROS_WARN_STREAM(x) ((void)0)
#define ROS_INFO_STREAM(x) ((void)0)
#define ROS_ERROR_STREAM(x) ((void)0)
