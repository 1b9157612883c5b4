// TEST INFRASTRUCTURE (oracle/_ref build only). Stand-in for the sliver of roscpp the in-place-compiled reference sources
// touch.  NodeHandle::subscribe keeps the callback in a registry keyed by topic, so that the golden-vector entry points can
// deliver "messages" to the reference's own callbacks (the cmd_vel rate limiter lives in one of them).
#pragma once
#include <functional>
#include <map>
#include <memory>
#include <string>
namespace ros {
struct Subscriber {};
struct Publisher {
  template <class M> void publish(const M&) const {}
};
struct Duration { explicit Duration(double = 0.0) {} };
struct Time { double t = 0.0; };
namespace ref_shim {
inline std::map<std::string, std::function<void(const std::shared_ptr<const void>&)>>& callbacks() {
  static std::map<std::string, std::function<void(const std::shared_ptr<const void>&)>> m;
  return m;
}
template <class M>
void deliver(const std::string& topic, const M& msg) {
  callbacks().at(topic)(std::static_pointer_cast<const void>(std::make_shared<const M>(msg)));
}
}  // namespace ref_shim
class NodeHandle {
 public:
  template <class M, class F>
  Subscriber subscribe(const std::string& topic, int, F cb) {
    ref_shim::callbacks()[topic] = [cb](const std::shared_ptr<const void>& p) { cb(std::static_pointer_cast<const M>(p)); };
    return Subscriber();
  }
  template <class M>
  Publisher advertise(const std::string&, int, bool = false) { return Publisher(); }
  template <class T>
  bool getParam(const std::string&, T&) const { return false; }
};
inline void init(int&, char**, const std::string&) {}
inline void spin() {}
}  // namespace ros
#define ROS_WARN(...) ((void)0)
#define ROS_INFO(...) ((void)0)
#define ROS_ERROR(...) ((void)0)
