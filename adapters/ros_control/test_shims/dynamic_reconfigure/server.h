// MOCK (tests only): dynamic_reconfigure::Server<Config> — keeps the callback, calls it once with the defaults like the real server
// does on setCallback, and lets the test harness push a new configuration (mock::reconfigure) the way rqt_reconfigure would.
#pragma once
#include <cstdint>
#include <functional>
#include <ros/ros.h>
namespace dynamic_reconfigure {
namespace mock {
inline std::function<void(void*, uint32_t)>& hook() { static std::function<void(void*, uint32_t)> h; return h; }
template <class Config>
bool reconfigure(Config cfg, uint32_t level = ~0u) {
  if (!hook()) return false;
  hook()(&cfg, level);
  return true;
}
}  // namespace mock
template <class Config>
class Server {
 public:
  using CallbackType = std::function<void(Config&, uint32_t)>;
  explicit Server(const ros::NodeHandle& = ros::NodeHandle()) {}
  void setCallback(const CallbackType& cb) {
    cb_ = cb;
    mock::hook() = [this](void* cfg, uint32_t level) { cb_(*static_cast<Config*>(cfg), level); };
    Config defaults = Config::__getDefault__();
    cb_(defaults, ~0u);
  }
 private:
  CallbackType cb_;
};
}  // namespace dynamic_reconfigure
