// TEST INFRASTRUCTURE (oracle/_ref build only).  dynamic_reconfigure::Server<Config> [ROS-knowledge]: setCallback invokes the callback
// once with the current configuration, as the ROS server does; the configuration is whatever the generator put into config().
#pragma once
#include <cstdint>
#include <functional>
#include <thread>
#include <boost/bind.hpp>
#include <ros/ros.h>
namespace dynamic_reconfigure {
template <class Config>
class Server {
 public:
  using CallbackType = std::function<void(Config&, uint32_t)>;
  explicit Server(const ros::NodeHandle&) {}
  static Config& config() { static Config c; return c; }
  void setCallback(const CallbackType& f) { cb_ = f; cb_(config(), 0); }
 private:
  CallbackType cb_;
};
}  // namespace dynamic_reconfigure
