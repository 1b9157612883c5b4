// MOCK for legged_common/include/legged_common/hardware_interface/ContactSensorInterface.h:18-52 (tests only).
#pragma once
#include <map>
#include <stdexcept>
#include <string>
namespace legged {
class ContactSensorHandle {
 public:
  ContactSensorHandle() {}
  ContactSensorHandle(const std::string& name, const bool* flag) : name_(name), flag_(flag) {}
  std::string getName() const { return name_; }
  bool isContact() const { return *flag_; }
 private:
  std::string name_;
  const bool* flag_ = nullptr;
};
class ContactSensorInterface {
 public:
  void registerHandle(const ContactSensorHandle& h) { map_[h.getName()] = h; }
  ContactSensorHandle getHandle(const std::string& n) {
    auto it = map_.find(n);
    if (it == map_.end()) throw std::runtime_error("no contact handle '" + n + "'");
    return it->second;
  }
 private:
  std::map<std::string, ContactSensorHandle> map_;
};
}  // namespace legged
