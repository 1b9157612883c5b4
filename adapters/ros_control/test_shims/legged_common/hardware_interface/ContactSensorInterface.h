// TEST STAND-IN for legged_common/include/legged_common/hardware_interface/ContactSensorInterface.h:18-52.
#pragma once
#include <string>
namespace legged {
class ContactSensorHandle {
 public:
  bool isContact() const { return true; }
};
class ContactSensorInterface {
 public:
  ContactSensorHandle getHandle(const std::string&) { return ContactSensorHandle(); }
};
}  // namespace legged
