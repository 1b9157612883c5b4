// MOCK pluginlib (tests only): PLUGINLIB_EXPORT_CLASS registers a factory under the class name, mock::createInstance
// instantiates by name the way controller_manager does through pluginlib::ClassLoader.
#pragma once
#include <functional>
#include <map>
#include <string>
namespace pluginlib_mock {
inline std::map<std::string, std::function<void*()>>& registry() { static std::map<std::string, std::function<void*()>> r; return r; }
struct Registrar { Registrar(const char* name, std::function<void*()> f) { registry()[name] = std::move(f); } };
template <class Base> Base* createInstance(const std::string& name) {
  auto it = registry().find(name);
  return it == registry().end() ? nullptr : static_cast<Base*>(it->second());
}
}  // namespace pluginlib_mock
#define PLUGINLIB_EXPORT_CLASS(cls, base) \
  static pluginlib_mock::Registrar pluginlib_mock_registrar_(#cls, []() -> void* { return static_cast<base*>(new cls()); });
