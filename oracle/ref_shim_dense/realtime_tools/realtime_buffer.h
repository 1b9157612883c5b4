#pragma once
namespace realtime_tools {
template <class T>
class RealtimeBuffer {
 public:
  void writeFromNonRT(const T& v) { v_ = v; }
  T* readFromRT() { return &v_; }
 private:
  T v_;
};
}  // namespace realtime_tools
