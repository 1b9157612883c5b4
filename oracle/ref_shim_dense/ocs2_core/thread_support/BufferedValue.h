// TEST INFRASTRUCTURE (oracle/_ref build only). Stand-in for OCS2's <ocs2_core/thread_support/BufferedValue.h>
// [OCS2-knowledge]: an active value plus a buffered one that replaces it at updateFromBuffer() (single-threaded here).
#pragma once
#include <memory>
#include <utility>
namespace ocs2 {
template <typename T>
class BufferedValue {
 public:
  explicit BufferedValue(T init) : activeValue_(std::move(init)) {}
  const T& get() const { return activeValue_; }
  T& get() { return activeValue_; }
  void setBuffer(const T& value) { buffer_.reset(new T(value)); }
  void setBuffer(T&& value) { buffer_.reset(new T(std::move(value))); }
  bool updateFromBuffer() {
    if (!buffer_) return false;
    activeValue_ = std::move(*buffer_);
    buffer_.reset();
    return true;
  }

 private:
  T activeValue_;
  std::unique_ptr<T> buffer_;
};
}  // namespace ocs2
