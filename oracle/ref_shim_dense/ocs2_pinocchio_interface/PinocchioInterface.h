// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for OCS2's PinocchioInterface (model + data holder).  Copies made
// FROM one interface are numbered in order: WbcBase's constructor copies its argument into the measured interface first and
// the desired interface second (member order, legged_wbc/include/legged_wbc/WbcBase.h:101), so copy k reads feed role k.
#pragma once
#include <memory>
#include <pinocchio/shim.hpp>
namespace ocs2 {
class PinocchioInterface {
 public:
  PinocchioInterface() : model_(std::make_shared<pinocchio::Model>()), data_(std::make_shared<pinocchio::Data>()) {}
  PinocchioInterface(const PinocchioInterface& o) : model_(o.model_), data_(std::make_shared<pinocchio::Data>()) {
    data_->role = o.copies_ < 2 ? o.copies_ : 0;
    ++o.copies_;
  }
  PinocchioInterface(PinocchioInterface&& o) noexcept : model_(o.model_), data_(o.data_) {}
  PinocchioInterface& operator=(const PinocchioInterface& o) { model_ = o.model_; data_ = std::make_shared<pinocchio::Data>(); data_->role = o.data_->role; return *this; }
  PinocchioInterface& operator=(PinocchioInterface&& o) noexcept { model_ = o.model_; data_ = o.data_; return *this; }
  const pinocchio::Model& getModel() const { return *model_; }
  pinocchio::Model& mutableModel() { return *model_; }
  pinocchio::Data& getData() { return *data_; }
  const pinocchio::Data& getData() const { return *data_; }
  void setRole(int r) { data_->role = r; }
 private:
  std::shared_ptr<pinocchio::Model> model_;
  std::shared_ptr<pinocchio::Data> data_;
  mutable int copies_ = 0;
};
}  // namespace ocs2
