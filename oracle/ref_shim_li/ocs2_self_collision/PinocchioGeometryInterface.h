// TEST INFRASTRUCTURE (oracle/_ref build only).  [OCS2-knowledge: PinocchioGeometryInterface(interface, link pairs, object pairs).]
// Holder of the collision pairs LeggedInterface::getSelfCollisionConstraint read from task.info.
#pragma once
#include <string>
#include <utility>
#include <vector>
#include <ocs2_pinocchio_interface/PinocchioInterface.h>
namespace ocs2 {
class PinocchioGeometryInterface {
 public:
  PinocchioGeometryInterface(const PinocchioInterface&, std::vector<std::pair<std::string, std::string>> linkPairs,
                             std::vector<std::pair<size_t, size_t>> objectPairs)
      : linkPairs(std::move(linkPairs)), objectPairs(std::move(objectPairs)) {}
  size_t getNumCollisionPairs() const { return linkPairs.size() + objectPairs.size(); }
  std::vector<std::pair<std::string, std::string>> linkPairs;
  std::vector<std::pair<size_t, size_t>> objectPairs;
};
}  // namespace ocs2
