#pragma once
#include <ocs2_ros_interfaces/command/TargetTrajectoriesRosPublisher.h>
