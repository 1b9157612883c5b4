#pragma once
#include <memory>
namespace std_msgs { struct Int32 { int data = 0; typedef std::shared_ptr<const Int32> ConstPtr; }; }
