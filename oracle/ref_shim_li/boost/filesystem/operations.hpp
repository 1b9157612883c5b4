#pragma once
#include <boost/filesystem/path.hpp>
