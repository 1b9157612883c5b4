#pragma once
// TEST INFRASTRUCTURE (oracle/_ref build only): included by LeggedInterface.cpp, nothing of it is used there.
namespace ocs2 { class SolverSynchronizedModule { public: virtual ~SolverSynchronizedModule() = default; }; }
