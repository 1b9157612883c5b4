#pragma once
// TEST INFRASTRUCTURE (oracle/_ref build only).  Included by the reference's SwitchedModelReferenceManager.h; nothing of it is used.
