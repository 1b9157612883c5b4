// TEST INFRASTRUCTURE (oracle/_ref build only).
#pragma once
#define PLUGINLIB_EXPORT_CLASS(cls, base)
