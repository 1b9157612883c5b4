// TEST INFRASTRUCTURE (oracle/_ref build only).  (included by LeggedController.h, unused)
#pragma once
