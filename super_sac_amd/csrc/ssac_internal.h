// Internal helpers shared by the .hip translation units of libssac_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "ssac_hip.h"

// record an error message (thread-local) and return a non-zero status
int ssac_fail(const char *msg);
// hipGetLastError() after a launch; non-zero + message on failure
int ssac_check_launch(const char *what);
