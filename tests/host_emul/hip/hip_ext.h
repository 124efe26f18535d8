// Host emulation: the extended launch call lives in hip_runtime.h here.
#pragma once
#include <hip/hip_runtime.h>
