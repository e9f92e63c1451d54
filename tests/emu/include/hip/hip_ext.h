/* hip_ext.h of the CPU execution harness: hipExtLaunchKernelGGL lives in hip_runtime.h here */
#pragma once
#include "hip_runtime.h"
