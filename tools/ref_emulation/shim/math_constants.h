// stand-in for CUDA's math_constants.h: only the constants the reference names
#pragma once
#include <cuda_runtime.h>
#define CUDART_PI_F 3.141592654f
#define CUDART_NAN_F __int_as_float(0x7fffffff)
#define CUDART_INF_F __int_as_float(0x7f800000)
