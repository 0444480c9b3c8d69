// stand-in for CUDA's math_constants.h: only the constant the reference names
#pragma once
#include <cuda_runtime.h>
#define CUDART_PI_F 3.141592654f
