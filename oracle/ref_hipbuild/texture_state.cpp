// the device-side descriptor of the one 2-D float texture reference (see shim/cuda_runtime.h)
#include <cuda_runtime.h>
__device__ CuemuTex2D cuemu_tex2d_float;
