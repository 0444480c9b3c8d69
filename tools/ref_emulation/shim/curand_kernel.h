// forwards to the single stand-in header (tools/ref_emulation/shim/cuda_runtime.h)
#pragma once
#include <cuda_runtime.h>
