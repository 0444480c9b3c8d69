// forwards to oracle/ref_hipbuild/shim/cuda_runtime.h
#pragma once
#include <cuda_runtime.h>
