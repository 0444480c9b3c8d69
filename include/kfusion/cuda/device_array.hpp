// Forwarding header: kfusion::cuda::DeviceArray / DeviceArray2D live in sobfu_amd/sobfu.hpp (reference: include/kfusion/cuda/device_array.hpp).
#pragma once
#include <sobfu_amd/sobfu.hpp>
