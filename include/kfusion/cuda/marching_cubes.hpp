// Forwarding header: kfusion::cuda::MarchingCubes lives in sobfu_amd/sobfu.hpp (reference: include/kfusion/cuda/marching_cubes.hpp).
#pragma once
#include <sobfu_amd/sobfu.hpp>
