// Forwarding header: the reference include path <kfusion/cuda/tsdf_volume.hpp> resolves to the MI355X shells.
#pragma once
#include <sobfu_amd/sobfu.hpp>
