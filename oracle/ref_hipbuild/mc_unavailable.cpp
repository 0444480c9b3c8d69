// The reference's marching cubes kernels (src/kfusion/cuda/marching_cubes.cu) are written for 32-wide warps (ballot masks, warp scans, a PTX lane query) and 1-D texture
// fetches; they are NOT part of the hipcc build of the reference (oracle/ref_hipbuild).  SobFusion (src/sobfu/sob_fusion.cpp) and the host class MarchingCubes
// (src/kfusion/marching_cubes.cpp) link against their five launchers, so the frames scenario of the driver needs the symbols: here they are, refusing to run.
// (Marching cubes is compared with the host emulation of those kernels instead: tests/golden/ref_mc_14x11x9.npz.)
#include <cstdio>
#include <cstdlib>

#include <cuda_runtime.h>
#include <kfusion/internal.hpp>

namespace {
[[noreturn]] void refuse(const char* what) {
    std::fprintf(stderr, "reference_hip: %s is not built for gfx950 (32-wide warp code): marching cubes is unavailable in this build\n", what);
    std::abort();
}
}  // namespace

namespace kfusion {
namespace device {
void bindTextures(const int*, const int*, const int*) { refuse("bindTextures"); }
void unbindTextures() { refuse("unbindTextures"); }
int getOccupiedVoxels(const TsdfVolume&, DeviceArray2D<int>&) { refuse("getOccupiedVoxels"); }
int computeOffsetsAndTotalVertices(DeviceArray2D<int>&) { refuse("computeOffsetsAndTotalVertices"); }
void generateTriangles(const TsdfVolume&, const DeviceArray2D<int>&, const float3&, const Aff3f&, DeviceArray<PointType>&, DeviceArray<PointType>&) { refuse("generateTriangles"); }
}  // namespace device
}  // namespace kfusion
