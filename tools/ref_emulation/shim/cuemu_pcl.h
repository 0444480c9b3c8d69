// Stand-ins for the PCL / Boost / Eigen names that appear in the reference's headers on the way to the hot path (point
// PODs, PolygonMesh, boost::shared_ptr).  Written for this repo; nothing here is executed by the fixtures except the PODs.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>
namespace boost {
template <class T>
using shared_ptr = std::shared_ptr<T>;
namespace filesystem {}
}  // namespace boost
namespace pcl {
struct alignas(16) PointXYZ { float x, y, z, pad; };
struct alignas(16) Normal { float normal_x, normal_y, normal_z, curvature; };
struct Vertices { std::vector<uint32_t> vertices; };
struct PCLPointCloud2 { std::vector<uint8_t> data; uint32_t width = 0, height = 0; };
struct PolygonMesh {
    typedef boost::shared_ptr<PolygonMesh> Ptr;
    PCLPointCloud2 cloud;
    std::vector<Vertices> polygons;
};
template <class P>
struct PointCloud {
    std::vector<P> points;
    uint32_t width = 0, height = 0;
    size_t size() const { return points.size(); }
    P& operator[](size_t i) { return points[i]; }
    const P& operator[](size_t i) const { return points[i]; }
};
template <class P>
inline void toPCLPointCloud2(const PointCloud<P>& c, PCLPointCloud2& out) {
    out.width = c.width, out.height = c.height;
    out.data.assign((const uint8_t*) c.points.data(), (const uint8_t*) (c.points.data() + c.points.size()));
}
}  // namespace pcl
