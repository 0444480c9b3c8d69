// Stand-in for cv::Affine3<T>: a rigid transform kept as (R, t).  OpenCV keeps a 4x4 matrix and inverts it by LU; for the
// rigid poses the reference builds (identity camera, translated volume: sob_fusion.cpp:31, tsdf_volume.cpp:96) both give
// the same R and t bit for bit, which tools/ref_emulation/driver.cpp asserts for the poses it uses.
#pragma once
#include <opencv2/core/core.hpp>
namespace cv {
template <class T>
class Affine3 {
public:
    typedef Matx<T, 3, 3> Mat3;
    typedef Vec<T, 3> Vec3;
    Affine3() : R_(Mat3::eye()) {}
    Affine3(const Mat3& R, const Vec3& t = Vec3()) : R_(R), t_(t) {}
    static Affine3 Identity() { return Affine3(); }
    Mat3 rotation() const { return R_; }
    Vec3 translation() const { return t_; }
    void rotation(const Mat3& R) { R_ = R; }
    void translation(const Vec3& t) { t_ = t; }
    Affine3 translate(const Vec3& t) const { return Affine3(R_, t_ + t); }
    Affine3 inv(int = 0) const {
        Mat3 Rt = R_.t();
        return Affine3(Rt, -(Rt * t_));
    }
    Vec3 operator*(const Vec3& v) const { return R_ * v + t_; }

private:
    Mat3 R_;
    Vec3 t_;
};
template <class T>
inline Affine3<T> operator*(const Affine3<T>& a, const Affine3<T>& b) {
    return Affine3<T>(a.rotation() * b.rotation(), a.rotation() * b.translation() + a.translation());
}
typedef Affine3<float> Affine3f;
}  // namespace cv
