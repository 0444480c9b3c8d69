// Minimal stand-in for the OpenCV core types the reference's hot-path sources name (Vec, Matx, Ptr, Mat).  Written for this
// repo; see cuda_runtime.h in this directory for what the emulation is for.  Only what is executed has a body: cv::Mat's
// algebra is declared for the reference's unused static get_3d_sobolev_filter (solver.cpp:107-158) and never linked.
#pragma once
#include <cstring>
#include <iosfwd>
#include <memory>
#include <vector>

#define CV_32FC1 5
#define CV_32FC4 29

namespace cv {
template <class T, int N>
struct Vec {
    T val[N];
    Vec() {
        for (int i = 0; i < N; ++i) val[i] = T(0);
    }
    Vec(T a, T b) : Vec() { val[0] = a, val[1] = b; }
    Vec(T a, T b, T c) : Vec() { val[0] = a, val[1] = b, val[2] = c; }
    T& operator[](int i) { return val[i]; }
    const T& operator[](int i) const { return val[i]; }
    template <class U>
    operator Vec<U, N>() const {
        Vec<U, N> r;
        for (int i = 0; i < N; ++i) r.val[i] = static_cast<U>(val[i]);
        return r;
    }
    static Vec all(T v) {
        Vec r;
        for (int i = 0; i < N; ++i) r.val[i] = v;
        return r;
    }
};
template <class T, int N>
inline Vec<T, N> operator+(const Vec<T, N>& a, const Vec<T, N>& b) {
    Vec<T, N> r;
    for (int i = 0; i < N; ++i) r.val[i] = a.val[i] + b.val[i];
    return r;
}
template <class T, int N>
inline Vec<T, N> operator-(const Vec<T, N>& a) {
    Vec<T, N> r;
    for (int i = 0; i < N; ++i) r.val[i] = -a.val[i];
    return r;
}
typedef Vec<int, 3> Vec3i;
typedef Vec<float, 3> Vec3f;
typedef Vec<double, 3> Vec3d;

template <class T, int M, int N>
struct Matx {
    T val[M * N];
    Matx() {
        for (int i = 0; i < M * N; ++i) val[i] = T(0);
    }
    T& operator()(int r, int c) { return val[r * N + c]; }
    const T& operator()(int r, int c) const { return val[r * N + c]; }
    static Matx eye() {
        Matx m;
        for (int i = 0; i < (M < N ? M : N); ++i) m(i, i) = T(1);
        return m;
    }
    Matx<T, N, M> t() const {
        Matx<T, N, M> r;
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < N; ++j) r(j, i) = (*this)(i, j);
        return r;
    }
};
template <class T, int M, int K, int N>
inline Matx<T, M, N> operator*(const Matx<T, M, K>& a, const Matx<T, K, N>& b) {
    Matx<T, M, N> r;
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            T s = T(0);
            for (int k = 0; k < K; ++k) s += a(i, k) * b(k, j);
            r(i, j) = s;
        }
    return r;
}
template <class T, int M, int N>
inline Vec<T, M> operator*(const Matx<T, M, N>& a, const Vec<T, N>& v) {
    Vec<T, M> r;
    for (int i = 0; i < M; ++i) {
        T s = T(0);
        for (int k = 0; k < N; ++k) s += a(i, k) * v[k];
        r[i] = s;
    }
    return r;
}
typedef Matx<float, 3, 3> Matx33f;

template <class T>
struct Ptr : std::shared_ptr<T> {
    Ptr() {}
    Ptr(T* p) : std::shared_ptr<T>(p) {}
    Ptr(const std::shared_ptr<T>& p) : std::shared_ptr<T>(p) {}
    bool empty() const { return !this->get(); }
    operator T*() const { return this->get(); }
};

// n-dimensional dense array: enough for `new cv::Mat(3, sizes, CV_32FC4)`, ptr<T>() and at<T>(i0, i1, i2)
class Mat {
public:
    Mat() {}
    Mat(int ndims, const int* sizes, int type) : dims_(sizes, sizes + ndims), data_(total(ndims, sizes) * (type == CV_32FC4 ? 16 : 4)) {}
    template <class T>
    T* ptr(int = 0) { return reinterpret_cast<T*>(data_.data()); }
    template <class T>
    T& at(int i0, int i1, int i2) { return ptr<T>()[((size_t) i0 * dims_[1] + i1) * dims_[2] + i2]; }
    template <class T>
    T& at(int i0, int i1);  // 2-D algebra below: declared only
    static Mat eye(int, int, int);
    static Mat zeros(int, int, int);

private:
    static size_t total(int n, const int* s) {
        size_t t = 1;
        for (int i = 0; i < n; ++i) t *= (size_t) s[i];
        return t;
    }
    std::vector<int> dims_;
    std::vector<unsigned char> data_;
};
Mat operator*(float, const Mat&);
Mat operator-(const Mat&, const Mat&);
std::ostream& operator<<(std::ostream&, const Mat&);
enum { DECOMP_LU = 0, DECOMP_SVD = 1 };
bool solve(const Mat&, const Mat&, Mat&, int);
}  // namespace cv
