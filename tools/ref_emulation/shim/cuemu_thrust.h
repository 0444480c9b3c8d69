// The two thrust calls the reference makes (marching_cubes.cu: exclusive_scan over device_ptr<int>), on host memory.
#pragma once
#include <cuda_runtime.h>
namespace thrust {
template <class T>
struct device_ptr {
    T* p;
    device_ptr(T* p_ = 0) : p(p_) {}
    T& operator*() const { return *p; }
    T& operator[](size_t i) const { return p[i]; }
    device_ptr operator+(ptrdiff_t n) const { return device_ptr(p + n); }
    device_ptr operator-(ptrdiff_t n) const { return device_ptr(p - n); }
    ptrdiff_t operator-(const device_ptr& o) const { return p - o.p; }
    bool operator!=(const device_ptr& o) const { return p != o.p; }
    T* get() const { return p; }
};
template <class T>
inline void exclusive_scan(device_ptr<T> first, device_ptr<T> last, device_ptr<T> out) {
    T run = T();
    for (T *s = first.p, *d = out.p; s != last.p; ++s, ++d) {
        T v = *s;  // in-place safe
        *d  = run;
        run = run + v;
    }
}
}  // namespace thrust
