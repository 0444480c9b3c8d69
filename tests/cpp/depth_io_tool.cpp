// CPU-only driver for include/sobfu_amd/depth_io.hpp (tests/test_depth_io.py):
//   depth_io_tool read <file> <rows> <cols> <out.raw>   decode a depth frame, write the pixels as raw little-endian uint16
//   depth_io_tool npy <out.npy> <d0> <d1> ...             write float32 ramp 0, 0.5, 1, ... of that shape
#include <sobfu_amd/depth_io.hpp>

int main(int argc, char** argv) {
    if (argc >= 6 && std::string(argv[1]) == "read") {
        std::vector<uint16_t> px;
        std::string why;
        if (!sobfu_amd::read_depth(argv[2], std::atoi(argv[3]), std::atoi(argv[4]), px, &why)) {
            std::printf("error: %s\n", why.c_str());
            return 1;
        }
        FILE* f = std::fopen(argv[5], "wb");
        if (!f) return 2;
        std::fwrite(px.data(), 2, px.size(), f);
        std::fclose(f);
        return 0;
    }
    if (argc >= 4 && std::string(argv[1]) == "npy") {
        std::vector<size_t> shape;
        size_t n = 1;
        for (int i = 3; i < argc; ++i) { shape.push_back((size_t) std::atol(argv[i])); n *= shape.back(); }
        std::vector<float> v(n);
        for (size_t i = 0; i < n; ++i) v[i] = 0.5f * (float) i;
        return sobfu_amd::write_npy(argv[2], v.data(), shape) ? 0 : 1;
    }
    std::printf("usage: depth_io_tool read <file> <rows> <cols> <out.raw> | npy <out.npy> <dims...>\n");
    return 2;
}
