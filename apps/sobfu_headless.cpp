// sobfu_headless -- headless counterpart of the reference app's frame loop (src/apps/demo.cpp:285-340) on the MI355X
// shells: reads a params .ini, feeds depth frames to SobFusion::operator(), prints per-frame volume statistics and can
// dump the fields.  No OpenCV / PCL / VTK: depth frames are binary 16-bit PGM ("P5", maxval 65535, big-endian) or raw
// little-endian uint16 files of rows*cols pixels, or a built-in synthetic translating sphere.
//
//   sobfu_headless <params.ini> [--max-iter N] [--verbose|--vverbose] [--dims N] [--dump DIR]
//                  (--synthetic FRAMES [--shift DX] | frame0.pgm frame1.pgm ...)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include <sobfu_amd/sobfu.hpp>

static bool load_depth(const std::string& path, int rows, int cols, std::vector<uint16_t>& out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    out.assign((size_t) rows * cols, 0);
    char magic[3] = {0, 0, 0};
    bool ok = false;
    if (std::fread(magic, 1, 2, f) == 2 && magic[0] == 'P' && magic[1] == '5') {
        int w = 0, h = 0, maxv = 0;
        if (std::fscanf(f, "%d %d %d", &w, &h, &maxv) == 3 && w == cols && h == rows && maxv > 255) {
            std::fgetc(f);
            std::vector<unsigned char> b((size_t) rows * cols * 2);
            ok = std::fread(b.data(), 1, b.size(), f) == b.size();
            for (size_t i = 0; ok && i < out.size(); ++i) out[i] = (uint16_t) ((b[2 * i] << 8) | b[2 * i + 1]);
        }
    } else {
        std::rewind(f);
        ok = std::fread(out.data(), 2, out.size(), f) == out.size();
    }
    std::fclose(f);
    return ok;
}

// uint16 mm depth of a sphere, same convention as sobfu_amd/synthetic.py::render_sphere_depth (float64, rint)
static void render_sphere(double cx, double cy, double cz, double r, const kfusion::Intr& in, int rows, int cols, std::vector<uint16_t>& out) {
    out.assign((size_t) rows * cols, 0);
    for (int v = 0; v < rows; ++v)
        for (int u = 0; u < cols; ++u) {
            double dx = (u - (double) in.cx) / (double) in.fx, dy = (v - (double) in.cy) / (double) in.fy;
            double a = dx * dx + dy * dy + 1.0, b = -2.0 * (dx * cx + dy * cy + cz), c = cx * cx + cy * cy + cz * cz - r * r;
            double disc = b * b - 4.0 * a * c;
            if (disc >= 0) out[(size_t) v * cols + u] = (uint16_t) std::nearbyint(1000.0 * (-b - std::sqrt(disc)) / (2.0 * a));
        }
}

static void stats(const char* name, kfusion::cuda::TsdfVolume& v) {
    cv::Vec3i d = v.getDims();
    std::vector<float2> h((size_t) d[0] * d[1] * d[2]);
    v.data().download(h.data());
    double st = 0, sw = 0;
    long nt = 0;
    for (auto& e : h) { st += e.x; sw += e.y; nt += (std::fabs(e.x) < 1.f && e.y > 0.f); }
    std::printf("%s: sum_tsdf=%.4f sum_weight=%.0f non_truncated_observed=%ld\n", name, st, sw, nt);
}

int main(int argc, char** argv) {
    if (argc < 3) {
        std::printf("usage: %s <params.ini> [--max-iter N] [--verbose|--vverbose] [--dims N] [--dump DIR] (--synthetic FRAMES [--shift DX] | depth files...)\n", argv[0]);
        return 2;
    }
    Params p;
    if (!sobfu_amd::read_params_ini(argv[1], p)) {
        std::printf("cannot read %s\n", argv[1]);
        return 2;
    }
    int synthetic = 0;
    double shift = 0.005;
    std::string dump;
    std::vector<std::string> files;
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--max-iter" && i + 1 < argc) p.max_iter = std::atoi(argv[++i]);
        else if (a == "--verbose") p.verbosity = 1;
        else if (a == "--vverbose") p.verbosity = 2;
        else if (a == "--dims" && i + 1 < argc) { int n = std::atoi(argv[++i]); p.volume_dims = cv::Vec3i::all(n); }
        else if (a == "--synthetic" && i + 1 < argc) synthetic = std::atoi(argv[++i]);
        else if (a == "--shift" && i + 1 < argc) shift = std::atof(argv[++i]);
        else if (a == "--dump" && i + 1 < argc) dump = argv[++i];
        else files.push_back(a);
    }
    if (argc > 2) {  // --dims changes the voxel size: re-derive the voxel-unit parameters
        std::map<std::string, std::string> kv;
        Params q = p;
        sobfu_amd::read_params_ini(argv[1], q, &kv);
        float tv = std::strtof(kv["TSDF_TRUNC_DIST"].c_str(), nullptr), ev = std::strtof(kv["ETA"].c_str(), nullptr);
        p.tsdf_trunc_dist = tv * p.voxel_sizes()[0];
        p.eta = ev * p.voxel_sizes()[0];
    }
    kfusion::cuda::setDevice(0);
    kfusion::cuda::printShortCudaDeviceInfo(0);
    SobFusion fusion(p);
    const int nframes = synthetic > 0 ? synthetic : (int) files.size();
    std::vector<uint16_t> img;
    kfusion::cuda::Depth depth;
    for (int n = 0; n < nframes; ++n) {
        if (synthetic > 0) render_sphere(shift * n, 0.0, 0.75, 0.1, p.intr, p.rows, p.cols, img);
        else if (!load_depth(files[n], p.rows, p.cols, img)) { std::printf("cannot read depth frame %s\n", files[n].c_str()); return 2; }
        depth.upload(img.data(), (size_t) p.cols * sizeof(uint16_t), p.rows, p.cols);  // demo.cpp:327-329
        fusion(depth);
        stats("phi_global", *fusion.phi_global);
        if (n > 0) {
            stats("phi_n", *fusion.phi_n);
            if (n >= p.start_frame) {
                stats("phi_n_psi", *fusion.phi_n_psi);
                stats("phi_global_psi_inv", *fusion.phi_global_psi_inv);
                const sobfu_hip_solver_report& r = fusion.solver->last_report;
                std::printf("solver: iterations=%d converged=%d last_max_update_norm=%g\n", r.iterations, r.converged, r.last_max_update_norm);
            }
        }
    }
    if (!dump.empty() && fusion.psi) {  // raw little-endian float32 dumps (replaces the reference's commented-out .vti writer)
        cv::Vec3i d = p.volume_dims;
        size_t n = (size_t) d[0] * d[1] * d[2];
        std::vector<float4> h(n);
        fusion.psi->get_data().download(h.data());
        FILE* f = std::fopen((dump + "/psi.f32").c_str(), "wb");
        if (f) { std::fwrite(h.data(), sizeof(float4), n, f); std::fclose(f); }
        std::vector<float2> t(n);
        fusion.phi_global->data().download(t.data());
        f = std::fopen((dump + "/phi_global.f32").c_str(), "wb");
        if (f) { std::fwrite(t.data(), sizeof(float2), n, f); std::fclose(f); }
    }
    return 0;
}
