// sobfu_headless -- headless counterpart of the reference app's frame loop (src/apps/demo.cpp:285-340) on the MI355X
// shells: reads a params .ini, feeds depth frames to SobFusion::operator(), prints per-frame volume statistics and can
// dump the fields.  No OpenCV / PCL / VTK: depth frames are 16-bit grayscale PNG (what the reference's datasets ship),
// binary 16-bit PGM or raw little-endian uint16 files of rows*cols pixels (sobfu_amd/depth_io.hpp), or a built-in
// synthetic translating sphere.  --dump DIR writes psi, psi_inv and the four TSDF volumes as .npy (float32); --mesh DIR
// writes marching-cubes meshes of the volumes per frame as legacy-ASCII .vtk polydata (the reference: demo.cpp:236-246).
//
//   sobfu_headless <params.ini> [--max-iter N] [--verbose|--vverbose] [--dims N] [--dump DIR] [--mesh DIR] [--no-stats]
//                  (--synthetic FRAMES [--shift DX] | frame0.pgm frame1.pgm ...)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include <sobfu_amd/depth_io.hpp>
#include <sobfu_amd/sobfu.hpp>

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
        std::printf("usage: %s <params.ini> [--max-iter N] [--verbose|--vverbose] [--dims N] [--dump DIR] [--mesh DIR] (--synthetic FRAMES [--shift DX] | depth files...)\n", argv[0]);
        return 2;
    }
    Params p;
    std::string why;
    if (!sobfu_amd::read_params_ini(argv[1], p, nullptr, &why)) {
        std::printf("bad parameter file: %s\n", why.c_str());
        return 2;
    }
    int synthetic = 0;
    double shift = 0.005;
    std::string dump, mesh_dir;
    bool print_stats = true;  // per-frame volume statistics download four volumes: --no-stats leaves only the frame loop (timing runs)
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
        else if (a == "--mesh" && i + 1 < argc) mesh_dir = argv[++i];
        else if (a == "--no-stats") print_stats = false;
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
    double time_ms = 0.0;
    for (int n = 0; n < nframes; ++n) {
        if (synthetic > 0) render_sphere(shift * n, 0.0, 0.75, 0.1, p.intr, p.rows, p.cols, img);
        else {
            std::string why;
            if (!sobfu_amd::read_depth(files[n], p.rows, p.cols, img, &why)) {
                std::printf("cannot read depth frame %s: %s\n", files[n].c_str(), why.c_str());
                return 2;
            }
        }
        depth.upload(img.data(), (size_t) p.cols * sizeof(uint16_t), p.rows, p.cols);  // demo.cpp:327-329
        {
            kfusion::SampledScopeTime fps(time_ms);  // demo.cpp:331 -- "avg. frame time" every 34 frames
            fusion(depth);
        }
        if (print_stats) stats("phi_global", *fusion.phi_global);
        auto save_mesh = [&](const char* name, const sobfu_amd::TriangleMesh& m) {  // demo.cpp:236-246 (name_frame.vtk)
            if (m.empty()) return;
            const std::string path = mesh_dir + "/" + name + "_" + std::to_string(n) + ".vtk";
            if (sobfu_amd::write_vtk(path, m)) std::printf("mesh %s: %zu triangles\n", name, m.triangles());
            else std::printf("cannot write %s\n", path.c_str());
        };
        if (!mesh_dir.empty()) {
            save_mesh("phi_global", fusion.get_phi_global_mesh());
            if (n > 0) save_mesh("phi_n", fusion.get_phi_n_mesh());
            if (n > 0 && n >= p.start_frame) {
                save_mesh("phi_n_psi", fusion.get_phi_n_psi_mesh());
                save_mesh("phi_global_psi_inv", fusion.get_phi_global_psi_inv_mesh());
            }
        }
        if (n > 0 && print_stats) {
            stats("phi_n", *fusion.phi_n);
            if (n >= p.start_frame) {
                stats("phi_n_psi", *fusion.phi_n_psi);
                stats("phi_global_psi_inv", *fusion.phi_global_psi_inv);
                const sobfu_hip_solver_report& r = fusion.solver->last_report;
                std::printf("solver: iterations=%d converged=%d last_max_update_norm=%g\n", r.iterations, r.converged, r.last_max_update_norm);
            }
        }
    }
    if (!dump.empty() && fusion.psi) {  // .npy dumps (replace the reference's commented-out .vti writer, demo.cpp:252-283)
        cv::Vec3i d = p.volume_dims;
        const size_t n = (size_t) d[0] * d[1] * d[2], Z = (size_t) d[2], Y = (size_t) d[1], X = (size_t) d[0];
        std::vector<float4> h(n);
        std::vector<float2> t(n);
        auto field = [&](const char* name, sobfu::cuda::DeformationField& f) {
            f.get_data().download(h.data());
            if (!sobfu_amd::write_npy(dump + "/" + name + ".npy", (const float*) h.data(), {Z, Y, X, 4})) std::printf("cannot write %s\n", name);
        };
        auto volume = [&](const char* name, kfusion::cuda::TsdfVolume& v) {
            v.data().download(t.data());
            if (!sobfu_amd::write_npy(dump + "/" + name + ".npy", (const float*) t.data(), {Z, Y, X, 2})) std::printf("cannot write %s\n", name);
        };
        field("psi", *fusion.psi);
        if (fusion.psi_inv) field("psi_inv", *fusion.psi_inv);
        volume("phi_global", *fusion.phi_global);
        if (fusion.phi_n) volume("phi_n", *fusion.phi_n);
        if (fusion.phi_n_psi) volume("phi_n_psi", *fusion.phi_n_psi);
        if (fusion.phi_global_psi_inv) volume("phi_global_psi_inv", *fusion.phi_global_psi_inv);
    }
    return 0;
}
