// Driver of the host emulation (see build.py / shim/cuda_runtime.h: SHIM EVIDENCE, build-container only).
//
// Every number it writes is computed by the reference's own functions, called through the reference's own headers:
// the driver only moves raw arrays between files and those calls.  Usage:  ref_emu <scenario> <dir> key=value ...
// Inputs are <dir>/<name>.bin (little-endian float32 / uint16), outputs <dir>/out_<name>.bin, the reference's stdout
// <dir>/out_log.txt.  tests/golden/make_reference_fixtures.py writes the inputs and packs the outputs into .npz files.
//
//   kernels  one call of each L1 launcher: TsdfDifferentiator / SecondOrderDifferentiator / Differentiator::calculate,
//            calculate_potential_gradient, set_convolution_kernel + convolution_{rows,columns,depth}, update_psi, apply,
//            init_identity + estimate_inverse, integrate(phi_global, phi_n_psi), Reductor::{data_energy,
//            reg_energy_sobolev, max_update_norm}     (solver.hpp:109-136, vector_fields.hpp:140-241, reductor.hpp:31-35)
//   solver   sobfu::cuda::Solver(params).estimate_psi(...) on uploaded volumes and a start field (solver.cpp:7-101), or on two
//            initSphere volumes from the identity (the set-up of the reference's test/solver_test.cpp:109-132)
//   tsdf     kfusion::cuda::TsdfVolume::init{Sphere,Box,Ellipsoid,Plane,Torus}                 (tsdf_volume.cpp:108-146)
//   depth    depthBilateralFilter -> depthTruncation -> computeDists -> TsdfVolume::integrate  (imgproc.cpp, tsdf_volume.cpp:95)
//   launchers  timing aid: every launcher above `repeat` times, nothing written (GPU build)
//   frames   SobFusion::operator() frame by frame (sob_fusion.cpp:71-145); its private volumes are read for the dumps
//   mc       kfusion::cuda::MarchingCubes::run                                                 (marching_cubes.cpp:24-79)
#include <cassert>
#include <cstdint>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include <chrono>

#include <cuda_runtime.h>
#define private public  // SobFusion keeps phi_global & co. private; the dumps need to read them
#include <sobfu/sob_fusion.hpp>
#undef private
#include <kfusion/cuda/marching_cubes.hpp>  // (oracle/ref_hipbuild compiles this driver too, without the marching cubes KERNELS -- 32-wide warp code -- and so without the mc scenario)
#include <kfusion/precomp.hpp>
#include <sobfu/solver.hpp>

static std::string g_dir;
static std::map<std::string, double> g_args;

static double arg(const char* k) {
    auto it = g_args.find(k);
    if (it == g_args.end()) fprintf(stderr, "ref_emu: missing argument %s\n", k), exit(2);
    return it->second;
}
static double arg(const char* k, double dflt) { return g_args.count(k) ? g_args[k] : dflt; }

template <class T>
static std::vector<T> read_bin(const std::string& name, size_t count) {
    std::vector<T> v(count);
    std::ifstream f(g_dir + "/" + name + ".bin", std::ios::binary);
    if (!f.read((char*) v.data(), count * sizeof(T)) || f.peek() != EOF) fprintf(stderr, "ref_emu: %s.bin has the wrong size\n", name.c_str()), exit(2);
    return v;
}
static void write_bin(const std::string& name, const void* p, size_t bytes) {
    std::ofstream f(g_dir + "/out_" + name + ".bin", std::ios::binary);
    f.write((const char*) p, bytes);
}
// digest=1: an array leaves as 8 bytes -- sum over its 32-bit words w_i of w_i * (2 i + 1) mod 2^64 (tests/ref_hip_runner.py computes the same) -- for sizes whose
// arrays are gigabytes (BASELINE config 5's 512^3 on the GPU build)
static bool g_digest = false;
static void dump(const std::string& name, const kfusion::cuda::CudaData& d) {
    std::vector<char> h(d.sizeBytes());
    d.download(h.data());
    if (!g_digest) return write_bin(name, h.data(), h.size());
    const uint32_t* w = (const uint32_t*) h.data();
    uint64_t sum = 0;
    for (size_t i = 0, n = h.size() / 4; i < n; ++i) sum += (uint64_t) w[i] * (2 * (uint64_t) i + 1);
    write_bin(name, &sum, sizeof sum);
}

static Params make_params() {
    Params p;
    p.volume_dims = cv::Vec3i((int) arg("X"), (int) arg("Y"), (int) arg("Z"));
    p.volume_size = cv::Vec3f((float) arg("size_x"), (float) arg("size_y"), (float) arg("size_z"));
    // demo.cpp:71-74: voxel-unit parameters and the volume pose
    p.tsdf_trunc_dist = (float) arg("trunc_vox") * p.voxel_sizes()[0];
    p.eta             = (float) arg("eta_vox") * p.voxel_sizes()[0];
    p.volume_pose     = cv::Affine3f().translate(cv::Vec3f(-p.volume_size[0] / 2.f, -p.volume_size[1] / 2.f, (float) arg("t_z", 0.0)));
    p.tsdf_max_weight = (float) arg("max_weight", 64.0);
    p.gradient_delta_factor = 0.1f;
    p.intr = kfusion::Intr((float) arg("fx", 1.0), (float) arg("fy", 1.0), (float) arg("cx", 0.0), (float) arg("cy", 0.0));
    p.rows = (int) arg("rows", 480), p.cols = (int) arg("cols", 640);
    p.icp_truncate_depth_dist = (float) arg("trunc_depth", 0.0);
    p.bilateral_kernel_size   = (int) arg("bilateral_ksz", 7);
    p.bilateral_sigma_spatial = (float) arg("bilateral_ss", 4.5);
    p.bilateral_sigma_depth   = (float) arg("bilateral_sd", 0.04);
    p.start_frame = (int) arg("start_frame", 1), p.verbosity = (int) arg("verbosity", 0);
    p.s = (int) arg("s", 7), p.lambda = (float) arg("lambda", 0.1), p.alpha = (float) arg("alpha", 0.0), p.w_reg = (float) arg("w_reg", 0.0);
    p.max_iter = (int) arg("max_iter", 0), p.max_update_norm = (float) arg("max_update_norm", -1.0);
    return p;
}

static void scenario_kernels() {
    const int X = (int) arg("X"), Y = (int) arg("Y"), Z = (int) arg("Z");
    const size_t N = (size_t) X * Y * Z;
    const int3 dims = make_int3(X, Y, Z);
    const float3 vsz = make_float3(1.f, 1.f, 1.f);  // not read by any of these kernels
    const float trunc = 1.f, eta = 1.f, max_weight = (float) arg("max_weight");
    auto up = [&](const char* name, size_t floats) {
        kfusion::cuda::CudaData d;
        d.upload(read_bin<float>(name, floats).data(), floats * 4);
        return d;
    };
    auto fresh = [&](size_t floats) { return kfusion::cuda::CudaData(floats * 4); };
    kfusion::cuda::CudaData vol = up("phi_n_psi", N * 2), pg = up("phi_global", N * 2), psi = up("psi", N * 4), fuse = up("fuse_in", N * 2);
    kfusion::cuda::CudaData taps = up("taps", 7), grad = fresh(N * 4), L = fresh(N * 4), J0 = fresh(N * 16), J1 = fresh(N * 16),
                            nU = fresh(N * 4), nUS = fresh(N * 4), warped = fresh(N * 2), inv = fresh(N * 4);

    kfusion::device::TsdfVolume vol_d(vol.ptr<float2>(), dims, vsz, trunc, eta, max_weight), pg_d(pg.ptr<float2>(), dims, vsz, trunc, eta, max_weight),
        warped_d(warped.ptr<float2>(), dims, vsz, trunc, eta, max_weight), fuse_d(fuse.ptr<float2>(), dims, vsz, trunc, eta, max_weight);
    sobfu::device::DeformationField psi_d(psi.ptr<float4>(), dims), inv_d(inv.ptr<float4>(), dims);
    sobfu::device::TsdfGradient grad_d(grad.ptr<float4>(), dims);
    sobfu::device::Laplacian L_d(L.ptr<float4>(), dims);
    sobfu::device::PotentialGradient nU_d(nU.ptr<float4>(), dims), nUS_d(nUS.ptr<float4>(), dims);
    sobfu::device::Jacobian J0_d(J0.ptr<Mat4f>(), dims), J1_d(J1.ptr<Mat4f>(), dims);

    sobfu::device::TsdfDifferentiator(vol_d).calculate(grad_d);
    sobfu::device::SecondOrderDifferentiator(psi_d).calculate(L_d);
    sobfu::device::Differentiator diff(psi_d);
    diff.calculate(J0_d);
    diff.calculate_deformation_jacobian(J1_d);
    dump("grad", grad), dump("laplacian", L), dump("jacobian0", J0), dump("jacobian1", J1);

    sobfu::device::calculate_potential_gradient(vol_d, pg_d, grad_d, L_d, nU_d, (float) arg("w_reg"));
    dump("nabla_U", nU);

    sobfu::device::set_convolution_kernel(taps.ptr<float>());
    sobfu::device::convolution_rows(nUS.ptr<float4>(), nU.ptr<float4>(), X, Y, Z);
    dump("conv_rows", nUS);
    sobfu::device::convolution_columns(nUS.ptr<float4>(), nU.ptr<float4>(), X, Y, Z);
    dump("conv_cols", nUS);
    sobfu::device::convolution_depth(nUS.ptr<float4>(), nU.ptr<float4>(), X, Y, Z);
    dump("conv_depth", nUS);

    sobfu::device::Reductor r(dims, vsz.x, trunc);
    sobfu::device::update_psi(psi_d, nUS_d, r.updates, (float) arg("alpha"));
    dump("psi_new", psi);
    write_bin("updates", r.updates, N * 16);

    sobfu::device::apply(vol_d, warped_d, psi_d);
    dump("warped", warped);
    sobfu::device::init_identity(inv_d);
    sobfu::device::estimate_inverse(psi_d, inv_d);
    dump("psi_inv", inv);
    kfusion::device::integrate(fuse_d, warped_d);
    dump("fused", fuse);

    float2 m        = r.max_update_norm();
    float scalars[] = {r.data_energy(pg.ptr<float2>(), vol.ptr<float2>()), r.reg_energy_sobolev(J1.ptr<Mat4f>()), m.x, m.y, (float) r.blocks, (float) r.threads};
    write_bin("scalars", scalars, sizeof scalars);
}

static void scenario_solver() {
    Params p = make_params();
    const size_t N = (size_t) p.volume_dims[0] * p.volume_dims[1] * p.volume_dims[2];
    cv::Ptr<kfusion::cuda::TsdfVolume> pg(new kfusion::cuda::TsdfVolume(p)), pgi(new kfusion::cuda::TsdfVolume(p)), pn(new kfusion::cuda::TsdfVolume(p)),
        pnp(new kfusion::cuda::TsdfVolume(p));
    auto psi = std::make_shared<sobfu::cuda::DeformationField>(p.volume_dims), psi_inv = std::make_shared<sobfu::cuda::DeformationField>(p.volume_dims);
    if (g_args.count("sphere_r")) {  // the set-up of the reference's own test/solver_test.cpp:109-132: two initSphere volumes, identity start
        pg->initSphere(make_float3((float) arg("sphere_cx"), (float) arg("sphere_cy"), (float) arg("sphere_cz")), (float) arg("sphere_r"));
        pn->initSphere(make_float3((float) arg("sphere2_cx"), (float) arg("sphere2_cy"), (float) arg("sphere2_cz")), (float) arg("sphere_r"));
        dump("phi_global", pg->data()), dump("phi_n", pn->data());
    } else if (g_args.count("ell_rx")) {  // two initEllipsoid volumes (no device libm on the way: every build gives the same bits), identity start
        pg->initEllipsoid(make_float3((float) arg("ell_rx"), (float) arg("ell_ry"), (float) arg("ell_rz")));
        pn->initEllipsoid(make_float3((float) arg("ell2_rx"), (float) arg("ell2_ry"), (float) arg("ell2_rz")));
        dump("phi_global", pg->data()), dump("phi_n", pn->data());
    } else {
        pg->data().upload(read_bin<float>("phi_global", N * 2).data(), N * 8);
        pn->data().upload(read_bin<float>("phi_n", N * 2).data(), N * 8);
        psi->get_data().upload(read_bin<float>("psi0", N * 4).data(), N * 16);
    }
    sobfu::cuda::Solver solver(p);
    solver.estimate_psi(pg, pgi, pn, pnp, psi, psi_inv);
    dump("psi", psi->get_data()), dump("phi_n_psi", pnp->data()), dump("psi_inv", psi_inv->get_data()), dump("phi_global_psi_inv", pgi->data());
}

static void scenario_tsdf() {
    Params p = make_params();
    kfusion::cuda::TsdfVolume v(p);
    v.initSphere(make_float3((float) arg("sphere_cx"), (float) arg("sphere_cy"), (float) arg("sphere_cz")), (float) arg("sphere_r"));
    dump("sphere", v.data());
    v.initBox(make_float3((float) arg("box_x"), (float) arg("box_y"), (float) arg("box_z")));
    dump("box", v.data());
    v.initEllipsoid(make_float3((float) arg("ell_x"), (float) arg("ell_y"), (float) arg("ell_z")));
    dump("ellipsoid", v.data());
    v.initPlane((float) arg("plane_z"));
    dump("plane", v.data());
    v.initTorus(make_float2((float) arg("torus_R"), (float) arg("torus_r")));
    dump("torus", v.data());
}

static std::vector<unsigned short> dense(const kfusion::cuda::Depth& d) {
    std::vector<unsigned short> h((size_t) d.rows() * d.cols());
    d.download(h.data(), d.cols() * sizeof(unsigned short));
    return h;
}

static void scenario_depth() {
    Params p = make_params();
    const int rows = p.rows, cols = p.cols;
    std::vector<unsigned short> raw = read_bin<unsigned short>("depth", (size_t) rows * cols);
    kfusion::cuda::Depth depth, filtered;
    depth.upload(raw.data(), cols * sizeof(unsigned short), rows, cols);
    kfusion::cuda::depthBilateralFilter(depth, filtered, p.bilateral_kernel_size, p.bilateral_sigma_spatial, p.bilateral_sigma_depth);
    std::vector<unsigned short> h = dense(filtered);
    write_bin("bilateral", h.data(), h.size() * 2);
    kfusion::cuda::depthTruncation(filtered, p.icp_truncate_depth_dist);
    h = dense(filtered);
    write_bin("truncated", h.data(), h.size() * 2);
    kfusion::cuda::Dists dists;
    kfusion::cuda::computeDists(filtered, dists, p.intr);
    std::vector<float> hd((size_t) rows * cols);
    dists.download(hd.data(), cols * sizeof(float));
    write_bin("dists", hd.data(), hd.size() * 4);
    kfusion::cuda::TsdfVolume v(p);
    v.integrate(dists, cv::Affine3f::Identity(), p.intr);
    dump("volume", v.data());
}

// timing aid for the GPU build (tests/reference_launcher_table.py reads the per-kernel times from rocprofv3): every L1 launcher of the kernels and
// depth scenarios `repeat` times back to back on the same uploaded arrays, nothing dumped
static void scenario_launchers() {
    Params p = make_params();
    const int X = p.volume_dims[0], Y = p.volume_dims[1], Z = p.volume_dims[2], rep = (int) arg("repeat", 10);
    const size_t N = (size_t) X * Y * Z;
    const int3 dims = make_int3(X, Y, Z);
    const float3 vsz = make_float3(1.f, 1.f, 1.f);
    const float trunc = 1.f, eta = 1.f, max_weight = p.tsdf_max_weight;
    auto up = [&](const char* name, size_t floats) {
        kfusion::cuda::CudaData d;
        d.upload(read_bin<float>(name, floats).data(), floats * 4);
        return d;
    };
    auto fresh = [&](size_t floats) { return kfusion::cuda::CudaData(floats * 4); };
    kfusion::cuda::CudaData vol = up("phi_n_psi", N * 2), pg = up("phi_global", N * 2), psi = up("psi", N * 4), fuse = up("fuse_in", N * 2);
    kfusion::cuda::CudaData taps = up("taps", 7), grad = fresh(N * 4), L = fresh(N * 4), J1 = fresh(N * 16), nU = fresh(N * 4), nUS = fresh(N * 4),
                            warped = fresh(N * 2), inv = fresh(N * 4);
    kfusion::device::TsdfVolume vol_d(vol.ptr<float2>(), dims, vsz, trunc, eta, max_weight), pg_d(pg.ptr<float2>(), dims, vsz, trunc, eta, max_weight),
        warped_d(warped.ptr<float2>(), dims, vsz, trunc, eta, max_weight), fuse_d(fuse.ptr<float2>(), dims, vsz, trunc, eta, max_weight);
    sobfu::device::DeformationField psi_d(psi.ptr<float4>(), dims), inv_d(inv.ptr<float4>(), dims);
    sobfu::device::TsdfGradient grad_d(grad.ptr<float4>(), dims);
    sobfu::device::Laplacian L_d(L.ptr<float4>(), dims);
    sobfu::device::PotentialGradient nU_d(nU.ptr<float4>(), dims), nUS_d(nUS.ptr<float4>(), dims);
    sobfu::device::Jacobian J1_d(J1.ptr<Mat4f>(), dims);
    sobfu::device::Reductor r(dims, vsz.x, trunc);
    sobfu::device::set_convolution_kernel(taps.ptr<float>());
    sobfu::device::Differentiator diff(psi_d);
    // flush=1: 512 MB are overwritten before every call, so that no call finds its inputs in the 256 MB Infinity Cache because the call before it left them there
    kfusion::cuda::CudaData flush_buf;
    const size_t flush_bytes = (size_t) 512 << 20;
    if (arg("flush", 0.0) != 0.0) flush_buf.create(flush_bytes);
#define REP(...) for (int i_ = 0; i_ < rep; ++i_) { if (!flush_buf.empty()) cudaMemset(flush_buf.ptr<char>(), i_, flush_bytes); __VA_ARGS__; }
    REP(sobfu::device::apply(vol_d, warped_d, psi_d))
    REP(sobfu::device::TsdfDifferentiator(vol_d).calculate(grad_d))
    REP(sobfu::device::SecondOrderDifferentiator(psi_d).calculate(L_d))
    REP(diff.calculate_deformation_jacobian(J1_d))
    REP(sobfu::device::calculate_potential_gradient(vol_d, pg_d, grad_d, L_d, nU_d, p.w_reg))
    REP(sobfu::device::convolution_rows(nUS.ptr<float4>(), nU.ptr<float4>(), X, Y, Z))
    REP(sobfu::device::convolution_columns(nUS.ptr<float4>(), nU.ptr<float4>(), X, Y, Z))
    REP(sobfu::device::convolution_depth(nUS.ptr<float4>(), nU.ptr<float4>(), X, Y, Z))
    sobfu::device::convolution_rows(nUS.ptr<float4>(), nU.ptr<float4>(), X, Y, Z);  // (bounded again after the accumulating repeats)
    REP(sobfu::device::update_psi(psi_d, nUS_d, r.updates, p.alpha))
    REP(r.max_update_norm())
    REP(r.data_energy(pg.ptr<float2>(), vol.ptr<float2>()))
    REP(r.reg_energy_sobolev(J1.ptr<Mat4f>()))
    REP(sobfu::device::init_identity(inv_d); sobfu::device::estimate_inverse(psi_d, inv_d))
    REP(kfusion::device::integrate(fuse_d, warped_d))
    // per-frame side: clear, an analytic volume, the depth pre-steps and integrate(depth)
    const int rows = p.rows, cols = p.cols;
    std::vector<unsigned short> raw = read_bin<unsigned short>("depth", (size_t) rows * cols);
    kfusion::cuda::Depth depth, filtered;
    depth.upload(raw.data(), cols * sizeof(unsigned short), rows, cols);
    kfusion::cuda::Dists dists;
    kfusion::cuda::TsdfVolume v(p);
    REP(v.clear())
    REP(v.initSphere(make_float3((float) arg("sphere_cx"), (float) arg("sphere_cy"), (float) arg("sphere_cz")), (float) arg("sphere_r")))
    REP(kfusion::cuda::depthBilateralFilter(depth, filtered, p.bilateral_kernel_size, p.bilateral_sigma_spatial, p.bilateral_sigma_depth))
    REP(kfusion::cuda::depthTruncation(filtered, p.icp_truncate_depth_dist))
    REP(kfusion::cuda::computeDists(filtered, dists, p.intr))
    REP(v.integrate(dists, cv::Affine3f::Identity(), p.intr))
#undef REP
    cudaDeviceSynchronize();
}

// wall time of Solver::estimate_psi as the reference runs it (a host sync and a 128 KB read-back per iteration, solver.cu:172):
// `repeat` solves of max_iter iterations from two initSphere volumes; prints seconds per solve (meaningful on the GPU build only)
static void scenario_time() {
    Params p = make_params();
    cv::Ptr<kfusion::cuda::TsdfVolume> pg(new kfusion::cuda::TsdfVolume(p)), pgi(new kfusion::cuda::TsdfVolume(p)), pn(new kfusion::cuda::TsdfVolume(p)),
        pnp(new kfusion::cuda::TsdfVolume(p));
    pg->initSphere(make_float3((float) arg("sphere_cx"), (float) arg("sphere_cy"), (float) arg("sphere_cz")), (float) arg("sphere_r"));
    pn->initSphere(make_float3((float) arg("sphere2_cx"), (float) arg("sphere2_cy"), (float) arg("sphere2_cz")), (float) arg("sphere_r"));
    sobfu::cuda::Solver solver(p);
    std::ofstream out(g_dir + "/out_time.txt");
    for (int r = 0; r < (int) arg("repeat", 3); ++r) {
        auto psi = std::make_shared<sobfu::cuda::DeformationField>(p.volume_dims), psi_inv = std::make_shared<sobfu::cuda::DeformationField>(p.volume_dims);
        cudaDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        solver.estimate_psi(pg, pgi, pn, pnp, psi, psi_inv);
        cudaDeviceSynchronize();
        out << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << "\n";
    }
}

static void scenario_frames() {
    Params p = make_params();
    const int n = (int) arg("frames");
    SobFusion fusion(p);
    for (int f = 0; f < n; ++f) {
        std::vector<unsigned short> raw = read_bin<unsigned short>("depth_" + std::to_string(f), (size_t) p.rows * p.cols);
        kfusion::cuda::Depth depth;
        depth.upload(raw.data(), p.cols * sizeof(unsigned short), p.rows, p.cols);  // demo.cpp:327-329
        fusion(depth);
        const std::string s = "_f" + std::to_string(f);
        dump("phi_global" + s, fusion.phi_global->data());
        if (f > 0) dump("phi_n" + s, fusion.phi_n->data());
        if (f >= p.start_frame && f > 0) {
            dump("psi" + s, fusion.psi->get_data()), dump("psi_inv" + s, fusion.psi_inv->get_data());
            dump("phi_n_psi" + s, fusion.phi_n_psi->data()), dump("phi_global_psi_inv" + s, fusion.phi_global_psi_inv->data());
        }
    }
}

#ifndef REF_HIP_BUILD
static void scenario_mc() {
    Params p = make_params();
    const size_t N = (size_t) p.volume_dims[0] * p.volume_dims[1] * p.volume_dims[2];
    kfusion::cuda::TsdfVolume v(p);
    v.data().upload(read_bin<float>("volume", N * 2).data(), N * 8);
    cuemu::max_threads = 1;  // getOccupiedVoxels appends by atomicAdd: the order of its output is the order blocks run in
    kfusion::cuda::MarchingCubes mc;
    mc.setPose(p.volume_pose);
    kfusion::cuda::DeviceArray<pcl::PointXYZ> vb((size_t) arg("buffer"));
    kfusion::cuda::DeviceArray<pcl::Normal> nb((size_t) arg("buffer"));
    kfusion::cuda::Surface s = mc.run(v, vb, nb);
    std::vector<pcl::PointXYZ> hv(s.vertices.size());
    std::vector<pcl::Normal> hn(s.normals.size());
    if (!hv.empty()) s.vertices.download(hv.data()), s.normals.download(hn.data());
    write_bin("vertices", hv.data(), hv.size() * 16);
    write_bin("normals", hn.data(), hn.size() * 16);
}

#endif  // REF_HIP_BUILD

int main(int argc, char** argv) {
    if (argc < 3) return fprintf(stderr, "usage: ref_emu <scenario> <dir> key=value ...\n"), 2;
    const std::string scenario = argv[1];
    g_dir = argv[2];
    for (int i = 3; i < argc; ++i) {
        const char* eq = strchr(argv[i], '=');
        if (!eq) return fprintf(stderr, "ref_emu: bad argument %s\n", argv[i]), 2;
        g_args[std::string(argv[i], (size_t) (eq - argv[i]))] = atof(eq + 1);
    }
    g_digest = arg("digest", 0.0) != 0.0;
    // identity poses must stay bit-exact identities through the stand-in Affine3 (shim/opencv2/core/affine.hpp)
    {
        cv::Affine3f pose = cv::Affine3f().translate(cv::Vec3f(-0.25f, -0.125f, 0.5f)), v2c = cv::Affine3f::Identity().inv() * pose;
        for (int i = 0; i < 9; ++i) assert(v2c.rotation().val[i] == (i % 4 == 0 ? 1.f : 0.f));
        assert(v2c.translation()[0] == -0.25f && v2c.translation()[1] == -0.125f && v2c.translation()[2] == 0.5f);
    }
    std::ofstream log(g_dir + "/out_log.txt");
    std::streambuf* old = std::cout.rdbuf(log.rdbuf());
    if (scenario == "kernels") scenario_kernels();
    else if (scenario == "solver") scenario_solver();
    else if (scenario == "tsdf") scenario_tsdf();
    else if (scenario == "depth") scenario_depth();
    else if (scenario == "launchers") scenario_launchers();
    else if (scenario == "time") scenario_time();
    else if (scenario == "frames") scenario_frames();
#ifndef REF_HIP_BUILD
    else if (scenario == "mc") scenario_mc();
#endif
    else return fprintf(stderr, "ref_emu: unknown scenario %s\n", scenario.c_str()), 2;
    std::cout.flush();
    std::cout.rdbuf(old);
    return 0;
}
