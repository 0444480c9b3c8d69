// Exercises the C++ shells (include/sobfu_amd/sobfu.hpp) the way the reference's gtest suite exercises the reference
// classes: the six value-pinning cases (test/deformation_field_test.cpp:92-336, test/reductions_test.cpp:86-101)
// restated against the shells, the three solver smoke set-ups (test/solver_test.cpp:109-208) with real assertions,
// and the DeviceMemory ownership rules (src/kfusion/device_memory.cpp:50-131).  Needs a GPU.  Exit code 0 = pass.
#include <cmath>
#include <cstdio>
#include <vector>

#include <sobfu_amd/sobfu.hpp>

static int g_fail = 0, g_checks = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        ++g_checks;                                                          \
        if (!(cond)) {                                                       \
            if (++g_fail <= 20) std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
        }                                                                    \
    } while (0)
#define CHECK_NEAR(a, b, tol) CHECK(std::fabs((double) (a) - (double) (b)) <= (tol))

using kfusion::cuda::CudaData;
using kfusion::cuda::TsdfVolume;

struct Fixture {  // the 64^3 / 0.25 m fixture every reference test builds in SetUp()
    Params params;
    int3 dims;
    float3 vs;
    int n;
    explicit Fixture(float trunc_vox = 10.f) {
        params.volume_dims = cv::Vec3i::all(64);
        params.volume_size = cv::Vec3f::all(0.25f);
        params.tsdf_trunc_dist = trunc_vox * params.volume_size[0] / 64.f;
        params.eta = 2.f * params.volume_size[0] / 64.f;
        params.tsdf_max_weight = 128.f;
        params.gradient_delta_factor = 0.1f;
        params.intr = kfusion::Intr(1.f, 1.f, 0.f, 0.f);
        params.max_iter = 8;
        params.max_update_norm = -1.f;
        params.s = 7;
        params.lambda = 0.1f;
        params.alpha = 0.01f;
        params.w_reg = 0.4f;
        dims = kfusion::device_cast<int3>(params.volume_dims);
        vs = kfusion::device_cast<float3>(params.voxel_sizes());
        n = 64 * 64 * 64;
    }
    size_t idx(int i, int j, int k) const { return (size_t) i + 64 * ((size_t) j + 64 * (size_t) k); }
};

static void upload_field(CudaData& d, const std::vector<float4>& h) { d.upload(h.data(), h.size() * sizeof(float4)); }

static void test_identity_and_memory() {
    Fixture f;
    auto psi = std::make_shared<sobfu::cuda::DeformationField>(f.params.volume_dims);
    std::vector<float4> h(f.n);
    CudaData data = psi->get_data();  // by value: shares the allocation
    CHECK(data.ptr<float4>() == psi->get_data().ptr<float4>());
    data.download(h.data());
    int bad = 0;
    for (int k = 0; k < 64; ++k) for (int j = 0; j < 64; ++j) for (int i = 0; i < 64; ++i) {
        float4 v = h[f.idx(i, j, k)];
        bad += !(v.x == (float) i && v.y == (float) j && v.z == (float) k && v.w == 0.f);
    }
    CHECK(bad == 0);
    // DeviceMemory: copy shares, copyTo clones, swap exchanges, release on last owner
    CudaData a;
    CHECK(a.empty());
    a.create(1024);
    CudaData b = a, c;
    CHECK(b.ptr<char>() == a.ptr<char>() && b.sizeBytes() == 1024);
    a.copyTo(c);
    CHECK(c.ptr<char>() != a.ptr<char>() && c.sizeBytes() == 1024);
    char* pa = a.ptr<char>();
    a.swap(c);
    CHECK(c.ptr<char>() == pa && b.ptr<char>() == pa);
    a.release();
    CHECK(a.empty() && !b.empty());
    CHECK(psi->get_no_nans() == 0);
}

static void test_workspace_structs() {
    // SpatialGradients / set_data / get_no_nans: source-compatibility members of the reference's host classes
    sobfu::cuda::SpatialGradients sg(cv::Vec3i(8, 6, 4));
    CHECK(sg.nabla_U->get_dims()[1] == 6 && sg.J != nullptr && sg.L_o_psi_inv->get_no_nans() == 0);
    sobfu::cuda::VectorField a(cv::Vec3i(8, 6, 4)), b(cv::Vec3i(8, 6, 4));
    kfusion::cuda::CudaData d = a.get_data();
    b.set_data(d);
    CHECK(b.get_data().ptr<float>() == a.get_data().ptr<float>());
}

static void test_tsdf_gradient() {
    Fixture f;
    cv::Ptr<TsdfVolume> phi(new TsdfVolume(f.params));
    phi->initSphere(make_float3(0.16f, 0.16f, 0.16f), 0.01f);
    kfusion::device::TsdfVolume vol = phi->pod();
    CudaData gd;
    gd.create(f.n * sizeof(float4));
    sobfu::device::TsdfGradient grad(gd.ptr<float4>(), f.dims);
    sobfu::device::TsdfDifferentiator diff(vol);
    diff.calculate(grad);
    std::vector<float2> t(f.n);
    std::vector<float4> g(f.n);
    phi->data().download(t.data());
    gd.download(g.data());
    int bad = 0, cnt = 0;
    for (int k = 1; k < 63; ++k) for (int j = 1; j < 63; ++j) for (int i = 1; i < 63; ++i) {
        size_t p = f.idx(i, j, k);
        if (std::fabs(t[p].x) < 1.f) {
            float nrm = std::sqrt(g[p].x * g[p].x + g[p].y * g[p].y + g[p].z * g[p].z);
            bad += !(std::fabs(nrm - f.vs.x / f.params.tsdf_trunc_dist) <= 0.15f);
            ++cnt;
        }
    }
    CHECK(cnt > 1000 && bad == 0);
}

static void test_jacobian_laplacian() {
    Fixture f;
    std::vector<float4> h(f.n);
    std::vector<Mat4f> J(f.n);
    CudaData pd, jd, ld;
    jd.create(f.n * sizeof(Mat4f));
    ld.create(f.n * sizeof(float4));
    auto run = [&](int mode_unused) {
        (void) mode_unused;
        upload_field(pd, h);
        sobfu::device::DeformationField psi(pd.ptr<float4>(), f.dims);
        sobfu::device::Jacobian Jd(jd.ptr<Mat4f>(), f.dims);
        sobfu::device::Differentiator diff(psi);
        diff.calculate(Jd);
        jd.download(J.data());
    };
    // uniform field -> J == 0 everywhere, boundary included
    for (auto& v : h) v = make_float4(1.f, 1.f, 1.f, 0.f);
    run(0);
    int bad = 0;
    for (int p = 0; p < f.n; ++p) for (int r = 0; r < 3; ++r)
        bad += !(std::fabs(J[p].data[r].x) <= 1e-5f && std::fabs(J[p].data[r].y) <= 1e-5f && std::fabs(J[p].data[r].z) <= 1e-5f);
    CHECK(bad == 0);
    // psi = (i, j, k) -> J == I on the interior
    for (int k = 0; k < 64; ++k) for (int j = 0; j < 64; ++j) for (int i = 0; i < 64; ++i) h[f.idx(i, j, k)] = make_float4(i, j, k, 0.f);
    run(0);
    bad = 0;
    for (int k = 1; k < 63; ++k) for (int j = 1; j < 63; ++j) for (int i = 1; i < 63; ++i) {
        const Mat4f& m = J[f.idx(i, j, k)];
        bad += !(std::fabs(m.data[0].x - 1) <= 1e-5f && std::fabs(m.data[1].y - 1) <= 1e-5f && std::fabs(m.data[2].z - 1) <= 1e-5f &&
                 std::fabs(m.data[0].y) <= 1e-5f && std::fabs(m.data[0].z) <= 1e-5f && std::fabs(m.data[1].x) <= 1e-5f &&
                 std::fabs(m.data[1].z) <= 1e-5f && std::fabs(m.data[2].x) <= 1e-5f && std::fabs(m.data[2].y) <= 1e-5f);
    }
    CHECK(bad == 0);
    // psi = (i(1-j), exp(-k)+j, k): analytic J and NEGATIVE Laplacian, tolerance 0.1
    for (int k = 0; k < 64; ++k) for (int j = 0; j < 64; ++j) for (int i = 0; i < 64; ++i)
        h[f.idx(i, j, k)] = make_float4(i * (1.f - j), std::exp(-(float) k) + j, k, 0.f);
    run(0);
    sobfu::device::DeformationField psi(pd.ptr<float4>(), f.dims);
    sobfu::device::Laplacian Ld(ld.ptr<float4>(), f.dims);
    sobfu::device::SecondOrderDifferentiator so(psi);
    so.calculate(Ld);
    std::vector<float4> L(f.n);
    ld.download(L.data());
    bad = 0;
    for (int k = 1; k < 63; ++k) for (int j = 1; j < 63; ++j) for (int i = 1; i < 63; ++i) {
        const Mat4f& m = J[f.idx(i, j, k)];
        const float e = std::exp(-(float) k), tol = 0.1f;
        bad += !(std::fabs(m.data[0].x - (1.f - j)) <= tol && std::fabs(m.data[0].y + i) <= tol && std::fabs(m.data[0].z) <= tol &&
                 std::fabs(m.data[1].x) <= tol && std::fabs(m.data[1].y - 1.f) <= tol && std::fabs(m.data[1].z + e) <= tol &&
                 std::fabs(m.data[2].x) <= tol && std::fabs(m.data[2].y) <= tol && std::fabs(m.data[2].z - 1.f) <= tol);
        const float4 l = L[f.idx(i, j, k)];
        bad += !(std::fabs(l.x) <= tol && std::fabs(l.y + e) <= tol && std::fabs(l.z) <= tol);
    }
    CHECK(bad == 0);
}

static void test_data_term() {
    Fixture f(5.f);
    cv::Ptr<TsdfVolume> pg(new TsdfVolume(f.params)), pn(new TsdfVolume(f.params));
    kfusion::device::TsdfVolume pn_dev = pn->pod();
    kfusion::device::clear_volume(pn_dev);
    pg->initSphere(make_float3(5.f, 5.f, 5.f), 0.01f);  // far outside: all ones
    sobfu::device::Reductor r(f.dims, f.vs.x, f.params.tsdf_trunc_dist);
    float e = r.data_energy(pg->data().ptr<float2>(), pn->data().ptr<float2>());
    CHECK_NEAR(e, 0.5f * f.n, 0.1);
    CHECK(r.blocks == 256 && r.threads == 512);
}

static double field_l2(sobfu::cuda::DeformationField& psi, int n) {
    std::vector<float4> h(n);
    psi.get_data().download(h.data());
    double s = 0;
    for (int k = 0; k < 64; ++k) for (int j = 0; j < 64; ++j) for (int i = 0; i < 64; ++i) {
        float4 v = h[(size_t) i + 64 * ((size_t) j + 64 * (size_t) k)];
        s += (double) (v.x - i) * (v.x - i) + (double) (v.y - j) * (v.y - j) + (double) (v.z - k) * (v.z - k);
    }
    return std::sqrt(s);
}

static void test_solver_alignment() {
    Fixture f;
    f.params.max_iter = 10;
    f.params.verbosity = 2;
    cv::Ptr<TsdfVolume> pg(new TsdfVolume(f.params)), pgi(new TsdfVolume(f.params)), pn(new TsdfVolume(f.params)), pnp(new TsdfVolume(f.params));
    auto psi = std::make_shared<sobfu::cuda::DeformationField>(f.params.volume_dims);
    auto psi_inv = std::make_shared<sobfu::cuda::DeformationField>(f.params.volume_dims);
    auto solver = std::make_shared<sobfu::cuda::Solver>(f.params);
    pg->initSphere(make_float3(0.13f, 0.13f, 0.13f), 0.012f);
    pn->initSphere(make_float3(0.125f, 0.13f, 0.13f), 0.012f);
    pnp->initSphere(make_float3(0.125f, 0.13f, 0.13f), 0.012f);
    solver->estimate_psi(pg, pgi, pn, pnp, psi, psi_inv);
    const sobfu_hip_solver_report& r = solver->last_report;
    CHECK(r.iterations == 10 && r.converged == 0);
    // SURVEY Appendix B run 1 (values printed by the reference for this set-up); init_sphere differs from the oracle by
    // <= a few ulp (powf), so compare to the printed precision only
    CHECK_NEAR(r.last_e_data, 24.4234, 2e-3);
    CHECK_NEAR(r.last_e_reg, 0.00274294, 2e-6);
    CHECK_NEAR(r.last_max_update_norm, 0.000380488, 2e-8);
    CHECK_NEAR(field_l2(*psi, f.n), 0.197894352, 2e-5);
    // serial second frame (SerialAlignmentTest shape): psi persists, apply() between the solves
    pn->clear();
    pnp->clear();
    pn->initSphere(make_float3(0.123f, 0.13f, 0.13f), 0.012f);
    psi->apply(pn, pnp);
    double before = field_l2(*psi, f.n);
    solver->estimate_psi(pg, pgi, pn, pnp, psi, psi_inv);
    CHECK(field_l2(*psi, f.n) > before);
    CHECK(psi->get_no_nans() == 0 && psi_inv->get_no_nans() == 0);
    // fusion through the class surface
    pg->integrate(*pnp);
    std::vector<float2> h(f.n);
    pg->data().download(h.data());
    float wmax = 0;
    for (auto& v : h) wmax = std::fmax(wmax, v.y);
    CHECK(wmax == 2.f);
}

static void test_depth_integration() {
    // params_advent.ini values on a flat wall 0.75 m away: every observed voxel's tsdf follows (0.75*ray_scale - z)/trunc
    Params p;
    p.volume_dims = cv::Vec3i::all(64);
    p.volume_size = cv::Vec3f::all(0.5f);
    p.tsdf_trunc_dist = 5.f * 0.5f / 64.f;
    p.eta = 2.f * 0.5f / 64.f;
    p.tsdf_max_weight = 128.f;
    p.intr = kfusion::Intr(570.342f, 570.342f, 320.f, 240.f);
    p.volume_pose = cv::Affine3f().translate(cv::Vec3f(-0.25f, -0.25f, 0.5f));
    std::vector<unsigned short> img(640 * 480, 750);
    kfusion::cuda::Depth depth, filtered;
    depth.upload(img.data(), 640 * sizeof(unsigned short), 480, 640);
    kfusion::cuda::depthBilateralFilter(depth, filtered, 7, 4.5f, 0.005f);
    kfusion::cuda::depthTruncation(filtered, 1.5f);
    kfusion::cuda::Dists dists;
    kfusion::cuda::computeDists(filtered, dists, p.intr);
    TsdfVolume vol(p);
    vol.integrate(dists, cv::Affine3f::Identity(), p.intr);
    std::vector<float2> h(64 * 64 * 64);
    vol.data().download(h.data());
    int pos = 0, neg = 0, mid = 0;
    for (auto& v : h) { pos += v.x == 1.f; neg += v.x == -1.f; mid += (v.x != 0.f && std::fabs(v.x) < 1.f); }
    CHECK(pos > 1000 && neg > 1000 && mid > 1000);
    // centre column: sign change where the voxel centre crosses z = 0.75 m (volume z origin at 0.5 m)
    int zc = (int) ((0.75f - 0.5f) / (0.5f / 64.f));
    CHECK(h[32 + 64 * (32 + 64 * (size_t) (zc - 2))].x > 0.f && h[32 + 64 * (32 + 64 * (size_t) (zc + 2))].x < 0.f);
}

static void test_marching_cubes() {
    // kfusion::cuda::MarchingCubes on an analytic sphere: closed surface (zero total area vector), vertices on the sphere,
    // enclosed volume; the pose translates the mesh; an unobserved volume gives an empty Surface
    Params p;
    p.volume_dims = cv::Vec3i::all(64);
    p.volume_size = cv::Vec3f::all(0.5f);
    p.tsdf_trunc_dist = 5.f * 0.5f / 64.f;
    p.eta = 2.f * 0.5f / 64.f;
    p.tsdf_max_weight = 128.f;
    TsdfVolume vol(p);
    kfusion::cuda::MarchingCubes mc;
    kfusion::cuda::DeviceArray<kfusion::cuda::Point> vb;
    kfusion::cuda::DeviceArray<kfusion::cuda::Normal> nb;
    CHECK(mc.run(vol, vb, nb).vertices.empty());
    const float cx = 0.25f, cy = 0.26f, cz = 0.24f, r = 0.1f;
    float3 c; c.x = cx; c.y = cy; c.z = cz;
    vol.initSphere(c, r);
    mc.setPose(cv::Affine3f().translate(cv::Vec3f(1.f, 2.f, 3.f)));
    kfusion::cuda::Surface s = mc.run(vol, vb, nb);
    std::vector<float4> v, n;
    s.vertices.download(v);
    s.normals.download(n);
    CHECK(v.size() == n.size() && v.size() % 3 == 0 && v.size() > 10000);
    double ax = 0, ay = 0, az = 0, volume = 0, rmin = 1e9, rmax = 0;
    for (size_t i = 0; i < v.size(); i += 3) {
        double q[3][3];
        for (int k = 0; k < 3; ++k) {  // undo store_point's flip and the pose translation
            q[k][0] = v[i + k].x - 1.0; q[k][1] = -v[i + k].y - 2.0; q[k][2] = -v[i + k].z - 3.0;
            double d = std::sqrt((q[k][0] - cx) * (q[k][0] - cx) + (q[k][1] - cy) * (q[k][1] - cy) + (q[k][2] - cz) * (q[k][2] - cz));
            rmin = std::min(rmin, d); rmax = std::max(rmax, d);
        }
        double e1[3] = {q[1][0] - q[0][0], q[1][1] - q[0][1], q[1][2] - q[0][2]}, e2[3] = {q[2][0] - q[0][0], q[2][1] - q[0][1], q[2][2] - q[0][2]};
        ax += 0.5 * (e1[1] * e2[2] - e1[2] * e2[1]); ay += 0.5 * (e1[2] * e2[0] - e1[0] * e2[2]); az += 0.5 * (e1[0] * e2[1] - e1[1] * e2[0]);
        volume += (q[0][0] * (q[1][1] * q[2][2] - q[1][2] * q[2][1]) - q[0][1] * (q[1][0] * q[2][2] - q[1][2] * q[2][0]) +
                   q[0][2] * (q[1][0] * q[2][1] - q[1][1] * q[2][0])) / 6.0;
    }
    const double vs = 0.5 / 64.0;
    CHECK(rmin > r - 0.05 * vs && rmax < r + 0.05 * vs);
    CHECK(std::fabs(ax) < 1e-7 && std::fabs(ay) < 1e-7 && std::fabs(az) < 1e-7);
    CHECK(std::fabs(std::fabs(volume) / (4.0 / 3.0 * 3.14159265358979 * r * r * r) - 1.0) < 5e-3);
    CHECK(v[0].w == 1.f && n[0].w == 1.f && std::fabs(n[0].x * n[0].x + n[0].y * n[0].y + n[0].z * n[0].z - 1.f) < 1e-5f);
}

int main() {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        std::printf("host_shell_tests: no HIP device\n");
        return 2;
    }
    kfusion::cuda::setDevice(0);
    test_identity_and_memory();
    test_workspace_structs();
    test_tsdf_gradient();
    test_jacobian_laplacian();
    test_data_term();
    test_solver_alignment();
    test_depth_integration();
    test_marching_cubes();
    std::printf("host_shell_tests: %d checks, %d failed\n", g_checks, g_fail);
    return g_fail ? 1 : 0;
}
