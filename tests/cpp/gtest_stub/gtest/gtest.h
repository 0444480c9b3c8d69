// Minimal stand-in for the part of GoogleTest the reference's test/*.cpp use (TEST_F, ASSERT_NEAR, ::testing::Test,
// InitGoogleTest, RUN_ALL_TESTS) -- the image has no gtest.  Test infrastructure only: it lets the reference's own test translation
// units be compiled, unchanged and from where they lie under /root/reference, against this repo's include/ (the drop-in claim), and
// run on the HIP path (tests/test_reference_callers.py).
#pragma once
#include <chrono>
#include <cmath>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

namespace testing {
class Test {
public:
    virtual ~Test() = default;
    virtual void SetUp() {}
    virtual void TearDown() {}
    virtual void TestBody() = 0;
    bool failed_ = false;
};
struct Registry {
    struct Entry {
        std::string suite, name;
        std::function<Test*()> make;
    };
    static std::vector<Entry>& all() {
        static std::vector<Entry> v;
        return v;
    }
    static int add(const char* s, const char* n, std::function<Test*()> m) {
        all().push_back(Entry{s, n, std::move(m)});
        return 0;
    }
};
inline void InitGoogleTest(int*, char**) {}
inline int run_all_tests() {
    int failed = 0;
    for (const Registry::Entry& e : Registry::all()) {
        std::printf("[ RUN      ] %s.%s\n", e.suite.c_str(), e.name.c_str());
        std::fflush(stdout);
        const auto t0 = std::chrono::steady_clock::now();
        Test* t = e.make();
        t->SetUp();
        t->TestBody();
        t->TearDown();
        const bool bad = t->failed_;
        delete t;
        const long ms = (long) std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        std::printf(bad ? "[  FAILED  ] %s.%s (%ld ms)\n" : "[       OK ] %s.%s (%ld ms)\n", e.suite.c_str(), e.name.c_str(), ms);
        failed += bad ? 1 : 0;
    }
    std::printf("[==========] %zu tests ran, %d failed\n", Registry::all().size(), failed);
    return failed == 0 ? 0 : 1;
}
}  // namespace testing

#define RUN_ALL_TESTS() ::testing::run_all_tests()
#define TEST_F(fixture, name)                                                                                              \
    class fixture##_##name##_Test : public fixture {                                                                       \
    public:                                                                                                                \
        void TestBody() override;                                                                                          \
    };                                                                                                                     \
    static int fixture##_##name##_registered =                                                                            \
        ::testing::Registry::add(#fixture, #name, [] { return static_cast<::testing::Test*>(new fixture##_##name##_Test()); }); \
    void fixture##_##name##_Test::TestBody()
// ASSERT_*: a failure ends the test body (gtest semantics)
#define ASSERT_NEAR(a, b, tol)                                                                                             \
    do {                                                                                                                   \
        const double gt_a = (double) (a), gt_b = (double) (b), gt_t = (double) (tol);                                      \
        if (!(std::fabs(gt_a - gt_b) <= gt_t)) {                                                                           \
            std::printf("%s:%d: Failure\nThe difference between %s and %s is %g, which exceeds %s (%g vs %g)\n", __FILE__, __LINE__, #a, #b, \
                        std::fabs(gt_a - gt_b), #tol, gt_a, gt_b);                                                         \
            this->failed_ = true;                                                                                          \
            return;                                                                                                        \
        }                                                                                                                  \
    } while (0)
