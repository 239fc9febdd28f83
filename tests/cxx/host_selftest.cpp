// Host-side self-test, built by tests/test_abi_cpu.py with -fsanitize=address,undefined (-fno-sanitize-recover): the header-only host code of
// x-slam_amd/host that needs no GPU — the flat YAML reader on well-formed and malformed text, the fixed-size complex algebra (4x4 / 3x3
// inverses against products, se3Exp against its inverse, the 6x6 solvers against their residuals), DoubleComplex arithmetic identities.
// Exit code 0 = every check held and the sanitizers saw nothing.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>

#include "DoubleComplex.h"
#include "flat_yaml.hpp"
#include "host_algebra.hpp"

using namespace xs_host;

static int failures = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

static void yaml() {
    const FlatYaml y = FlatYaml::Load("# comment only\n\nkey_a: 12   # trailing\nkey_b: \"quoted # not a comment\"\n  spaced  :   3.5  \nflag: true\n"
                                      "no_colon_line\n: value_without_key\nempty_value:\nlast_line_no_newline: 7");
    CHECK(y.as<int>("key_a") == 12);
    CHECK(y.as<std::string>("key_b") == "quoted # not a comment");
    CHECK(y.as<float>("spaced") == 3.5f);
    CHECK(y.as<bool>("flag") && !y.as<bool>("key_a", false) && y.as<bool>("absent", true));
    CHECK(y.has("empty_value") && y.as<std::string>("empty_value").empty() && y.as<int>("empty_value") == 0);
    CHECK(y.as<int>("last_line_no_newline") == 7 && !y.has("no_colon_line") && y.items().size() == 6);
    bool threw = false;
    try { (void)y.as<int>("absent"); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
    threw = false;
    try { (void)FlatYaml::LoadFile("/nonexistent/dir/config.yaml"); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
    // hostile shapes: a lone quote, only separators, very long lines, embedded NULs, CR LF endings
    const FlatYaml q = FlatYaml::Load("a: \"\nb: '\nc: \"x\nd:\"\"\n:::\n\"\"\"\n#\n");
    CHECK(q.as<std::string>("a") == "\"" && q.as<std::string>("d").empty());
    std::string big(1 << 16, 'k');
    big += ": ";
    big += std::string(1 << 16, 'v');
    CHECK(FlatYaml::Load(big).items().size() == 1);
    const FlatYaml crlf = FlatYaml::Load("x: 1\r\ny: two\r\n");
    CHECK(crlf.as<int>("x") == 1 && crlf.as<std::string>("y") == "two");
    std::string nul("k: a");
    nul.push_back('\0');
    nul += "b\nz: 9\n";
    CHECK(FlatYaml::Load(nul).as<int>("z") == 9);
    CHECK(FlatYaml::Load("").items().empty());
}

static float cabs_max(const Matrix4cf &a, const Matrix4cf &b) {
    float m = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m = std::fmax(m, std::abs(a.m[i][j] - b.m[i][j]));
    return m;
}

static void algebra() {
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    Matrix4cf eye;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) eye.m[i][j] = hostComplex(i == j ? 1.f : 0.f, 0.f);
    for (int trial = 0; trial < 200; ++trial) {
        hostComplex xi[6];
        for (int k = 0; k < 6; ++k) xi[k] = hostComplex(u(rng) * (k < 3 ? 1.f : 0.5f), trial % 3 == 0 ? 1e-7f * u(rng) : 0.f);
        if (trial % 10 == 0) for (int k = 3; k < 6; ++k) xi[k] = hostComplex(0.f, k == 3 + trial % 3 ? 1e-7f : 0.f);   // (the small-angle branch)
        const Matrix4cf T = se3Exp(xi);
        const Matrix4cf Ti = inverse(T);
        CHECK(cabs_max(T * Ti, eye) < 2e-5f);
        hostComplex neg[6];
        for (int k = 0; k < 6; ++k) neg[k] = -xi[k];
        CHECK(cabs_max(se3Exp(neg), Ti) < 2e-5f);
        const Matrix3cf R = GetRotation(T), Ri = inverse(R);
        const Matrix3cf P = R * Ri;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) CHECK(std::abs(P.m[i][j] - hostComplex(i == j ? 1.f : 0.f, 0.f)) < 2e-5f);
        Vector3cf x; for (int i = 0; i < 3; ++i) x.v[i] = hostComplex(u(rng), 0.f);
        const Vector3cf back = Ri * (R * x);
        for (int i = 0; i < 3; ++i) CHECK(std::abs(back.v[i] - x.v[i]) < 2e-5f);
        for (int axis = 0; axis < 3; ++axis) {
            const Matrix3cf A = angle_axis(hostComplex(u(rng), 0.f), axis);
            const Matrix3cf AAi = A * inverse(A);
            for (int i = 0; i < 3; ++i) CHECK(std::abs(AAi.m[i][i] - hostComplex(1.f, 0.f)) < 2e-5f);
        }
        // 6 x 6 systems: J^T J + small ridge, complex (the ICP's) and real (the Gauss-Newton loop's)
        double J[12][6], A[36] = {0}, b[6] = {0}, xs[6];
        for (auto &row : J) for (double &e : row) e = u(rng);
        for (int r = 0; r < 12; ++r) for (int i = 0; i < 6; ++i) { b[i] += J[r][i] * u(rng); for (int j = 0; j < 6; ++j) A[i * 6 + j] += J[r][i] * J[r][j]; }
        for (int i = 0; i < 6; ++i) A[i * 6 + i] += 1e-3;
        CHECK(solve_spd6(A, b, xs));
        for (int i = 0; i < 6; ++i) { double r = -b[i]; for (int j = 0; j < 6; ++j) r += A[i * 6 + j] * xs[j]; CHECK(std::fabs(r) < 1e-9); }
        hostComplexICP Ac[36], bc[6], xc[6];
        for (int i = 0; i < 36; ++i) Ac[i] = hostComplexICP(A[i], 1e-9 * u(rng));
        for (int i = 0; i < 6; ++i) bc[i] = hostComplexICP(b[i], 1e-9 * u(rng));
        llt_solve6(Ac, bc, xc);
        for (int i = 0; i < 6; ++i) { hostComplexICP r = -bc[i]; for (int j = 0; j < 6; ++j) r += Ac[i * 6 + j] * xc[j]; CHECK(std::abs(r) < 1e-8); }
        CHECK(real_determinant6(Ac) > 0);
    }
    double Z[36] = {0}, zb[6] = {1, 1, 1, 1, 1, 1}, zx[6];
    CHECK(!solve_spd6(Z, zb, zx));            // a singular system is refused, not divided through
    Z[0] = -1; CHECK(!solve_spd6(Z, zb, zx));
}

static void double_complex() {
    std::mt19937 rng(11);
    std::uniform_real_distribution<float> u(0.5f, 2.f);
    for (int trial = 0; trial < 200; ++trial) {
        const DoubleComplex a(u(rng), 1e-6f * u(rng), 1e-6f * u(rng), 0.f), b(u(rng), 1e-6f * u(rng), 1e-6f * u(rng), 0.f);
        const DoubleComplex q = (a * b) / b;
        CHECK(std::fabs(q.real().real() - a.real().real()) < 1e-5f);
        const DoubleComplex s = sqrt(a * a);
        CHECK(std::fabs(s.real().real() - a.real().real()) < 1e-5f);
        const DoubleComplex e = log(exp(a));
        CHECK(std::fabs(e.real().real() - a.real().real()) < 1e-5f);
        DoubleComplex c = a; c += b; c -= b; c *= 2.f; c /= 2.f;
        CHECK(std::fabs(c.real().real() - a.real().real()) < 1e-5f);
        DoubleComplex p(u(rng)); p.addPerturbation();
        const DoubleComplex cube = p * p * p;                       // d/dx x^3 = 3 x^2, d2/dx2 = 6 x
        const float x = p.real().real();
        CHECK(std::fabs(cube.real().imag() / 1e-6f - 3 * x * x) < 1e-3f * 3 * x * x);
        CHECK(std::fabs(cube.imag().imag() / 1e-12f - 6 * x) < 2e-2f * 6 * x);
        p.clearPerturbation();
        CHECK(p.real().imag() == 0 && p.imag().real() == 0);
    }
}

int main() {
    yaml();
    algebra();
    double_complex();
    if (failures) { std::printf("%d checks failed\n", failures); return 1; }
    std::printf("host self-test: all checks held\n");
    return 0;
}
