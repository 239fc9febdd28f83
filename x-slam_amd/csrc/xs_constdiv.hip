// xs_constdiv.hip — the exhaustive check behind xs::ConstDiv (xs_device.h): for a constant c, every one of the 2^32 float
// operands through the short division and through the IEEE divide, compared bit for bit on the device; the verdicts are kept in a
// small host-side table that the launchers consult (xs_const_div_get).  A constant that has not been prepared, or that fails, is
// divided by with the IEEE sequence — never silently with the short form.
#include "xs_device.h"
#include "xs_env.h"
#include "../../include/xslam_amd.h"
#include <math.h>
#include <mutex>

using namespace xs;

__global__ void __launch_bounds__(256) k_const_div_check(float c, float rc, unsigned long long *bad) {
    // bad[0]: operands in the short form's domain (2^-60 <= |x| <= 2^60, +-0) where it differs from x / c
    // bad[1]: operands with |x| <= 2^60 where floor(short form) differs from floor(x / c)
    const unsigned base = (blockIdx.x * blockDim.x + threadIdx.x) * 64u;
    unsigned b0 = 0, b1 = 0;
    for (unsigned i = 0; i < 64; ++i) {
        const unsigned bits = base + i;
        const float x = __uint_as_float(bits);
        if (!(fabsf(x) <= 0x1p60f)) continue;
        const float q = div_short(x, c, rc), d = x / c;
        const unsigned u = bits & 0x7fffffffu;
        if ((u == 0u || u - 0x21800000u <= 0x5d800000u - 0x21800000u) && __float_as_uint(q) != __float_as_uint(d)) ++b0;
        if (__float2int_rd(q) != __float2int_rd(d)) ++b1;
    }
    if (b0) atomicAdd(bad, (unsigned long long)b0);
    if (b1) atomicAdd(bad + 1, (unsigned long long)b1);
}

namespace {
struct Entry { unsigned bits; unsigned ok; };
std::mutex g_mu;
Entry g_tab[32];
int g_n = 0;
bool g_enabled = !xs::exp_env_set("XS_CONST_DIV_OFF");   // (measurement aid of an XS_EXPERIMENTS build; the product switch is xs_const_div_enable)
}

/* Test aid: with on = 0 every launcher divides (xs_const_div_state reads 0) whatever has been prepared; returns the previous setting. */
extern "C" int xs_const_div_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    const int was = g_enabled ? 1 : 0;
    g_enabled = on != 0;
    return was;
}

/* 0 = not prepared (or failed): divide.  Host only, no device work. */
extern "C" unsigned xs_const_div_state(float c) {
    const float m = fabsf(c);
    union { float f; unsigned u; } cv; cv.f = m;
    const unsigned bits = cv.u;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_enabled) return 0;
    for (int i = 0; i < g_n; ++i) if (g_tab[i].bits == bits) return g_tab[i].ok;
    return 0;
}
/* The launchers' view: |c|, RN(1 / |c|) and the verdict (the caller applies the sign of c). */
xs::ConstDiv xs_const_div_get(float c) {
    xs::ConstDiv d;
    d.c = fabsf(c); d.rc = 1.0f / d.c; d.ok = xs_const_div_state(c);
    return d;
}
/* Checks constant c on the current device (all 2^32 operands, ~2 ms + one synchronisation, on the null stream) unless it is already
 * in the table; returns its verdict bits (1: short division exact on its domain, 2: floor exact), 0 if the device could not be used.
 * Call once per constant, at set-up time: the orchestrator does for voxel_size, fx and fy.
 * The table is per process, keyed by the constant alone: a verdict found on one device is used on every device of the process — sound
 * because the check exercises IEEE operations (multiply, fma, divide, floor) that every gfx942 / gfx950 device evaluates identically
 * (this library builds for nothing else).  It holds 32 constants; a 33rd is still checked and its verdict returned, but not kept:
 * xs_const_div_state then reports 0 for it and the launchers take the IEEE divide (correct, just not the short form). */
extern "C" unsigned xs_const_div_prepare(float c) {
    const float m = fabsf(c);
    if (!(m >= 0x1p-20f && m <= 0x1p20f)) return 0;   // (also rejects NaN / 0 / inf; the bound is what floor_div_by's saturation argument needs)
    union { float f; unsigned u; } cv; cv.f = m;
    const unsigned bits = cv.u;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (int i = 0; i < g_n; ++i) if (g_tab[i].bits == bits) return g_tab[i].ok;
    }
    unsigned long long *bad = nullptr, h[2] = {~0ull, ~0ull};
    if (hipMalloc(&bad, 16) != hipSuccess) { (void)hipGetLastError(); return 0; }
    unsigned ok = 0;
    if (hipMemset(bad, 0, 16) == hipSuccess) {
        hipLaunchKernelGGL(k_const_div_check, dim3(1u << 18), dim3(256), 0, 0, m, 1.0f / m, bad);
        if (hipGetLastError() == hipSuccess && hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost) == hipSuccess)
            ok = (h[0] == 0 ? 1u : 0u) | (h[1] == 0 ? 2u : 0u);
    }
    (void)hipFree(bad);
    (void)hipGetLastError();
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < g_n; ++i) if (g_tab[i].bits == bits) return g_tab[i].ok;
    if (g_n < 32) g_tab[g_n++] = Entry{bits, ok};
    return ok;
}
