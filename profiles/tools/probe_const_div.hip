// Exhaustive check of x / c == fma(fma(q, c, -x), -r, q), q = x * r, r = RN(1/c), over every float x, for a list of constants c > 0:
// prints, per constant, the number of operands (by magnitude class) where the two differ bitwise.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
__global__ void k(float c, float r, unsigned long long *out) {
    // out[0]: mismatches with |x| <= 2^60 (incl. zeros, denormals); out[1]: of those, with |x| >= 2^-60; out[2]: min |x| bits of a mismatch in range; out[3]: max
    const unsigned base = (blockIdx.x * blockDim.x + threadIdx.x);
    unsigned long long bad = 0, bad_mid = 0; unsigned lo = 0xffffffffu, hi = 0;
    for (unsigned i = 0; i < 64; ++i) {
        const unsigned bits = base * 64u + i;
        const float x = __uint_as_float(bits);
        const float t = fabsf(x);
        if (!(t <= 0x1p60f)) continue;
        float q = x * r;
        const float e = __builtin_fmaf(q, c, -x);
        q = __builtin_fmaf(e, -r, q);
        const float d = x / c;
        if (__float_as_uint(q) != __float_as_uint(d)) { ++bad; if (t >= 0x1p-60f) ++bad_mid; lo = min(lo, __float_as_uint(t)); hi = max(hi, __float_as_uint(t)); }
    }
    if (bad) { atomicAdd(out, bad); atomicAdd(out + 1, bad_mid); atomicMin((unsigned *)(out + 2), lo); atomicMax((unsigned *)(out + 3), hi); }
}
int main(int argc, char **argv) {
    unsigned long long *d; hipMalloc(&d, 32);
    for (int a = 1; a < argc; ++a) {
        const float c = (float)atof(argv[a]), r = 1.0f / c;
        unsigned long long h[4] = {0, 0, 0xffffffffull, 0};
        hipMemcpy(d, h, 32, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(1u << 18), dim3(256), 0, 0, c, r, d);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        float lo, hi; unsigned l = (unsigned)h[2], u = (unsigned)h[3]; memcpy(&lo, &l, 4); memcpy(&hi, &u, 4);
        printf("c = %.9g  r = %.9g: %llu mismatches among |x| <= 2^60 (%llu of them with |x| >= 2^-60), |x| of mismatches in [%g, %g]; %.2f ms\n", c, r, h[0], h[1], h[0] ? lo : 0.f, h[0] ? hi : 0.f, ms);
    }
    return 0;
}
