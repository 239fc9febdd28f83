#include <hip/hip_runtime.h>
__global__ void k(const float *p, float *q, int n) {
    __shared__ unsigned s[4][3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s[wave][0][lane] = 0x80000000u; s[wave][1][lane] = 0x80000000u; s[wave][2][lane] = 0x80000000u;
    const unsigned lds = (unsigned)(size_t)&s[wave][0][0];
    const unsigned l0 = __builtin_amdgcn_readfirstlane(lds), l1 = l0 + 256, l2 = l0 + 512;
    const unsigned voff = lane * 4 + wave * 256;
    unsigned keep;
    const float *p1 = p + n, *p2 = p + 2 * n;
    if (lane & 1)
    asm volatile("s_mov_b32 %0, m0\n\ts_waitcnt lgkmcnt(0)\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\t"
                 "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dword %1, %4\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(p), "s"(p1), "s"(p2), "s"(l0), "s"(l1), "s"(l2) : "memory");
    unsigned w;
    int spins = 0;
    do { w = *(volatile unsigned *)&s[wave][2][lane]; if (++spins > 100000) break; } while (__builtin_amdgcn_ballot_w64((lane & 1) && w == 0x80000000u));
    q[threadIdx.x] = __uint_as_float(s[wave][0][lane]) + __uint_as_float(s[wave][1][lane]) + __uint_as_float(w) + spins * 1000.f;
}
int main() {
    float *p, *q; int n = 1024;
    hipMalloc(&p, 3 * n * 4); hipMalloc(&q, 256 * 4);
    float h[3 * 1024]; for (int i = 0; i < 3 * n; ++i) h[i] = i % 1024 + (i / 1024) * 0.25f;
    hipMemcpy(p, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, p, q, n);
    float r[256]; hipMemcpy(r, q, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) { float e = (i & 1) ? 3.f * i + 0.75f : 0.f; float got = r[i] - 1000.f * (int)(r[i] / 1000.f); if ((i & 1) && fabsf(got - e) > 1e-3) { ++bad; if (bad < 5) printf("lane %d got %f (raw %f) want %f\n", i, got, r[i], e); } }
    printf("bad %d  spins(lane1) %d r[1]=%f r[0]=%f\n", bad, (int)(r[1] / 1000.f), r[1], r[0]);
    return bad != 0;
}
