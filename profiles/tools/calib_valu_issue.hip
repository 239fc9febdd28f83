// VALU issue calibration: cycles per wave-instruction for the instruction classes the integrate kernel is made of,
// with 1, 2, 4 and 8 waves per SIMD resident (256-thread workgroups, a grid of CUs x waves-per-SIMD), every wave running a long independent
// stream of ONE instruction class.  cycles = s_memtime delta of the loop / instructions issued by the wave; the per-SIMD cost of one
// wave-instruction is cycles / (waves per SIMD): that is the figure an issue-bound kernel's floor is computed with.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *cyc, int iters) {
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, dt = 0.0;
    const double dk = 1.0000001;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (OP == 1) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(*(double *)&a0), "+v"(*(double *)&a2), "+v"(*(double *)&a4), "+v"(*(double *)&a6) : "v"(*(const double *)&b), "v"(*(const double *)&c));) }
        if (OP == 2) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 3) { REP16(asm volatile("v_div_fixup_f32 %0, %0, %8, %9\n v_div_fixup_f32 %1, %1, %8, %9\n v_div_fixup_f32 %2, %2, %8, %9\n v_div_fixup_f32 %3, %3, %8, %9\n v_div_fixup_f32 %4, %4, %8, %9\n v_div_fixup_f32 %5, %5, %8, %9\n v_div_fixup_f32 %6, %6, %8, %9\n v_div_fixup_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (OP == 4) { REP16(asm volatile("v_cvt_i32_f32 %0, %0\n v_cvt_f32_i32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_f32_i32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_f32_i32 %2, %2\n v_cvt_i32_f32 %3, %3\n v_cvt_f32_i32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 5) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(*(unsigned long long *)&a0), "+v"(*(unsigned long long *)&a2) : "v"(b), "v"(c) : "vcc");) }
        if (OP == 6) { REP16(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 3\n v_readlane_b32 s20, %0, 5\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 5\n v_readlane_b32 s23, %3, 5" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23");) }
        if (OP == 7) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %5, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %5, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");) }
        // (round 5: the double-precision classes of the ICP fold — float entries widened and added in double)
        if (OP == 8) { REP16(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n v_cvt_f64_f32 %0, %5\n v_cvt_f64_f32 %1, %6\n v_cvt_f64_f32 %2, %7\n v_cvt_f64_f32 %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));) }
        if (OP == 9) { REP16(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dk));) }
        if (OP == 10) { REP16(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dk));) }
        if (OP == 11) { REP16(asm volatile("v_cvt_f64_f32 %4, %5\n v_add_f64 %0, %0, %4\n v_cvt_f64_f32 %4, %6\n v_add_f64 %1, %1, %4\n v_cvt_f64_f32 %4, %7\n v_add_f64 %2, %2, %4\n v_cvt_f64_f32 %4, %8\n v_add_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(dt) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));) }
        if (OP == 12) { REP16(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + dt);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int OP> static void run(const char *name, int cus) {
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = cus * wps;   // 256-thread workgroups: one wave on each SIMD of a CU per workgroup
        float *out; unsigned long long *cyc;
        hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&cyc, (size_t)blocks * 4 * 8);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 10);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        // wall-clock cross-check: every SIMD of the chip issues (waves per SIMD) x iters x 128 instructions during the launch
        const double ns_per = ms * 1e6 / ((double)wps * iters * 128.0);
        std::vector<unsigned long long> h((size_t)blocks * 4);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2], per = med / (iters * 128.0);
        printf("%-26s %d waves/SIMD: %7.2f s_memtime ticks per wave-instruction as the wave sees it (%5.2f per SIMD slot); wall clock: %6.3f ns of its SIMD per wave-instruction = %5.2f cycles at 2.4 GHz (launch %.1f us)\n", name, wps, per, per / wps, ns_per, ns_per * 2.4, ms * 1e3);
        hipFree(out); hipFree(cyc);
    }
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs\n", p.name, cus);
    run<0>("v_fma_f32", cus); run<1>("v_pk_fma_f32", cus); run<2>("v_rcp_f32", cus); run<3>("v_div_fixup_f32", cus);
    run<4>("v_cvt_i32_f32 / f32_i32", cus); run<5>("v_mad_u64_u32", cus); run<6>("v_readlane_b32", cus); run<7>("v_cmp + v_cndmask", cus);
    run<8>("v_cvt_f64_f32", cus); run<9>("v_add_f64", cus); run<10>("v_fma_f64", cus); run<11>("v_cvt_f64_f32 + v_add_f64", cus); run<12>("v_sqrt_f32", cus);
    return 0;
}
