// Calibration for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters in the integrate kernel's own
// access pattern: one dword per lane (256 B per wave-instruction), three arrays read and three
// written, a known number of bytes well past the 256 MiB Infinity Cache.  Not part of the
// product; built and run on the GPU box by profiles/tools/collect_pmc.sh.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256) k_calib_state_update(float *v, int *w, float *g, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float a = v[i], b = g[i];
        const int c = w[i];
        v[i] = a * 0.5f + 1.0f; g[i] = b * 0.5f; w[i] = c + 1;
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main() {
    const size_t n = (size_t)96 << 20;  // 96 Mi elements: 384 MiB per array, 1152 MiB read + 1152 MiB written
    float *v, *g; int *w;
    CK(hipMalloc(&v, n * 4)); CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&w, n * 4));
    CK(hipMemset(v, 0, n * 4)); CK(hipMemset(g, 0, n * 4)); CK(hipMemset(w, 0, n * 4));
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k_calib_state_update, dim3(8192), dim3(256), 0, 0, v, w, g, n);
    CK(hipDeviceSynchronize());
    printf("{\"calib_bytes_read\": %zu, \"calib_bytes_written\": %zu}\n", n * 12, n * 12);
    return 0;
}
