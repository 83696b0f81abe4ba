// Micro-benchmark (development aid, not loaded by the library): issue rate of v_fma_f32, v_fma_f64, v_rcp_f64 and the fp64
// division sequence, per SIMD and clock, eight independent chains per lane so that latency does not bound it.
// hipcc -O3 --offload-arch=gfx950 -o tools/fma_rate tools/fma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <class T, int OP>
__global__ void __launch_bounds__(256) k(T *out, T a, T b, int iters) {
    T x[8];
    for (int k = 0; k < 8; k++) x[k] = (T)(threadIdx.x + k) * (T)1e-3 + (T)1;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (OP == 0) x[k] = x[k] * a + b;                // one fma
            if (OP == 1) x[k] = b / x[k] + a;                // one division (and an add)
        }
    }
    T s = 0;
    for (int k = 0; k < 8; k++) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class T, int OP>
double run(const char *name, int iters) {
    T *d = nullptr;
    const int blocks = 256 * 8, threads = 256;              // 8 blocks of 4 waves per CU: 8 waves per SIMD
    if (hipMalloc(&d, sizeof(T) * blocks * threads) != hipSuccess) return -1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<T, OP>), dim3(blocks), dim3(threads), 0, 0, d, (T)0.999, (T)0.001, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<T, OP>), dim3(blocks), dim3(threads), 0, 0, d, (T)0.999, (T)0.001, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * threads / 64.0 * iters * 8;      // wave-level operations
    printf("%-28s %8.3f ms  %8.2f G wave-ops/s  = %6.3f per SIMD and ns\n", name, ms, ops / ms * 1e-6, ops / ms * 1e-6 / 1024.0);
    hipFree(d);
    return ms;
}
int main() {
    const int it = 20000;
    run<float, 0>("fp32 fma (not packed)", it);
    run<double, 0>("fp64 fma", it);
    run<float, 1>("fp32 division + add", it / 4);
    run<double, 1>("fp64 division + add", it / 4);
    return 0;
}
