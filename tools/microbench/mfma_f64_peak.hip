// FP64 matrix-core peak of the device: every wave issues independent chains of
// v_mfma_f64_16x16x4_f64 from registers (no memory traffic).  Prints TFLOP/s.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak mfma_f64_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ __launch_bounds__(256) void mfma_loop(double *out, int iters, long long *cyc)
{
    const long long t0 = clock64();
    d4 acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) acc[c] = (d4){0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = clock64() - t0;   // shader clocks of wave 0
}

template <int CHAINS> static void run(int waves_per_simd)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int blocks = cus * waves_per_simd;      // 256 threads = 4 waves = one per SIMD
    const int iters = 20000;
    double *out;
    long long *cyc, hcyc = 0;
    (void)hipMalloc(&out, (size_t)blocks * 256 * sizeof(double));
    (void)hipMalloc(&cyc, sizeof(long long));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, 100, cyc);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&hcyc, cyc, sizeof(hcyc), hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 * iters * CHAINS * 2.0 * 16 * 16 * 4;
    printf("CUs %d nominal %.0f MHz, waves/SIMD %d, chains %d: %.3f ms, %.2f TFLOP/s; wave 0: %.1f clocks per MFMA, "
           "counter rate %.0f MHz\n", cus, p.clockRate / 1e3, waves_per_simd, CHAINS, ms, flops / (ms * 1e-3) / 1e12,
           (double)hcyc / ((double)iters * CHAINS), (double)hcyc / (ms * 1e-3) / 1e6);
    (void)hipFree(out);
}

int main()
{
    run<1>(1); run<2>(1); run<4>(1); run<1>(2); run<4>(2); run<4>(4);
    return 0;
}
