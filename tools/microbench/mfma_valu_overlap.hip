// Can one wave overlap v_mfma_f64_16x16x4_f64 with independent f64 VALU work?  Three loops per
// launch configuration (512 threads = 2 waves per SIMD, one workgroup per CU):
//   mfma only, valu only, both in one basic block (compiler order, and interleaved with
//   sched_group_barrier: 1 MFMA then VPER VALU instructions).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NM = 16;     // MFMAs per iteration (one dependent chain)
constexpr int NV = 8;      // independent VALU chains
constexpr int LV = 24;     // fma per chain per iteration  -> NV * LV = 192 v_fma_f64

template <int MODE>
__global__ __launch_bounds__(512) void loop(double *out, int iters)
{
    d4 acc = {0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    double v[NV];
    for (int k = 0; k < NV; k++) v[k] = 0.5 + k * 1e-3 + threadIdx.x * 1e-6;
    const double c = 0.999999, d = 1e-7;
    // MODE 4: waves 0-3 of the workgroup only issue MFMAs, waves 4-7 only VALU (each SIMD hosts one of each)
    const bool do_m = MODE == 4 ? (threadIdx.x < 256) : (MODE == 0 || MODE >= 2);
    const bool do_v = MODE == 4 ? (threadIdx.x >= 256) : (MODE == 1 || MODE >= 2);
    for (int i = 0; i < iters; i++) {
        if (do_m) {
#pragma unroll
            for (int m = 0; m < NM; m++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int l = 0; l < LV; l++)
#pragma unroll
                for (int k = 0; k < NV; k++) v[k] = v[k] * c + d;
        }
        if (MODE == 3) {
            // 1 MFMA, then 12 VALU, sixteen times
#pragma unroll
            for (int m = 0; m < NM; m++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);   // VALU
            }
        }
        if (MODE == 1) __builtin_amdgcn_sched_barrier(0);
    }
    double s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int k = 0; k < NV; k++) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> static double run(const char *name)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount, iters = 20000;
    double *out;
    (void)hipMalloc(&out, (size_t)blocks * 512 * sizeof(double));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(loop<MODE>, dim3(blocks), dim3(512), 0, 0, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(loop<MODE>, dim3(blocks), dim3(512), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %.3f ms  (%.1f ns per iteration)\n", name, ms, ms * 1e6 / iters);
    (void)hipFree(out);
    return ms;
}

int main()
{
    double m = run<0>("mfma only (16 per iter)");
    double v = run<1>("valu only (192 fma per iter)");
    double b = run<2>("both, compiler order");
    double s = run<3>("both, 1 MFMA : 12 VALU");
    run<4>("split: 4 mfma + 4 valu waves");
    printf("sum %.3f ms, max %.3f ms, both %.3f ms, interleaved %.3f ms\n", m + v, m > v ? m : v, b, s);
    return 0;
}
