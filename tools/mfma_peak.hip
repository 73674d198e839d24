// Micro-benchmark: sustained rate of v_mfma_f32_32x32x2_f32 with 1/2/4 accumulators per wave and
// 1..4 waves per SIMD (blocks of 256 threads = 1 wave per SIMD per block).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n)
        for (int e = 0; e < 16; ++e) acc[n][e] = threadIdx.x * 0.001f + n;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n)
        for (int e = 0; e < 16; ++e) s += acc[n][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu) {
    int blocks = 256 * blocks_per_cu, iters = 2000;
    float* out;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 16 * NACC * 4096.0;
    printf("acc/wave %d  waves/SIMD %d : %.1f TFLOP/s  (%.3f ms)\n", NACC, blocks_per_cu, flops / ms / 1e9, ms);
    hipFree(out);
}

int main() {
    run<1>(1); run<1>(2); run<1>(4);
    run<2>(1); run<2>(2);
    run<4>(1); run<4>(2);
    return 0;
}
