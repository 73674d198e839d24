// Does VALU work overlap with the f32 MFMA on gfx950?  Per iteration: 16 dependent MFMAs + NV
// independent v_fma_f32 per lane.  If the time grows with NV from the start, they share the pipe.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d\n", (int)e_); return; } } while (0)

template <int NV, bool MF>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = threadIdx.x * 0.001f;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.5f + j;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MF) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], a, b);
        }
    }
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += acc[e];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, bool MF>
void run(int bpc) {
    int blocks = 256 * bpc, iters = 1000;
    float* out;
    CK(hipMalloc(&out, blocks * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<NV, MF>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f, 0.5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<NV, MF>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double cyc_per_iter = ms * 1e-3 * 2.4e9 / iters / bpc;
    printf("mfma %d  valu/mfma %2d  waves/SIMD %d : %.3f ms  (%.0f cycles per 16-MFMA iteration per wave-slot)\n", (int)MF, NV, bpc, ms, cyc_per_iter);
    CK(hipFree(out));
}

int main() {
    run<0, true>(1); run<4, true>(1); run<8, true>(1); run<16, true>(1);
    run<8, false>(1); run<16, false>(1);
    run<0, true>(2); run<8, true>(2); run<16, true>(2);
    return 0;
}
