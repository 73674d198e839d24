// Shared device helpers for libpmhip (gfx950 / CDNA4 only: 64-lane waves, f32 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/pmhip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PM_WAVE 64

extern thread_local char pm_err_text[256];
int pm_check_launch(const char* what);
// Name of the kernel variant a C-ABI call launched, as rocprofv3 prints it (bench.py groups its live HIP-event times by
// it, so the live table and the rocprof summary have the same rows).  Off unless pm_kernel_names_enable(1).
extern bool pm_ktag_on;
void pm_ktagf(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
#define PM_KTAG(...) do { if (pm_ktag_on) pm_ktagf(__VA_ARGS__); } while (0)
// A dispatch-relevant property of the call that the kernel name does not carry ("masked": a sub-kernel of a masked
// convolution, kws > KW).  tests/test_gpu_zz_coverage.py keys on (name, variant).
void pm_kvarf(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
#define PM_KVAR(...) do { if (pm_ktag_on) pm_kvarf(__VA_ARGS__); } while (0)
// Zero-fill by a plain kernel (ptr and nbytes multiples of 4).  hipMemsetAsync is avoided on purpose:
// captured as a memset node of a HIP graph on ROCm 7.2 it was observed to leave every fourth dword
// of the range non-zero on replay (tests/test_gpu_vqvae.py::test_vqvae_train_steps_match_oracle).
int pm_zero_async(hipStream_t stream, void* ptr, size_t nbytes);

__device__ __forceinline__ float pm_gelu_tanh(float x) {      // jax.nn.gelu(approximate=True)
    return 0.5f * x * (1.f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float pm_gelu_tanh_d(float x) {
    const float t = tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x));
    return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * 0.7978845608028654f * (1.f + 0.134145f * x * x);
}
__device__ __forceinline__ float pm_act(float v, int act, float slope) {
    if (act == PM_ACT_LEAKY) return v >= 0.f ? v : slope * v;  // jax.nn.leaky_relu: where(x >= 0, x, a*x)
    if (act == PM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == PM_ACT_GELU) return pm_gelu_tanh(v);
    return v;
}
// derivative expressed on the activation's OUTPUT (sign is preserved by both activations) or on
// its input: leaky' = 1 for v >= 0 else slope; relu' = 1 for v > 0 else 0 (jax's relu jvp).
__device__ __forceinline__ float pm_dact(float v, int act, float slope) {
    if (act == PM_ACT_LEAKY) return v >= 0.f ? 1.f : slope;
    if (act == PM_ACT_RELU) return v > 0.f ? 1.f : 0.f;
    if (act == PM_ACT_GELU) return pm_gelu_tanh_d(v);   // on the activation's INPUT only (gelu is not monotone)
    return 1.f;
}

// The shared epilogue of every gather-GEMM / thin kernel:
//   out = act( v * act'(aux) + res )            (default)
//   out = act( (v + res) * act'(aux) )          (aux_act | PM_AUX_AFTER_RES)
// The second form is the data gradient into a tensor that was stored AFTER its activation and
// also feeds a residual path (relu(enc_3) entering ConvResidualStack, reference vqvae.py:215-217).
__device__ __forceinline__ float pm_epilogue(float v, const float* __restrict__ aux, const float* __restrict__ res,
                                             size_t o, int aux_act, int out_act, float slope) {
    const bool after = (aux_act & PM_AUX_AFTER_RES) != 0;
    if (res && after) v += res[o];
    if (aux) v *= pm_dact(aux[o], aux_act & (PM_AUX_AFTER_RES - 1), slope);
    if (res && !after) v += res[o];
    return pm_act(v, out_act, slope);
}

// The same epilogue for one 32-column accumulator tile of a GEMM-class kernel: 16 elements per lane at output offsets
// ro[e] + n (ro[e] < 0: no such row).  Phase 1 issues every aux / res load (16 + 16 in flight, clamped addresses for the
// missing rows), phase 2 does the arithmetic, phase 3 only stores.  Written element by element, every load waits for the
// stores in front of it (vmcnt counts loads and stores alike, and the compiler sinks the arithmetic into the conditional
// store blocks): 16 memory round trips at the end of every data-gradient kernel.
typedef float pm_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void pm_epilogue_tile(const pm_f32x16& acc, const int (&ro)[16], int n, float bv,
                                                 const float* __restrict__ aux, const float* __restrict__ res,
                                                 float* __restrict__ out, float* __restrict__ out2, int act2, int aux_act,
                                                 int out_act, float slope) {
    float av[16], rv[16], v[16];
    if (aux) {
#pragma unroll
        for (int e = 0; e < 16; ++e) av[e] = aux[(size_t)(ro[e] >= 0 ? ro[e] : 0) + n];
    }
    if (res) {
#pragma unroll
        for (int e = 0; e < 16; ++e) rv[e] = res[(size_t)(ro[e] >= 0 ? ro[e] : 0) + n];
    }
    const bool after = (aux_act & PM_AUX_AFTER_RES) != 0;
    const int dact = aux_act & (PM_AUX_AFTER_RES - 1);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        float x = acc[e] + bv;
        if (res && after) x += rv[e];
        if (aux) x *= pm_dact(av[e], dact, slope);
        if (res && !after) x += rv[e];
        x = pm_act(x, out_act, slope);
        asm volatile("" : "+v"(x));          // materialised here, not inside the conditional store blocks below
        v[e] = x;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        if (ro[e] < 0) continue;
        const size_t o = (size_t)ro[e] + n;
        out[o] = v[e];
        if (out2) out2[o] = pm_act(v[e], act2, slope);
    }
}

// The same epilogue for a TRANSPOSED accumulator tile: the kernel issued its MFMAs with the operands swapped (weights as A,
// activations as B), so this lane holds ONE output row (position) and its 16 registers are the columns
// c(e) = (e & 3) + 8 * (e >> 2) + 4 * h of the 32-column tile starting at n0 - four runs of 4 consecutive columns.  Every
// aux / res load and every store is then a 16-byte access (4 + 4 loads, 4 stores per lane instead of 16 + 16 + 16 dwords):
// the store tail of a GEMM-class kernel is store-ISSUE bound (MI355X_MICROARCH.md, cycle constants, "epilogue store tail").
// row_off: offset of the row's first column in out / aux / res (< 0: no such row); N % 4 == 0, 16-byte aligned tensors.
typedef float pm_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pm_epilogue_tile_t(const pm_f32x16& acc, long long row_off, int n0, int h, int N,
                                                   const float* __restrict__ bias, const float* __restrict__ aux,
                                                   const float* __restrict__ res, float* __restrict__ out,
                                                   float* __restrict__ out2, int act2, int aux_act, int out_act, float slope) {
    pm_f32x4 av[4], rv[4], bv[4];
    const bool row_ok = row_off >= 0;
    const long long ro = row_ok ? row_off : 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = n0 + 8 * g + 4 * h;
        const int cc = c < N ? c : 0;                        // N % 4 == 0: a run of 4 columns is all there or not at all
        bv[g] = bias ? *reinterpret_cast<const pm_f32x4*>(bias + cc) : pm_f32x4{0.f, 0.f, 0.f, 0.f};
        if (aux) av[g] = *reinterpret_cast<const pm_f32x4*>(aux + ro + cc);
        if (res) rv[g] = *reinterpret_cast<const pm_f32x4*>(res + ro + cc);
    }
    const bool after = (aux_act & PM_AUX_AFTER_RES) != 0;
    const int dact = aux_act & (PM_AUX_AFTER_RES - 1);
    pm_f32x4 v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float x = acc[4 * g + q] + bv[g][q];
            if (res && after) x += rv[g][q];
            if (aux) x *= pm_dact(av[g][q], dact, slope);
            if (res && !after) x += rv[g][q];
            x = pm_act(x, out_act, slope);
            asm volatile("" : "+v"(x));
            v[g][q] = x;
        }
    if (!row_ok) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = n0 + 8 * g + 4 * h;
        if (c >= N) continue;
        *reinterpret_cast<pm_f32x4*>(out + ro + c) = v[g];
        if (out2) {
            pm_f32x4 w;
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = pm_act(v[g][q], act2, slope);
            *reinterpret_cast<pm_f32x4*>(out2 + ro + c) = w;
        }
    }
}

__device__ __forceinline__ float pm_softplus(float x) {  // logaddexp(x, 0)
    return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float pm_sigmoid(float x) {
    return x >= 0.f ? 1.f / (1.f + expf(-x)) : expf(x) / (1.f + expf(x));
}

__device__ __forceinline__ float pm_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float pm_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Philox4x32-10 (Random123), counter (c0, c1, c2, c3), key = the two words of `seed`: the generator of pm_normal_fill /
// pm_dropout_mask / pm_gumbel_fill (pm_optim.hip), here for kernels that draw their own numbers in place.
__device__ __forceinline__ void pm_philox4x32_10(unsigned (&c)[4], unsigned long long seed) {
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (unsigned)p1;
        c[3] = (unsigned)p0;
        c[0] = n0;
        c[2] = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
