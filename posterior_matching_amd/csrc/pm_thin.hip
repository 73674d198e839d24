// "Thin" convolutions of the PM-VAE: the layers that touch the 1- or 2-channel image.
//   encoder conv_0 (1 -> 32), partial-encoder conv_0 (2 -> 32)       reference networks.py:30-36
//   decoder conv_t_5 (32 -> 1) forward / data-gradient                     networks.py:62-68
// With K = 25..50 or N = 1 these are no matrix problems: padded to MFMA tiles they waste 32x the
// work.  They are HBM / VALU-bound and get plain f32 FMA kernels (coalesced NHWC rows, weights in
// LDS), using the same index rule  src*d = dst*a + tap*cs + off  (d = 1 only) as the GEMM engine.
#include "pm_common.h"

namespace {

struct ThinArgs {
    int B, IH, IW, C, OH, OW, N, KH, KW;
    int a, cs, off;
    int wts, wcs, wns;
    int M;
    int in_act, out_act, aux_act;
    float slope;
};

bool fill_thin(const pm_gather_desc* d, ThinArgs& t) {
    if (!d || d->groups != 1 || d->d != 1 || d->B <= 0 || d->a <= 0) return false;
    if (d->cs != 1 && d->cs != -1) return false;
    if (d->off_x != d->off || d->kws != d->KW) return false;   // no sub-kernel form here
    long long M = (long long)d->B * d->OH * d->OW;
    if (M >= 0x7fffffffLL / 64 || (long long)d->B * d->IH * d->IW * d->C >= 0x7fffffffLL) return false;
    t.B = d->B; t.IH = d->IH; t.IW = d->IW; t.C = d->C; t.OH = d->OH; t.OW = d->OW; t.N = d->N;
    t.KH = d->KH; t.KW = d->KW; t.a = d->a; t.cs = d->cs; t.off = d->off;
    t.wts = d->wts; t.wcs = d->wcs; t.wns = d->wns; t.M = (int)M;
    t.in_act = d->in_act; t.out_act = d->out_act; t.aux_act = d->aux_act; t.slope = d->slope;
    return true;
}

constexpr int THIN_MAXKK = 64;  // taps * C of the thin operand

// out[m][n] = epi( sum_kk in[src(m, tap)][c] * w[kk][n] ),  N in {8,16,24,32}, taps*C <= 64.
// 256 threads = 64 output positions x 4 groups of 8 channels; weights staged in LDS.
__global__ __launch_bounds__(256) void thin_conv_kernel(ThinArgs t, const float* __restrict__ in,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          const float* __restrict__ aux, const float* __restrict__ res,
                                                          float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float Ws[THIN_MAXKK * 32];
    const int KK = t.KH * t.KW * t.C;
    for (int e = threadIdx.x; e < KK * 32; e += 256) {
        int kk = e >> 5, n = e & 31;
        int tap = kk / t.C, c = kk - tap * t.C;
        Ws[e] = n < t.N ? w[tap * t.wts + c * t.wcs + n * t.wns] : 0.f;
    }
    __syncthreads();
    const int ng = threadIdx.x & 3;
    const int m = blockIdx.x * 64 + (threadIdx.x >> 2);
    if (m >= t.M || ng * 8 >= t.N) return;
    const int q = m % t.OW;
    const int r = m / t.OW;
    const int p = r % t.OH;
    const int b = r / t.OH;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const float* ib = in + (size_t)b * t.IH * t.IW * t.C;
    for (int ky = 0; ky < t.KH; ++ky) {
        const int sy = p * t.a + ky * t.cs + t.off;
        if ((unsigned)sy >= (unsigned)t.IH) continue;
        for (int kx = 0; kx < t.KW; ++kx) {
            const int sx = q * t.a + kx * t.cs + t.off;
            if ((unsigned)sx >= (unsigned)t.IW) continue;
            const float* ip = ib + (sy * t.IW + sx) * t.C;
            const float* wp = Ws + ((ky * t.KW + kx) * t.C) * 32 + ng * 8;
            for (int c = 0; c < t.C; ++c) {
                const float v = pm_act(ip[c], t.in_act, t.slope);
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp + c * 32);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(wp + c * 32 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[e] = fmaf(v, w0[e], acc[e]);
                    acc[4 + e] = fmaf(v, w1[e], acc[4 + e]);
                }
            }
        }
    }
    const size_t o = (size_t)m * t.N + ng * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = acc[e] + (bias ? bias[ng * 8 + e] : 0.f);
        out[o + e] = pm_epilogue(v, aux, res, o + e, t.aux_act, t.out_act, t.slope);
    }
}

// out[b,p,q] = act( bias + sum_tap T[b, src(p,q,tap), tap] ): the second half of a wide -> 1-channel
// transposed conv whose per-tap dot products T = in . w[tap] were produced by one GEMM (N = taps).
__global__ __launch_bounds__(256) void tap_shift_add_kernel(ThinArgs t, const float* __restrict__ T, int ldt,
                                                              const float* __restrict__ bias, float* __restrict__ out) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= t.M) return;
    const int q = m % t.OW;
    const int r = m / t.OW;
    const int p = r % t.OH;
    const int b = r / t.OH;
    const float* tb = T + (size_t)b * t.IH * t.IW * ldt;
    float s = bias ? bias[0] : 0.f;
    for (int ky = 0; ky < t.KH; ++ky) {
        const int sy = p * t.a + ky * t.cs + t.off;
        if ((unsigned)sy >= (unsigned)t.IH) continue;
        for (int kx = 0; kx < t.KW; ++kx) {
            const int sx = q * t.a + kx * t.cs + t.off;
            if ((unsigned)sx >= (unsigned)t.IW) continue;
            s += tb[(sy * t.IW + sx) * ldt + ky * t.KW + kx];
        }
    }
    out[m] = pm_act(s, t.out_act, t.slope);
}

}  // namespace

extern "C" int pm_thin_conv(pm_stream_t stream, const pm_gather_desc* d, const float* in, const float* w,
                            const float* bias, const float* aux, const float* res, float* out) {
    ThinArgs t;
    if (!fill_thin(d, t) || !in || !w || !out) return PM_EINVAL;
    if (t.KH * t.KW * t.C > THIN_MAXKK || t.N > 32 || t.N % 8 != 0) return PM_EINVAL;
    hipLaunchKernelGGL(thin_conv_kernel, dim3((t.M + 63) / 64), dim3(256), 0, (hipStream_t)stream, t, in, w, bias, aux,
                       res, out);
    return pm_check_launch("pm_thin_conv");
}

extern "C" int pm_tap_shift_add(pm_stream_t stream, const pm_gather_desc* d, const float* T, int ldt,
                                const float* bias, float* out) {
    ThinArgs t;
    if (!fill_thin(d, t) || !T || !out || d->N != 1 || ldt < d->KH * d->KW) return PM_EINVAL;
    hipLaunchKernelGGL(tap_shift_add_kernel, dim3((t.M + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, T, ldt,
                       bias, out);
    return pm_check_launch("pm_tap_shift_add");
}
