// Evaluation paths of the PM-VAE: imputation and importance-sampled likelihoods.
//   PosteriorMatchingVAE.impute       reference vae.py:146-169
//   PosteriorMatchingVAE.is_log_prob  reference vae.py:171-226
//   _AutoregressiveDistribution._sample_n  reference distributions.py:168-190 (one latent dimension per network pass)
// S samples per example live on the row axis, sample-minor: row b*S + s.  Everything here is HBM-bound row work.
#include "pm_common.h"

namespace {

constexpr float kLog2Pi = 1.8378770664093453f;

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = pm_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// dst[(b*S + s), :] = src[b, :]
__global__ __launch_bounds__(256) void repeat_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            long long total, long long n, int S) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        const long long row = e / n, j = e - row * n;
        dst[e] = src[(row / S) * n + j];
    }
}

__global__ __launch_bounds__(256) void sigmoid_kernel(const float* __restrict__ in, float* __restrict__ out, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) out[e] = pm_sigmoid(in[e]);
}

// ll[b*S + s] = sum_j w[b, j] * log Bernoulli(x[b, j]; logits[b*S + s, j]);  w NULL: 1;  Dw in {D, 1 per pixel group}:
// w index = j / (D / Dw)  (mask with one channel per pixel against C-channel data)
__global__ __launch_bounds__(256) void bernoulli_ll_rep_kernel(const float* __restrict__ logits, const float* __restrict__ x,
                                                                 const float* __restrict__ w, float* __restrict__ ll,
                                                                 int S, int D, int Dw) {
    __shared__ float red[4];
    const int row = blockIdx.x, b = row / S;
    const size_t lb = (size_t)row * D, xb = (size_t)b * D, wb = (size_t)b * Dw;
    const int grp = D / Dw;
    float s = 0.f;
    for (int j = threadIdx.x; j < D; j += 256) {
        const float l = logits[lb + j], t = x[xb + j];
        const float v = t * (-pm_softplus(-l)) + (1.f - t) * (-pm_softplus(l));
        s += w ? w[wb + j / grp] * v : v;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) ll[row] = s;
}

__global__ __launch_bounds__(256) void normal_ll_rep_kernel(const float* __restrict__ loc, const float* __restrict__ x,
                                                              const float* __restrict__ log_scale,
                                                              const float* __restrict__ w, float* __restrict__ ll, int S,
                                                              int D, int Dw, float scale_eps) {
    __shared__ float red[4];
    const int row = blockIdx.x, b = row / S;
    const size_t lb = (size_t)row * D, xb = (size_t)b * D, wb = (size_t)b * Dw;
    const int grp = D / Dw;
    const float sigma = expf(log_scale[0]) + scale_eps;
    const float inv = 1.f / sigma, ls = logf(sigma);
    float s = 0.f;
    for (int j = threadIdx.x; j < D; j += 256) {
        const float u = (x[xb + j] - loc[lb + j]) * inv;
        const float v = -0.5f * u * u - ls - 0.5f * kLog2Pi;
        s += w ? w[wb + j / grp] * v : v;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) ll[row] = s;
}

// lp[r] = log N(z[r]; 0, I)
__global__ __launch_bounds__(256) void std_normal_logprob_kernel(const float* __restrict__ z, float* __restrict__ lp,
                                                                   long long R, int k) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    float s = 0.f;
    for (int j = 0; j < k; ++j) {
        const float v = z[r * k + j];
        s += v * v;
    }
    lp[r] = -0.5f * s - 0.5f * kLog2Pi * (float)k;
}

// out[b] = log mean_s exp(a[b,s] + bb[b,s] - c[b,s])   (tfp.math.reduce_logmeanexp over the sample axis)
__global__ __launch_bounds__(64) void logmeanexp3_kernel(const float* __restrict__ a, const float* __restrict__ bb,
                                                           const float* __restrict__ c, float* __restrict__ out, int S,
                                                           long long sb, long long ss) {
    const int b = blockIdx.x;
    const size_t base = (size_t)b * sb;
    float m = -INFINITY;
    for (int s = threadIdx.x; s < S; s += 64) {
        float v = a[base + s * ss];
        if (bb) v += bb[base + s * ss];
        if (c) v -= c[base + s * ss];
        m = fmaxf(m, v);
    }
    m = pm_wave_max(m);
    float acc = 0.f;
    for (int s = threadIdx.x; s < S; s += 64) {
        float v = a[base + s * ss];
        if (bb) v += bb[base + s * ss];
        if (c) v -= c[base + s * ss];
        acc += (m == -INFINITY) ? 0.f : expf(v - m);
    }
    acc = pm_wave_sum(acc);
    if (threadIdx.x == 0) out[b] = (m == -INFINITY) ? -INFINITY : m + logf(acc) - logf((float)S);
}

// One step of the autoregressive sampler: head rows [i*R, (i+1)*R) hold (logits | means | raw scales) of latent
// dimension i given z[:, :i];  z[r, i] = mean[c] + (softplus(raw[c]) + 1e-5) * eps[r, i],  c = argmax(logits + gumbel[r, i, :])
__global__ __launch_bounds__(256) void gmm_sample_step_kernel(const float* __restrict__ head, const float* __restrict__ gumbel,
                                                                const float* __restrict__ eps, float* __restrict__ z,
                                                                long long R, int k, int nc, int i) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float* h = head + ((size_t)i * R + r) * 3 * nc;
    const float* g = gumbel + ((size_t)r * k + i) * nc;
    int best = 0;
    float bv = h[0] + g[0];
    for (int c = 1; c < nc; ++c) {
        const float v = h[c] + g[c];
        if (v > bv) {
            bv = v;
            best = c;
        }
    }
    z[r * k + i] = h[nc + best] + (pm_softplus(h[2 * nc + best]) + 1e-5f) * eps[r * k + i];
}

// out[r / P] += sign * sum_j log N(z[r, j]; loc[r, j], softplus(raw[r, j]) + 1e-5): params rows of stride ld = (loc | raw | ...)
__global__ __launch_bounds__(256) void diag_logprob_acc_kernel(const float* __restrict__ params, int ld,
                                                                const float* __restrict__ z, float* __restrict__ out,
                                                                long long R, int Z, int P, float sign) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float t = 0.f;
    long long r = 0;
    if (i < R * Z) {
        r = i / Z;
        const int j = (int)(i - r * Z);
        const float loc = params[r * ld + j], sc = pm_softplus(params[r * ld + Z + j]) + 1e-5f;
        const float u = (z[i] - loc) / sc;
        t = sign * (-0.5f * u * u - logf(sc) - 0.5f * kLog2Pi);
    }
    if (Z == 16) {       // the 16 lanes of a row reduce by shuffle, then one atomic per row
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (i < R * Z && (threadIdx.x & 15) == 0) atomicAdd(out + r / P, t);
    } else if (i < R * Z) {
        atomicAdd(out + r / P, t);
    }
}

// out[b] (+)= sign * sum_p v[b*P + p] * (w ? w[b*P + p] : 1)
__global__ __launch_bounds__(256) void segment_wsum_kernel(const float* __restrict__ v, const float* __restrict__ w,
                                                            float* __restrict__ out, int P, float sign, int accumulate) {
    __shared__ float red[4];
    const size_t base = (size_t)blockIdx.x * P;
    float s = 0.f;
    for (int p = threadIdx.x; p < P; p += 256) s += w ? v[base + p] * w[base + p] : v[base + p];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = (accumulate ? out[blockIdx.x] : 0.f) + sign * s;
}

// ---- PosteriorMatchingVAE.expected_info_gains (reference vae.py:228-290) ------------------------------------------------
// The candidate batch of one decoder sample (vae.py:257-276): row c of [F + 1] rows is the instance with mask
//   m_0 = b,   m_c = max(b, onehot_{c-1})   (feature index = position * Cb + mask channel)
// and values where(b == 1, x, x_u) * m_c (x_o = x * b equals x there), concatenated with m_c: out [F + 1, P, C + Cb].
__global__ __launch_bounds__(256) void info_gain_inputs_kernel(const float* __restrict__ x, const float* __restrict__ b,
                                                                const float* __restrict__ x_u, float* __restrict__ out,
                                                                int P, int C, int Cb) {
    const int c = blockIdx.y;                       // candidate row: 0 = the current mask, c >= 1 = feature c - 1 acquired
    const int W = C + Cb;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < P * W; e += gridDim.x * 256) {
        const int pos = e / W, j = e - pos * W;
        const int cb = j < C ? (Cb == C ? j : 0) : j - C;
        const float bv = b[pos * Cb + cb];
        const float m = (c >= 1 && c - 1 == pos * Cb + cb) ? fmaxf(bv, 1.f) : bv;
        float v = m;
        if (j < C) v = (bv == 1.f ? x[pos * C + j] : x_u[pos * C + j]) * m;   // where(b == 1, x * b, x_u) * m
        out[(size_t)c * P * W + e] = v;
    }
}

// entropy of a Gaussian head per row: 0.5 k (1 + log 2 pi) + sum_i log scale_i; scale_i = softplus(raw_i) + 1e-5 is the
// diagonal of FillScaleTriL (tril != 0: parameters [loc k | fill_triangular k(k+1)/2]) or of the diagonal head [loc | raw]
__global__ __launch_bounds__(256) void gaussian_entropy_kernel(const float* __restrict__ params, float* __restrict__ ent,
                                                                 long long R, int k, int tril, int ent_stride) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const int np = tril ? k + k * (k + 1) / 2 : 2 * k;
    const float* prow = params + (size_t)r * np;
    float s = 0.f;
    for (int i = 0; i < k; ++i) {
        int idx;
        if (tril) {
            const int m = k * (k + 1) / 2, t = i * k + i;       // TFP fill_triangular index of element (i, i)
            idx = k + (t < m - k ? k + t : 2 * m - k - 1 - t);
        } else {
            idx = k + i;
        }
        s += logf(pm_softplus(prow[idx]) + 1e-5f);
    }
    ent[(size_t)r * ent_stride] = 0.5f * k * (1.f + kLog2Pi) + s;
}

// gains[f] = mean_s ents[s, 0] - mean_s ents[s, 1 + f] where b[f] == 0, else -inf   (vae.py:278-288)
__global__ __launch_bounds__(256) void info_gain_finish_kernel(const float* __restrict__ ents, const float* __restrict__ b,
                                                                 float* __restrict__ gains, int S, int F) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    float before = 0.f, after = 0.f;
    for (int s = 0; s < S; ++s) {
        before += ents[(size_t)s * (F + 1)];
        after += ents[(size_t)s * (F + 1) + 1 + f];
    }
    gains[f] = b[f] == 0.f ? before / S - after / S : -INFINITY;
}

}  // namespace

extern "C" int pm_diag_logprob_acc(pm_stream_t stream, const float* params, int ld, const float* z, float* out,
                                   long long rows, int Z, int P, float sign) {
    if (!params || !z || !out || rows <= 0 || Z <= 0 || P <= 0 || ld < 2 * Z) return PM_EINVAL;
    hipLaunchKernelGGL(diag_logprob_acc_kernel, dim3((unsigned)((rows * Z + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       params, ld, z, out, rows, Z, P, sign);
    return pm_check_launch("pm_diag_logprob_acc");
}

extern "C" int pm_segment_wsum(pm_stream_t stream, const float* v, const float* w, float* out, int B, int P, float sign,
                               int accumulate) {
    if (!v || !out || B <= 0 || P <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(segment_wsum_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, v, w, out, P, sign, accumulate);
    return pm_check_launch("pm_segment_wsum");
}

extern "C" int pm_repeat_rows(pm_stream_t stream, const float* src, float* dst, long long B, int S, long long n) {
    if (!src || !dst || B <= 0 || S <= 0 || n <= 0) return PM_EINVAL;
    const long long total = B * S * n;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(repeat_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, total, n, S);
    return pm_check_launch("pm_repeat_rows");
}

extern "C" int pm_sigmoid(pm_stream_t stream, const float* in, float* out, long long n) {
    if (!in || !out || n <= 0) return PM_EINVAL;
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sigmoid_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, n);
    return pm_check_launch("pm_sigmoid");
}

extern "C" int pm_bernoulli_ll_rep_fwd(pm_stream_t stream, const float* logits, const float* x, const float* w, float* ll,
                                       int B, int S, int D, int Dw) {
    if (!logits || !x || !ll || B <= 0 || S <= 0 || D <= 0 || Dw <= 0 || D % Dw != 0) return PM_EINVAL;
    hipLaunchKernelGGL(bernoulli_ll_rep_kernel, dim3(B * S), dim3(256), 0, (hipStream_t)stream, logits, x, w, ll, S, D, Dw);
    return pm_check_launch("pm_bernoulli_ll_rep_fwd");
}

extern "C" int pm_normal_ll_rep_fwd(pm_stream_t stream, const float* loc, const float* x, const float* log_scale,
                                    const float* w, float* ll, int B, int S, int D, int Dw, float scale_eps) {
    if (!loc || !x || !log_scale || !ll || B <= 0 || S <= 0 || D <= 0 || Dw <= 0 || D % Dw != 0) return PM_EINVAL;
    hipLaunchKernelGGL(normal_ll_rep_kernel, dim3(B * S), dim3(256), 0, (hipStream_t)stream, loc, x, log_scale, w, ll, S,
                       D, Dw, scale_eps);
    return pm_check_launch("pm_normal_ll_rep_fwd");
}

extern "C" int pm_std_normal_logprob(pm_stream_t stream, const float* z, float* lp, long long R, int k) {
    if (!z || !lp || R <= 0 || k <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(std_normal_logprob_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z,
                       lp, R, k);
    return pm_check_launch("pm_std_normal_logprob");
}

extern "C" int pm_logmeanexp3(pm_stream_t stream, const float* a, const float* b, const float* c, float* out, int B,
                              int S, int sample_major) {
    if (!a || !out || B <= 0 || S <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(logmeanexp3_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, a, b, c, out, S,
                       sample_major ? 1LL : (long long)S, sample_major ? (long long)B : 1LL);
    return pm_check_launch("pm_logmeanexp3");
}

extern "C" int pm_gmm_sample_step(pm_stream_t stream, const float* head, const float* gumbel, const float* eps, float* z,
                                  long long R, int k, int nc, int i) {
    if (!head || !gumbel || !eps || !z || R <= 0 || k <= 0 || nc <= 0 || i < 0 || i >= k) return PM_EINVAL;
    hipLaunchKernelGGL(gmm_sample_step_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, (hipStream_t)stream, head,
                       gumbel, eps, z, R, k, nc, i);
    return pm_check_launch("pm_gmm_sample_step");
}

extern "C" int pm_info_gain_inputs(pm_stream_t stream, const float* x, const float* b, const float* x_u, float* out, int P,
                                   int C, int Cb) {
    if (!x || !b || !x_u || !out || P <= 0 || C <= 0 || (Cb != C && Cb != 1)) return PM_EINVAL;
    const int F = P * Cb;
    int gx = (P * (C + Cb) + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(info_gain_inputs_kernel, dim3(gx, F + 1), dim3(256), 0, (hipStream_t)stream, x, b, x_u, out, P, C, Cb);
    return pm_check_launch("pm_info_gain_inputs");
}

extern "C" int pm_gaussian_entropy(pm_stream_t stream, const float* params, float* ent, long long R, int k, int tril,
                                   int ent_stride) {
    if (!params || !ent || R <= 0 || k <= 0 || ent_stride <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(gaussian_entropy_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params,
                       ent, R, k, tril, ent_stride);
    return pm_check_launch("pm_gaussian_entropy");
}

extern "C" int pm_info_gain_finish(pm_stream_t stream, const float* ents, const float* b, float* gains, int S, int F) {
    if (!ents || !b || !gains || S <= 0 || F <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(info_gain_finish_kernel, dim3((F + 255) / 256), dim3(256), 0, (hipStream_t)stream, ents, b, gains, S, F);
    return pm_check_launch("pm_info_gain_finish");
}
