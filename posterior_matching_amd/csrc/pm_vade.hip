// VaDE mixture prior (reference posterior_matching/models/vade.py:40-57,96-150): pi = Categorical(logits), components
// MultivariateNormalDiag(mu_c, exp(log_scale_c)).  With s_c = log p(z | c) + log pi_c and gamma = softmax(s) the five terms of
// VADE.elbo collapse:  sum_c gamma_c (s_c - log gamma_c) = logsumexp_c s_c = log p(z)  (oracle/vade_oracle.py checks the
// identity), so the ELBO is rec_ll + log p(z) - log q(z | x) and the kernels here are the mixture log-density, its gradient
// and the cluster probabilities.  One wave per latent row, lane c = component c (C <= 64), a loop over the k latent
// dimensions; tiny problems (B = 128, k = C = 10 at configs/vade_mnist.py) - latency, not throughput.
#include "pm_common.h"

namespace {

constexpr float HALF_LOG_2PI = 0.9189385332046727f;

// log pi_c = logits_c - logsumexp(logits)   (distrax.Categorical(logits).logits is normalised)
__device__ __forceinline__ float log_pi_lane(const float* __restrict__ logits, int c, int C) {
    const float l = c < C ? logits[c] : -INFINITY;
    const float m = pm_wave_max(l);
    const float s = pm_wave_sum(c < C ? expf(l - m) : 0.f);
    return l - (m + logf(s));
}

// s_c for this lane's component; lanes c >= C return -inf
__device__ __forceinline__ float comp_score(const float* __restrict__ zrow, const float* __restrict__ mu,
                                            const float* __restrict__ ls, float lpi, int c, int k, int C) {
    if (c >= C) return -INFINITY;
    float acc = 0.f;
    for (int j = 0; j < k; ++j) {
        const float l = ls[c * k + j];
        const float d = (zrow[j] - mu[c * k + j]) * expf(-l);
        acc += -0.5f * d * d - l - HALF_LOG_2PI;
    }
    return acc + lpi;
}

__global__ __launch_bounds__(256) void vade_prior_fwd_kernel(const float* __restrict__ z, const float* __restrict__ mu,
                                                              const float* __restrict__ ls, const float* __restrict__ logits,
                                                              float* __restrict__ lp, long long rows, int k, int C) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int c = threadIdx.x & 63;
    if (r >= rows) return;
    const float s = comp_score(z + r * k, mu, ls, log_pi_lane(logits, c, C), c, k, C);
    const float m = pm_wave_max(s);
    const float t = pm_wave_sum(c < C ? expf(s - m) : 0.f);
    if (c == 0) lp[r] = m + logf(t);
}

// d (sum_r g_r log p(z_r)): dz [rows, k] written; dmu / dlog_scale / dlogits accumulated (atomics: rows adders per address)
__global__ __launch_bounds__(256) void vade_prior_bwd_kernel(const float* __restrict__ z, const float* __restrict__ mu,
                                                              const float* __restrict__ ls, const float* __restrict__ logits,
                                                              const float* __restrict__ g, float* __restrict__ dz,
                                                              float* __restrict__ dmu, float* __restrict__ dls,
                                                              float* __restrict__ dlogits, long long rows, int k, int C) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int c = threadIdx.x & 63;
    if (r >= rows) return;
    const float lpi = log_pi_lane(logits, c, C);
    const float* zrow = z + r * k;
    const float s = comp_score(zrow, mu, ls, lpi, c, k, C);
    const float m = pm_wave_max(s);
    const float e = c < C ? expf(s - m) : 0.f;
    const float gamma = e / pm_wave_sum(e);
    const float gr = g[r];
    for (int j = 0; j < k; ++j) {
        float t = 0.f, dm = 0.f;
        if (c < C) {
            const float l = ls[c * k + j];
            dm = zrow[j] - mu[c * k + j];
            t = dm * expf(-2.f * l);                       // (z - mu) / sigma^2
            if (dmu) atomicAdd(dmu + c * k + j, gr * gamma * t);
            if (dls) atomicAdd(dls + c * k + j, gr * gamma * (t * dm - 1.f));
        }
        const float dzj = pm_wave_sum(-gamma * t);
        if (c == 0 && dz) dz[r * k + j] = gr * dzj;
    }
    if (c < C && dlogits) atomicAdd(dlogits + c, gr * (gamma - expf(lpi)));
}

// q(c | x)_b = mean over the S samples z[b*S + s] of softmax_c(s_c)   (VADE.predict_cluster, vade.py:96-115)
__global__ __launch_bounds__(256) void vade_cluster_probs_kernel(const float* __restrict__ z, const float* __restrict__ mu,
                                                                  const float* __restrict__ ls, const float* __restrict__ logits,
                                                                  float* __restrict__ probs, long long B, int S, int k, int C) {
    const long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int c = threadIdx.x & 63;
    if (b >= B) return;
    const float lpi = log_pi_lane(logits, c, C);
    float acc = 0.f;
    for (int s = 0; s < S; ++s) {
        const float sc = comp_score(z + (b * S + s) * k, mu, ls, lpi, c, k, C);
        const float m = pm_wave_max(sc);
        const float e = c < C ? expf(sc - m) : 0.f;
        acc += e / pm_wave_sum(e);
    }
    if (c < C) probs[b * C + c] = acc / (float)S;
}

}  // namespace

extern "C" int pm_vade_prior_fwd(pm_stream_t stream, const float* z, const float* mu, const float* log_scale,
                                 const float* logits, float* lp, long long rows, int k, int C) {
    if (!z || !mu || !log_scale || !logits || !lp || rows <= 0 || k <= 0 || C <= 0 || C > 64) return PM_EINVAL;
    hipLaunchKernelGGL(vade_prior_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, z, mu,
                       log_scale, logits, lp, rows, k, C);
    return pm_check_launch("pm_vade_prior_fwd");
}

extern "C" int pm_vade_prior_bwd(pm_stream_t stream, const float* z, const float* mu, const float* log_scale,
                                 const float* logits, const float* g, float* dz, float* dmu, float* dlog_scale,
                                 float* dlogits, long long rows, int k, int C) {
    if (!z || !mu || !log_scale || !logits || !g || rows <= 0 || k <= 0 || C <= 0 || C > 64) return PM_EINVAL;
    hipLaunchKernelGGL(vade_prior_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, z, mu,
                       log_scale, logits, g, dz, dmu, dlog_scale, dlogits, rows, k, C);
    return pm_check_launch("pm_vade_prior_bwd");
}

extern "C" int pm_vade_cluster_probs(pm_stream_t stream, const float* z, const float* mu, const float* log_scale,
                                     const float* logits, float* probs, long long B, int S, int k, int C) {
    if (!z || !mu || !log_scale || !logits || !probs || B <= 0 || S <= 0 || k <= 0 || C <= 0 || C > 64) return PM_EINVAL;
    hipLaunchKernelGGL(vade_cluster_probs_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, z, mu,
                       log_scale, logits, probs, B, S, k, C);
    return pm_check_launch("pm_vade_cluster_probs");
}
