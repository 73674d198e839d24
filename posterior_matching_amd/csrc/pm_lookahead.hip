// Lookahead posteriors for active feature acquisition (reference posterior_matching/models/lookahead.py:14-227).  The heavy
// part of LookaheadPosterior.__call__ is evaluation of the frozen PM-VAE (partial encoder -> samples -> decoder -> partial
// encoder again on B * model_samples * lookahead_subsample masked images), which runs on the engine's existing layers; the
// kernels here are what is specific to the model:
//   pm_lookahead_inputs      the re-encoder's input rows [x_look | b_look] (lookahead.py:154-176)
//   pm_lookahead_ll_fwd/bwd  mean over the model samples of log N(z; loc_f, softplus(raw_f) + 1e-5) for the subsampled features,
//                            masked by `valid`, averaged over the valid ones (lookahead.py:188-203) and its gradient w.r.t. the
//                            LookaheadBlock's Linear output
//   pm_lookahead_info_gains  expected_info_gains (lookahead.py:205-227): current entropy - lookahead entropies, -inf where observed
// All of them are tiny (B = 32, k = 10, 16 of 256 features per step): one workgroup per example, lane = latent dimension.
#include "pm_common.h"

namespace {

constexpr float HALF_LOG_2PI = 0.9189385332046727f;
constexpr float kScaleShift = 1e-5f;

// out[((b*Z + z)*S + s), p, 0..C-1] = imp[b, z, p, c] * bl,  out[.., C] = bl,  bl = max(b[b, p], p == inds[s])
__global__ __launch_bounds__(256) void lookahead_inputs_kernel(const float* __restrict__ imp, const float* __restrict__ b,
                                                                const int* __restrict__ inds, float* __restrict__ out,
                                                                long long total, int Z, int S, int P, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;            // (row, p)
    if (i >= total) return;
    const int p = (int)(i % P);
    const long long row = i / P;
    const int s = (int)(row % S);
    const long long bz = row / S;
    const long long bb = bz / Z;
    const float bl = fmaxf(b[bb * P + p], p == inds[s] ? 1.f : 0.f);
    const float* src = imp + (bz * P + p) * C;
    float* dst = out + i * (C + 1);
    for (int c = 0; c < C; ++c) dst[c] = src[c] * bl;
    dst[C] = bl;
}

// valid[s] of example b: the subsampled feature is not observed yet (max(b + onehot) < 2, lookahead.py:166-173)
__device__ __forceinline__ bool feature_valid(const float* __restrict__ brow, int f) { return brow[f] + 1.f < 2.f; }

// One workgroup per example; wave w takes the subsampled features s = w, w + 4, ...; lane j = latent dimension (k <= 64).
template <bool BWD>
__global__ __launch_bounds__(256) void lookahead_ll_kernel(const float* __restrict__ params, const int* __restrict__ inds,
                                                            const float* __restrict__ zs, const float* __restrict__ b,
                                                            const float* __restrict__ g, float* __restrict__ ll,
                                                            float* __restrict__ dparams, int F, int Z, int S, int k) {
    __shared__ float red[4];
    const int bb = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* brow = b + (size_t)bb * F;
    int nvalid = 0;
    for (int s = 0; s < S; ++s) nvalid += feature_valid(brow, inds[s]) ? 1 : 0;          // denom (every thread: S is small)
    float wsum = 0.f;
    for (int s = wave; s < S; s += 4) {
        const int f = inds[s];
        if (!feature_valid(brow, f)) continue;                                            // wave-uniform
        const float* pr = params + ((size_t)bb * F + f) * 2 * k;
        const float loc = lane < k ? pr[lane] : 0.f;
        const float raw = lane < k ? pr[k + lane] : 0.f;
        const float sc = pm_softplus(raw) + kScaleShift;
        const float inv = 1.f / sc;
        float a1 = 0.f, a2 = 0.f;                                                         // sum_z (z - loc), sum_z (z - loc)^2
        if (lane < k)
            for (int z = 0; z < Z; ++z) {
                const float d = zs[(((size_t)bb * Z + z) * S + s) * k + lane] - loc;
                a1 += d;
                a2 += d * d;
            }
        if (!BWD) {
            float t = lane < k ? -0.5f * a2 * inv * inv / (float)Z - logf(sc) - HALF_LOG_2PI : 0.f;
            wsum += pm_wave_sum(t);
        } else if (lane < k) {
            const float coef = g[bb] / (float)nvalid;
            const float dloc = coef * a1 * inv * inv / (float)Z;
            const float dsc = coef * (a2 * inv * inv * inv / (float)Z - inv);
            float* dp = dparams + ((size_t)bb * F + f) * 2 * k;
            dp[lane] = dloc;
            dp[k + lane] = dsc * pm_sigmoid(raw);                                         // d softplus
        }
    }
    if (BWD) return;
    if (lane == 0) red[wave] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) ll[bb] = nvalid ? (red[0] + red[1] + red[2] + red[3]) / (float)nvalid : 0.f;
}

// gains[f] = b[f] == 0 ? cur_ent - H(N(loc_f, diag scale_f^2)) : -inf ;  H = sum_j log scale_j + k/2 (1 + log 2 pi)
__global__ __launch_bounds__(256) void lookahead_info_gains_kernel(const float* __restrict__ params, const float* __restrict__ cur_ent,
                                                                    const float* __restrict__ b, float* __restrict__ gains, int F,
                                                                    int k) {
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (f >= F) return;
    const float t = lane < k ? logf(pm_softplus(params[(size_t)f * 2 * k + k + lane]) + kScaleShift) : 0.f;
    const float h = pm_wave_sum(t) + 0.5f * (float)k * (1.f + 2.f * HALF_LOG_2PI);
    if (lane == 0) gains[f] = b[f] == 0.f ? cur_ent[0] - h : -INFINITY;
}

// Greedy acquisition (reference posterior_matching/acquisition.py:38-58): logits = where(gains == -inf, -1e10, gains);
// action = distrax.Categorical(logits).mode() = argmax (lowest index among ties); probs = softmax(logits).  One workgroup.
__global__ __launch_bounds__(256) void acquisition_policy_kernel(const float* __restrict__ gains, float* __restrict__ probs,
                                                                  int* __restrict__ action, int F) {
    __shared__ float smax[4], ssum[4];
    __shared__ int sarg[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float m = -INFINITY;
    int am = 0x7fffffff;
    for (int f = threadIdx.x; f < F; f += 256) {
        const float g = gains[f];
        const float l = g == -INFINITY ? -1e10f : g;
        if (l > m) { m = l; am = f; }                    // ascending f per thread: the first maximum stays
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64);
        const int oa = __shfl_xor(am, o, 64);
        if (om > m || (om == m && oa < am)) { m = om; am = oa; }
    }
    if (lane == 0) { smax[wave] = m; sarg[wave] = am; }
    __syncthreads();
    for (int w = 0; w < 4; ++w)
        if (smax[w] > m || (smax[w] == m && sarg[w] < am)) { m = smax[w]; am = sarg[w]; }
    float t = 0.f;
    for (int f = threadIdx.x; f < F; f += 256) {
        const float g = gains[f];
        t += expf((g == -INFINITY ? -1e10f : g) - m);
    }
    t = pm_wave_sum(t);
    if (lane == 0) ssum[wave] = t;
    __syncthreads();
    const float z = ssum[0] + ssum[1] + ssum[2] + ssum[3];
    for (int f = threadIdx.x; f < F; f += 256) {
        const float g = gains[f];
        probs[f] = expf((g == -INFINITY ? -1e10f : g) - m) / z;
    }
    if (threadIdx.x == 0) action[0] = am;
}

// recon[d] = mean_s imp[s, d];  rmse = sqrt(mean_d (x[d] - recon[d])^2 (1 - b[d / C]))   (acquisition.py:13-15, 50-53); one workgroup
__global__ __launch_bounds__(256) void reconstruction_rmse_kernel(const float* __restrict__ imp, const float* __restrict__ x,
                                                                   const float* __restrict__ b, float* __restrict__ recon,
                                                                   float* __restrict__ rmse, int S, int D, int Cr) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        float m = 0.f;
        for (int s = 0; s < S; ++s) m += imp[(size_t)s * D + d];
        m /= (float)S;
        recon[d] = m;
        const float e = x[d] - m;
        acc += e * e * (1.f - b[d / Cr]);
    }
    acc = pm_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) rmse[0] = sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)D);
}

}  // namespace

extern "C" int pm_acquisition_policy(pm_stream_t stream, const float* gains, float* probs, int* action, int F) {
    if (!gains || !probs || !action || F <= 0) return PM_EINVAL;
    PM_KTAG("acquisition_policy_kernel");
    hipLaunchKernelGGL(acquisition_policy_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, gains, probs, action, F);
    return pm_check_launch("pm_acquisition_policy");
}

extern "C" int pm_reconstruction_rmse(pm_stream_t stream, const float* imp, const float* x, const float* b, float* recon,
                                      float* rmse, int S, int D, int C, int Cm) {
    if (!imp || !x || !b || !recon || !rmse || S <= 0 || D <= 0 || C <= 0 || (Cm != C && Cm != 1) || D % C != 0) return PM_EINVAL;
    PM_KTAG("reconstruction_rmse_kernel");
    hipLaunchKernelGGL(reconstruction_rmse_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, imp, x, b, recon, rmse, S, D,
                       Cm == 1 ? C : 1);
    return pm_check_launch("pm_reconstruction_rmse");
}

extern "C" int pm_lookahead_inputs(pm_stream_t stream, const float* imp, const float* b, const int* inds, float* out,
                                   long long B, int Z, int S, int P, int C) {
    if (!imp || !b || !inds || !out || B <= 0 || Z <= 0 || S <= 0 || P <= 0 || C <= 0) return PM_EINVAL;
    const long long total = B * Z * S * P;
    if (total * (C + 1) >= (1LL << 40)) return PM_EINVAL;
    PM_KTAG("lookahead_inputs_kernel");
    hipLaunchKernelGGL(lookahead_inputs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, imp, b,
                       inds, out, total, Z, S, P, C);
    return pm_check_launch("pm_lookahead_inputs");
}

extern "C" int pm_lookahead_ll_fwd(pm_stream_t stream, const float* params, const int* inds, const float* zs, const float* b,
                                   float* ll, int B, int F, int Z, int S, int k) {
    if (!params || !inds || !zs || !b || !ll || B <= 0 || F <= 0 || Z <= 0 || S <= 0 || k <= 0 || k > 64) return PM_EINVAL;
    PM_KTAG("lookahead_ll_kernel<false>");
    hipLaunchKernelGGL(lookahead_ll_kernel<false>, dim3(B), dim3(256), 0, (hipStream_t)stream, params, inds, zs, b,
                       (const float*)nullptr, ll, (float*)nullptr, F, Z, S, k);
    return pm_check_launch("pm_lookahead_ll_fwd");
}

extern "C" int pm_lookahead_ll_bwd(pm_stream_t stream, const float* params, const int* inds, const float* zs, const float* b,
                                   const float* g, float* dparams, int B, int F, int Z, int S, int k) {
    if (!params || !inds || !zs || !b || !g || !dparams || B <= 0 || F <= 0 || Z <= 0 || S <= 0 || k <= 0 || k > 64)
        return PM_EINVAL;
    if (pm_zero_async((hipStream_t)stream, dparams, (size_t)B * F * 2 * k * sizeof(float))) return PM_ELAUNCH;
    PM_KTAG("lookahead_ll_kernel<true>");
    hipLaunchKernelGGL(lookahead_ll_kernel<true>, dim3(B), dim3(256), 0, (hipStream_t)stream, params, inds, zs, b, g,
                       (float*)nullptr, dparams, F, Z, S, k);
    return pm_check_launch("pm_lookahead_ll_bwd");
}

extern "C" int pm_lookahead_info_gains(pm_stream_t stream, const float* params, const float* cur_ent, const float* b, float* gains,
                                       int F, int k) {
    if (!params || !cur_ent || !b || !gains || F <= 0 || k <= 0 || k > 64) return PM_EINVAL;
    PM_KTAG("lookahead_info_gains_kernel");
    hipLaunchKernelGGL(lookahead_info_gains_kernel, dim3((F + 3) / 4), dim3(256), 0, (hipStream_t)stream, params, cur_ent, b, gains,
                       F, k);
    return pm_check_launch("pm_lookahead_info_gains");
}
