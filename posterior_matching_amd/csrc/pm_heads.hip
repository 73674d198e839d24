// Distribution heads of the PM-VAE step: HBM / latency-bound wave-level kernels.
//   TriLGaussian sample + KL, MVN-TriL log-prob       (reference distributions.py:101-113, vae.py:124,130,138)
//   Bernoulli / Normal log-likelihood with row sums    (distributions.py:20-25,41-55, vae.py:127-128)
//   AR-GMM input builder, GMM log-prob                 (distributions.py:124-134,152-166)
//   mask-concat                                        (vae.py:132-133)
// Reductions use 64-lane wave shuffles; one wave owns one example wherever a row fits a wave.
#include "pm_common.h"

namespace {

constexpr float kLog2Pi = 1.8378770664093453f;
constexpr float kDiagShift = 1e-5f;  // tfb.FillScaleTriL diag_shift

// TFP fill_triangular (lower): element (r, c <= r) of the k x k matrix comes from
// v[k + t] if t < m - k else v[2m - k - 1 - t], t = r*k + c, m = k(k+1)/2.
__device__ __forceinline__ int tril_index(int r, int c, int k) {
    int m = k * (k + 1) / 2;
    int t = r * k + c;
    return t < m - k ? k + t : 2 * m - k - 1 - t;
}

constexpr int TRIL_MAXK = 64;

// Stage one example's scale factor into LDS as Lp[r][c] (row stride k+1, conflict-free per-row reads),
// softplus + shift already applied on the diagonal.  `v` points at params + k.
// Loads are issued 16 at a time with clamped indices, then consumed: a load under `if (c <= r)` makes the compiler drain
// vmcnt at the join, i.e. one memory round trip per loop trip (16 of them for k = 32: the whole kernel was 17 us of latency).
__device__ __forceinline__ void tril_stage(const float* __restrict__ v, float* Lp, int k, int lane) {
    constexpr int NB = 16;
    for (int t0 = lane; t0 < k * k; t0 += 64 * NB) {
        float x[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int t = t0 + 64 * j;
            const int r = t / k, c = t - r * k;
            x[j] = v[(t < k * k && c <= r) ? tril_index(r, c, k) : 0];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int t = t0 + 64 * j;
            const int r = t / k, c = t - r * k;
            if (t < k * k && c <= r) Lp[r * (k + 1) + c] = c == r ? pm_softplus(x[j]) + kDiagShift : x[j];
        }
    }
}

__global__ __launch_bounds__(256) void tril_sample_kl_fwd_kernel(const float* __restrict__ params,
                                                                   const float* __restrict__ eps,
                                                                   float* __restrict__ z, float* __restrict__ kl,
                                                                   int B, int k) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b_raw = blockIdx.x * 4 + wave;
    const bool active = b_raw < B;
    const int b = active ? b_raw : B - 1;  // inactive waves shadow the last row and store nothing
    const int P = k + k * (k + 1) / 2;
    float* Lp = sm + wave * (k * (k + 1) + k);
    float* es = Lp + k * (k + 1);
    const float* prow = params + (size_t)b * P;
    tril_stage(prow + k, Lp, k, lane);
    if (lane < k) es[lane] = eps[(size_t)b * k + lane];
    __syncthreads();
    float part = 0.f;
    if (lane < k) {
        float mu = prow[lane];
        float acc = mu, sq = mu * mu;
        for (int c = 0; c <= lane; ++c) {
            float l = Lp[lane * (k + 1) + c];
            acc += l * es[c];
            sq += l * l;
        }
        if (active) z[(size_t)b * k + lane] = acc;
        part = 0.5f * (sq - 1.f) - logf(Lp[lane * (k + 1) + lane]);
    }
    part = pm_wave_sum(part);
    if (lane == 0 && active) kl[b] = part;
}

__global__ __launch_bounds__(256) void tril_sample_kl_bwd_kernel(const float* __restrict__ params,
                                                                   const float* __restrict__ eps,
                                                                   const float* __restrict__ dz,
                                                                   const float* __restrict__ dz2,
                                                                   const float* __restrict__ g_kl,
                                                                   float* __restrict__ dparams, int B, int k) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b_raw = blockIdx.x * 4 + wave;
    const bool active = b_raw < B;
    const int b = active ? b_raw : B - 1;  // inactive waves shadow the last row and store nothing
    const int P = k + k * (k + 1) / 2;
    float* es = sm + wave * 2 * k;
    float* dzs = es + k;
    const float* prow = params + (size_t)b * P;
    float* drow = dparams + (size_t)b * P;
    if (lane < k) {
        es[lane] = eps[(size_t)b * k + lane];
        // dz2 (may be NULL): a second gradient w.r.t. z that the caller would otherwise add with its own launch (the
        // posterior-matching branch's d matching_ll / dz beside the decoder's, vae.py:136-140)
        dzs[lane] = dz[(size_t)b * k + lane] + (dz2 ? dz2[(size_t)b * k + lane] : 0.f);
    }
    __syncthreads();
    const float gk = g_kl[b];
    if (!active) return;
    const float mu = lane < k ? prow[lane] : 0.f;
    // loads (16 per lane, clamped), arithmetic, stores in three phases: interleaved, every load waits for the store in front
    // of it (vmcnt counts both) - 16 round trips
    constexpr int NB = 16;
    for (int t0 = lane; t0 < k * k; t0 += 64 * NB) {
        float raw[NB], gv[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int t = t0 + 64 * j;
            const int r = t / k, c = t - r * k;
            raw[j] = prow[(t < k * k && c <= r) ? k + tril_index(r, c, k) : 0];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int t = t0 + 64 * j;
            const int tt = t < k * k ? t : 0;
            const int r = tt / k, c = tt - r * k;
            float g;
            if (c == r) {
                const float l = pm_softplus(raw[j]) + kDiagShift;
                g = (dzs[r] * es[c] + gk * (l - 1.f / l)) * pm_sigmoid(raw[j]);
            } else {
                g = dzs[r] * es[c] + gk * raw[j];
            }
            asm volatile("" : "+v"(g));
            gv[j] = g;
        }
        if (t0 == lane && lane < k) drow[lane] = dzs[lane] + gk * mu;  // d loc
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int t = t0 + 64 * j;
            const int r = t / k, c = t - r * k;
            if (t < k * k && c <= r) drow[k + tril_index(r, c, k)] = gv[j];
        }
    }
}

// MultivariateNormalTriL.log_prob via forward substitution; optional gradient (back substitution).
template <bool BWD>
__global__ __launch_bounds__(256) void tril_logprob_kernel(const float* __restrict__ params,
                                                             const float* __restrict__ z, const float* __restrict__ g,
                                                             float* __restrict__ lp, float* __restrict__ dparams,
                                                             float* __restrict__ dzo, int B, int k) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b_raw = blockIdx.x * 4 + wave;
    const bool active = b_raw < B;
    const int b = active ? b_raw : B - 1;  // inactive waves shadow the last row and store nothing
    const int P = k + k * (k + 1) / 2;
    float* Lp = sm + wave * (k * (k + 1) + 2 * k);
    float* ys = Lp + k * (k + 1);
    float* ws = ys + k;
    const float* prow = params + (size_t)b * P;
    tril_stage(prow + k, Lp, k, lane);
    __syncthreads();
    float dr = lane < k ? z[(size_t)b * k + lane] - prow[lane] : 0.f;
    float ldiag = lane < k ? Lp[lane * (k + 1) + lane] : 1.f;
    float y = 0.f;
    for (int c = 0; c < k; ++c) {  // column-oriented forward substitution
        float yc = __shfl(dr / ldiag, c, 64);
        if (lane == c) y = yc;
        if (lane > c && lane < k) dr -= Lp[lane * (k + 1) + c] * yc;
    }
    float part = lane < k ? -0.5f * y * y - logf(ldiag) - 0.5f * kLog2Pi : 0.f;
    part = pm_wave_sum(part);
    if (!BWD) {
        if (lane == 0 && active) lp[b] = part;
        return;
    }
    // w = L^-T y ; d lp/d z = -w, d lp/d loc = +w, d lp/d L[r][c] = w_r y_c - [r==c]/L_rr
    float yr = y, w = 0.f;
    for (int c = k - 1; c >= 0; --c) {
        float wc = __shfl(yr / ldiag, c, 64);
        if (lane == c) w = wc;
        if (lane < c) yr -= Lp[c * (k + 1) + lane] * wc;
    }
    if (lane < k) {
        ys[lane] = y;
        ws[lane] = w;
    }
    __syncthreads();
    const float gb = g[b];
    float* drow = dparams + (size_t)b * P;
    if (!active) return;
    if (lane < k) {
        drow[lane] = gb * w;
        if (dzo) dzo[(size_t)b * k + lane] = -gb * w;
    }
    for (int t = lane; t < k * k; t += 64) {
        int r = t / k, c = t - r * k;
        if (c > r) continue;
        int idx = k + tril_index(r, c, k);
        float gl = ws[r] * ys[c];
        if (c == r) gl = (gl - 1.f / Lp[r * (k + 1) + r]) * pm_sigmoid(prow[idx]);
        drow[idx] = gb * gl;
    }
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = pm_wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return s;
}

__global__ __launch_bounds__(256) void bernoulli_ll_fwd_kernel(const float* __restrict__ logits,
                                                                 const float* __restrict__ x, float* __restrict__ ll,
                                                                 int D) {
    __shared__ float red[4];
    const size_t base = (size_t)blockIdx.x * D;
    float s = 0.f;
    for (int j = threadIdx.x; j < D; j += 256) {
        float l = logits[base + j], t = x[base + j];
        // x * (-softplus(-l)) + (1 - x) * (-softplus(l))
        s += t * (-pm_softplus(-l)) + (1.f - t) * (-pm_softplus(l));
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) ll[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void bernoulli_ll_bwd_kernel(const float* __restrict__ logits,
                                                                 const float* __restrict__ x, const float* __restrict__ g,
                                                                 float* __restrict__ dpre, long long total, int D, int act,
                                                                 float slope) {
    long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    float l = logits[idx];
    dpre[idx] = g[idx / D] * (x[idx] - pm_sigmoid(l)) * pm_dact(l, act, slope);
}

// Forward AND backward of the Bernoulli log-likelihood in one pass over the logits: d loss / d ll_b = g[b] is known before the
// forward pass runs (train_pm_vae.py:62-70: the loss is linear in reconstruction_ll with coefficient -1 / B), so the
// gradient w.r.t. the decoder's last pre-activation leaves the same kernel that sums ll - the loss kernel and a second
// pass over logits / x are off the step's dependent chain.
__global__ __launch_bounds__(256) void bernoulli_ll_fwd_bwd_kernel(const float* __restrict__ logits,
                                                                     const float* __restrict__ x, const float* __restrict__ g,
                                                                     float* __restrict__ ll, float* __restrict__ dpre, int D,
                                                                     int act, float slope) {
    __shared__ float red[4];
    const size_t base = (size_t)blockIdx.x * D;
    const float gb = g[blockIdx.x];
    float s = 0.f;
    for (int j = threadIdx.x; j < D; j += 256) {
        const float l = logits[base + j], t = x[base + j];
        s += t * (-pm_softplus(-l)) + (1.f - t) * (-pm_softplus(l));
        dpre[base + j] = gb * (t - pm_sigmoid(l)) * pm_dact(l, act, slope);
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) ll[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void normal_ll_fwd_kernel(const float* __restrict__ loc, const float* __restrict__ x,
                                                              const float* __restrict__ log_scale,
                                                              float* __restrict__ ll, int D, float scale_eps) {
    __shared__ float red[4];
    const size_t base = (size_t)blockIdx.x * D;
    const float sigma = expf(log_scale[0]) + scale_eps;
    const float inv = 1.f / sigma;
    const float ls = logf(sigma);
    float s = 0.f;
    for (int j = threadIdx.x; j < D; j += 256) {
        float u = (x[base + j] - loc[base + j]) * inv;
        s += -0.5f * u * u - ls - 0.5f * kLog2Pi;
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) ll[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void normal_ll_bwd_kernel(const float* __restrict__ loc, const float* __restrict__ x,
                                                              const float* __restrict__ log_scale,
                                                              const float* __restrict__ g, float* __restrict__ dloc,
                                                              float* __restrict__ d_log_scale, int D, float scale_eps,
                                                              float* __restrict__ scratch) {
    __shared__ float red[4];
    const size_t base = (size_t)blockIdx.x * D;
    const float e = expf(log_scale[0]);
    const float inv = 1.f / (e + scale_eps);
    const float gb = g[blockIdx.x];
    float s = 0.f;
    for (int j = threadIdx.x; j < D; j += 256) {
        float u = (x[base + j] - loc[base + j]) * inv;
        dloc[base + j] = gb * u * inv;
        s += gb * (u * u - 1.f);  // sigma * d/d sigma of (-0.5 u^2 - log sigma)
    }
    s = block_sum_256(s, red);
    if (!scratch) {
        if (threadIdx.x == 0) atomicAdd(d_log_scale, s * e * inv);  // d sigma / d log_scale = exp(log_scale)
        return;
    }
    // fixed-order form: every example's term goes to scratch[b]; the workgroup that draws the last ticket adds them in example
    // order and makes the ONE += - the same bits on every run (the atomic form adds the B terms in arrival order)
    __shared__ int last;
    unsigned* ticket = reinterpret_cast<unsigned*>(scratch + gridDim.x);
    if (threadIdx.x == 0) {
        __hip_atomic_store(scratch + blockIdx.x, s * e * inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int nb = (int)gridDim.x;
    const int per = (nb + 255) / 256;                       // thread t: examples [t * per, (t + 1) * per)
    float t = 0.f;
    for (int i = threadIdx.x * per; i < nb && i < (threadIdx.x + 1) * per; ++i)
        t += __hip_atomic_load(scratch + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __shared__ float fin[256];
    fin[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int i = 0; i < 256; ++i) tot += fin[i];
        d_log_scale[0] += tot;
        *ticket = 0u;
    }
}

// DiagonalGaussian head (reference distributions.py:58-84): params [B, 2k] = (loc | raw), scale = softplus(raw) + 1e-5.
// One wave per example.  MODE 0: z = loc + scale*eps, out[b] = KL(N(loc, scale) || N(0, I));  MODE 1: out[b] = log N(z; loc, scale).
template <int MODE>
__global__ __launch_bounds__(256) void diag_gaussian_fwd_kernel(const float* __restrict__ params,
                                                                 const float* __restrict__ ez, float* __restrict__ z,
                                                                 float* __restrict__ out, int B, int k) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float part = 0.f;
    for (int j = lane; j < k; j += 64) {
        const float mu = params[(size_t)b * 2 * k + j];
        const float sc = pm_softplus(params[(size_t)b * 2 * k + k + j]) + kDiagShift;
        const float e = ez[(size_t)b * k + j];
        if (MODE == 0) {
            z[(size_t)b * k + j] = mu + sc * e;
            part += -logf(sc) + 0.5f * (sc * sc + mu * mu - 1.f);
        } else {
            const float u = (e - mu) / sc;
            part += -0.5f * u * u - logf(sc) - 0.5f * kLog2Pi;
        }
    }
    part = pm_wave_sum(part);
    if (lane == 0) out[b] = part;
}

// MODE 0: dparams from dz [B,k] and g (d loss / d kl[b]);  MODE 1: dparams and dz (may be NULL) from g (d loss / d lp[b]).
template <int MODE>
__global__ __launch_bounds__(256) void diag_gaussian_bwd_kernel(const float* __restrict__ params,
                                                                 const float* __restrict__ ez, const float* __restrict__ dz,
                                                                 const float* __restrict__ g, float* __restrict__ dparams,
                                                                 float* __restrict__ dzo, long long total, int k) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long b = i / k;
    const int j = (int)(i - b * k);
    const float mu = params[b * 2 * k + j], raw = params[b * 2 * k + k + j];
    const float sc = pm_softplus(raw) + kDiagShift, e = ez[i], gb = g[b];
    if (MODE == 0) {
        const float gz = dz[i];
        dparams[b * 2 * k + j] = gz + gb * mu;
        dparams[b * 2 * k + k + j] = (gz * e + gb * (sc - 1.f / sc)) * pm_sigmoid(raw);
    } else {
        const float u = (e - mu) / sc;
        dparams[b * 2 * k + j] = gb * u / sc;
        dparams[b * 2 * k + k + j] = gb * (u * u - 1.f) / sc * pm_sigmoid(raw);
        if (dzo) dzo[i] = -gb * u / sc;
    }
}

__global__ __launch_bounds__(256) void mask_concat_kernel(const float* __restrict__ x, const float* __restrict__ b,
                                                            float* __restrict__ out, long long R, int C, int Cb) {
    const int W = C + Cb;
    long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= R * W) return;
    long long r = idx / W;
    int c = (int)(idx - r * W);
    if (c < C)
        out[idx] = x[r * C + c] * b[r * Cb + (Cb == 1 ? 0 : c)];
    else
        out[idx] = b[r * Cb + (c - C)];
}

__global__ __launch_bounds__(256) void argmm_build_input_kernel(const float* __restrict__ z,
                                                                  const float* __restrict__ ctx, float* __restrict__ inp,
                                                                  int B, int k, int ctx_dim) {
    const int W = 2 * k + ctx_dim;
    long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    long long total = (long long)k * B * W;
    if (idx >= total) return;
    long long row = idx / W;
    int col = (int)(idx - row * W);
    int i = (int)(row / B), b = (int)(row - (long long)i * B);
    float v;
    if (col < k)
        v = col < i ? z[(size_t)b * k + col] : 0.f;
    else if (col < 2 * k)
        v = (col - k) < i ? 1.f : 0.f;
    else
        v = ctx[(size_t)b * ctx_dim + (col - 2 * k)];
    inp[idx] = v;
}

__global__ __launch_bounds__(256) void argmm_input_bwd_kernel(const float* __restrict__ dinp, float* __restrict__ dz,
                                                                float* __restrict__ dctx, int B, int k, int ctx_dim,
                                                                int accumulate_dz, const float* __restrict__ ctx,
                                                                int ctx_act, float slope) {
    const int W = 2 * k + ctx_dim;
    const int cols = k + ctx_dim;
    long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * cols) return;
    int b = (int)(idx / cols);
    int j = (int)(idx - (long long)b * cols);
    if (j < k) {
        if (!dz) return;
        float s = 0.f;
        for (int i = j + 1; i < k; ++i) s += dinp[((size_t)i * B + b) * W + j];
        if (accumulate_dz) s += dz[(size_t)b * k + j];
        dz[(size_t)b * k + j] = s;
    } else {
        int c = j - k;
        float s = 0.f;
        for (int i = 0; i < k; ++i) s += dinp[((size_t)i * B + b) * W + 2 * k + c];
        if (ctx) s *= pm_dact(ctx[(size_t)b * ctx_dim + c], ctx_act, slope);
        dctx[(size_t)b * ctx_dim + c] = s;
    }
}

constexpr int GMM_MAXC = 16;

// one thread per (example b, latent dim i); kp = k rounded up to a power of two lanes of a wave hold one example (k <= 64; the
// lanes i >= k of a group only take part in the shuffles: configs/pm_vade_mnist.py has latent_dim = 10)
template <bool BWD>
__global__ __launch_bounds__(256) void gmm_logprob_kernel(const float* __restrict__ head, const float* __restrict__ z,
                                                            const float* __restrict__ g, float* __restrict__ mll,
                                                            float* __restrict__ dhead, float* __restrict__ dz, int B,
                                                            int k, int kp, int nc, int accumulate_dz) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long bb = idx / kp;
    int ii = (int)(idx - bb * kp);
    if (BWD) {
        // backward: nothing crosses threads, so thread t takes head row t ([k][B] rows of 3 nc floats): neighbouring threads read
        // neighbouring rows (the forward mapping - a wave = two examples' latent dims - puts them B rows = 30 KB apart: 30
        // scattered 4-byte loads per thread on 32 workgroups, 25 us for 1 MB)
        ii = (int)(idx / B);
        bb = idx - (long long)ii * B;
    }
    const bool ok = bb < B && ii < k;
    int b = ok ? (int)bb : 0;
    int i = ok ? ii : 0;
    const float* hrow = head + ((size_t)i * B + b) * 3 * nc;
    const float zi = z[(size_t)b * k + i];
    float term[GMM_MAXC];
    float mx = -INFINITY, lmx = -INFINITY;
    for (int c = 0; c < nc; ++c) lmx = fmaxf(lmx, hrow[c]);
    float lse_logits = 0.f;
    for (int c = 0; c < nc; ++c) lse_logits += expf(hrow[c] - lmx);
    lse_logits = lmx + logf(lse_logits);
#pragma unroll
    for (int c = 0; c < GMM_MAXC; ++c) {
        if (c < nc) {
            float s = pm_softplus(hrow[2 * nc + c]) + 1e-5f;
            float u = (zi - hrow[nc + c]) / s;
            term[c] = (hrow[c] - lse_logits) - 0.5f * u * u - logf(s) - 0.5f * kLog2Pi;
            mx = fmaxf(mx, term[c]);
        }
    }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < GMM_MAXC; ++c)
        if (c < nc) se += expf(term[c] - mx);
    float lpv = mx + logf(se);
    if (!BWD) {
        float v = ok ? lpv : 0.f;
        for (int o = kp >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (ok && i == 0) mll[b] = v;
        return;
    }
    if (!ok) return;
    const float gb = g[b];
    float* drow = dhead + ((size_t)i * B + b) * 3 * nc;
    float dzi = 0.f;
#pragma unroll
    for (int c = 0; c < GMM_MAXC; ++c) {
        if (c < nc) {
            float r = expf(term[c] - lpv);               // responsibility
            float pi = expf(hrow[c] - lse_logits);        // softmax(logits)
            float raw = hrow[2 * nc + c];
            float s = pm_softplus(raw) + 1e-5f;
            float u = (zi - hrow[nc + c]) / s;
            drow[c] = gb * (r - pi);
            drow[nc + c] = gb * r * u / s;
            drow[2 * nc + c] = gb * r * (u * u - 1.f) / s * pm_sigmoid(raw);
            dzi -= r * u / s;
        }
    }
    if (dz) {
        float v = gb * dzi;
        if (accumulate_dz) v += dz[(size_t)b * k + i];
        dz[(size_t)b * k + i] = v;
    }
}

}  // namespace

extern "C" int pm_tril_sample_kl_fwd(pm_stream_t stream, const float* params, const float* eps, float* z, float* kl,
                                     int B, int k) {
    if (!params || !eps || !z || !kl || B <= 0 || k <= 0 || k > TRIL_MAXK) return PM_EINVAL;
    size_t sh = 4 * (size_t)(k * (k + 1) + k) * sizeof(float);
    hipLaunchKernelGGL(tril_sample_kl_fwd_kernel, dim3((B + 3) / 4), dim3(256), sh, (hipStream_t)stream, params, eps,
                       z, kl, B, k);
    return pm_check_launch("pm_tril_sample_kl_fwd");
}

extern "C" int pm_tril_sample_kl_bwd(pm_stream_t stream, const float* params, const float* eps, const float* dz,
                                     const float* g_kl, float* dparams, int B, int k) {
    if (!params || !eps || !dz || !g_kl || !dparams || B <= 0 || k <= 0 || k > TRIL_MAXK) return PM_EINVAL;
    size_t sh = 4 * (size_t)(2 * k) * sizeof(float);
    hipLaunchKernelGGL(tril_sample_kl_bwd_kernel, dim3((B + 3) / 4), dim3(256), sh, (hipStream_t)stream, params, eps,
                       dz, (const float*)nullptr, g_kl, dparams, B, k);
    return pm_check_launch("pm_tril_sample_kl_bwd");
}

extern "C" int pm_tril_sample_kl_bwd2(pm_stream_t stream, const float* params, const float* eps, const float* dz,
                                      const float* dz2, const float* g_kl, float* dparams, int B, int k) {
    if (!params || !eps || !dz || !g_kl || !dparams || B <= 0 || k <= 0 || k > TRIL_MAXK) return PM_EINVAL;
    size_t sh = 4 * (size_t)(2 * k) * sizeof(float);
    hipLaunchKernelGGL(tril_sample_kl_bwd_kernel, dim3((B + 3) / 4), dim3(256), sh, (hipStream_t)stream, params, eps,
                       dz, dz2, g_kl, dparams, B, k);
    return pm_check_launch("pm_tril_sample_kl_bwd2");
}

extern "C" int pm_tril_logprob_fwd(pm_stream_t stream, const float* params, const float* z, float* lp, int B, int k) {
    if (!params || !z || !lp || B <= 0 || k <= 0 || k > TRIL_MAXK) return PM_EINVAL;
    size_t sh = 4 * (size_t)(k * (k + 1) + 2 * k) * sizeof(float);
    hipLaunchKernelGGL(tril_logprob_kernel<false>, dim3((B + 3) / 4), dim3(256), sh, (hipStream_t)stream, params, z,
                       (const float*)nullptr, lp, (float*)nullptr, (float*)nullptr, B, k);
    return pm_check_launch("pm_tril_logprob_fwd");
}

extern "C" int pm_tril_logprob_bwd(pm_stream_t stream, const float* params, const float* z, const float* g,
                                   float* dparams, float* dz, int B, int k) {
    if (!params || !z || !g || !dparams || B <= 0 || k <= 0 || k > TRIL_MAXK) return PM_EINVAL;
    size_t sh = 4 * (size_t)(k * (k + 1) + 2 * k) * sizeof(float);
    hipLaunchKernelGGL(tril_logprob_kernel<true>, dim3((B + 3) / 4), dim3(256), sh, (hipStream_t)stream, params, z, g,
                       (float*)nullptr, dparams, dz, B, k);
    return pm_check_launch("pm_tril_logprob_bwd");
}

extern "C" int pm_bernoulli_ll_fwd(pm_stream_t stream, const float* logits, const float* x, float* ll, int B, int D) {
    if (!logits || !x || !ll || B <= 0 || D <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(bernoulli_ll_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, x, ll, D);
    return pm_check_launch("pm_bernoulli_ll_fwd");
}

extern "C" int pm_bernoulli_ll_bwd(pm_stream_t stream, const float* logits, const float* x, const float* g,
                                   float* dpre, int B, int D, int act, float slope) {
    if (!logits || !x || !g || !dpre || B <= 0 || D <= 0) return PM_EINVAL;
    long long total = (long long)B * D;
    hipLaunchKernelGGL(bernoulli_ll_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, logits, x, g, dpre, total, D, act, slope);
    return pm_check_launch("pm_bernoulli_ll_bwd");
}

extern "C" int pm_bernoulli_ll_fwd_bwd(pm_stream_t stream, const float* logits, const float* x, const float* g, float* ll,
                                       float* dpre, int B, int D, int act, float slope) {
    if (!logits || !x || !g || !ll || !dpre || B <= 0 || D <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(bernoulli_ll_fwd_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, x, g, ll, dpre, D, act,
                       slope);
    return pm_check_launch("pm_bernoulli_ll_fwd_bwd");
}

extern "C" int pm_normal_ll_fwd(pm_stream_t stream, const float* loc, const float* x, const float* log_scale,
                                float* ll, int B, int D, float scale_eps) {
    if (!loc || !x || !log_scale || !ll || B <= 0 || D <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(normal_ll_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, loc, x, log_scale, ll, D,
                       scale_eps);
    return pm_check_launch("pm_normal_ll_fwd");
}

extern "C" int pm_normal_ll_bwd(pm_stream_t stream, const float* loc, const float* x, const float* log_scale,
                                const float* g, float* dloc, float* d_log_scale, int B, int D, float scale_eps) {
    if (!loc || !x || !log_scale || !g || !dloc || !d_log_scale || B <= 0 || D <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(normal_ll_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, loc, x, log_scale, g, dloc,
                       d_log_scale, D, scale_eps, (float*)nullptr);
    return pm_check_launch("pm_normal_ll_bwd");
}

// scratch: B + 1 floats, the last one zero before the first call (the kernel leaves it zero)
extern "C" int pm_normal_ll_bwd_det(pm_stream_t stream, const float* loc, const float* x, const float* log_scale,
                                    const float* g, float* dloc, float* d_log_scale, int B, int D, float scale_eps,
                                    float* scratch) {
    if (!loc || !x || !log_scale || !g || !dloc || !d_log_scale || !scratch || B <= 0 || D <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(normal_ll_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, loc, x, log_scale, g, dloc,
                       d_log_scale, D, scale_eps, scratch);
    return pm_check_launch("pm_normal_ll_bwd_det");
}

extern "C" int pm_mask_concat(pm_stream_t stream, const float* x, const float* b, float* out, long long R, int C,
                              int Cb) {
    if (!x || !b || !out || R <= 0 || C <= 0 || (Cb != C && Cb != 1)) return PM_EINVAL;
    long long total = R * (C + Cb);
    hipLaunchKernelGGL(mask_concat_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       b, out, R, C, Cb);
    return pm_check_launch("pm_mask_concat");
}

extern "C" int pm_argmm_build_input(pm_stream_t stream, const float* z, const float* ctx, float* inp, int B, int k,
                                    int ctx_dim) {
    if (!z || !ctx || !inp || B <= 0 || k <= 0 || ctx_dim <= 0) return PM_EINVAL;
    long long total = (long long)k * B * (2 * k + ctx_dim);
    hipLaunchKernelGGL(argmm_build_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, z, ctx, inp, B, k, ctx_dim);
    return pm_check_launch("pm_argmm_build_input");
}

extern "C" int pm_argmm_input_bwd(pm_stream_t stream, const float* dinp, float* dz, float* dctx, int B, int k,
                                  int ctx_dim, int accumulate_dz, const float* ctx, int ctx_act, float slope) {
    if (!dinp || !dctx || B <= 0 || k <= 0 || ctx_dim <= 0) return PM_EINVAL;
    long long total = (long long)B * (k + ctx_dim);
    hipLaunchKernelGGL(argmm_input_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, dinp, dz, dctx, B, k, ctx_dim, accumulate_dz, ctx, ctx_act, slope);
    return pm_check_launch("pm_argmm_input_bwd");
}

static bool gmm_shape_ok(int B, int k, int nc) {
    return B > 0 && k > 0 && k <= 64 && nc > 0 && nc <= GMM_MAXC;
}
static int gmm_pow2(int k) {
    int kp = 1;
    while (kp < k) kp <<= 1;
    return kp;
}

extern "C" int pm_gmm_logprob_fwd(pm_stream_t stream, const float* head, const float* z, float* mll, int B, int k,
                                  int nc) {
    if (!head || !z || !mll || !gmm_shape_ok(B, k, nc)) return PM_EINVAL;
    const int kp = gmm_pow2(k);
    long long total = (long long)B * kp;
    hipLaunchKernelGGL(gmm_logprob_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, head, z, (const float*)nullptr, mll, (float*)nullptr, (float*)nullptr, B, k,
                       kp, nc, 0);
    return pm_check_launch("pm_gmm_logprob_fwd");
}

extern "C" int pm_gmm_logprob_bwd(pm_stream_t stream, const float* head, const float* z, const float* g, float* dhead,
                                  float* dz, int B, int k, int nc, int accumulate_dz) {
    if (!head || !z || !g || !dhead || !gmm_shape_ok(B, k, nc)) return PM_EINVAL;
    const int kp = gmm_pow2(k);
    long long total = (long long)B * kp;
    total = (long long)B * k;                              // one thread per head row, 64-thread workgroups: 128 of them at B = 256
    hipLaunchKernelGGL(gmm_logprob_kernel<true>, dim3((unsigned)((total + 63) / 64)), dim3(64), 0,
                       (hipStream_t)stream, head, z, g, (float*)nullptr, dhead, dz, B, k, kp, nc, accumulate_dz);
    return pm_check_launch("pm_gmm_logprob_bwd");
}

extern "C" int pm_diag_gaussian_sample_kl_fwd(pm_stream_t stream, const float* params, const float* eps, float* z,
                                              float* kl, int B, int k) {
    if (!params || !eps || !z || !kl || B <= 0 || k <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(diag_gaussian_fwd_kernel<0>, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, params, eps, z, kl, B, k);
    return pm_check_launch("pm_diag_gaussian_sample_kl_fwd");
}

extern "C" int pm_diag_gaussian_sample_kl_bwd(pm_stream_t stream, const float* params, const float* eps, const float* dz,
                                              const float* g_kl, float* dparams, int B, int k) {
    if (!params || !eps || !dz || !g_kl || !dparams || B <= 0 || k <= 0) return PM_EINVAL;
    const long long total = (long long)B * k;
    hipLaunchKernelGGL(diag_gaussian_bwd_kernel<0>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       params, eps, dz, g_kl, dparams, (float*)nullptr, total, k);
    return pm_check_launch("pm_diag_gaussian_sample_kl_bwd");
}

extern "C" int pm_diag_gaussian_logprob_fwd(pm_stream_t stream, const float* params, const float* z, float* lp, int B,
                                            int k) {
    if (!params || !z || !lp || B <= 0 || k <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(diag_gaussian_fwd_kernel<1>, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, params, z,
                       (float*)nullptr, lp, B, k);
    return pm_check_launch("pm_diag_gaussian_logprob_fwd");
}

extern "C" int pm_diag_gaussian_logprob_bwd(pm_stream_t stream, const float* params, const float* z, const float* g,
                                            float* dparams, float* dz, int B, int k) {
    if (!params || !z || !g || !dparams || B <= 0 || k <= 0) return PM_EINVAL;
    const long long total = (long long)B * k;
    hipLaunchKernelGGL(diag_gaussian_bwd_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       params, z, (const float*)nullptr, g, dparams, dz, total, k);
    return pm_check_launch("pm_diag_gaussian_logprob_bwd");
}
