// Row-wise (HBM / latency-bound) pieces of the Posterior-Matching VDVAE, reference
// posterior_matching/models/vdvae.py.  All 1x1 / 3x3 convolutions run in the gather-GEMM engine.
//   gelu            jax.nn.gelu (tanh form) of [a | b] before a Block's first conv (:282-292); gelu' for the
//                   concatenated inputs of the posterior blocks (single-source gelu' is a GEMM epilogue)
//   avgpool         hk.AvgPool(k, k, VALID) (:297)
//   resize_nearest  x += jax.image.resize(xs[mixin][..., :W], NEAREST) (:676-680)
//   diag_sample_kl  z = loc + (softplus(raw)+1e-5) eps ; KL(posterior || prior), both diagonal (:541-563)
//   diag_tril_kl    KL(stop_grad(posterior) || MultivariateNormalTriL(masked posterior)) (:546-569)
//   affine          final_fn: x * gain + bias (:807-813)
//   dmol            discretised mixture of logistics log-prob / mean, one channel (:351-435)
#include <cstdlib>
#include "pm_common.h"

namespace {

constexpr float kDiagShift = 1e-5f;

__device__ __forceinline__ float gelu_f(float x) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.f + tanhf(u));
}
__device__ __forceinline__ float gelu_d(float x) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    const float t = tanhf(u);
    return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * 0.7978845608028654f * (1.f + 3.f * 0.044715f * x * x);
}
__device__ __forceinline__ float log_sigmoid(float x) { return -pm_softplus(-x); }

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ out, long long R, int Ca, int Cb) {
    const int C = Ca + Cb;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * C) return;
    const long long r = i / C;
    const int c = (int)(i - r * C);
    out[i] = gelu_f(c < Ca ? a[r * Ca + c] : b[r * Cb + (c - Ca)]);
}

// da / db (+)= dout * gelu'(x), x = [a | b]
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ dout, float* __restrict__ da,
                                                        float* __restrict__ db, long long R, int Ca, int Cb,
                                                        int accumulate) {
    const int C = Ca + Cb;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * C) return;
    const long long r = i / C;
    const int c = (int)(i - r * C);
    const bool first = c < Ca;
    const size_t src = first ? (size_t)r * Ca + c : (size_t)r * Cb + (c - Ca);
    float* dst = first ? da : db;
    if (!dst) return;
    const float g = dout[i] * gelu_d(first ? a[src] : b[src]);
    dst[src] = accumulate ? dst[src] + g : g;
}

__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           long long total, int H, int W, int C, int k) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int OH = H / k, OW = W / k;
    const int c = (int)(i % C);
    long long r = i / C;
    const int q = (int)(r % OW);
    r /= OW;
    const int p = (int)(r % OH);
    const long long b = r / OH;
    float s = 0.f;
    for (int dy = 0; dy < k; ++dy)
        for (int dx = 0; dx < k; ++dx) s += x[((b * H + p * k + dy) * W + q * k + dx) * C + c];
    out[i] = s / (float)(k * k);
}

__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx,
                                                           long long total, int H, int W, int C, int k) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // over dx elements
    if (i >= total) return;
    const int OH = H / k, OW = W / k;
    const int c = (int)(i % C);
    long long r = i / C;
    const int xq = (int)(r % W);
    r /= W;
    const int yp = (int)(r % H);
    const long long b = r / H;
    const int p = yp / k, q = xq / k;
    dx[i] = (p < OH && q < OW) ? dout[((b * OH + p) * OW + q) * C + c] / (float)(k * k) : 0.f;
}

// NEAREST source index floor((i + 0.5) * in / out)
__device__ __forceinline__ int nearest_src(int i, int in, int out) {
    int s = (int)floorf(((float)i + 0.5f) * (float)in / (float)out);
    return s < in ? s : in - 1;
}

// dst[b,y,x,c] += src[b, sy, sx, c]  for c < C (src has Cs >= C channels)
__global__ __launch_bounds__(256) void resize_add_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                          long long total, int h, int w, int Cs, int H, int W, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    long long r = i / C;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const long long b = r / H;
    dst[i] += src[((b * h + nearest_src(y, h, H)) * w + nearest_src(x, w, W)) * Cs + c];
}

// dsrc[b,sy,sx,c] += sum over the dst pixels that read it
__global__ __launch_bounds__(256) void resize_add_bwd_kernel(const float* __restrict__ ddst, float* __restrict__ dsrc,
                                                              long long total, int h, int w, int Cs, int H, int W, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // over (b, sy, sx, c < C)
    if (i >= total) return;
    const int c = (int)(i % C);
    long long r = i / C;
    const int sx = (int)(r % w);
    r /= w;
    const int sy = (int)(r % h);
    const long long b = r / h;
    float s = 0.f;
    for (int y = 0; y < H; ++y) {
        if (nearest_src(y, h, H) != sy) continue;
        for (int x = 0; x < W; ++x)
            if (nearest_src(x, w, W) == sx) s += ddst[((b * H + y) * W + x) * C + c];
    }
    dsrc[((b * h + sy) * w + sx) * Cs + c] += s;
}

// out[b, :] = src[:] for every b (x_bias broadcast over the batch)
__global__ __launch_bounds__(256) void broadcast_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                              long long total, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) dst[i] = src[i % n];
}

// out[r, c] = a[r, c] + b[r*ldb + bcol + c]   (x += h with h the trailing columns of the prior block's output, :558)
__global__ __launch_bounds__(256) void add_cols_kernel(const float* __restrict__ a, const float* __restrict__ b, int ldb,
                                                        int bcol, float* __restrict__ out, long long total, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long r = i / C;
    out[i] = a[i] + b[r * ldb + bcol + (int)(i - r * C)];
}
// dst[r*ldd + dcol + c] = src[r, c]
__global__ __launch_bounds__(256) void copy_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, int ldd,
                                                         int dcol, long long total, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long r = i / C;
    dst[r * ldd + dcol + (int)(i - r * C)] = src[i];
}
__global__ __launch_bounds__(256) void scale_shift_kernel(const float* __restrict__ x, float a, float c,
                                                           float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = x[i] * a + c;
}

// One thread per (row, latent dim).  post [R, 2Z] = (loc | raw scale); prior: row stride ldp, (loc | raw) first.
__global__ __launch_bounds__(256) void diag_sample_kl_fwd_kernel(const float* __restrict__ post,
                                                                  const float* __restrict__ prior, int ldp,
                                                                  const float* __restrict__ eps, float* __restrict__ z,
                                                                  float* __restrict__ kl, long long R, int Z, int P) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float t = 0.f;
    long long r = 0;
    if (i < R * Z) {
        r = i / Z;
        const int j = (int)(i - r * Z);
        const float mq = post[r * 2 * Z + j], sq = pm_softplus(post[r * 2 * Z + Z + j]) + kDiagShift;
        const float mp = prior[r * ldp + j], sp = pm_softplus(prior[r * ldp + Z + j]) + kDiagShift;
        z[i] = mq + sq * eps[i];
        const float d = mq - mp;
        t = logf(sp) - logf(sq) + (sq * sq + d * d) / (2.f * sp * sp) - 0.5f;
    }
    // rows of one example are contiguous.  784 rows of an example adding to ONE address one by one serialise in the L2
    // atomic unit (measured: 70 us for 12 544 rows, the arithmetic is ~3 us): when the workgroup's first and last element
    // belong to the same example - always, except at the few-position resolutions - the workgroup adds ONE value.
    __shared__ float wsum[4];
    const long long i_first = (long long)blockIdx.x * 256;
    long long i_last = i_first + 255;
    if (i_last > R * Z - 1) i_last = R * Z - 1;
    const bool one_example = (i_first / Z) / P == (i_last / Z) / P;
    if (one_example) {
        const float w = pm_wave_sum(t);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = w;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(kl + (i_first / Z) / P, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
    } else if (Z == 16) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (i < R * Z && (threadIdx.x & 15) == 0) atomicAdd(kl + r / P, t);
    } else if (i < R * Z) {
        atomicAdd(kl + r / P, t);
    }
}

// dpost [R,2Z] ; dprior written into columns [0, 2Z) of a row-stride-ldp buffer
__global__ __launch_bounds__(256) void diag_sample_kl_bwd_kernel(const float* __restrict__ post,
                                                                  const float* __restrict__ prior, int ldp,
                                                                  const float* __restrict__ eps,
                                                                  const float* __restrict__ dz, float g_kl,
                                                                  float* __restrict__ dpost, float* __restrict__ dprior,
                                                                  long long R, int Z) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * Z) return;
    const long long r = i / Z;
    const int j = (int)(i - r * Z);
    const float rq = post[r * 2 * Z + Z + j], rp = prior[r * ldp + Z + j];
    const float mq = post[r * 2 * Z + j], sq = pm_softplus(rq) + kDiagShift;
    const float mp = prior[r * ldp + j], sp = pm_softplus(rp) + kDiagShift;
    const float d = mq - mp, e = eps[i], gz = dz[i];
    const float isp2 = 1.f / (sp * sp);
    dpost[r * 2 * Z + j] = gz + g_kl * d * isp2;
    dpost[r * 2 * Z + Z + j] = (gz * e + g_kl * (-1.f / sq + sq * isp2)) * pm_sigmoid(rq);
    dprior[r * ldp + j] = -g_kl * d * isp2;
    dprior[r * ldp + Z + j] = g_kl * (1.f / sp - (sq * sq + d * d) * isp2 / sp) * pm_sigmoid(rp);
}

// ---- a decoder block's three launches between its Blocks, fused (reference vdvae.py:532-563) ---------------------------------
//   forward :  z = loc_q + scale_q * eps,  kl[b] += KL(q || p),  x2 = x_in + h + z W_z + b_z          (x += h; x += z_proj(z))
//   backward:  dz = dx2 W_z^T,  d(posterior, prior parameters) of the sample + KL,  d h = dx2 (copied into the prior gradient)
// were  add_cols + diag_sample_kl_fwd + a 16 -> 192 1x1 convolution  and  that convolution's data gradient +
// diag_sample_kl_bwd + copy_cols: three launches of 8 - 13 us each on a dependent chain of 20 decoder blocks, for a few
// hundred FMAs per position.  One wave per position row (SP_RW rows per wave), W_z in LDS, f32 FMAs.
// Rows are dealt one per wave, SP_NW waves per workgroup: the first form (4 waves x 4 rows one after the other) spent its 25 us
// in four serial load -> softplus / log -> wave sum -> store chains per wave and lost to the three separate launches; one row
// per wave runs the chains of a workgroup's 16 rows side by side, W_z is staged once per 16 rows, and a workgroup makes ONE
// atomic add to its example's KL (49 per example at 28 x 28, as the unfused kernel does).
#ifndef PM_SP_RW
#define PM_SP_RW 1
#endif
#ifndef PM_SP_NW
#define PM_SP_NW 16
#endif
constexpr int SP_RW = PM_SP_RW;
constexpr int SP_NW = PM_SP_NW;
__global__ __launch_bounds__(64 * SP_NW) void sample_project_fwd_kernel(const float* __restrict__ post, const float* __restrict__ prior,
                                                                  int ldp, const float* __restrict__ eps,
                                                                  const float* __restrict__ x_in, const float* __restrict__ wz,
                                                                  const float* __restrict__ bz, float* __restrict__ z,
                                                                  float* __restrict__ x2, float* __restrict__ kl, long long R,
                                                                  int Z, int W, int P) {
    extern __shared__ float sp_lds[];
    float* Wl = sp_lds;                  // [Z][W]
    float* bl = Wl + Z * W;              // [W]
    float* zr = bl + W;                  // [SP_NW][Z]
    __shared__ float wsum[SP_NW];
    for (int e = threadIdx.x; e < Z * W; e += 64 * SP_NW) Wl[e] = wz[e];
    for (int e = threadIdx.x; e < W; e += 64 * SP_NW) bl[e] = bz ? bz[e] : 0.f;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long r_first = (long long)blockIdx.x * SP_NW * SP_RW;
    long long r_last = r_first + SP_NW * SP_RW - 1;
    if (r_last > R - 1) r_last = R - 1;
    const bool one_example = r_first / P == r_last / P;      // 784 rows adding to ONE address one by one serialise in L2
    float klacc = 0.f;
    for (int it = 0; it < SP_RW; ++it) {
        const long long r = r_first + (long long)wave * SP_RW + it;
        if (r >= R) break;                                   // wave-uniform
        float t = 0.f;
        if (lane < Z) {
            const float mq = post[r * 2 * Z + lane], sq = pm_softplus(post[r * 2 * Z + Z + lane]) + kDiagShift;
            const float mp = prior[r * ldp + lane], spv = pm_softplus(prior[r * ldp + Z + lane]) + kDiagShift;
            const float zj = mq + sq * eps[r * Z + lane];
            z[r * Z + lane] = zj;
            zr[wave * Z + lane] = zj;
            const float d = mq - mp;
            t = logf(spv) - logf(sq) + (sq * sq + d * d) / (2.f * spv * spv) - 0.5f;
        }
        t = pm_wave_sum(t);
        if (one_example) klacc += t;
        else if (lane == 0) atomicAdd(kl + r / P, t);
        for (int w = lane; w < W; w += 64) {
            float acc = x_in[r * W + w] + prior[r * ldp + 2 * Z + w] + bl[w];
            for (int j = 0; j < Z; ++j) acc = fmaf(zr[wave * Z + j], Wl[j * W + w], acc);
            x2[r * W + w] = acc;
        }
    }
    if (one_example) {
        if (lane == 0) wsum[wave] = klacc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int w = 0; w < SP_NW; ++w) t += wsum[w];
            atomicAdd(kl + r_first / P, t);
        }
    }
}

__global__ __launch_bounds__(64 * SP_NW) void sample_project_bwd_kernel(const float* __restrict__ post, const float* __restrict__ prior,
                                                                  int ldp, const float* __restrict__ eps,
                                                                  const float* __restrict__ dx2, const float* __restrict__ wz,
                                                                  float g_kl, float* __restrict__ dpost,
                                                                  float* __restrict__ dprior, long long R, int Z, int W) {
    extern __shared__ float sp_lds[];
    const int WP = W + 1;                // padded pitch: the Z rows of a column fall into different banks
    float* Wl = sp_lds;                  // [Z][W + 1]
    float* dr = Wl + Z * WP;             // [SP_NW][W]
    for (int e = threadIdx.x; e < Z * W; e += 64 * SP_NW) Wl[(e / W) * WP + e % W] = wz[e];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // lane -> (latent j, one of NCH column ranges): NCH * Z = 64 for Z = 16; other Z: NCH = 1 and lanes >= Z idle in the dot product
    const int NCH = Z <= 16 ? 64 / 16 : 1;
    const int Zs = Z <= 16 ? 16 : Z;     // lanes per column range
    const int j = lane % Zs, ch = lane / Zs;
    const int wlen = (W + NCH - 1) / NCH;
    float* drow = dr + wave * W;
    for (int it = 0; it < SP_RW; ++it) {
        const long long r = ((long long)blockIdx.x * SP_NW + wave) * SP_RW + it;
        if (r >= R) break;                                   // wave-uniform
        for (int w = lane; w < W; w += 64) {
            const float v = dx2[r * W + w];
            drow[w] = v;
            dprior[r * ldp + 2 * Z + w] = v;                 // d h = d x1 = dx2
        }
        float s = 0.f;
        if (j < Z && ch < NCH) {
            const int w0 = ch * wlen, w1 = w0 + wlen < W ? w0 + wlen : W;
            for (int w = w0; w < w1; ++w) s = fmaf(drow[w], Wl[j * WP + w], s);
        }
        if (NCH == 4) {                                      // sum over the four column ranges: lanes j, j + 16, j + 32, j + 48
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
        }
        if (lane < Z) {
            const float gz = s;
            const float rq = post[r * 2 * Z + Z + lane], rp = prior[r * ldp + Z + lane];
            const float mq = post[r * 2 * Z + lane], sq = pm_softplus(rq) + kDiagShift;
            const float mp = prior[r * ldp + lane], spv = pm_softplus(rp) + kDiagShift;
            const float d = mq - mp, e = eps[r * Z + lane];
            const float isp2 = 1.f / (spv * spv);
            dpost[r * 2 * Z + lane] = gz + g_kl * d * isp2;
            dpost[r * 2 * Z + Z + lane] = (gz * e + g_kl * (-1.f / sq + sq * isp2)) * pm_sigmoid(rq);
            dprior[r * ldp + lane] = -g_kl * d * isp2;
            dprior[r * ldp + Z + lane] = g_kl * (1.f / spv - (sq * sq + d * d) * isp2 / spv) * pm_sigmoid(rp);
        }
    }
}

// TFP fill_triangular index (see pm_heads.hip)
__device__ __forceinline__ int tril_index(int r, int c, int k) {
    int m = k * (k + 1) / 2;
    int t = r * k + c;
    return t < m - k ? k + t : 2 * m - k - 1 - t;
}

// KL( N(mu_a, diag(s_a)^2) || N(mu_b, L L^T) ) per row, one wave per row, Z <= 16.
//   f = 0.5 [ sum_j s_j^2 |M e_j|^2 + |M d|^2 - Z ] + sum log L_ii - sum log s_i,  M = L^-1, d = mu_b - mu_a
// Backward (w.r.t. the masked-posterior parameters only: the posterior enters with stop_gradient, vdvae.py:546-551):
//   Q = M diag(s^2) M^T + u u^T (u = M d);  dL = -M^T Q (lower part) + diag(1/L_ii);  d mu_b = M^T u
constexpr int DT_SPW = 4;   // rows per wave of the forward form
template <bool BWD>
__global__ __launch_bounds__(256) void diag_tril_kl_kernel(const float* __restrict__ post, const float* __restrict__ mp,
                                                            float* __restrict__ kl, float g, float* __restrict__ dmp,
                                                            long long R, int Z, int P) {
    constexpr int ZM = 16;
    __shared__ float Ls[4][ZM * (ZM + 1)], Ms[4][ZM * (ZM + 1)], Qs[4][ZM * (ZM + 1)], vs[4][4 * ZM];
    __shared__ float wsum[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // forward: a workgroup owns DT_SPW * 4 consecutive rows (a wave walks DT_SPW of them) and adds ONE value per example it
    // touches - one by one, the 784 rows of an example serialise in the L2 atomic unit (83 us measured for 12 544 rows)
    constexpr int SPW = BWD ? 1 : DT_SPW;
    const long long wg_first = (long long)blockIdx.x * 4 * SPW;
    long long wg_last = wg_first + 4 * SPW - 1;
    if (wg_last > R - 1) wg_last = R - 1;
    const bool one_example = !BWD && wg_first / P == wg_last / P;
    float wave_total = 0.f;
    for (int it = 0; it < SPW; ++it) {
    const long long r_raw = wg_first + (long long)wave * SPW + it;
    const bool active = r_raw < R;
    const long long r = active ? r_raw : R - 1;
    const int NP = Z + Z * (Z + 1) / 2;
    if (it > 0) __syncthreads();
    float* L = Ls[wave];
    float* M = Ms[wave];
    float* Q = Qs[wave];
    float* sa = vs[wave];          // s_a
    float* dd = sa + ZM;           // d = mu_b - mu_a
    float* uu = dd + ZM;           // u = M d
    const float* prow = mp + (size_t)r * NP;
    for (int t = lane; t < Z * Z; t += 64) {
        const int i = t / Z, c = t - i * Z;
        float x = 0.f;
        if (c <= i) {
            x = prow[Z + tril_index(i, c, Z)];
            if (c == i) x = pm_softplus(x) + kDiagShift;
        }
        L[i * (ZM + 1) + c] = x;
        M[i * (ZM + 1) + c] = 0.f;
    }
    if (lane < Z) {
        sa[lane] = pm_softplus(post[(size_t)r * 2 * Z + Z + lane]) + kDiagShift;
        dd[lane] = prow[lane] - post[(size_t)r * 2 * Z + lane];
    }
    __syncthreads();
    // M = L^-1 row by row: M[i][j] = (delta_ij - sum_{k=j}^{i-1} L[i][k] M[k][j]) / L[i][i]
    for (int i = 0; i < Z; ++i) {
        if (lane <= i) {
            float s = lane == i ? 1.f : 0.f;
            for (int k = lane; k < i; ++k) s -= L[i * (ZM + 1) + k] * M[k * (ZM + 1) + lane];
            M[i * (ZM + 1) + lane] = s / L[i * (ZM + 1) + i];
        }
        __syncthreads();
    }
    float part = 0.f;
    if (lane < Z) {
        float u = 0.f, cn = 0.f;
        for (int j = 0; j <= lane; ++j) u += M[lane * (ZM + 1) + j] * dd[j];
        for (int i = lane; i < Z; ++i) cn += M[i * (ZM + 1) + lane] * M[i * (ZM + 1) + lane];   // |M e_lane|^2
        uu[lane] = u;
        part = 0.5f * (sa[lane] * sa[lane] * cn + u * u - 1.f) + logf(L[lane * (ZM + 1) + lane]) - logf(sa[lane]);
    }
    part = pm_wave_sum(part);
    if (!BWD) {
        if (one_example) wave_total += active ? part : 0.f;
        else if (lane == 0 && active) atomicAdd(kl + r / P, part);
        continue;
    }
    __syncthreads();
    for (int t = lane; t < Z * Z; t += 64) {   // Q[i][c] = sum_j M[i][j] s_j^2 M[c][j] + u_i u_c
        const int i = t / Z, c = t - i * Z;
        float s = uu[i] * uu[c];
        const int jm = i < c ? i : c;
        for (int j = 0; j <= jm; ++j) s += M[i * (ZM + 1) + j] * sa[j] * sa[j] * M[c * (ZM + 1) + j];
        Q[i * (ZM + 1) + c] = s;
    }
    __syncthreads();
    float* drow = dmp + (size_t)r * NP;
    if (!active) return;
    if (lane < Z) {                             // d mu_b = M^T u
        float s = 0.f;
        for (int i = lane; i < Z; ++i) s += M[i * (ZM + 1) + lane] * uu[i];
        drow[lane] = g * s;
    }
    for (int t = lane; t < Z * Z; t += 64) {
        const int rr = t / Z, c = t - rr * Z;
        if (c > rr) continue;
        float s = 0.f;
        for (int i = rr; i < Z; ++i) s -= M[i * (ZM + 1) + rr] * Q[i * (ZM + 1) + c];
        const int idx = Z + tril_index(rr, c, Z);
        if (c == rr) s = (s + 1.f / L[rr * (ZM + 1) + rr]) * pm_sigmoid(prow[idx]);
        drow[idx] = g * s;
    }
    }   // rows of this wave
    if (!BWD && one_example) {
        if (lane == 0) wsum[wave] = wave_total;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(kl + wg_first / P, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
    }
}

// The same quantity for Z = 16 with a 16-lane group per site (4 sites per wave, 16 per workgroup) and no barrier inside the
// triangular algebra.  The wave-per-site form above keeps 16 of 64 lanes busy and crosses a workgroup barrier per row of the
// inverse (64 / 39 us forward / backward at 12 544 sites: 13 % of the VDVAE step).  Here lane j of a group owns COLUMN j of
// M = L^-1: forward substitution L x = e_j runs in its registers, L's entries come as broadcast LDS reads (the 16 lanes of
// a group read the same address), u = M d is solved redundantly by every lane off the same L reads.  Backward: the columns
// go to LDS once; lane c then builds column c of Q = M diag(s^2) M^T + u u^T and, in the same pass over the rows of M,
// column c of M^T Q (rows of M are broadcast reads); results leave through an LDS row image so that stores are coalesced.
constexpr int T16 = 16;
constexpr int T16_LP = 20;                        // LDS row pitch of L / M (floats): 16-byte aligned rows
constexpr int T16_SITE = T16 * T16_LP + 4;        // floats per site: the 4-float skew moves the 4 groups of a wave to different banks
constexpr int T16_NP = T16 + T16 * (T16 + 1) / 2; // parameters per site: mean + fill_triangular entries
template <bool BWD>
__global__ __launch_bounds__(256) void diag_tril_kl16_kernel(const float* __restrict__ post, const float* __restrict__ mp,
                                                              float* __restrict__ kl, float g, float* __restrict__ dmp,
                                                              long long R, int P) {
    constexpr int Z = T16;
    __shared__ __attribute__((aligned(16))) float Ls[16 * T16_SITE];
    __shared__ __attribute__((aligned(16))) float Ms[BWD ? 16 * T16_SITE : 4];
    __shared__ float sv[16][3 * Z];               // s_a^2, d = mu_b - mu_a, raw diagonal parameters
    __shared__ float ob[BWD ? 16 * T16_NP : 4];   // the output rows of the workgroup's 16 sites
    __shared__ float wsum[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane >> 4, j = lane & 15;
    const int sl = wave * 4 + grp;                // site slot inside the workgroup
    const long long r_raw = (long long)blockIdx.x * 16 + sl;
    const bool active = r_raw < R;
    const long long r = active ? r_raw : R - 1;
    float* L = Ls + sl * T16_SITE;
    const float* prow = mp + (size_t)r * T16_NP;
#pragma unroll
    for (int it = 0; it < Z; ++it) {              // element (i, c) = (it, j) of L
        float x = 0.f;
        if (j <= it) {
            x = prow[Z + tril_index(it, j, Z)];
            if (j == it) {
                sv[sl][2 * Z + j] = x;
                x = pm_softplus(x) + kDiagShift;
            }
        }
        L[it * T16_LP + j] = x;
    }
    {
        const float sa = pm_softplus(post[(size_t)r * 2 * Z + Z + j]) + kDiagShift;
        sv[sl][j] = sa * sa;
        sv[sl][Z + j] = prow[j] - post[(size_t)r * 2 * Z + j];
    }
    __syncthreads();

    // forward substitution: x = column j of M (x[i] = 0 for i < j falls out of the recurrence), u = M d
    float x[Z], u[Z];
    float ljj = 1.f, uj = 0.f;
#pragma unroll
    for (int i = 0; i < Z; ++i) {
        float lrow[Z];
#pragma unroll
        for (int q = 0; q < Z / 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(L + i * T16_LP + 4 * q);
            lrow[4 * q] = v[0]; lrow[4 * q + 1] = v[1]; lrow[4 * q + 2] = v[2]; lrow[4 * q + 3] = v[3];
        }
        float sx = i == j ? 1.f : 0.f, su = sv[sl][Z + i];
#pragma unroll
        for (int k = 0; k < i; ++k) {
            sx -= lrow[k] * x[k];
            su -= lrow[k] * u[k];
        }
        const float inv = 1.f / lrow[i];
        x[i] = sx * inv;
        u[i] = su * inv;
        if (i == j) { ljj = lrow[i]; uj = u[i]; }
    }
    float cn = 0.f;
#pragma unroll
    for (int i = 0; i < Z; ++i) cn += x[i] * x[i];                       // |M e_j|^2
    const float sa2 = sv[sl][j];
    float part = 0.5f * (sa2 * cn + uj * uj - 1.f) + logf(ljj) - 0.5f * logf(sa2);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);     // over the 16 lanes of the group
    if (!BWD) {
        // one value per example a workgroup touches: added site by site the 784 sites of an example serialise in the L2
        // atomic unit
        const long long wg_first = (long long)blockIdx.x * 16;
        long long wg_last = wg_first + 15;
        if (wg_last > R - 1) wg_last = R - 1;
        if (wg_first / P == wg_last / P) {
            float t = (active && j == 0) ? part : 0.f;
            t = pm_wave_sum(t);
            if (lane == 0) wsum[wave] = t;
            __syncthreads();
            if (threadIdx.x == 0) atomicAdd(kl + wg_first / P, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
        } else if (j == 0 && active) {
            atomicAdd(kl + r / P, part);
        }
        return;
    }
    if constexpr (BWD) {
        float* M = Ms + sl * T16_SITE;
#pragma unroll
        for (int i = 0; i < Z; ++i) M[i * T16_LP + j] = x[i];            // column j
        __syncthreads();
        // mcs[k] = M[c][k] * s_k^2 (row c = this lane's own row of M), c = j
        float mcs[Z];
#pragma unroll
        for (int q = 0; q < Z / 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(M + j * T16_LP + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) mcs[4 * q + e] = v[e] * sv[sl][4 * q + e];
        }
        float G[Z];                                                      // G[rr] = sum_i M[i][rr] Q[i][c]
#pragma unroll
        for (int rr = 0; rr < Z; ++rr) G[rr] = 0.f;
        float dmu = 0.f;                                                 // d mu_b[c] = sum_i M[i][c] u_i
#pragma unroll
        for (int i = 0; i < Z; ++i) {
            float mrow[Z];
#pragma unroll
            for (int q = 0; q < Z / 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(M + i * T16_LP + 4 * q);
                mrow[4 * q] = v[0]; mrow[4 * q + 1] = v[1]; mrow[4 * q + 2] = v[2]; mrow[4 * q + 3] = v[3];
            }
            float qi = u[i] * uj;                                        // Q[i][c]
#pragma unroll
            for (int k = 0; k <= i; ++k) qi += mrow[k] * mcs[k];         // M[i][k] = 0 for k > i, mcs[k] = 0 for k > c
#pragma unroll
            for (int rr = 0; rr <= i; ++rr) G[rr] += mrow[rr] * qi;
            dmu += x[i] * u[i];
        }
        float* orow = ob + sl * T16_NP;
        orow[j] = g * dmu;
#pragma unroll
        for (int rr = 0; rr < Z; ++rr) {
            if (rr < j) continue;                                        // lower part only: entries (rr, c = j), rr >= j
            float sres = -G[rr];
            if (rr == j) sres = (sres + 1.f / ljj) * pm_sigmoid(sv[sl][2 * Z + j]);
            orow[Z + tril_index(rr, j, Z)] = g * sres;
        }
        __syncthreads();
        const long long row0 = (long long)blockIdx.x * 16;
        long long nrows = R - row0;
        if (nrows > 16) nrows = 16;
        float* dst = dmp + (size_t)row0 * T16_NP;
        for (int t = threadIdx.x; t < (int)nrows * T16_NP; t += 256) dst[t] = ob[t];
    }
}

__global__ __launch_bounds__(256) void affine_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gain,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          long long total, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) out[i] = x[i] * gain[i % C] + bias[i % C];
}

// dx = dout * gain ; dgain[c] += sum_r dout*x ; dbias[c] += sum_r dout.  Block = 256 rows chunk x all channels.
__global__ __launch_bounds__(256) void affine_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gain,
                                                          const float* __restrict__ dout, float* __restrict__ dx,
                                                          float* __restrict__ dgain, float* __restrict__ dbias,
                                                          long long R, int C, int rows_per_block, long long part_stride) {
    // part_stride > 0 (partial-sum mode, pm_affine_bwd_part): dgain / dbias are arenas, this workgroup STORES its sums into
    // slot blockIdx.x (one writer per element; pm_reduce_partials / the optimizer add the slots in a fixed order)
    const bool part = part_stride > 0;
    if (part) {
        dgain += (size_t)blockIdx.x * part_stride;
        dbias += (size_t)blockIdx.x * part_stride;
    }
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    for (int c = threadIdx.x; c < C; c += 256) {
        float sg = 0.f, sb = 0.f;
        const float gc = gain[c];
        for (int j = 0; j < rows_per_block && r0 + j < R; ++j) {
            const size_t o = (size_t)(r0 + j) * C + c;
            const float d = dout[o];
            dx[o] = d * gc;
            sg += d * x[o];
            sb += d;
        }
        if (part) {
            dgain[c] = sg;
            dbias[c] = sb;
        } else {
            atomicAdd(dgain + c, sg);
            atomicAdd(dbias + c, sb);
        }
    }
}

// Discretised logistic mixture, one channel: params [R, nm, 3] = (logit, loc, raw scale); value in [0, 255].
constexpr int DMOL_MAXM = 16;
template <bool BWD>
__global__ __launch_bounds__(256) void dmol_kernel(const float* __restrict__ params, const float* __restrict__ value,
                                                    float* __restrict__ ll, const float g, float* __restrict__ dparams,
                                                    long long R, int nm, int P, float low, float high) {
    const long long r_raw = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool active = r_raw < R;
    if (BWD && !active) return;
    const long long r = active ? r_raw : R - 1;            // forward: idle lanes redo the last row and add nothing (wave sums below)
    const float* pr = params + (size_t)r * nm * 3;
    const float half = 0.5f * (high - low);
    float y = value[r];
    y = fminf(fmaxf(y, low), high);
    float lw[DMOL_MAXM], comp[DMOL_MAXM], dup[DMOL_MAXM], ddn[DMOL_MAXM];
    float mx = -INFINITY;
    for (int k = 0; k < nm; ++k) mx = fmaxf(mx, pr[3 * k]);
    float se = 0.f;
    for (int k = 0; k < nm; ++k) se += expf(pr[3 * k] - mx);
    const float lse_w = mx + logf(se);
    float best = -INFINITY;
    for (int k = 0; k < nm; ++k) {
        const float loc = low + half * (pr[3 * k + 1] + 1.f);
        const float sc = (pm_softplus(pr[3 * k + 2]) + expf(-7.f)) * half;
        const float up = (y + 0.5f - loc) / sc, dn = (y - 0.5f - loc) / sc;
        float c, a_up = 0.f, a_dn = 0.f;      // a_up = dc/dup, a_dn = dc/ddn
        if (y >= high) {                       // P = 1 - F(y - .5) = sigmoid(-dn)
            c = log_sigmoid(-dn);
            a_dn = -pm_sigmoid(dn);
        } else if (y <= low) {                 // P = F(y + .5) = sigmoid(up)
            c = log_sigmoid(up);
            a_up = pm_sigmoid(-up);
        } else {
            const float lcu = log_sigmoid(up), lsu = log_sigmoid(-up), lcd = log_sigmoid(dn), lsd = log_sigmoid(-dn);
            const bool use_sf = lsu < lcu;     // TFP: difference of the smaller pair (log-survival vs log-cdf)
            const float big = use_sf ? lsd : lcu, small = use_sf ? lsu : lcd;
            c = big + log1pf(-expf(fminf(small - big, 0.f)));
            a_up = expf(lcu + lsu - c);        // sigmoid'(up) / P
            a_dn = -expf(lcd + lsd - c);
        }
        comp[k] = c;
        lw[k] = pr[3 * k] - lse_w;
        dup[k] = a_up;
        ddn[k] = a_dn;
        best = fmaxf(best, lw[k] + c);
    }
    float s = 0.f;
    for (int k = 0; k < nm; ++k) s += expf(lw[k] + comp[k] - best);
    const float lp = best + logf(s);
    if (!BWD) {
        // one atomic add per WAVE when its 64 rows belong to one example (rows are example-major): a thread-per-row add put
        // P = 784 adds on one address per example - 135 us per call, twice per step, on the link between the VDVAE's forward and
        // backward pass
        const int e = active ? (int)(r / P) : -1;
        const int e0 = __shfl(e, 0, 64);
        if (__all(!active || e == e0)) {
            const float t = pm_wave_sum(active ? lp : 0.f);
            if ((threadIdx.x & 63) == 0 && e0 >= 0) atomicAdd(ll + e0, t);
        } else if (active) {
            atomicAdd(ll + e, lp);
        }
        return;
    }
    float* dr = dparams + (size_t)r * nm * 3;
    for (int k = 0; k < nm; ++k) {
        const float resp = expf(lw[k] + comp[k] - lp);            // posterior responsibility of component k
        const float raw = pr[3 * k + 2];
        const float sc = (pm_softplus(raw) + expf(-7.f)) * half;
        const float loc = low + half * (pr[3 * k + 1] + 1.f);
        const float up = (y + 0.5f - loc) / sc, dn = (y - 0.5f - loc) / sc;
        dr[3 * k] = g * (resp - expf(lw[k]));
        dr[3 * k + 1] = g * resp * (-(dup[k] + ddn[k]) * half / sc);
        dr[3 * k + 2] = g * resp * (-(up * dup[k] + dn * ddn[k]) / sc) * half * pm_sigmoid(raw);
    }
}

__global__ __launch_bounds__(256) void dmol_mean_kernel(const float* __restrict__ params, float* __restrict__ out,
                                                         long long R, int nm, float low, float high) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float* pr = params + (size_t)r * nm * 3;
    float mx = -INFINITY;
    for (int k = 0; k < nm; ++k) mx = fmaxf(mx, pr[3 * k]);
    float se = 0.f, acc = 0.f;
    for (int k = 0; k < nm; ++k) {
        const float w = expf(pr[3 * k] - mx);
        se += w;
        acc += w * pr[3 * k + 1];
    }
    const float loc = fminf(fmaxf(acc / se, -1.f), 1.f);
    out[r] = rintf(low + 0.5f * (high - low) * (loc + 1.f));    // jnp.round: half to even
}

// out[0..4] = {loss, mean rec_ll, mean kl, mean pm_kl, bpd}   (train_pm_vdvae.py:109-120)
__global__ __launch_bounds__(256) void vdvae_loss_kernel(const float* __restrict__ rec, const float* __restrict__ kl,
                                                          const float* __restrict__ pmkl, int B, float dims,
                                                          float* __restrict__ out) {
    __shared__ float red[3][4];
    float a = 0.f, b = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) {
        a += rec[i];
        b += kl[i];
        c += pmkl[i];
    }
    a = pm_wave_sum(a);
    b = pm_wave_sum(b);
    c = pm_wave_sum(c);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = a;
        red[1][threadIdx.x >> 6] = b;
        red[2][threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float mr = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / B;
        const float mk = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / B;
        const float mp = (red[2][0] + red[2][1] + red[2][2] + red[2][3]) / B;
        const float elbo = mr - mk;
        out[0] = -elbo + mp;
        out[1] = mr;
        out[2] = mk;
        out[3] = mp;
        out[4] = -elbo / (dims * 0.6931471805599453f);
    }
}

}  // namespace

extern "C" int pm_gelu_fwd(pm_stream_t stream, const float* a, const float* b, float* out, long long rows, int Ca, int Cb) {
    if (!a || !out || rows <= 0 || Ca <= 0 || Cb < 0 || (Cb > 0 && !b)) return PM_EINVAL;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(blocks_for(rows * (Ca + Cb))), dim3(256), 0, (hipStream_t)stream, a, b, out,
                       rows, Ca, Cb);
    return pm_check_launch("pm_gelu_fwd");
}

extern "C" int pm_gelu_bwd(pm_stream_t stream, const float* a, const float* b, const float* dout, float* da, float* db,
                           long long rows, int Ca, int Cb, int accumulate) {
    if (!a || !dout || rows <= 0 || Ca <= 0 || Cb < 0 || (Cb > 0 && !b)) return PM_EINVAL;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(blocks_for(rows * (Ca + Cb))), dim3(256), 0, (hipStream_t)stream, a, b, dout,
                       da, db, rows, Ca, Cb, accumulate);
    return pm_check_launch("pm_gelu_bwd");
}

extern "C" int pm_avgpool_fwd(pm_stream_t stream, const float* x, float* out, int B, int H, int W, int C, int k) {
    if (!x || !out || B <= 0 || H < k || W < k || C <= 0 || k <= 0) return PM_EINVAL;
    const long long total = (long long)B * (H / k) * (W / k) * C;
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, out, total, H, W,
                       C, k);
    return pm_check_launch("pm_avgpool_fwd");
}

extern "C" int pm_avgpool_bwd(pm_stream_t stream, const float* dout, float* dx, int B, int H, int W, int C, int k) {
    if (!dout || !dx || B <= 0 || H < k || W < k || C <= 0 || k <= 0) return PM_EINVAL;
    const long long total = (long long)B * H * W * C;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dout, dx, total, H,
                       W, C, k);
    return pm_check_launch("pm_avgpool_bwd");
}

extern "C" int pm_resize_nearest_add(pm_stream_t stream, const float* src, float* dst, int B, int h, int w, int Cs, int H,
                                     int W, int C) {
    if (!src || !dst || B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || C <= 0 || Cs < C) return PM_EINVAL;
    const long long total = (long long)B * H * W * C;
    hipLaunchKernelGGL(resize_add_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, total, h, w,
                       Cs, H, W, C);
    return pm_check_launch("pm_resize_nearest_add");
}

extern "C" int pm_resize_nearest_add_bwd(pm_stream_t stream, const float* ddst, float* dsrc, int B, int h, int w, int Cs,
                                         int H, int W, int C) {
    if (!ddst || !dsrc || B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || C <= 0 || Cs < C) return PM_EINVAL;
    const long long total = (long long)B * h * w * C;
    hipLaunchKernelGGL(resize_add_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, ddst, dsrc, total,
                       h, w, Cs, H, W, C);
    return pm_check_launch("pm_resize_nearest_add_bwd");
}

extern "C" int pm_broadcast_rows(pm_stream_t stream, const float* src, float* dst, long long B, long long n) {
    if (!src || !dst || B <= 0 || n <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(broadcast_rows_kernel, dim3(blocks_for(B * n)), dim3(256), 0, (hipStream_t)stream, src, dst, B * n, n);
    return pm_check_launch("pm_broadcast_rows");
}

extern "C" int pm_diag_sample_kl_fwd(pm_stream_t stream, const float* post, const float* prior, int ldp, const float* eps,
                                     float* z, float* kl, long long rows, int Z, int P) {
    if (!post || !prior || !eps || !z || !kl || rows <= 0 || Z <= 0 || P <= 0 || ldp < 2 * Z) return PM_EINVAL;
    hipLaunchKernelGGL(diag_sample_kl_fwd_kernel, dim3(blocks_for(rows * Z)), dim3(256), 0, (hipStream_t)stream, post, prior,
                       ldp, eps, z, kl, rows, Z, P);
    return pm_check_launch("pm_diag_sample_kl_fwd");
}

extern "C" int pm_diag_sample_kl_bwd(pm_stream_t stream, const float* post, const float* prior, int ldp, const float* eps,
                                     const float* dz, float g_kl, float* dpost, float* dprior, long long rows, int Z) {
    if (!post || !prior || !eps || !dz || !dpost || !dprior || rows <= 0 || Z <= 0 || ldp < 2 * Z) return PM_EINVAL;
    hipLaunchKernelGGL(diag_sample_kl_bwd_kernel, dim3(blocks_for(rows * Z)), dim3(256), 0, (hipStream_t)stream, post, prior,
                       ldp, eps, dz, g_kl, dpost, dprior, rows, Z);
    return pm_check_launch("pm_diag_sample_kl_bwd");
}

extern "C" int pm_sample_project_fwd(pm_stream_t stream, const float* post, const float* prior, int ldp, const float* eps,
                                     const float* x_in, const float* wz, const float* bz, float* z, float* x2, float* kl,
                                     long long rows, int Z, int W, int P) {
    if (!post || !prior || !eps || !x_in || !wz || !z || !x2 || !kl || rows <= 0 || Z <= 0 || Z > 64 || W <= 0 || P <= 0 ||
        ldp < 2 * Z + W)
        return PM_EINVAL;
    const size_t lds = ((size_t)Z * W + W + SP_NW * Z) * sizeof(float);
    if (lds > 60 * 1024) return PM_EINVAL;
    hipLaunchKernelGGL(sample_project_fwd_kernel, dim3((unsigned)((rows + SP_NW * SP_RW - 1) / (SP_NW * SP_RW))), dim3(64 * SP_NW), lds,
                       (hipStream_t)stream, post, prior, ldp, eps, x_in, wz, bz, z, x2, kl, rows, Z, W, P);
    return pm_check_launch("pm_sample_project_fwd");
}

extern "C" int pm_sample_project_bwd(pm_stream_t stream, const float* post, const float* prior, int ldp, const float* eps,
                                     const float* dx2, const float* wz, float g_kl, float* dpost, float* dprior, long long rows,
                                     int Z, int W) {
    if (!post || !prior || !eps || !dx2 || !wz || !dpost || !dprior || rows <= 0 || Z <= 0 || Z > 64 || W <= 0 ||
        ldp < 2 * Z + W)
        return PM_EINVAL;
    const size_t lds = ((size_t)Z * (W + 1) + SP_NW * W) * sizeof(float);
    if (lds > 60 * 1024) return PM_EINVAL;
    hipLaunchKernelGGL(sample_project_bwd_kernel, dim3((unsigned)((rows + SP_NW * SP_RW - 1) / (SP_NW * SP_RW))), dim3(64 * SP_NW), lds,
                       (hipStream_t)stream, post, prior, ldp, eps, dx2, wz, g_kl, dpost, dprior, rows, Z, W);
    return pm_check_launch("pm_sample_project_bwd");
}

extern "C" int pm_diag_tril_kl_fwd(pm_stream_t stream, const float* post, const float* masked_params, float* kl,
                                   long long rows, int Z, int P) {
    if (!post || !masked_params || !kl || rows <= 0 || Z <= 0 || Z > 16 || P <= 0) return PM_EINVAL;
    static const bool old_form = getenv("PM_TRIL_KL_WAVE") != nullptr;     // A/B switch for measurements
    PM_KTAG(Z == 16 && !old_form ? "diag_tril_kl16_kernel<false>" : "diag_tril_kl_kernel<false>");
    if (Z == 16 && !old_form)
        hipLaunchKernelGGL(diag_tril_kl16_kernel<false>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, (hipStream_t)stream, post,
                           masked_params, kl, 0.f, (float*)nullptr, rows, P);
    else
        hipLaunchKernelGGL(diag_tril_kl_kernel<false>, dim3((unsigned)((rows + 4 * DT_SPW - 1) / (4 * DT_SPW))), dim3(256), 0, (hipStream_t)stream, post,
                       masked_params, kl, 0.f, (float*)nullptr, rows, Z, P);
    return pm_check_launch("pm_diag_tril_kl_fwd");
}

extern "C" int pm_diag_tril_kl_bwd(pm_stream_t stream, const float* post, const float* masked_params, float g,
                                   float* dmasked_params, long long rows, int Z, int P) {
    if (!post || !masked_params || !dmasked_params || rows <= 0 || Z <= 0 || Z > 16 || P <= 0) return PM_EINVAL;
    static const bool old_form = getenv("PM_TRIL_KL_WAVE") != nullptr;
    PM_KTAG(Z == 16 && !old_form ? "diag_tril_kl16_kernel<true>" : "diag_tril_kl_kernel<true>");
    if (Z == 16 && !old_form)
        hipLaunchKernelGGL(diag_tril_kl16_kernel<true>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, (hipStream_t)stream, post,
                           masked_params, (float*)nullptr, g, dmasked_params, rows, P);
    else
        hipLaunchKernelGGL(diag_tril_kl_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, post,
                       masked_params, (float*)nullptr, g, dmasked_params, rows, Z, P);
    return pm_check_launch("pm_diag_tril_kl_bwd");
}

extern "C" int pm_affine_fwd(pm_stream_t stream, const float* x, const float* gain, const float* bias, float* out,
                             long long rows, int C) {
    if (!x || !gain || !bias || !out || rows <= 0 || C <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(affine_fwd_kernel, dim3(blocks_for(rows * C)), dim3(256), 0, (hipStream_t)stream, x, gain, bias, out,
                       rows * C, C);
    return pm_check_launch("pm_affine_fwd");
}

extern "C" int pm_affine_bwd(pm_stream_t stream, const float* x, const float* gain, const float* dout, float* dx,
                             float* dgain, float* dbias, long long rows, int C) {
    if (!x || !gain || !dout || !dx || !dgain || !dbias || rows <= 0 || C <= 0) return PM_EINVAL;
    const int rpb = 64;
    hipLaunchKernelGGL(affine_bwd_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream, x,
                       gain, dout, dx, dgain, dbias, rows, C, rpb, 0LL);
    return pm_check_launch("pm_affine_bwd");
}

extern "C" int pm_affine_bwd_part_slots(long long rows, int* nslots) {
    if (!nslots || rows <= 0) return PM_EINVAL;
    *nslots = (int)((rows + 63) / 64);
    return PM_OK;
}

extern "C" int pm_affine_bwd_part(pm_stream_t stream, const float* x, const float* gain, const float* dout, float* dx,
                                  float* part_gain, float* part_bias, long long part_stride, int nslots, long long rows, int C) {
    if (!x || !gain || !dout || !dx || !part_gain || !part_bias || rows <= 0 || C <= 0 || part_stride < C) return PM_EINVAL;
    const int rpb = 64;
    if (nslots != (int)((rows + rpb - 1) / rpb)) return PM_EINVAL;
    hipLaunchKernelGGL(affine_bwd_kernel, dim3((unsigned)nslots), dim3(256), 0, (hipStream_t)stream, x, gain, dout, dx, part_gain,
                       part_bias, rows, C, rpb, part_stride);
    return pm_check_launch("pm_affine_bwd_part");
}

extern "C" int pm_dmol_ll_fwd(pm_stream_t stream, const float* params, const float* value, float* ll, long long rows,
                              int num_mixtures, int P, float low, float high) {
    if (!params || !value || !ll || rows <= 0 || num_mixtures <= 0 || num_mixtures > DMOL_MAXM || P <= 0 || rows % P)
        return PM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (pm_zero_async(s, ll, (size_t)(rows / P) * sizeof(float))) return PM_ELAUNCH;
    hipLaunchKernelGGL(dmol_kernel<false>, dim3(blocks_for(rows)), dim3(256), 0, s, params, value, ll, 0.f, (float*)nullptr,
                       rows, num_mixtures, P, low, high);
    return pm_check_launch("pm_dmol_ll_fwd");
}

extern "C" int pm_dmol_ll_bwd(pm_stream_t stream, const float* params, const float* value, float g, float* dparams,
                              long long rows, int num_mixtures, int P, float low, float high) {
    if (!params || !value || !dparams || rows <= 0 || num_mixtures <= 0 || num_mixtures > DMOL_MAXM || P <= 0)
        return PM_EINVAL;
    hipLaunchKernelGGL(dmol_kernel<true>, dim3(blocks_for(rows)), dim3(256), 0, (hipStream_t)stream, params, value,
                       (float*)nullptr, g, dparams, rows, num_mixtures, P, low, high);
    return pm_check_launch("pm_dmol_ll_bwd");
}

extern "C" int pm_dmol_mean(pm_stream_t stream, const float* params, float* out, long long rows, int num_mixtures, float low,
                            float high) {
    if (!params || !out || rows <= 0 || num_mixtures <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(dmol_mean_kernel, dim3(blocks_for(rows)), dim3(256), 0, (hipStream_t)stream, params, out, rows,
                       num_mixtures, low, high);
    return pm_check_launch("pm_dmol_mean");
}

extern "C" int pm_vdvae_loss(pm_stream_t stream, const float* rec, const float* kl, const float* pm_kl, int B,
                             float num_dims, float* out) {
    if (!rec || !kl || !pm_kl || !out || B <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(vdvae_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rec, kl, pm_kl, B, num_dims, out);
    return pm_check_launch("pm_vdvae_loss");
}

extern "C" int pm_add_cols(pm_stream_t stream, const float* a, const float* b, int ldb, int bcol, float* out, long long rows,
                           int C) {
    if (!a || !b || !out || rows <= 0 || C <= 0 || bcol < 0 || ldb < bcol + C) return PM_EINVAL;
    hipLaunchKernelGGL(add_cols_kernel, dim3(blocks_for(rows * C)), dim3(256), 0, (hipStream_t)stream, a, b, ldb, bcol, out,
                       rows * C, C);
    return pm_check_launch("pm_add_cols");
}

extern "C" int pm_copy_cols(pm_stream_t stream, const float* src, float* dst, int ldd, int dcol, long long rows, int C) {
    if (!src || !dst || rows <= 0 || C <= 0 || dcol < 0 || ldd < dcol + C) return PM_EINVAL;
    hipLaunchKernelGGL(copy_cols_kernel, dim3(blocks_for(rows * C)), dim3(256), 0, (hipStream_t)stream, src, dst, ldd, dcol,
                       rows * C, C);
    return pm_check_launch("pm_copy_cols");
}

extern "C" int pm_scale_shift(pm_stream_t stream, const float* x, float a, float c, float* out, long long n) {
    if (!x || !out || n <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(scale_shift_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, a, c, out, n);
    return pm_check_launch("pm_scale_shift");
}
