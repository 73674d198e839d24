// Row-wise (HBM-bound) pieces of the PixelCNN partial posterior, reference
// posterior_matching/models/pixel_cnn.py:372-553.  The masked convolutions and hk.Linear layers run in
// the gather-GEMM engine (pm_conv.hip); these kernels are what sits between them:
//   embed        hk.Embed lookup of the code indices (:401) and its scatter-add gradient
//   concat_elu   elu(concat[x, -x]) (:373-374), optionally times the dropout keep mask (:446, :508)
//   gate         x += h_projection; sigmoid gating; residual (:455-460, :516-522, :565-574)
//   rows_sum     gradient of the broadcast conditional projection: sum over the H*W positions
//   elu          the final activation (:548)
//   categorical  tfd.Categorical(logits).log_prob(value) summed per example (:53-63, :553)
#include "pm_common.h"

namespace {

__device__ __forceinline__ float elu_f(float v) { return v > 0.f ? v : expm1f(v); }
__device__ __forceinline__ float elu_d(float v) { return v > 0.f ? 1.f : expf(v); }

__global__ __launch_bounds__(256) void embed_fwd_kernel(const int* __restrict__ idx, const float* __restrict__ table,
                                                         float* __restrict__ out, long long total, int F, int K) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long r = i / F;
    const int f = (int)(i - r * F);
    int k = idx[r];
    k = k < 0 ? 0 : (k >= K ? K - 1 : k);          // jnp indexing clamps out-of-range indices
    out[i] = table[(size_t)k * F + f];
}

__global__ __launch_bounds__(256) void embed_bwd_kernel(const int* __restrict__ idx, const float* __restrict__ dout,
                                                         float* __restrict__ dtable, long long total, int F, int K) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long r = i / F;
    const int f = (int)(i - r * F);
    int k = idx[r];
    k = k < 0 ? 0 : (k >= K ? K - 1 : k);
    atomicAdd(dtable + (size_t)k * F + f, dout[i]);
}

// The same scatter-add with an ORDER-INDEPENDENT result (round 4): every contribution is converted to 64-bit fixed point and
// added with integer atomics - integer addition is associative, so the sum has the same bits whatever order the memory side
// applies the atomics in - then converted back once per table element.  The scale is a power of two chosen per call from
// max |dout| (itself order-independent) so that rows x max cannot overflow 62 bits: the fixed-point sum is EXACT to
// 2^-(62 - log2 rows) of the largest possible sum, i.e. more accurate than an f32 accumulation.  A non-finite dout gives NaN rows
// (as the f32 sum would).  (One wave per table row scanning the index vector - no atomics at all - was measured first: 11.5 ms
// per pm_vqvae_mnist step against 9.75: early in training a handful of codes own thousands of rows each, and their waves add
// those rows one after the other.)
__global__ __launch_bounds__(256) void embed_amax_kernel(const float* __restrict__ dout, long long total, unsigned* __restrict__ amax) {
    unsigned m = 0u;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const unsigned u = __builtin_bit_cast(unsigned, dout[i]) & 0x7fffffffu;     // |x| as bits: monotone, NaN above inf
        m = u > m ? u : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = __shfl_xor(m, o, 64);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(amax, m);
}

// scale = 2^e with rows * max|dout| * 2^e < 2^61 (e from the exponent bits of the maximum; 0 when everything is zero)
__device__ __forceinline__ bool embed_scale(unsigned amax_bits, long long rows, double& scale) {
    if (amax_bits >= 0x7f800000u) return false;                 // inf / NaN somewhere
    const int ex = (int)(amax_bits >> 23) - 127 + 1;            // max|dout| < 2^ex
    int lr = 0;
    while ((1LL << lr) < rows) ++lr;                            // rows <= 2^lr
    scale = ldexp(1.0, 61 - lr - ex);
    return true;
}

__global__ __launch_bounds__(256) void embed_bwd_fixed_kernel(const int* __restrict__ idx, const float* __restrict__ dout,
                                                               unsigned long long* __restrict__ acc, const unsigned* __restrict__ amax,
                                                               long long total, long long rows, int F, int K) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    double scale;
    if (!embed_scale(amax[0], rows, scale)) return;             // the finalize kernel writes NaN
    const long long r = i / F;
    const int f = (int)(i - r * F);
    int k = idx[r];
    k = k < 0 ? 0 : (k >= K ? K - 1 : k);
    const long long q = __double2ll_rn((double)dout[i] * scale);
    if (q) atomicAdd(acc + (size_t)k * F + f, (unsigned long long)q);
}

__global__ __launch_bounds__(256) void embed_finalize_kernel(unsigned long long* __restrict__ acc, const unsigned* __restrict__ amax,
                                                              float* __restrict__ dtable, long long n, long long rows) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double scale;
    if (!embed_scale(amax[0], rows, scale)) {
        dtable[i] = __builtin_nanf("");
        return;
    }
    const long long q = (long long)acc[i];
    if (q) {
        dtable[i] += (float)((double)q / scale);
        acc[i] = 0ull;                                          // left zero for the next call
    }
}

__global__ void embed_reset_kernel(unsigned* amax) { amax[0] = 0u; }

// logical x = [a | b] (widths Ca, Cb; b may be absent) ; out[r] = [elu(x) | elu(-x)] * drop[r]
__global__ __launch_bounds__(256) void concat_elu_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              const float* __restrict__ drop, float* __restrict__ out,
                                                              long long R, int Ca, int Cb) {
    const int C = Ca + Cb;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * C) return;
    const long long r = i / C;
    const int c = (int)(i - r * C);
    const float v = c < Ca ? a[r * Ca + c] : b[r * Cb + (c - Ca)];
    const size_t o = (size_t)r * 2 * C + c;
    float p = elu_f(v), n = elu_f(-v);
    if (drop) {
        p *= drop[o];
        n *= drop[o + C];
    }
    out[o] = p;
    out[o + C] = n;
}

// da / db (+)= dout_pos * elu'(x) * drop_pos - dout_neg * elu'(-x) * drop_neg
__global__ __launch_bounds__(256) void concat_elu_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              const float* __restrict__ drop,
                                                              const float* __restrict__ dout, float* __restrict__ da,
                                                              float* __restrict__ db, long long R, int Ca, int Cb,
                                                              int accumulate, const float* __restrict__ add_a) {
    const int C = Ca + Cb;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * C) return;
    const long long r = i / C;
    const int c = (int)(i - r * C);
    const bool first = c < Ca;
    const size_t src = first ? (size_t)r * Ca + c : (size_t)r * Cb + (c - Ca);
    const float v = first ? a[src] : b[src];
    const size_t o = (size_t)r * 2 * C + c;
    float gp = dout[o], gn = dout[o + C];
    if (drop) {
        gp *= drop[o];
        gn *= drop[o + C];
    }
    const float g = gp * elu_d(v) - gn * elu_d(-v);
    float* dst = first ? da : db;
    if (!dst) return;
    float r_ = accumulate ? dst[src] + g : g;
    if (first && add_a) r_ += add_a[src];            // the block's residual path (d_in += dout) folded into this pass
    dst[src] = r_;
}

// out = input + sigmoid(y_g + h_g) * (y_a + h_a);  y [R, 2F] = [activation | gate], h [B, 2F] broadcast over P rows
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* __restrict__ y, const float* __restrict__ h,
                                                        const float* __restrict__ input, float* __restrict__ out,
                                                        long long R, int F, int P) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * F) return;
    const long long r = i / F;
    const int f = (int)(i - r * F);
    float act = y[r * 2 * F + f], gate = y[r * 2 * F + F + f];
    if (h) {
        const long long bb = r / P;
        act += h[bb * 2 * F + f];
        gate += h[bb * 2 * F + F + f];
    }
    out[i] = input[i] + pm_sigmoid(gate) * act;
}

// dy [R, 2F]: d act = dout * s ; d gate = dout * act * s * (1 - s)
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ y, const float* __restrict__ h,
                                                        const float* __restrict__ dout, float* __restrict__ dy,
                                                        long long R, int F, int P) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * F) return;
    const long long r = i / F;
    const int f = (int)(i - r * F);
    float act = y[r * 2 * F + f], gate = y[r * 2 * F + F + f];
    if (h) {
        const long long bb = r / P;
        act += h[bb * 2 * F + f];
        gate += h[bb * 2 * F + F + f];
    }
    const float s = pm_sigmoid(gate);
    const float g = dout[i];
    dy[r * 2 * F + f] = g * s;
    dy[r * 2 * F + F + f] = g * act * s * (1.f - s);
}

// out[b, n] = sum_{p < P} x[(b*P + p), n].  One workgroup per (example, 32 columns): 8 row groups walk the P positions
// with 128-byte coalesced loads and meet in LDS (a thread per (b, n) walking P strided rows serially left 16 workgroups
// on the chip at the CelebA size: 60 us per call, 16 % of that step).
__global__ __launch_bounds__(1024) void rows_sum_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                         long long total, int N, int P) {
    // 32 row groups (1024 threads): at the CelebA size (B = 16, P = 256, N = 256: 128 workgroups) a thread adds 8 rows, four
    // loads in flight at a time, instead of 32 rows one after the other (15.6 -> ~8 us per call, 48 calls per step)
    __shared__ float red[32][33];
    const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const long long b = blockIdx.y;
    const int n = blockIdx.x * 32 + col;
    float s = 0.f;
    if (n < N) {
        const float* p = x + (size_t)b * P * N + n;
        int j = rg;
        for (; j + 96 < P; j += 128) {
            const float v0 = p[(size_t)j * N], v1 = p[(size_t)(j + 32) * N], v2 = p[(size_t)(j + 64) * N], v3 = p[(size_t)(j + 96) * N];
            s += (v0 + v1) + (v2 + v3);
        }
        for (; j < P; j += 32) s += p[(size_t)j * N];
    }
    red[rg][col] = s;
    __syncthreads();
    if (rg == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) t += red[r][col];
        out[b * N + n] = t;
    }
}

// G tensors of the same shape in ONE launch (blockIdx.z picks the tensor): the row sums of every gated block's dy - the
// conditional projection's gradient, read only at the end of the backward pass - were one 4 us launch-floor kernel per block
// (48 per step at the CelebA PixelCNN's size).
constexpr int RSM_MAX = 64;
struct RowsSumMulti { const float* x[RSM_MAX]; };
__global__ __launch_bounds__(1024) void rows_sum_multi_kernel(RowsSumMulti a, float* __restrict__ out, long long B, int N, int P) {
    __shared__ float red[32][33];
    const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const long long b = blockIdx.y;
    const int n = blockIdx.x * 32 + col;
    float s = 0.f;
    if (n < N) {
        const float* p = a.x[blockIdx.z] + (size_t)b * P * N + n;
        int j = rg;
        for (; j + 96 < P; j += 128) {
            const float v0 = p[(size_t)j * N], v1 = p[(size_t)(j + 32) * N], v2 = p[(size_t)(j + 64) * N], v3 = p[(size_t)(j + 96) * N];
            s += (v0 + v1) + (v2 + v3);
        }
        for (; j < P; j += 32) s += p[(size_t)j * N];
    }
    red[rg][col] = s;
    __syncthreads();
    if (rg == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) t += red[r][col];
        out[((size_t)blockIdx.z * B + b) * N + n] = t;
    }
}

// The same sum with 16-byte loads, one workgroup per (example, 256 columns): 64 column quads x 4 row groups, every thread's
// P / 4 loads are independent (the 32-column form above is 2048 workgroups of 6 KB each at the mnist PixelCNN's size: 18 us
// for 12.8 MB).  N % 4 == 0, 16-byte aligned x / out.
__global__ __launch_bounds__(256) void rows_sum_v4_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int P) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ f4 red[4][64];
    const int cq = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const long long b = blockIdx.y;
    const int n = blockIdx.x * 256 + 4 * cq;
    f4 s = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        const float* p = x + (size_t)b * P * N + n;
        for (int j = rg; j < P; j += 4) s += *reinterpret_cast<const f4*>(p + (size_t)j * N);
    }
    red[rg][cq] = s;
    __syncthreads();
    if (rg == 0 && n < N) *reinterpret_cast<f4*>(out + b * N + n) = (red[0][cq] + red[1][cq]) + (red[2][cq] + red[3][cq]);
}

// out[i] = sum_{g < G} x[g*stride + i]
__global__ __launch_bounds__(256) void groups_sum_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          long long n, int G, long long stride, int accumulate) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = accumulate ? out[i] : 0.f;
    for (int g = 0; g < G; ++g) s += x[(size_t)g * stride + i];
    out[i] = s;
}

__global__ __launch_bounds__(256) void elu_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = elu_f(x[i]);
}
__global__ __launch_bounds__(256) void elu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dout,
                                                       float* __restrict__ dx, long long n, int accumulate) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g = dout[i] * elu_d(x[i]);
    dx[i] = accumulate ? dx[i] + g : g;
}

// One wave per row: log_softmax(logits[r])[idx[r]] ; ll[b] = sum over the P rows of example b (atomic).
// One row per wave, 16 rows per workgroup; the workgroup adds its rows' terms per EXAMPLE first (rows are example-major, so at
// most two examples meet in a workgroup when P >= 16) and makes one atomic add per (workgroup, example): one add per ROW put
// P = 256 adds on one address (the CelebA code grid: 66 us on the step's critical link between forward and backward pass).
constexpr int CLL_NW = 16;
__global__ __launch_bounds__(64 * CLL_NW) void categorical_ll_fwd_kernel(const float* __restrict__ logits,
                                                                          const int* __restrict__ idx, float* __restrict__ lse,
                                                                          float* __restrict__ ll, long long R, int K, int P) {
    __shared__ float term[CLL_NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.x * CLL_NW;
    const long long r = r0 + wave;
    if (r < R) {                                       // wave-uniform
        const float* lr = logits + (size_t)r * K;
        float mx = -INFINITY;
        for (int k = lane; k < K; k += 64) mx = fmaxf(mx, lr[k]);
        mx = pm_wave_max(mx);
        float s = 0.f;
        for (int k = lane; k < K; k += 64) s += expf(lr[k] - mx);
        s = pm_wave_sum(s);
        if (lane == 0) {
            const float l = mx + logf(s);
            lse[r] = l;
            int k = idx[r];
            k = k < 0 ? 0 : (k >= K ? K - 1 : k);
            term[wave] = lr[k] - l;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long cur = r0 / P;
        float acc = 0.f;
        for (int w = 0; w < CLL_NW && r0 + w < R; ++w) {
            const long long e = (r0 + w) / P;
            if (e != cur) {
                atomicAdd(ll + cur, acc);
                acc = 0.f;
                cur = e;
            }
            acc += term[w];
        }
        atomicAdd(ll + cur, acc);
    }
}

// dlogits[r, k] = g[b] * (onehot(idx[r])[k] - softmax(logits[r])[k])
__global__ __launch_bounds__(256) void categorical_ll_bwd_kernel(const float* __restrict__ logits,
                                                                  const int* __restrict__ idx,
                                                                  const float* __restrict__ lse,
                                                                  const float* __restrict__ g,
                                                                  float* __restrict__ dlogits, long long R, int K, int P) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * K) return;
    const long long r = i / K;
    const int k = (int)(i - r * K);
    int t = idx[r];
    t = t < 0 ? 0 : (t >= K ? K - 1 : t);
    const float p = expf(logits[i] - lse[r]);
    dlogits[i] = g[r / P] * ((k == t ? 1.f : 0.f) - p);
}

// loss = -mean_b ll[b] ; g_ll[b] = -grad_scale
__global__ __launch_bounds__(256) void neg_mean_loss_kernel(const float* __restrict__ ll, int B, float grad_scale,
                                                             float* __restrict__ out, float* __restrict__ g_ll) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) {
        s += ll[i];
        if (g_ll) g_ll[i] = -grad_scale;
    }
    s = pm_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = -(red[0] + red[1] + red[2] + red[3]) / (float)B;
}

// One wave per example: idx[b*P + pos] = argmax_k(logits[b*P + pos, k] + gumbel[b, k]) (Gumbel-max =
// jax.random.categorical); ties -> lowest k.
__global__ __launch_bounds__(256) void categorical_sample_kernel(const float* __restrict__ logits,
                                                                  const float* __restrict__ gumbel, int* __restrict__ idx,
                                                                  long long B, int K, int P, int pos) {
    const int lane = threadIdx.x & 63;
    const long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* lr = logits + ((size_t)b * P + pos) * K;
    const float* gr = gumbel + (size_t)b * K;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int k = lane; k < K; k += 64) {
        const float v = lr[k] + gr[k];
        if (v > best) { best = v; bi = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) idx[(size_t)b * P + pos] = bi >= K ? 0 : bi;
}

// imp[b, s, :] = clip(where(mask[b, :], x[b, :], imp[b, s, :]), 0, 1) in place (vqvae.py:304-312); mask has Cm
// channels per pixel (1 = broadcast over the C image channels)
__global__ __launch_bounds__(256) void impute_blend_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                            float* __restrict__ imp, long long total, long long D, int S,
                                                            int C, int Cm, float lo, float hi) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long j = i % D;
    const long long b = i / (D * S);
    const float m = mask[b * (D / C * Cm) + (Cm == C ? j : j / C)];
    const float v = m != 0.f ? x[b * D + j] : imp[i];
    imp[i] = lo <= hi ? fminf(fmaxf(v, lo), hi) : v;
}

// psnr[b] = -10 log10( mean_j (mean_s imp[b,s,j] - x[b,j])^2 )   (eval_pm_vqvae.py:133-136)
__global__ __launch_bounds__(256) void imputation_psnr_kernel(const float* __restrict__ imp, const float* __restrict__ x,
                                                               float* __restrict__ psnr, long long D, int S, float scale) {
    __shared__ float red[4];
    const long long b = blockIdx.x;
    float acc = 0.f;
    for (long long j = threadIdx.x; j < D; j += 256) {
        float m = 0.f;
        for (int s = 0; s < S; ++s) m += imp[(b * S + s) * D + j];
        const float d = (m / (float)S - x[b * D + j]) * scale;
        acc = fmaf(d, d, acc);
    }
    acc = pm_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) psnr[b] = -10.f * log10f((red[0] + red[1] + red[2] + red[3]) / (float)D);
}


// ---- 16-byte forms of the four row-wise kernels above (channel counts % 4 == 0, < 2^31 elements, 16-byte aligned operands).
// The scalar forms spend a 64-bit division and four 4-byte accesses per element: at the mnist PixelCNN's 12544 x 256 rows
// that is ~150 VALU instructions per element - they were VALU-bound (15 us for 38 MB), not HBM-bound.
typedef float pc_f32x4 __attribute__((ext_vector_type(4)));

// hk.dropout keep mask of the 4 elements of output vector `o`, drawn in place: exactly the values pm_dropout_mask writes to
// element quad `o` of a mask tensor (same Philox counter / key, same threshold), so the mask never exists in HBM
struct PhiloxDrop { float rate, keep_scale; unsigned long long seed; const int* step_dev; int stream_id; };
__device__ __forceinline__ pc_f32x4 philox_keep(const PhiloxDrop& d, unsigned step, unsigned o) {
    unsigned c[4] = {o, 0u, step, (unsigned)d.stream_id};
    pm_philox4x32_10(c, d.seed);
    pc_f32x4 m;
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = ((float)c[e] * 2.3283064365386963e-10f >= d.rate) ? d.keep_scale : 0.f;
    return m;
}

// DROP 0: no dropout; 1: keep mask read from `drop`; 2: keep mask drawn in place (PhiloxDrop)
template <int DROP>
__global__ __launch_bounds__(256) void concat_elu_fwd_v4_kernel(const pc_f32x4* __restrict__ a, const pc_f32x4* __restrict__ b,
                                                                 const pc_f32x4* __restrict__ drop, pc_f32x4* __restrict__ out,
                                                                 unsigned nv, unsigned Ca4, unsigned Cb4, PhiloxDrop pd) {
    const unsigned C4 = Ca4 + Cb4;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nv) return;
    const unsigned r = i / C4, c4 = i - r * C4;
    const pc_f32x4 v = c4 < Ca4 ? a[r * Ca4 + c4] : b[r * Cb4 + (c4 - Ca4)];
    const unsigned o = r * 2u * C4 + c4;
    pc_f32x4 pos, neg;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        pos[e] = elu_f(v[e]);
        neg[e] = elu_f(-v[e]);
    }
    if (DROP == 1) {
        pos *= drop[o];
        neg *= drop[o + C4];
    } else if (DROP == 2) {
        const unsigned step = pd.step_dev ? (unsigned)pd.step_dev[0] : 0u;
        pos *= philox_keep(pd, step, o);
        neg *= philox_keep(pd, step, o + C4);
    }
    out[o] = pos;
    out[o + C4] = neg;
}

template <int DROP>
__global__ __launch_bounds__(256) void concat_elu_bwd_v4_kernel(const pc_f32x4* __restrict__ a, const pc_f32x4* __restrict__ b,
                                                                 const pc_f32x4* __restrict__ drop, const pc_f32x4* __restrict__ dout,
                                                                 pc_f32x4* __restrict__ da, pc_f32x4* __restrict__ db, unsigned nv,
                                                                 unsigned Ca4, unsigned Cb4, int accumulate,
                                                                 const pc_f32x4* __restrict__ add_a, PhiloxDrop pd) {
    const unsigned C4 = Ca4 + Cb4;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nv) return;
    const unsigned r = i / C4, c4 = i - r * C4;
    const bool first = c4 < Ca4;
    pc_f32x4* dst = first ? da : db;
    if (!dst) return;
    const unsigned src = first ? r * Ca4 + c4 : r * Cb4 + (c4 - Ca4);
    const pc_f32x4 v = first ? a[src] : b[src];
    const unsigned o = r * 2u * C4 + c4;
    pc_f32x4 gp = dout[o], gn = dout[o + C4];
    if (DROP == 1) {
        gp *= drop[o];
        gn *= drop[o + C4];
    } else if (DROP == 2) {
        const unsigned step = pd.step_dev ? (unsigned)pd.step_dev[0] : 0u;
        gp *= philox_keep(pd, step, o);
        gn *= philox_keep(pd, step, o + C4);
    }
    pc_f32x4 g;
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = gp[e] * elu_d(v[e]) - gn[e] * elu_d(-v[e]);
    if (accumulate) g += dst[src];
    if (first && add_a) g += add_a[src];
    dst[src] = g;
}

template <bool BWD>
__global__ __launch_bounds__(256) void gate_v4_kernel(const pc_f32x4* __restrict__ y, const pc_f32x4* __restrict__ h,
                                                       const pc_f32x4* __restrict__ in_or_dout, pc_f32x4* __restrict__ out,
                                                       unsigned nv, unsigned F4, unsigned P) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nv) return;
    const unsigned r = i / F4, f4 = i - r * F4;
    pc_f32x4 act = y[r * 2u * F4 + f4], gate = y[r * 2u * F4 + F4 + f4];
    if (h) {
        const unsigned bb = r / P;
        act += h[bb * 2u * F4 + f4];
        gate += h[bb * 2u * F4 + F4 + f4];
    }
    pc_f32x4 s;
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = pm_sigmoid(gate[e]);
    const pc_f32x4 x = in_or_dout[i];
    if (BWD) {                                       // out = dy [R, 2F]: d act = dout * s ; d gate = dout * act * s * (1 - s)
        out[r * 2u * F4 + f4] = x * s;
        out[r * 2u * F4 + F4 + f4] = x * act * s * (1.f - s);
    } else {
        out[i] = x + s * act;
    }
}

// gate forward AND the next block's concat_elu in one pass: out as gate_v4_kernel<false>, ce [R, 2F] = (elu(out) | elu(-out)) - the
// tensor the next gated block of the same stack opens with (pixel_cnn.py:429-431): its separate launch and its re-read of `out`
// (one link of a dependent chain of ~10 us launches per block) go away.
__global__ __launch_bounds__(256) void gate_ce_v4_kernel(const pc_f32x4* __restrict__ y, const pc_f32x4* __restrict__ h,
                                                          const pc_f32x4* __restrict__ in, pc_f32x4* __restrict__ out,
                                                          pc_f32x4* __restrict__ ce, unsigned nv, unsigned F4, unsigned P) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nv) return;
    const unsigned r = i / F4, f4 = i - r * F4;
    pc_f32x4 act = y[r * 2u * F4 + f4], gate = y[r * 2u * F4 + F4 + f4];
    if (h) {
        const unsigned bb = r / P;
        act += h[bb * 2u * F4 + f4];
        gate += h[bb * 2u * F4 + F4 + f4];
    }
    pc_f32x4 s;
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = pm_sigmoid(gate[e]);
    const pc_f32x4 v = in[i] + s * act;
    out[i] = v;
    pc_f32x4 pos, neg;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        pos[e] = elu_f(v[e]);
        neg[e] = elu_f(-v[e]);
    }
    ce[r * 2u * F4 + f4] = pos;
    ce[r * 2u * F4 + F4 + f4] = neg;
}

// gate backward AND the conditional projection's gradient in one pass: dy as gate_v4_kernel<true>, dh[b] = sum over the P
// positions of image b of dy (what pm_rows_sum makes of dy afterwards: 12.8 MB re-read and one more launch per gated block
// at the mnist PixelCNN's size).  One workgroup per image: thread (fq, rg) owns the activation / gate quads fq of the rows
// rg, rg + RG, ... and keeps their sums in registers; the RG partial sums meet in LDS.  F4 = F / 4 divides 256, F4 <= 64.
__global__ __launch_bounds__(256) void gate_bwd_rows_sum_kernel(const pc_f32x4* __restrict__ y, const pc_f32x4* __restrict__ h,
                                                                 const pc_f32x4* __restrict__ dout, pc_f32x4* __restrict__ dy,
                                                                 pc_f32x4* __restrict__ dh, unsigned F4, unsigned P) {
    __shared__ pc_f32x4 red[256][2];
    const unsigned b = blockIdx.x, RG = 256u / F4;
    const unsigned fq = threadIdx.x % F4, rg = threadIdx.x / F4;
    pc_f32x4 ha = {0.f, 0.f, 0.f, 0.f}, hg = ha, sa = ha, sg = ha;
    if (h) {
        ha = h[b * 2u * F4 + fq];
        hg = h[b * 2u * F4 + F4 + fq];
    }
    for (unsigned j = rg; j < P; j += RG) {
        const unsigned r = b * P + j;
        const pc_f32x4 act = y[r * 2u * F4 + fq] + ha, gate = y[r * 2u * F4 + F4 + fq] + hg;
        pc_f32x4 s;
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = pm_sigmoid(gate[e]);
        const pc_f32x4 x = dout[r * F4 + fq];
        const pc_f32x4 da = x * s, dg = x * act * s * (1.f - s);
        dy[r * 2u * F4 + fq] = da;
        dy[r * 2u * F4 + F4 + fq] = dg;
        sa += da;
        sg += dg;
    }
    red[threadIdx.x][0] = sa;
    red[threadIdx.x][1] = sg;
    __syncthreads();
    if (threadIdx.x < 2u * F4) {                 // quad q of dh[b]: activation half first, then the gate half
        const unsigned half = threadIdx.x / F4, q = threadIdx.x % F4;
        pc_f32x4 t = red[q][half];
        for (unsigned g = 1; g < RG; ++g) t += red[g * F4 + q][half];
        dh[b * 2u * F4 + half * F4 + q] = t;
    }
}

// anonymous-namespace helpers of the launchers
inline bool al16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }
inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int pm_embed_fwd(pm_stream_t stream, const int* idx, const float* table, float* out, long long rows, int F,
                            int K) {
    if (!idx || !table || !out || rows <= 0 || F <= 0 || K <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(blocks_for(rows * F)), dim3(256), 0, (hipStream_t)stream, idx, table, out,
                       rows * F, F, K);
    return pm_check_launch("pm_embed_fwd");
}

extern "C" int pm_embed_bwd(pm_stream_t stream, const int* idx, const float* dout, float* dtable, long long rows, int F,
                            int K) {
    if (!idx || !dout || !dtable || rows <= 0 || F <= 0 || K <= 0) return PM_EINVAL;
    PM_KTAG("embed_bwd_kernel");
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(blocks_for(rows * F)), dim3(256), 0, (hipStream_t)stream, idx, dout, dtable,
                       rows * F, F, K);
    return pm_check_launch("pm_embed_bwd");
}

extern "C" int pm_embed_bwd_exact(pm_stream_t stream, const int* idx, const float* dout, float* dtable, long long rows, int F,
                                  int K, void* scratch) {
    if (!idx || !dout || !dtable || !scratch || rows <= 0 || F <= 0 || K <= 0 || (reinterpret_cast<size_t>(scratch) & 7))
        return PM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(scratch);
    unsigned* amax = reinterpret_cast<unsigned*>(acc + (size_t)K * F);
    const long long total = rows * F;
    long long ab = (total + 255) / 256;
    if (ab > 1024) ab = 1024;
    hipLaunchKernelGGL(embed_reset_kernel, dim3(1), dim3(1), 0, s, amax);
    hipLaunchKernelGGL(embed_amax_kernel, dim3((unsigned)ab), dim3(256), 0, s, dout, total, amax);
    PM_KTAG("embed_bwd_fixed_kernel");
    hipLaunchKernelGGL(embed_bwd_fixed_kernel, dim3(blocks_for(total)), dim3(256), 0, s, idx, dout, acc, amax, total, rows, F, K);
    hipLaunchKernelGGL(embed_finalize_kernel, dim3(blocks_for((long long)K * F)), dim3(256), 0, s, acc, amax, dtable,
                       (long long)K * F, rows);
    return pm_check_launch("pm_embed_bwd_exact");
}

extern "C" int pm_concat_elu_fwd(pm_stream_t stream, const float* a, const float* b, const float* drop, float* out,
                                 long long rows, int Ca, int Cb) {
    if (!a || !out || rows <= 0 || Ca <= 0 || Cb < 0 || (Cb > 0 && !b)) return PM_EINVAL;
    const long long total = rows * 2 * (Ca + Cb);
    if (Ca % 4 == 0 && Cb % 4 == 0 && total < 0x7fffffffLL && al16(a) && al16(b) && al16(drop) && al16(out)) {
        const unsigned nv = (unsigned)(rows * (Ca + Cb) / 4);
        typedef const pc_f32x4* cp;
        PM_KTAG("concat_elu_fwd_v4_kernel<%d>", drop ? 1 : 0);
        const PhiloxDrop none{};
        if (drop) hipLaunchKernelGGL(concat_elu_fwd_v4_kernel<1>, dim3((nv + 255) / 256), dim3(256), 0, (hipStream_t)stream, (cp)a, (cp)b, (cp)drop, (pc_f32x4*)out, nv, (unsigned)Ca / 4, (unsigned)Cb / 4, none);
        else hipLaunchKernelGGL(concat_elu_fwd_v4_kernel<0>, dim3((nv + 255) / 256), dim3(256), 0, (hipStream_t)stream, (cp)a, (cp)b, (cp)drop, (pc_f32x4*)out, nv, (unsigned)Ca / 4, (unsigned)Cb / 4, none);
        return pm_check_launch("pm_concat_elu_fwd");
    }
    PM_KTAG("concat_elu_fwd_kernel");
    hipLaunchKernelGGL(concat_elu_fwd_kernel, dim3(blocks_for(rows * (Ca + Cb))), dim3(256), 0, (hipStream_t)stream, a, b,
                       drop, out, rows, Ca, Cb);
    return pm_check_launch("pm_concat_elu_fwd");
}

extern "C" int pm_concat_elu_bwd(pm_stream_t stream, const float* a, const float* b, const float* drop,
                                 const float* dout, float* da, float* db, long long rows, int Ca, int Cb,
                                 int accumulate, const float* add_a) {
    if (!a || !dout || rows <= 0 || Ca <= 0 || Cb < 0 || (Cb > 0 && !b) || (add_a && !da)) return PM_EINVAL;
    const long long total = rows * 2 * (Ca + Cb);
    if (Ca % 4 == 0 && Cb % 4 == 0 && total < 0x7fffffffLL && al16(a) && al16(b) && al16(drop) && al16(dout) && al16(da) &&
        al16(db) && al16(add_a)) {
        const unsigned nv = (unsigned)(rows * (Ca + Cb) / 4);
        typedef const pc_f32x4* cp;
        PM_KTAG("concat_elu_bwd_v4_kernel<%d>", drop ? 1 : 0);
        const PhiloxDrop none{};
        if (drop) hipLaunchKernelGGL(concat_elu_bwd_v4_kernel<1>, dim3((nv + 255) / 256), dim3(256), 0, (hipStream_t)stream, (cp)a, (cp)b, (cp)drop, (cp)dout, (pc_f32x4*)da, (pc_f32x4*)db, nv, (unsigned)Ca / 4, (unsigned)Cb / 4, accumulate, (cp)add_a, none);
        else hipLaunchKernelGGL(concat_elu_bwd_v4_kernel<0>, dim3((nv + 255) / 256), dim3(256), 0, (hipStream_t)stream, (cp)a, (cp)b, (cp)drop, (cp)dout, (pc_f32x4*)da, (pc_f32x4*)db, nv, (unsigned)Ca / 4, (unsigned)Cb / 4, accumulate, (cp)add_a, none);
        return pm_check_launch("pm_concat_elu_bwd");
    }
    PM_KTAG("concat_elu_bwd_kernel");
    hipLaunchKernelGGL(concat_elu_bwd_kernel, dim3(blocks_for(rows * (Ca + Cb))), dim3(256), 0, (hipStream_t)stream, a, b,
                       drop, dout, da, db, rows, Ca, Cb, accumulate, add_a);
    return pm_check_launch("pm_concat_elu_bwd");
}

// concat_elu + hk.dropout with the keep mask drawn in place: out / da, db as pm_concat_elu_fwd / _bwd with the mask tensor
// pm_dropout_mask(n = rows * 2 (Ca + Cb), rate, seed, step_dev, stream_id) would hold.  16-byte form only (Ca, Cb multiples
// of 4, 16-byte aligned operands): callers fall back to the mask tensor otherwise (PM_EINVAL).
extern "C" int pm_concat_elu_fwd_philox(pm_stream_t stream, const float* a, const float* b, float* out, long long rows,
                                        int Ca, int Cb, float rate, unsigned long long seed, const int* step_dev,
                                        int stream_id) {
    if (!a || !out || rows <= 0 || Ca <= 0 || Cb < 0 || (Cb > 0 && !b) || !(rate >= 0.f) || !(rate < 1.f)) return PM_EINVAL;
    const long long total = rows * 2 * (Ca + Cb);
    if (Ca % 4 || Cb % 4 || total >= 0x7fffffffLL || !al16(a) || !al16(b) || !al16(out)) return PM_EINVAL;
    const unsigned nv = (unsigned)(rows * (Ca + Cb) / 4);
    typedef const pc_f32x4* cp;
    const PhiloxDrop pd{rate, 1.f / (1.f - rate), seed, step_dev, stream_id};
    PM_KTAG("concat_elu_fwd_v4_kernel<2>");
    hipLaunchKernelGGL(concat_elu_fwd_v4_kernel<2>, dim3((nv + 255) / 256), dim3(256), 0, (hipStream_t)stream, (cp)a, (cp)b,
                       (cp) nullptr, (pc_f32x4*)out, nv, (unsigned)Ca / 4, (unsigned)Cb / 4, pd);
    return pm_check_launch("pm_concat_elu_fwd_philox");
}

extern "C" int pm_concat_elu_bwd_philox(pm_stream_t stream, const float* a, const float* b, const float* dout, float* da,
                                        float* db, long long rows, int Ca, int Cb, int accumulate, const float* add_a,
                                        float rate, unsigned long long seed, const int* step_dev, int stream_id) {
    if (!a || !dout || rows <= 0 || Ca <= 0 || Cb < 0 || (Cb > 0 && !b) || (add_a && !da) || !(rate >= 0.f) || !(rate < 1.f))
        return PM_EINVAL;
    const long long total = rows * 2 * (Ca + Cb);
    if (Ca % 4 || Cb % 4 || total >= 0x7fffffffLL || !al16(a) || !al16(b) || !al16(dout) || !al16(da) || !al16(db) ||
        !al16(add_a))
        return PM_EINVAL;
    const unsigned nv = (unsigned)(rows * (Ca + Cb) / 4);
    typedef const pc_f32x4* cp;
    const PhiloxDrop pd{rate, 1.f / (1.f - rate), seed, step_dev, stream_id};
    PM_KTAG("concat_elu_bwd_v4_kernel<2>");
    hipLaunchKernelGGL(concat_elu_bwd_v4_kernel<2>, dim3((nv + 255) / 256), dim3(256), 0, (hipStream_t)stream, (cp)a, (cp)b,
                       (cp) nullptr, (cp)dout, (pc_f32x4*)da, (pc_f32x4*)db, nv, (unsigned)Ca / 4, (unsigned)Cb / 4, accumulate,
                       (cp)add_a, pd);
    return pm_check_launch("pm_concat_elu_bwd_philox");
}

extern "C" int pm_gate_fwd(pm_stream_t stream, const float* y, const float* h, const float* input, float* out,
                           long long rows, int F, int P) {
    if (!y || !input || !out || rows <= 0 || F <= 0 || P <= 0) return PM_EINVAL;
    if (F % 4 == 0 && rows * 2 * F < 0x7fffffffLL && al16(y) && al16(h) && al16(input) && al16(out)) {
        const unsigned nv = (unsigned)(rows * F / 4);
        typedef const pc_f32x4* cp;
        PM_KTAG("gate_v4_kernel<false>");
        hipLaunchKernelGGL(gate_v4_kernel<false>, dim3((nv + 255) / 256), dim3(256), 0, (hipStream_t)stream, (cp)y, (cp)h, (cp)input, (pc_f32x4*)out, nv, (unsigned)F / 4, (unsigned)P);
        return pm_check_launch("pm_gate_fwd");
    }
    PM_KTAG("gate_fwd_kernel");
    hipLaunchKernelGGL(gate_fwd_kernel, dim3(blocks_for(rows * F)), dim3(256), 0, (hipStream_t)stream, y, h, input, out,
                       rows, F, P);
    return pm_check_launch("pm_gate_fwd");
}

extern "C" int pm_gate_fwd_ce(pm_stream_t stream, const float* y, const float* h, const float* input, float* out, float* ce,
                              long long rows, int F, int P) {
    if (!y || !input || !out || !ce || rows <= 0 || F <= 0 || P <= 0) return PM_EINVAL;
    if (F % 4 != 0 || rows * 2 * F >= 0x7fffffffLL || !al16(y) || !al16(h) || !al16(input) || !al16(out) || !al16(ce))
        return PM_EINVAL;                                      // 16-byte form only: callers keep the two launches otherwise
    const unsigned nv = (unsigned)(rows * F / 4);
    typedef const pc_f32x4* cp;
    PM_KTAG("gate_ce_v4_kernel");
    hipLaunchKernelGGL(gate_ce_v4_kernel, dim3((nv + 255) / 256), dim3(256), 0, (hipStream_t)stream, (cp)y, (cp)h, (cp)input,
                       (pc_f32x4*)out, (pc_f32x4*)ce, nv, (unsigned)F / 4, (unsigned)P);
    return pm_check_launch("pm_gate_fwd_ce");
}

extern "C" int pm_gate_bwd(pm_stream_t stream, const float* y, const float* h, const float* dout, float* dy,
                           long long rows, int F, int P) {
    if (!y || !dout || !dy || rows <= 0 || F <= 0 || P <= 0) return PM_EINVAL;
    if (F % 4 == 0 && rows * 2 * F < 0x7fffffffLL && al16(y) && al16(h) && al16(dout) && al16(dy)) {
        const unsigned nv = (unsigned)(rows * F / 4);
        typedef const pc_f32x4* cp;
        PM_KTAG("gate_v4_kernel<true>");
        hipLaunchKernelGGL(gate_v4_kernel<true>, dim3((nv + 255) / 256), dim3(256), 0, (hipStream_t)stream, (cp)y, (cp)h, (cp)dout, (pc_f32x4*)dy, nv, (unsigned)F / 4, (unsigned)P);
        return pm_check_launch("pm_gate_bwd");
    }
    PM_KTAG("gate_bwd_kernel");
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(blocks_for(rows * F)), dim3(256), 0, (hipStream_t)stream, y, h, dout, dy,
                       rows, F, P);
    return pm_check_launch("pm_gate_bwd");
}

extern "C" int pm_gate_bwd_rows_sum(pm_stream_t stream, const float* y, const float* h, const float* dout, float* dy,
                                    float* dh, long long B, int F, int P) {
    if (!y || !dout || !dy || !dh || B <= 0 || F <= 0 || P <= 0) return PM_EINVAL;
    const int F4 = F / 4;
    if (F % 4 || F4 > 64 || 256 % F4 || B * P * 2 * F >= 0x7fffffffLL || B > 0x7fffffffLL || !al16(y) || !al16(h) ||
        !al16(dout) || !al16(dy) || !al16(dh))
        return PM_EINVAL;
    typedef const pc_f32x4* cp;
    PM_KTAG("gate_bwd_rows_sum_kernel");
    hipLaunchKernelGGL(gate_bwd_rows_sum_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, (cp)y, (cp)h, (cp)dout,
                       (pc_f32x4*)dy, (pc_f32x4*)dh, (unsigned)F4, (unsigned)P);
    return pm_check_launch("pm_gate_bwd_rows_sum");
}

extern "C" int pm_rows_sum_multi(pm_stream_t stream, const float* const* xs, int G, float* out, long long B, int N, int P) {
    if (!xs || !out || G <= 0 || G > RSM_MAX || B <= 0 || B > 65535 || N <= 0 || P <= 0) return PM_EINVAL;
    RowsSumMulti a;
    for (int g = 0; g < RSM_MAX; ++g) {
        a.x[g] = xs[g < G ? g : 0];
        if (!a.x[g]) return PM_EINVAL;
    }
    PM_KTAG("rows_sum_multi_kernel");
    hipLaunchKernelGGL(rows_sum_multi_kernel, dim3((unsigned)((N + 31) / 32), (unsigned)B, (unsigned)G), dim3(1024), 0,
                       (hipStream_t)stream, a, out, B, N, P);
    return pm_check_launch("pm_rows_sum_multi");
}

extern "C" int pm_rows_sum(pm_stream_t stream, const float* x, float* out, long long B, int N, int P) {
    if (!x || !out || B <= 0 || N <= 0 || P <= 0) return PM_EINVAL;
    if (B > 65535) return PM_EINVAL;
    if (N % 4 == 0 && N >= 128 && B * ((N + 255) / 256) >= 128 && al16(x) && al16(out)) {     // enough examples to fill the chip
        PM_KTAG("rows_sum_v4_kernel");
        hipLaunchKernelGGL(rows_sum_v4_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream, x,
                           out, N, P);
        return pm_check_launch("pm_rows_sum");
    }
    PM_KTAG("rows_sum_kernel");
    hipLaunchKernelGGL(rows_sum_kernel, dim3((unsigned)((N + 31) / 32), (unsigned)B), dim3(1024), 0, (hipStream_t)stream, x, out,
                       B * N, N, P);
    return pm_check_launch("pm_rows_sum");
}

extern "C" int pm_groups_sum(pm_stream_t stream, const float* x, float* out, long long n, int G, long long stride,
                             int accumulate) {
    if (!x || !out || n <= 0 || G <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(groups_sum_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, out, n, G, stride,
                       accumulate);
    return pm_check_launch("pm_groups_sum");
}

extern "C" int pm_elu_fwd(pm_stream_t stream, const float* x, float* out, long long n) {
    if (!x || !out || n <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(elu_fwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, out, n);
    return pm_check_launch("pm_elu_fwd");
}

extern "C" int pm_elu_bwd(pm_stream_t stream, const float* x, const float* dout, float* dx, long long n, int accumulate) {
    if (!x || !dout || !dx || n <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(elu_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, dout, dx, n, accumulate);
    return pm_check_launch("pm_elu_bwd");
}

extern "C" int pm_categorical_ll_fwd(pm_stream_t stream, const float* logits, const int* idx, float* lse, float* ll,
                                     long long rows, int K, int P) {
    if (!logits || !idx || !lse || !ll || rows <= 0 || K <= 0 || P <= 0 || rows % P != 0) return PM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (pm_zero_async(s, ll, (size_t)(rows / P) * sizeof(float))) return PM_ELAUNCH;
    hipLaunchKernelGGL(categorical_ll_fwd_kernel, dim3((unsigned)((rows + CLL_NW - 1) / CLL_NW)), dim3(64 * CLL_NW), 0, s, logits,
                       idx, lse, ll, rows, K, P);
    return pm_check_launch("pm_categorical_ll_fwd");
}

extern "C" int pm_categorical_ll_bwd(pm_stream_t stream, const float* logits, const int* idx, const float* lse,
                                     const float* g, float* dlogits, long long rows, int K, int P) {
    if (!logits || !idx || !lse || !g || !dlogits || rows <= 0 || K <= 0 || P <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(categorical_ll_bwd_kernel, dim3(blocks_for(rows * K)), dim3(256), 0, (hipStream_t)stream, logits,
                       idx, lse, g, dlogits, rows, K, P);
    return pm_check_launch("pm_categorical_ll_bwd");
}

extern "C" int pm_neg_mean_loss(pm_stream_t stream, const float* ll, int B, float grad_scale, float* out, float* g_ll) {
    if (!ll || !out || B <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(neg_mean_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ll, B, grad_scale, out, g_ll);
    return pm_check_launch("pm_neg_mean_loss");
}

extern "C" int pm_categorical_sample(pm_stream_t stream, const float* logits, const float* gumbel, int* idx, long long B,
                                     int K, int P, int pos) {
    if (!logits || !gumbel || !idx || B <= 0 || K <= 0 || P <= 0 || pos < 0 || pos >= P) return PM_EINVAL;
    hipLaunchKernelGGL(categorical_sample_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits,
                       gumbel, idx, B, K, P, pos);
    return pm_check_launch("pm_categorical_sample");
}

extern "C" int pm_impute_blend(pm_stream_t stream, const float* x, const float* mask, float* imp, long long B, int S,
                               long long D, int C, int Cm, float lo, float hi) {
    if (!x || !mask || !imp || B <= 0 || S <= 0 || D <= 0 || C <= 0 || (Cm != C && Cm != 1) || D % C != 0)
        return PM_EINVAL;
    const long long total = B * S * D;
    hipLaunchKernelGGL(impute_blend_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, mask, imp, total,
                       D, S, C, Cm, lo, hi);
    return pm_check_launch("pm_impute_blend");
}

extern "C" int pm_imputation_psnr(pm_stream_t stream, const float* imp, const float* x, float* psnr, long long B, int S,
                                  long long D, float scale) {
    if (!imp || !x || !psnr || B <= 0 || S <= 0 || D <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(imputation_psnr_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, imp, x, psnr, D, S, scale);
    return pm_check_launch("pm_imputation_psnr");
}
