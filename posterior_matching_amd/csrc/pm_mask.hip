// Device-side observed-feature masks (1 = observed, 0 = missing): the generators of the reference's
// masking.py evaluated on the GPU from a counter-based Philox4x32-10 stream, so that the input pipeline
// never leaves the device (the reference builds every mask in a Python loop inside
// tf.py_function, masking.py:44-47,338-342).
//   image mixtures  masking.py:24-47 (per-example component choice), :94-104 (pixel Bernoulli),
//                   :107-140 (random rectangle, area-bounded by rejection), :143-157 (fixed rectangle),
//                   :160-174 (random square), :235-286 (MNIST / Omniglot / CIFAR-10 mixtures)
//   feature masks   masking.py:84-91 (Bernoulli), :50-81 (uniform: q observed features, q uniform)
// The reference's generators own un-seeded NumPy RandomStates, so its bit streams cannot be reproduced by
// anyone; what is reproduced is each generator's distribution.  The streams below are defined by
// oracle/masking_oracle.py (same Philox counters), against which the kernels are bit-exact.
#include "pm_common.h"

namespace {

__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
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

// uniform integer in [0, n) from one 32-bit draw (multiply-shift)
__device__ __forceinline__ int rand_below(unsigned r, int n) { return (int)(((unsigned long long)r * (unsigned)n) >> 32); }
// Bernoulli(p) from one 32-bit draw: u = r * 2^-32 < p, evaluated in integers (threshold = floor(p * 2^32))
__device__ __forceinline__ bool bern(unsigned r, unsigned long long thr) { return (unsigned long long)r < thr; }

struct MixArgs {
    pm_mask_component comp[PM_MASK_MAX_COMPONENTS];
    int ncomp;
    int B, H, W;
    unsigned k0, k1;
    int stream_id;
};

constexpr unsigned DESC_TAG = 0x80000000u;   // descriptor draws live on their own Philox stream id
constexpr int RECT_MAX_TRIES = 256;

// One workgroup per example.  Thread 0 draws the example's component and its parameters:
//   counter (b, 0, step, stream|TAG)      word 0 -> component (inverse CDF over cum_weight), words 1,2 -> square x, y
//   counter (b, 1 + t, step, stream|TAG)  try t of the random rectangle: x1, x2, y1, y2
// then every thread fills pixels; pixel-Bernoulli examples draw word (e & 3) of counter (e >> 2, 0, step, stream)
// with e the element's flat index in [B, H, W].
// ---- RandomPatternMaskGenerator (masking.py:177-232) without the 10 000 x 10 000 cache ------------------------------
// The reference upsamples a (resolution * max_size)^2 field of uniform noise bicubically (PIL) to max_size^2, thresholds
// it at `density` and hands out random windows whose covered fraction lies within density +- density_std.  Here the
// noise field is a Philox function of (cell, epoch) and only the requested window is interpolated, with Pillow's
// arithmetic (Resample.c precompute_coeffs + the 32bpc horizontal-then-vertical passes: double-precision sums, float32
// intermediate), so a window equals the crop of the full resize bit for bit (oracle/masking_oracle.py restates it and is
// pinned against Pillow itself in tests/test_oracle_kat.py).
constexpr unsigned PATTERN_TAG = 0x50415454u;   // "PATT": word 2 of the noise counters
constexpr int PAT_MAX_DIM = 128;                // window height / width supported by the LDS plan
constexpr int PAT_MAX_LOW = 24;                 // low-resolution rows / columns a window may touch
constexpr int PATTERN_MAX_TRIES = 256;

#pragma clang fp contract(off)
__device__ double pat_bicubic(double x) {       // Pillow's bicubic_filter, a = -0.5
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0;
    if (x < 2.0) return (((x - 5.0) * x + 8.0) * x - 4.0) * a;
    return 0.0;
}

// taps of output coordinate `o` of an in_size -> out_size upscale: first input index and 5 normalised weights
__device__ void pat_taps(int o, int in_size, int out_size, int* lo_out, double* w) {
    const double scale = (double)in_size / (double)out_size;
    const double centre = ((double)o + 0.5) * scale;
    int lo = (int)(centre - 2.0 + 0.5);
    if (lo < 0) lo = 0;
    int hi = (int)(centre + 2.0 + 0.5);
    if (hi > in_size) hi = in_size;
    const int n = hi - lo;
    double tot = 0.0;
    for (int k = 0; k < 5; ++k) {
        const double v = k < n ? pat_bicubic((double)(k + lo) - centre + 0.5) : 0.0;
        w[k] = v;
        tot = tot + v;
    }
    if (tot != 0.0)
        for (int k = 0; k < 5; ++k) w[k] = w[k] / tot;
    *lo_out = lo;
}

struct PatternLds {
    double xw[PAT_MAX_DIM][5], yw[PAT_MAX_DIM][5];
    int xlo[PAT_MAX_DIM], ylo[PAT_MAX_DIM];
    float low[PAT_MAX_LOW][PAT_MAX_LOW];
    float hor[PAT_MAX_LOW][PAT_MAX_DIM];
    unsigned char bits[PAT_MAX_DIM * PAT_MAX_DIM];
    int pos[4];        // y0, x0, accepted, count
    int red[4];
};

// One try: window (y0, x0) of the pattern of `epoch` -> bits[] (1 = inside a blob) and the blob pixel count.  All threads.
__device__ int pattern_window(PatternLds& L, const MixArgs& a, const pm_mask_component& m, unsigned epoch, int y0, int x0) {
    const int H = a.H, W = a.W, LS = m.y1, M = m.size, tid = threadIdx.x;
    for (int i = tid; i < W + H; i += 256) {
        if (i < W) pat_taps(x0 + i, LS, M, &L.xlo[i], L.xw[i]);
        else pat_taps(y0 + i - W, LS, M, &L.ylo[i - W], L.yw[i - W]);
    }
    __syncthreads();
    const int r0 = L.ylo[0], c0 = L.xlo[0];                 // taps are monotone in the output coordinate
    int r1 = L.ylo[H - 1] + 5, c1 = L.xlo[W - 1] + 5;
    if (r1 > LS) r1 = LS;
    if (c1 > LS) c1 = LS;
    const int nr = r1 - r0, nc = c1 - c0;
    for (int i = tid; i < nr * nc; i += 256) {
        const int r = i / nc, c = i - r * nc;
        const unsigned long long cell = (unsigned long long)(r0 + r) * (unsigned)LS + (unsigned)(c0 + c);
        unsigned ctr[4] = {(unsigned)cell, epoch, PATTERN_TAG, (unsigned)a.stream_id | DESC_TAG};
        philox4x32_10(ctr, a.k0, a.k1);
        L.low[r][c] = (float)(ctr[0] >> 8) * 5.9604644775390625e-08f;
    }
    __syncthreads();
    for (int i = tid; i < nr * W; i += 256) {              // horizontal pass: double sums, float32 result
        const int r = i / W, x = i - r * W;
        double ssum = 0.0;
        for (int k = 0; k < 5; ++k) {
            int c = L.xlo[x] + k;
            if (c > LS - 1) c = LS - 1;
            ssum = ssum + (double)L.low[r][c - c0] * L.xw[x][k];
        }
        L.hor[r][x] = (float)ssum;
    }
    __syncthreads();
    int cnt = 0;
    const float dens = m.p;
    for (int i = tid; i < H * W; i += 256) {               // vertical pass + threshold
        const int y = i / W, x = i - y * W;
        double ssum = 0.0;
        for (int k = 0; k < 5; ++k) {
            int r = L.ylo[y] + k;
            if (r > LS - 1) r = LS - 1;
            ssum = ssum + (double)L.hor[r - r0][x] * L.yw[y][k];
        }
        const unsigned char bit = ((float)ssum < dens) ? 1 : 0;
        L.bits[i] = bit;
        cnt += bit;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((tid & 63) == 0) L.red[tid >> 6] = cnt;
    __syncthreads();
    const int total = L.red[0] + L.red[1] + L.red[2] + L.red[3];
    __syncthreads();
    return total;
}
#pragma clang fp contract(fast)

__global__ __launch_bounds__(256) void image_mask_mixture_kernel(MixArgs a, const int* __restrict__ step_dev,
                                                                  float* __restrict__ mask, int* __restrict__ desc_out,
                                                                  unsigned long long* __restrict__ pattern_state) {
    __shared__ int d[6];   // kind, y1, x1, y2, x2 (exclusive ends), component
    extern __shared__ __attribute__((aligned(16))) unsigned char pat_raw[];   // PatternLds when a PATTERN component exists
    const unsigned step = step_dev ? (unsigned)step_dev[0] : 0u;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        unsigned c[4] = {(unsigned)b, 0u, step, (unsigned)a.stream_id | DESC_TAG};
        philox4x32_10(c, a.k0, a.k1);
        const float u = (float)(c[0] >> 8) * 5.9604644775390625e-08f;   // 24-bit uniform in [0, 1): exact in f32
        int ci = a.ncomp - 1;
        for (int i = 0; i < a.ncomp; ++i)
            if (u < a.comp[i].cum_weight) {
                ci = i;
                break;
            }
        const pm_mask_component& m = a.comp[ci];
        int y1 = 0, x1 = 0, y2 = 0, x2 = 0;
        if (m.kind == PM_MASK_FIXED_RECT) {
            y1 = m.y1; x1 = m.x1; y2 = m.y2; x2 = m.x2;
        } else if (m.kind == PM_MASK_SQUARE) {
            x1 = rand_below(c[1], a.W - m.size);
            y1 = rand_below(c[2], a.H - m.size);
            x2 = x1 + m.size;
            y2 = y1 + m.size;
        } else if (m.kind == PM_MASK_RECT) {
            const float area = (float)(a.W * a.H);
            for (int t = 0; t < RECT_MAX_TRIES; ++t) {
                unsigned r[4] = {(unsigned)b, (unsigned)(1 + t), step, (unsigned)a.stream_id | DESC_TAG};
                philox4x32_10(r, a.k0, a.k1);
                int xa = rand_below(r[0], a.W), xb = rand_below(r[1], a.W);
                int ya = rand_below(r[2], a.H), yb = rand_below(r[3], a.H);
                x1 = xa < xb ? xa : xb; x2 = xa < xb ? xb : xa;
                y1 = ya < yb ? ya : yb; y2 = ya < yb ? yb : ya;
                const float cover = (float)((x2 - x1 + 1) * (y2 - y1 + 1));
                if (m.min_prop * area <= cover && cover <= m.max_prop * area) break;
            }
            x2 += 1;   // the reference blanks [y1 : y2 + 1, x1 : x2 + 1]
            y2 += 1;
        }
        d[0] = m.kind; d[1] = y1; d[2] = x1; d[3] = y2; d[4] = x2; d[5] = ci;
        if (desc_out) {
            int* o = desc_out + 6 * b;
            o[0] = m.kind; o[1] = y1; o[2] = x1; o[3] = y2; o[4] = x2; o[5] = ci;
        }
    }
    __syncthreads();
    const int kind = d[0], y1 = d[1], x1 = d[2], y2 = d[3], x2 = d[4];
    const int hw = a.H * a.W;
    float* mb = mask + (size_t)b * hw;
    if (kind == PM_MASK_PATTERN) {
        // counter (b, 1 + t, step, stream|TAG): try t -> x0 = word 0, y0 = word 1 (uniform over the valid origins);
        // accepted when density - std < blob pixels / (H*W) < density + std; after PATTERN_MAX_TRIES the last window stands
        PatternLds& L = *reinterpret_cast<PatternLds*>(pat_raw);
        const pm_mask_component& m = a.comp[d[5]];
        const unsigned epoch = pattern_state ? (unsigned)pattern_state[0] : 0u;
        const double lo_cnt = ((double)m.p - (double)m.min_prop) * (double)hw, hi_cnt = ((double)m.p + (double)m.min_prop) * (double)hw;
        int y0 = 0, x0 = 0;
        for (int t = 0; t < PATTERN_MAX_TRIES; ++t) {
            unsigned r[4] = {(unsigned)b, (unsigned)(1 + t), step, (unsigned)a.stream_id | DESC_TAG};
            philox4x32_10(r, a.k0, a.k1);
            x0 = rand_below(r[0], m.size - a.W + 1);
            y0 = rand_below(r[1], m.size - a.H + 1);
            const int cnt = pattern_window(L, a, m, epoch, y0, x0);
            if ((double)cnt > lo_cnt && (double)cnt < hi_cnt) break;
        }
        for (int e = threadIdx.x; e < hw; e += 256) mb[e] = L.bits[e] ? 0.f : 1.f;
        if (threadIdx.x == 0) {
            if (pattern_state) atomicAdd(&pattern_state[1], (unsigned long long)hw);
            if (desc_out) {
                desc_out[6 * b + 1] = y0;
                desc_out[6 * b + 2] = x0;
            }
        }
        return;
    }
    if (kind == PM_MASK_PIXEL_BERNOULLI) {
        const unsigned long long thr = (unsigned long long)((double)a.comp[d[5]].p * 4294967296.0);
        const long long e0 = (long long)b * hw;
        for (long long q = (e0 >> 2) + threadIdx.x; q * 4 < e0 + hw; q += 256) {
            unsigned c[4] = {(unsigned)q, (unsigned)(q >> 32), step, (unsigned)a.stream_id};
            philox4x32_10(c, a.k0, a.k1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long e = q * 4 + j;
                if (e >= e0 && e < e0 + hw) mb[e - e0] = bern(c[j], thr) ? 1.f : 0.f;
            }
        }
    } else {
        for (int e = threadIdx.x; e < hw; e += 256) {
            const int y = e / a.W, x = e - y * a.W;
            mb[e] = (y >= y1 && y < y2 && x >= x1 && x < x2) ? 0.f : 1.f;
        }
    }
}

__global__ __launch_bounds__(256) void bernoulli_mask_kernel(float* __restrict__ mask, long long n, unsigned long long thr,
                                                               unsigned k0, unsigned k1, const int* __restrict__ step_dev,
                                                               int stream_id) {
    const unsigned step = step_dev ? (unsigned)step_dev[0] : 0u;
    const long long quads = (n + 3) / 4;
    const long long stride = (long long)gridDim.x * 256;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < quads; q += stride) {
        unsigned c[4] = {(unsigned)q, (unsigned)(q >> 32), step, (unsigned)stream_id};
        philox4x32_10(c, k0, k1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (q * 4 + j < n) mask[q * 4 + j] = bern(c[j], thr) ? 1.f : 0.f;
    }
}

// Row b observes q = lo + floor(u * span) features chosen uniformly without replacement: feature i is observed iff
// fewer than q features have a smaller (key, index).  Keys: word (i & 3) of counter (b*ceil(D/4) + (i >> 2), 0, ...);
// q from word 0 of the descriptor counter (b, 0, step, stream|TAG).
__global__ __launch_bounds__(256) void uniform_mask_kernel(float* __restrict__ mask, int B, int D, int lo, int span,
                                                             unsigned k0, unsigned k1, const int* __restrict__ step_dev,
                                                             int stream_id) {
    extern __shared__ unsigned keys[];
    const unsigned step = step_dev ? (unsigned)step_dev[0] : 0u;
    const int b = blockIdx.x;
    const int dq = (D + 3) / 4;
    for (int q = threadIdx.x; q < dq; q += 256) {
        const long long ctr = (long long)b * dq + q;
        unsigned c[4] = {(unsigned)ctr, (unsigned)(ctr >> 32), step, (unsigned)stream_id};
        philox4x32_10(c, k0, k1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (q * 4 + j < D) keys[q * 4 + j] = c[j];
    }
    unsigned dsc[4] = {(unsigned)b, 0u, step, (unsigned)stream_id | DESC_TAG};
    philox4x32_10(dsc, k0, k1);
    int nobs = lo + rand_below(dsc[0], span);
    if (nobs > D) nobs = D;
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += 256) {
        const unsigned ki = keys[i];
        int rank = 0;
        for (int j = 0; j < D; ++j) {
            const unsigned kj = keys[j];
            rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0;
        }
        mask[(size_t)b * D + i] = rank < nobs ? 1.f : 0.f;
    }
}

// ---- uint8 image pipeline (reference utils.py:36-58: tfds uint8 images -> shuffle -> batch -> cast / 255) -----------------
// The whole dataset stays resident in HBM as uint8 (MNIST 47 MB, CelebA 64x64 2.4 GB of the 288 GB); a step draws its batch
// indices from Philox and gathers + converts the rows in one pass: no host work, a quarter of the bytes of f32 storage.
__global__ __launch_bounds__(256) void random_indices_kernel(int* __restrict__ idx, int B, int N, unsigned k0, unsigned k1,
                                                               const int* __restrict__ step_dev, int stream_id) {
    const unsigned step = step_dev ? (unsigned)step_dev[0] : 0u;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q * 4 >= B) return;
    unsigned c[4] = {(unsigned)q, 0u, step, (unsigned)stream_id};
    philox4x32_10(c, k0, k1);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (q * 4 + j < B) idx[q * 4 + j] = rand_below(c[j], N);
}

// dst[b, :] = scale * float(src[idx[b], :]) (idx NULL: row b itself); 16 bytes of uint8 per lane per step when D % 16 == 0
__global__ __launch_bounds__(256) void gather_u8_rows_kernel(const unsigned char* __restrict__ src, const int* __restrict__ idx,
                                                               float* __restrict__ dst, int B, long long D, float scale) {
    const int b = blockIdx.y;
    const long long row = idx ? (long long)idx[b] : (long long)b;
    const unsigned char* s = src + row * D;
    float* o = dst + (long long)b * D;
    if ((D & 15) == 0 && ((reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst)) & 15) == 0) {
        for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v * 16 < D; v += (long long)gridDim.x * 256) {
            const uint4 raw = *reinterpret_cast<const uint4*>(s + v * 16);
            const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 f = {(float)(w[j] & 255u), (float)((w[j] >> 8) & 255u), (float)((w[j] >> 16) & 255u), (float)(w[j] >> 24)};
                f *= scale;
                *reinterpret_cast<f32x4*>(o + v * 16 + j * 4) = f;
            }
        }
    } else {
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < D; e += (long long)gridDim.x * 256)
            o[e] = scale * (float)s[e];
    }
}

// masking.py:226-228: the noise is redrawn once update_freq * max_size^2 pixels have been handed out.  The reference checks
// after every mask; here the check runs once per batch (state[0] = epoch the NEXT launch reads, state[1] = pixels handed out).
__global__ void pattern_advance_kernel(unsigned long long* state, unsigned long long threshold) {
    if (state[1] > threshold) {
        state[0] += 1ull;
        state[1] = 0ull;
    }
}

}  // namespace

extern "C" int pm_image_mask_mixture(pm_stream_t stream, float* mask, int B, int H, int W,
                                     const pm_mask_component* comps, int ncomp, unsigned long long seed,
                                     const int* step_dev, int stream_id, int* desc_out, unsigned long long* pattern_state,
                                     unsigned long long pattern_refresh) {
    if (!mask || !comps || B <= 0 || H <= 0 || W <= 0 || ncomp <= 0 || ncomp > PM_MASK_MAX_COMPONENTS) return PM_EINVAL;
    bool has_pattern = false;
    if ((long long)B * H * W >= (1LL << 40) || stream_id < 0) return PM_EINVAL;
    MixArgs a;
    float prev = 0.f;
    for (int i = 0; i < ncomp; ++i) {
        const pm_mask_component& m = comps[i];
        if (m.cum_weight < prev || m.cum_weight > 1.0001f) return PM_EINVAL;
        prev = m.cum_weight;
        switch (m.kind) {
            case PM_MASK_PIXEL_BERNOULLI: if (!(m.p >= 0.f && m.p <= 1.f)) return PM_EINVAL; break;
            case PM_MASK_FIXED_RECT: break;
            case PM_MASK_SQUARE: if (m.size <= 0 || m.size >= H || m.size >= W) return PM_EINVAL; break;
            case PM_MASK_RECT: if (!(m.min_prop <= m.max_prop) || m.max_prop <= 0.f) return PM_EINVAL; break;
            case PM_MASK_PATTERN: {   // size = max_size, y1 = low-resolution size, p = density, min_prop = density_std
                if (m.size < H || m.size < W || m.y1 < 2 || m.y1 > m.size || H > PAT_MAX_DIM || W > PAT_MAX_DIM) return PM_EINVAL;
                const double scale = (double)m.y1 / (double)m.size;
                if ((double)(H > W ? H : W) * scale + 7.0 > (double)PAT_MAX_LOW || !pattern_state) return PM_EINVAL;
                has_pattern = true;
                break;
            }
            default: return PM_EINVAL;
        }
        a.comp[i] = m;
    }
    a.ncomp = ncomp; a.B = B; a.H = H; a.W = W;
    a.k0 = (unsigned)seed; a.k1 = (unsigned)(seed >> 32); a.stream_id = stream_id;
    const size_t lds = has_pattern ? sizeof(PatternLds) : 0;
    hipLaunchKernelGGL(image_mask_mixture_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, a, step_dev, mask, desc_out,
                       pattern_state);
    if (has_pattern && pattern_refresh > 0)
        hipLaunchKernelGGL(pattern_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, pattern_state, pattern_refresh);
    return pm_check_launch("pm_image_mask_mixture");
}

extern "C" int pm_bernoulli_mask(pm_stream_t stream, float* mask, long long n, float p, unsigned long long seed,
                                 const int* step_dev, int stream_id) {
    if (!mask || n <= 0 || !(p >= 0.f && p <= 1.f) || stream_id < 0) return PM_EINVAL;
    long long blocks = ((n + 3) / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    const unsigned long long thr = (unsigned long long)((double)p * 4294967296.0);
    hipLaunchKernelGGL(bernoulli_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mask, n, thr,
                       (unsigned)seed, (unsigned)(seed >> 32), step_dev, stream_id);
    return pm_check_launch("pm_bernoulli_mask");
}

extern "C" int pm_uniform_mask(pm_stream_t stream, float* mask, int B, int D, int lo, int span, unsigned long long seed,
                               const int* step_dev, int stream_id) {
    if (!mask || B <= 0 || D <= 0 || D > 8192 || lo < 0 || span <= 0 || stream_id < 0) return PM_EINVAL;
    hipLaunchKernelGGL(uniform_mask_kernel, dim3(B), dim3(256), (size_t)D * 4, (hipStream_t)stream, mask, B, D, lo, span,
                       (unsigned)seed, (unsigned)(seed >> 32), step_dev, stream_id);
    return pm_check_launch("pm_uniform_mask");
}

extern "C" int pm_random_indices(pm_stream_t stream, int* idx, int B, int N, unsigned long long seed, const int* step_dev,
                                 int stream_id) {
    if (!idx || B <= 0 || N <= 0 || stream_id < 0) return PM_EINVAL;
    hipLaunchKernelGGL(random_indices_kernel, dim3((unsigned)(((B + 3) / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx,
                       B, N, (unsigned)seed, (unsigned)(seed >> 32), step_dev, stream_id);
    return pm_check_launch("pm_random_indices");
}

extern "C" int pm_gather_u8_rows(pm_stream_t stream, const unsigned char* src, const int* idx, float* dst, int B, long long D,
                                 float scale) {
    if (!src || !dst || B <= 0 || B > 65535 || D <= 0) return PM_EINVAL;
    long long bx = (D / 16 + 255) / 256;
    if (bx < 1) bx = 1;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(gather_u8_rows_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, (hipStream_t)stream, src, idx, dst, B,
                       D, scale);
    return pm_check_launch("pm_gather_u8_rows");
}
