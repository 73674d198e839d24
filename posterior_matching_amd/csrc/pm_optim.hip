// Loss composition, beta / learning-rate schedules and the fused optimizer of the PM-VAE step.
//   loss_fn + beta schedules      reference train_pm_vae.py:28-43,58-72, utils.py:124-136
//   optax chain (Adam, decayed weights, exponential-decay schedule, apply_updates)
//                                 reference train_pm_vae.py:74-83 (optax 0.1.0 semantics)
// The step counter lives in device memory so that a captured HIP graph can be replayed unchanged.
#include <cstdio>
#include "pm_common.h"

namespace {

__device__ float beta_from_step(const pm_loss_cfg& c, int step) {
    if (c.beta_kind == 1) {  // optax.linear_schedule(low, high, steps, begin)
        if (c.period_or_steps <= 0) return c.high;
        int cnt = step - c.delay_or_begin;
        cnt = cnt < 0 ? 0 : (cnt > c.period_or_steps ? c.period_or_steps : cnt);
        float frac = 1.f - (float)cnt / (float)c.period_or_steps;
        return (c.low - c.high) * frac + c.high;
    }
    if (c.beta_kind == 2) {  // cyclical_annealing_schedule (utils.py:127-134)
        int period = c.period_or_steps, delay = c.delay_or_begin;
        int cnt = step - delay;
        int md = cnt % period;
        if (md < 0) md += period;  // jnp `%` is floor-mod
        int half = period / 2;
        cnt = md < 0 ? 0 : (md > half ? half : md);
        float frac = 1.f - (float)cnt / (float)half;
        float x = (c.low - c.high) * frac + c.high;
        return step >= delay ? x : 0.f;
    }
    return 1.f;
}

__global__ __launch_bounds__(256) void pmvae_loss_kernel(const float* __restrict__ rec, const float* __restrict__ kl,
                                                           const float* __restrict__ mll, int B, pm_loss_cfg cfg,
                                                           const int* __restrict__ step_dev, float* __restrict__ out,
                                                           float* __restrict__ g_rec, float* __restrict__ g_kl,
                                                           float* __restrict__ g_mll) {
    __shared__ float red[3][4];
    const int step = step_dev ? step_dev[0] : 0;
    const float beta = beta_from_step(cfg, step);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        if (rec) {
            s0 += rec[b];
            s1 += kl[b];
            s2 += mll[b];
        }
        if (g_rec) g_rec[b] = -cfg.grad_scale;
        if (g_kl) g_kl[b] = beta * cfg.grad_scale;
        if (g_mll) g_mll[b] = -cfg.matching_coef * cfg.grad_scale;
    }
    if (!rec) return;                    // gradients-only call (see pm_pmvae_loss)
    s0 = pm_wave_sum(s0);
    s1 = pm_wave_sum(s1);
    s2 = pm_wave_sum(s2);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        red[0][wave] = s0;
        red[1][wave] = s1;
        red[2][wave] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float m0 = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / B;
        float m1 = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / B;
        float m2 = (red[2][0] + red[2][1] + red[2][2] + red[2][3]) / B;
        out[0] = -(m0 - beta * m1) + cfg.matching_coef * (-m2);
        out[1] = m0;
        out[2] = m1;
        out[3] = m2;
        out[4] = beta;
    }
}

__device__ __forceinline__ float pm_lr(const pm_adam_cfg& c, int count) {
    if (c.lr_kind == 1) {   // optax.linear_schedule: polynomial_schedule(power = 1)
        const float f = fminf(fmaxf((float)count / c.lr_transition_steps, 0.f), 1.f);
        return (c.lr_init - c.lr_end) * (1.f - f) + c.lr_end;
    }
    return c.lr_init * powf(c.lr_decay_rate, (float)count / c.lr_transition_steps);
}

// One Adam element: the arithmetic of optax.scale_by_adam -> add_decayed_weights -> scale_by_schedule -> scale(-1), in
// the order the scalar kernels of rounds 1-2 used (results are bit-identical to them).  gi arrives scaled.
__device__ __forceinline__ void adam_one(float& pi, float gi, float& mi, float& vi, bool decay, const pm_adam_cfg& c,
                                         float bc1, float bc2, float lr) {
    mi = c.b1 * mi + (1.f - c.b1) * gi;
    vi = c.b2 * vi + (1.f - c.b2) * gi * gi;
    float u = (mi / bc1) / (sqrtf(vi / bc2) + c.eps);
    if (decay) u += c.weight_decay * pi;
    pi = pi - lr * u;
}

// HBM-bound: p, g, m, v read, p, m, v written (+ g zeroed with c.zero_grad, + ema read and written).  One 16-byte vector
// per thread and tensor, two vectors of every tensor in flight per thread (8-10 loads outstanding), grid-stride over
// 2 x 256 vectors per workgroup.  CLIP_EMA adds train_pm_vdvae.py:131-154: clip_by_global_norm (gnorm_sq = sum g^2 of the
// already reduced gradient), Trainer(skip_nonfinite_updates=True, ema_rate); `count` is the optimizer's own update counter
// (it does not advance on a skipped step).  VEC = false: element-wise form for buffers that are not 16-byte aligned.
template <bool CLIP_EMA, bool VEC>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, float* __restrict__ ema, long long n,
                                                     long long n_decay, const int* __restrict__ count_dev,
                                                     const float* __restrict__ gnorm_sq, pm_adam_cfg c, float clip,
                                                     float ema_rate, int skip_nonfinite) {
    const bool zero_g = c.zero_grad != 0;
    float gscale = c.grad_scale;
    bool skip = false;
    float cs = 1.f;
    if (CLIP_EMA) {
        const float gn = sqrtf(gnorm_sq[0]) * c.grad_scale;
        skip = skip_nonfinite && !isfinite(gn);                      // every thread sees the same value
        cs = (clip > 0.f && !(gn < clip)) ? clip / gn : 1.f;         // optax: where(g_norm < max_norm, g, g/g_norm*max_norm)
    }
    if (skip && !zero_g) return;
    const int count = count_dev[0];
    const float t = (float)(count + 1);
    const float bc1 = 1.f - powf(c.b1, t);
    const float bc2 = 1.f - powf(c.b2, t);
    const float lr = pm_lr(c, count);
    const long long nv = VEC ? (n >> 2) : 0;
    if (VEC) {
        f32x4* p4 = reinterpret_cast<f32x4*>(p);
        f32x4* g4 = reinterpret_cast<f32x4*>(g);
        f32x4* m4 = reinterpret_cast<f32x4*>(m);
        f32x4* v4 = reinterpret_cast<f32x4*>(v);
        f32x4* e4 = reinterpret_cast<f32x4*>(ema);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const long long stride = (long long)gridDim.x * 512;
        for (long long i0 = (long long)blockIdx.x * 512 + threadIdx.x; i0 < nv; i0 += stride) {
            const long long i1 = i0 + 256;
            const bool two = i1 < nv;
            const long long j1 = two ? i1 : i0;
            if (skip) {                                              // a skipped step still hands back a zeroed gradient
                g4[i0] = zero;
                if (two) g4[i1] = zero;
                continue;
            }
            f32x4 ga = g4[i0], gb = g4[j1], ma = m4[i0], mb = m4[j1], va = v4[i0], vb = v4[j1], pa = p4[i0], pb = p4[j1];
            f32x4 ea = zero, eb = zero;
            if (CLIP_EMA && ema) { ea = e4[i0]; eb = e4[j1]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float p0 = pa[e], m0 = ma[e], v0 = va[e], p1 = pb[e], m1 = mb[e], v1 = vb[e];
                adam_one(p0, CLIP_EMA ? ga[e] * gscale * cs : ga[e] * gscale, m0, v0, 4 * i0 + e < n_decay, c, bc1, bc2, lr);
                adam_one(p1, CLIP_EMA ? gb[e] * gscale * cs : gb[e] * gscale, m1, v1, 4 * j1 + e < n_decay, c, bc1, bc2, lr);
                pa[e] = p0; ma[e] = m0; va[e] = v0; pb[e] = p1; mb[e] = m1; vb[e] = v1;
                if (CLIP_EMA) {
                    ea[e] = ema_rate * ea[e] + (1.f - ema_rate) * p0;
                    eb[e] = ema_rate * eb[e] + (1.f - ema_rate) * p1;
                }
            }
            m4[i0] = ma; v4[i0] = va; p4[i0] = pa;
            if (CLIP_EMA && ema) e4[i0] = ea;
            if (zero_g) g4[i0] = zero;
            if (two) {
                m4[i1] = mb; v4[i1] = vb; p4[i1] = pb;
                if (CLIP_EMA && ema) e4[i1] = eb;
                if (zero_g) g4[i1] = zero;
            }
        }
    }
    // element-wise: the n % 4 tail of the vector form (last workgroup), or everything
    const long long first = nv << 2;
    const long long estride = VEC ? n : (long long)gridDim.x * 256;
    long long i = first + (VEC ? (blockIdx.x == gridDim.x - 1 ? (long long)threadIdx.x : n) : (long long)blockIdx.x * 256 + threadIdx.x);
    for (; i < n; i += estride) {
        if (!skip) {
            float pi = p[i], mi = m[i], vi = v[i];
            adam_one(pi, CLIP_EMA ? g[i] * gscale * cs : g[i] * gscale, mi, vi, i < n_decay, c, bc1, bc2, lr);
            p[i] = pi; m[i] = mi; v[i] = vi;
            if (CLIP_EMA && ema) ema[i] = ema_rate * ema[i] + (1.f - ema_rate) * pi;
        }
        if (zero_g) g[i] = 0.f;
    }
}

template <bool CLIP_EMA>
int launch_adam(hipStream_t s, float* p, float* g, float* m, float* v, float* ema, long long n, long long n_decay,
                const int* count_dev, const float* gnorm_sq, const pm_adam_cfg& c, float clip, float ema_rate, int skip) {
    const bool vec = !((reinterpret_cast<size_t>(p) | reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(m) |
                        reinterpret_cast<size_t>(v) | reinterpret_cast<size_t>(ema)) & 15);
    long long blocks = vec ? ((n >> 2) + 511) / 512 : (n + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    if (vec)
        hipLaunchKernelGGL((adam_kernel<CLIP_EMA, true>), dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, ema, n, n_decay,
                           count_dev, gnorm_sq, c, clip, ema_rate, skip);
    else
        hipLaunchKernelGGL((adam_kernel<CLIP_EMA, false>), dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, ema, n, n_decay,
                           count_dev, gnorm_sq, c, clip, ema_rate, skip);
    return 0;
}

// pm_adam_step_jobs: the same update driven by a job table that covers the whole flat buffer (pm_reduce_job; nslots == 0: a
// run without partial sums).  A job's gradient is g[i] + sum over its slots, added in the fixed order of
// reduce_partials_kernel - the weight-gradient kernels' partial sums are read ONCE, by the optimizer, and the separate
// reduction launch with its write and re-read of g is gone.  Runs start on 16-byte boundaries and their buffers are padded to
// whole vectors (ParamStore pads every parameter to 4 floats, arenas to 4 floats per slot), so every access is a 16-byte one.
__global__ __launch_bounds__(256) void adam_jobs_kernel(const pm_reduce_job* __restrict__ jobs, float* __restrict__ p,
                                                          float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                          long long n_decay, const int* __restrict__ count_dev, pm_adam_cfg c) {
    __shared__ f32x4 red[3][64];
    const pm_reduce_job j = jobs[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* __restrict__ src = j.src;
    const int count = j.count;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int e;                                  // this thread's first element of the job (a vector of 4), valid when `mine`
    bool mine;
    if (j.nslots >= 8) {                    // <= 256 elements: the waves deal the slots, wave 0 finishes
        e = 4 * lane;
        const bool in = e < count;
        if (in) {
            const float* sp = src + e;
            int s = wave;
            for (; s + 28 < j.nslots; s += 32) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(sp + (size_t)s * j.stride);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(sp + (size_t)(s + 4) * j.stride);
                const f32x4 a2 = *reinterpret_cast<const f32x4*>(sp + (size_t)(s + 8) * j.stride);
                const f32x4 a3 = *reinterpret_cast<const f32x4*>(sp + (size_t)(s + 12) * j.stride);
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(sp + (size_t)(s + 16) * j.stride);
                const f32x4 a5 = *reinterpret_cast<const f32x4*>(sp + (size_t)(s + 20) * j.stride);
                const f32x4 a6 = *reinterpret_cast<const f32x4*>(sp + (size_t)(s + 24) * j.stride);
                const f32x4 a7 = *reinterpret_cast<const f32x4*>(sp + (size_t)(s + 28) * j.stride);
                acc += ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
            }
            for (; s < j.nslots; s += 4) acc += *reinterpret_cast<const f32x4*>(sp + (size_t)s * j.stride);
        }
        if (wave > 0) red[wave - 1][lane] = acc;
        __syncthreads();
        mine = in && wave == 0;
        if (mine) acc = (acc + red[0][lane]) + (red[1][lane] + red[2][lane]);
    } else {                                // <= 1024 elements: every thread walks the slots of its own vector
        e = 4 * threadIdx.x;
        mine = e < count;
        if (mine)
            for (int s = 0; s < j.nslots; ++s) acc += *reinterpret_cast<const f32x4*>(src + (size_t)s * j.stride + e);
    }
    if (!mine) return;
    const int cnt = count_dev[0];
    const float t = (float)(cnt + 1);
    const float bc1 = 1.f - powf(c.b1, t);
    const float bc2 = 1.f - powf(c.b2, t);
    const float lr = pm_lr(c, cnt);
    const long long i0 = j.g_off + e;
    f32x4* g4 = reinterpret_cast<f32x4*>(g + i0);
    f32x4* p4 = reinterpret_cast<f32x4*>(p + i0);
    f32x4* m4 = reinterpret_cast<f32x4*>(m + i0);
    f32x4* v4 = reinterpret_cast<f32x4*>(v + i0);
    f32x4 gv = *g4, pv = *p4, mv = *m4, vv = *v4;
    gv += acc;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float p0 = pv[k], m0 = mv[k], v0 = vv[k];
        adam_one(p0, gv[k] * c.grad_scale, m0, v0, i0 + k < n_decay, c, bc1, bc2, lr);
        pv[k] = p0; mv[k] = m0; vv[k] = v0;
    }
    *m4 = mv; *v4 = vv; *p4 = pv;
    if (c.zero_grad) *g4 = f32x4{0.f, 0.f, 0.f, 0.f};
}

__global__ void counter_increment_kernel(int* c) { c[0] += 1; }

// sum of squares with a non-finite guard: out[0] += sum x^2 (NaN / inf propagate into the sum)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long long n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) s = fmaf(x[i], x[i], s);
    s = pm_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

__global__ void counter_increment_if_finite_kernel(int* c, const float* gnorm_sq) {
    if (isfinite(gnorm_sq[0])) c[0] += 1;
}

// out[n] += sum_m x[m][n]: bias gradient of a transposed-conv layer (its weight gradient runs with
// the roles of x and dy swapped, so the column sums of dy are taken here).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, long long M,
                                                       int N, int rows_per_block) {
    extern __shared__ float sums[];
    for (int n = threadIdx.x; n < N; n += 256) sums[n] = 0.f;
    __syncthreads();
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    const long long e0 = r0 * N, e1 = r1 * N;
    if (N % 4 == 0 && (1024 % N == 0) && ((e0 & 3) == 0)) {
        // 16-byte loads; a thread's four columns never change (1024 % N == 0), so it keeps private sums
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
        long long q = (e0 >> 2) + threadIdx.x;
        const long long q1 = e1 >> 2;
        for (; q + 1792 < q1; q += 2048) {                // 8 independent loads in flight
            const f32x4 a = x4[q], b = x4[q + 256], c = x4[q + 512], d = x4[q + 768];
            const f32x4 e = x4[q + 1024], f = x4[q + 1280], g = x4[q + 1536], h = x4[q + 1792];
            acc += ((a + b) + (c + d)) + ((e + f) + (g + h));
        }
        for (; q + 768 < q1; q += 1024) {
            const f32x4 a = x4[q], b = x4[q + 256], c = x4[q + 512], d = x4[q + 768];
            acc += (a + b) + (c + d);
        }
        for (; q < q1; q += 256) acc += x4[q];
        const int c0 = (int)(((e0 >> 2) + threadIdx.x) * 4 % N);
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(&sums[c0 + j], acc[j]);
    } else if (256 % N == 0) {
        float acc = 0.f;
        for (long long e = e0 + threadIdx.x; e < e1; e += 256) acc += x[e];
        atomicAdd(&sums[(int)((e0 + threadIdx.x) % N)], acc);
    } else {
        for (long long e = e0 + threadIdx.x; e < e1; e += 256) atomicAdd(&sums[(int)(e % N)], x[e]);
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += 256) atomicAdd(out + n, sums[n]);
}

// The same column sums without atomics (pm_colsum_part): every workgroup STORES its sums into slot blockIdx.x of an arena
// (pm_reduce_partials adds the slots in a fixed order), and inside the workgroup the per-thread sums meet in LDS in thread
// order - run-to-run identical results.  N % 4 == 0, 1024 % N == 0 (a thread's four columns never change).
__device__ __forceinline__ void colsum_part_body(const float* __restrict__ x, float* __restrict__ part, long long part_stride,
                                                 long long M, int N, int rows_per_block, unsigned blk, f32x4* accs) {
    const long long r0 = (long long)blk * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    if (!(N % 4 == 0 && 1024 % N == 0)) {
        // any other width: 256 / N row lanes (one for N > 256) of N column threads each, every thread a fixed subset of the
        // rows; the row lanes meet in LDS in lane order
        float* red = reinterpret_cast<float*>(accs);          // 1024 floats
        const int lanes = N <= 256 ? 256 / N : 1;
        for (int n0 = 0; n0 < N; n0 += 256) {                 // column passes (one unless N > 256)
            const int n = n0 + (N <= 256 ? (int)threadIdx.x % N : (int)threadIdx.x);
            const int rl = N <= 256 ? (int)threadIdx.x / N : 0;
            float sacc = 0.f;
            if (rl < lanes && n < N)
                for (long long r = r0 + rl; r < r1; r += lanes) sacc += x[r * N + n];
            __syncthreads();
            if (rl < lanes && n < N) red[rl * (N <= 256 ? N : 256) + (n - n0)] = sacc;
            __syncthreads();
            if (rl == 0 && n < N) {
                float t = 0.f;
                for (int l = 0; l < lanes; ++l) t += red[l * (N <= 256 ? N : 256) + (n - n0)];
                part[(size_t)blk * part_stride + n] = t;
            }
        }
        return;
    }
    const long long e0 = r0 * N, e1 = r1 * N;                 // rows_per_block * N % 1024 == 0: e0 % 4 == 0
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    long long q = (e0 >> 2) + threadIdx.x;
    const long long q1 = e1 >> 2;
    for (; q + 1792 < q1; q += 2048) {                        // 8 independent loads in flight
        const f32x4 a = x4[q], b = x4[q + 256], c = x4[q + 512], d = x4[q + 768];
        const f32x4 e = x4[q + 1024], f = x4[q + 1280], g = x4[q + 1536], h = x4[q + 1792];
        acc += ((a + b) + (c + d)) + ((e + f) + (g + h));
    }
    for (; q + 768 < q1; q += 1024) {
        const f32x4 a = x4[q], b = x4[q + 256], c = x4[q + 512], d = x4[q + 768];
        acc += (a + b) + (c + d);
    }
    for (; q < q1; q += 256) acc += x4[q];
    accs[threadIdx.x] = acc;
    __syncthreads();
    // thread t holds columns 4 * ((e0 / 4 + t) % (N / 4)) ..: column quad qd is held by threads t0, t0 + N/4, t0 + 2N/4, ...
    const int nq = N >> 2;
    if ((int)threadIdx.x < N) {
        const int n = threadIdx.x, qd = n >> 2, j = n & 3;
        int t0 = (int)((qd - (e0 >> 2)) % nq);
        if (t0 < 0) t0 += nq;
        float sacc = 0.f;
        for (int t = t0; t < 256; t += nq) sacc += accs[t][j];
        part[(size_t)blk * part_stride + n] = sacc;
    }
}

__global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                            long long part_stride, long long M, int N, int rows_per_block) {
    __shared__ f32x4 accs[256];
    colsum_part_body(x, part, part_stride, M, N, rows_per_block, blockIdx.x, accs);
}

// Up to 8 column-sum problems in ONE launch (pm_colsum_part_multi): the bias gradients of a stack of transposed convolutions
// are only read by the optimizer, so the decoder's five 8 - 13 us launches (14.8 MB each: ramp-bound at 1.7 TB/s) become one
// launch of 74 MB.  Workgroup -> (job, slot) by the jobs' slot counts; every slot is computed exactly as pm_colsum_part does.
constexpr int CSM_MAX = 8;
struct ColsumMulti {
    const float* x[CSM_MAX];
    float* part[CSM_MAX];
    long long stride[CSM_MAX], M[CSM_MAX];
    int N[CSM_MAX], rows[CSM_MAX], first[CSM_MAX + 1];
    int njobs;
};
__global__ __launch_bounds__(256) void colsum_part_multi_kernel(ColsumMulti a) {
    __shared__ f32x4 accs[256];
    int j = 0;
#pragma unroll
    for (int q = 1; q < CSM_MAX; ++q)
        if (q < a.njobs && (int)blockIdx.x >= a.first[q]) j = q;
    colsum_part_body(a.x[j], a.part[j], a.stride[j], a.M[j], a.N[j], a.rows[j], blockIdx.x - (unsigned)a.first[j], accs);
}

// ---- sum of partial-sum slots (pm_reduce_partials) -------------------------------------------------------------------
// Weight-gradient kernels in partial-sum mode store one slot per m-split / persistent workgroup; here g[off + i] +=
// sum_s src[s * stride + i] in a FIXED order, so gradients are run-to-run identical and no f32 atomic reaches memory (atomics
// execute at the memory side at ~1.3 TB/s chip-wide and count twice in HBM traffic; plain stores and these reads stream).
// A job is a run of consecutive elements of one parameter.  nslots >= 8: at most 256 elements, the four waves deal the slots
// (wave w: w, w + 4, ...; a lane = one 16-byte vector, 8 loads in flight) and their four partial vectors meet in LDS;
// nslots < 8: at most 1024 elements, every thread walks all slots of its own vector.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const pm_reduce_job* __restrict__ jobs, float* __restrict__ g) {
    __shared__ f32x4 red[3][256];
    const pm_reduce_job j = jobs[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* __restrict__ src = j.src;
    const bool vec = ((reinterpret_cast<size_t>(src) | (size_t)(j.stride * 4) | (size_t)(j.g_off * 4)) & 15) == 0;
    if (j.nslots < 8) {
        // thread t: elements 4 t .. 4 t + 3 of the job, all slots in order
        const int e = 4 * threadIdx.x;
        if (e >= j.count) return;
        const int rem = j.count - e;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (vec && rem >= 4) {
            for (int s = 0; s < j.nslots; ++s) acc += *reinterpret_cast<const f32x4*>(src + (size_t)s * j.stride + e);
            f32x4* gp = reinterpret_cast<f32x4*>(g + j.g_off + e);
            *gp = *gp + acc;
        } else {
            for (int k = 0; k < 4 && k < rem; ++k) {
                float a = 0.f;
                for (int s = 0; s < j.nslots; ++s) a += src[(size_t)s * j.stride + e + k];
                g[j.g_off + e + k] += a;
            }
        }
        return;
    }
    // count <= 256: lane l of every wave covers elements 4 l .. 4 l + 3; wave w sums slots w, w + 4, ... (8 loads in flight)
    const int e = 4 * lane;
    const int rem = j.count - e;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (rem > 0) {
        if (vec && rem >= 4) {
            const float* sp = src + e;
            int s = wave;
            for (; s + 28 < j.nslots; s += 32) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(sp + (size_t)s * j.stride);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(sp + (size_t)(s + 4) * j.stride);
                const f32x4 a2 = *reinterpret_cast<const f32x4*>(sp + (size_t)(s + 8) * j.stride);
                const f32x4 a3 = *reinterpret_cast<const f32x4*>(sp + (size_t)(s + 12) * j.stride);
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(sp + (size_t)(s + 16) * j.stride);
                const f32x4 a5 = *reinterpret_cast<const f32x4*>(sp + (size_t)(s + 20) * j.stride);
                const f32x4 a6 = *reinterpret_cast<const f32x4*>(sp + (size_t)(s + 24) * j.stride);
                const f32x4 a7 = *reinterpret_cast<const f32x4*>(sp + (size_t)(s + 28) * j.stride);
                acc += ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
            }
            for (; s < j.nslots; s += 4) acc += *reinterpret_cast<const f32x4*>(sp + (size_t)s * j.stride);
        } else {
            for (int s = wave; s < j.nslots; s += 4)
                for (int k = 0; k < 4 && k < rem; ++k) acc[k] += src[(size_t)s * j.stride + e + k];
        }
    }
    if (wave > 0) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave == 0 && rem > 0) {
        acc = (acc + red[0][lane]) + (red[1][lane] + red[2][lane]);
        if (vec && rem >= 4) {
            f32x4* gp = reinterpret_cast<f32x4*>(g + j.g_off + e);
            *gp = *gp + acc;
        } else {
            for (int k = 0; k < 4 && k < rem; ++k) g[j.g_off + e + k] += acc[k];
        }
    }
}

// Philox4x32-10 counter-based generator + Box-Muller: eps ~ N(0,1) for posterior.sample
// (vae.py:124).  Bit parity with JAX's threefry streams is not a goal; parity tests
// pass eps explicitly.  key = (seed, stream_id), counter = (element/4, step).
__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0,
                                             unsigned k1) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    c1 = (unsigned)p1;
    c3 = (unsigned)p0;
    c0 = n0;
    c2 = n2;
}

__global__ __launch_bounds__(256) void normal_fill_kernel(float* __restrict__ out, long long n,
                                                            unsigned long long seed, const int* __restrict__ step_dev,
                                                            int stream_id) {
    const unsigned step = step_dev ? (unsigned)step_dev[0] : 0u;
    const long long quads = (n + 3) / 4;
    const long long stride = (long long)gridDim.x * 256;
    for (long long qd = (long long)blockIdx.x * 256 + threadIdx.x; qd < quads; qd += stride) {
        unsigned c0 = (unsigned)qd, c1 = (unsigned)(qd >> 32), c2 = step, c3 = (unsigned)stream_id;
        unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c0, c1, c2, c3, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        const float s = 2.3283064365386963e-10f;  // 2^-32
        float u0 = ((float)c0 + 0.5f) * s, u1 = ((float)c1 + 0.5f) * s;
        float u2 = ((float)c2 + 0.5f) * s, u3 = ((float)c3 + 0.5f) * s;
        u0 = fminf(fmaxf(u0, 1e-10f), 1.f);
        u2 = fminf(fmaxf(u2, 1e-10f), 1.f);
        float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
        float v[4];
        sincosf(6.283185307179586f * u1, &v[1], &v[0]);
        sincosf(6.283185307179586f * u3, &v[3], &v[2]);
        v[0] *= r0; v[1] *= r0; v[2] *= r1; v[3] *= r1;
        for (int e = 0; e < 4; ++e)
            if (qd * 4 + e < n) out[qd * 4 + e] = v[e];
    }
}

// hk.dropout keep mask, pre-scaled: out = (u >= rate) / (1 - rate), u ~ U[0,1) from the same Philox stream
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ out, long long n, float rate,
                                                             unsigned long long seed, const int* __restrict__ step_dev,
                                                             int stream_id) {
    const unsigned step = step_dev ? (unsigned)step_dev[0] : 0u;
    const long long quads = (n + 3) / 4;
    const long long stride = (long long)gridDim.x * 256;
    const float keep_scale = 1.f / (1.f - rate);
    for (long long qd = (long long)blockIdx.x * 256 + threadIdx.x; qd < quads; qd += stride) {
        unsigned c[4] = {(unsigned)qd, (unsigned)(qd >> 32), step, (unsigned)stream_id};
        unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c[0], c[1], c[2], c[3], k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        float m[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = ((float)c[e] * 2.3283064365386963e-10f >= rate) ? keep_scale : 0.f;
        if (qd * 4 + 3 < n && (reinterpret_cast<size_t>(out) & 15) == 0) {      // one 16-byte store per quad
            typedef float f4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<f4*>(out + qd * 4) = f4{m[0], m[1], m[2], m[3]};
        } else {
            for (int e = 0; e < 4; ++e)
                if (qd * 4 + e < n) out[qd * 4 + e] = m[e];
        }
    }
}

__global__ __launch_bounds__(256) void gumbel_fill_kernel(float* __restrict__ out, long long n, unsigned long long seed,
                                                            const int* __restrict__ step_dev, int stream_id) {
    const unsigned step = step_dev ? (unsigned)step_dev[0] : 0u;
    const long long quads = (n + 3) / 4;
    const long long stride = (long long)gridDim.x * 256;
    for (long long qd = (long long)blockIdx.x * 256 + threadIdx.x; qd < quads; qd += stride) {
        unsigned c[4] = {(unsigned)qd, (unsigned)(qd >> 32), step, (unsigned)stream_id};
        unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c[0], c[1], c[2], c[3], k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        for (int e = 0; e < 4; ++e)
            if (qd * 4 + e < n) {
                const float u = ((float)c[e] + 0.5f) * 2.3283064365386963e-10f;   // (0, 1)
                out[qd * 4 + e] = -logf(-logf(fminf(fmaxf(u, 1e-20f), 0.99999994f)));
            }
    }
}

__global__ __launch_bounds__(256) void axpy1_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] += x[i];
}

}  // namespace

extern "C" int pm_pmvae_loss(pm_stream_t stream, const float* rec, const float* kl, const float* mll, int B,
                             const pm_loss_cfg* cfg, const int* step_dev, float* out, float* g_rec, float* g_kl,
                             float* g_mll) {
    if (!cfg || B <= 0) return PM_EINVAL;
    if (rec ? (!kl || !mll || !out) : (kl || mll || out)) return PM_EINVAL;     // all four, or none of them
    if (cfg->beta_kind == 2 && cfg->period_or_steps < 2) return PM_EINVAL;
    hipLaunchKernelGGL(pmvae_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rec, kl, mll, B, *cfg, step_dev,
                       out, g_rec, g_kl, g_mll);
    return pm_check_launch("pm_pmvae_loss");
}

extern "C" int pm_adam_step(pm_stream_t stream, float* p, float* g, float* m, float* v, long long n,
                            long long n_decay, const int* count_dev, const pm_adam_cfg* cfg) {
    if (!p || !g || !m || !v || !count_dev || !cfg || n <= 0) return PM_EINVAL;
    launch_adam<false>((hipStream_t)stream, p, g, m, v, nullptr, n, n_decay, count_dev, nullptr, *cfg, 0.f, 0.f, 0);
    return pm_check_launch("pm_adam_step");
}

extern "C" int pm_adam_step_jobs(pm_stream_t stream, const pm_reduce_job* jobs_dev, int njobs, float* p, float* g, float* m,
                                 float* v, long long n_decay, const int* count_dev, const pm_adam_cfg* cfg) {
    if (!jobs_dev || njobs <= 0 || !p || !g || !m || !v || !count_dev || !cfg) return PM_EINVAL;
    if ((reinterpret_cast<size_t>(p) | reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(m) | reinterpret_cast<size_t>(v)) & 15)
        return PM_EINVAL;
    PM_KTAG("adam_jobs_kernel");
    hipLaunchKernelGGL(adam_jobs_kernel, dim3((unsigned)njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev, p, g, m, v, n_decay,
                       count_dev, *cfg);
    return pm_check_launch("pm_adam_step_jobs");
}

// The same sum in a FIXED order (pm_sumsq_det): every workgroup stores its partial sum, the workgroup whose ticket says it is
// the last one adds the partials in index order - the global gradient norm behind the VDVAE's clip / non-finite skip then has
// the same bits in every run.  scratch: >= 1026 words (1024 partials, the ticket counter, which the last workgroup resets).
__global__ __launch_bounds__(256) void sumsq_det_kernel(const float* __restrict__ x, long long n, float* __restrict__ out,
                                                          float* __restrict__ scratch) {
    __shared__ float red[4];
    __shared__ int last;
    float s = 0.f;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) s += x[i] * x[i];
    s = pm_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    unsigned* ticket = reinterpret_cast<unsigned*>(scratch + 1024);
    if (threadIdx.x == 0) {
        __hip_atomic_store(scratch + blockIdx.x, (red[0] + red[1]) + (red[2] + red[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    float t = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 256)        // thread t: partials t, t + 256, ... (<= 4 of them)
        t += __hip_atomic_load(scratch + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __shared__ float fin[256];
    fin[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int i = 0; i < 256; ++i) tot += fin[i];
        out[0] = tot;
        *ticket = 0u;
    }
}

extern "C" int pm_sumsq_det(pm_stream_t stream, const float* x, long long n, float* out, float* scratch) {
    if (!x || !out || !scratch || n <= 0) return PM_EINVAL;
    long long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sumsq_det_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, out, scratch);
    return pm_check_launch("pm_sumsq_det");
}

extern "C" int pm_sumsq(pm_stream_t stream, const float* x, long long n, float* out) {
    if (!x || !out || n <= 0) return PM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (pm_zero_async(s, out, sizeof(float))) return PM_ELAUNCH;
    long long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n, out);
    return pm_check_launch("pm_sumsq");
}

extern "C" int pm_adam_step_clip_ema(pm_stream_t stream, float* p, float* g, float* m, float* v, float* ema,
                                     long long n, long long n_decay, int* count_dev, const float* gnorm_sq,
                                     const pm_adam_cfg* cfg, float clip, float ema_rate, int skip_nonfinite) {
    if (!p || !g || !m || !v || !count_dev || !gnorm_sq || !cfg || n <= 0) return PM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    launch_adam<true>(s, p, g, m, v, ema, n, n_decay, count_dev, gnorm_sq, *cfg, clip, ema_rate, skip_nonfinite);
    if (skip_nonfinite) hipLaunchKernelGGL(counter_increment_if_finite_kernel, dim3(1), dim3(1), 0, s, count_dev, gnorm_sq);
    else hipLaunchKernelGGL(counter_increment_kernel, dim3(1), dim3(1), 0, s, count_dev);
    return pm_check_launch("pm_adam_step_clip_ema");
}

// Device-side time stamp: dst[0] = the GPU's constant-rate real-time counter (100 MHz) when this one-thread kernel runs.
// tools/stamp_timeline.py puts one behind every launch of a replayed step: the stream order makes it the end time of the
// kernel in front of it - a timeline of the CONCURRENT step without a profiler attached (rocprofv3 slows the host's launches
// enough to change how the two streams interleave).
__global__ void stamp_kernel(unsigned long long* dst) { dst[0] = wall_clock64(); }

extern "C" int pm_stamp(pm_stream_t stream, unsigned long long* dst) {
    if (!dst) return PM_EINVAL;
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dst);
    return pm_check_launch("pm_stamp");
}

// snap = count; count += 1: the optimizer of step k reads `snap` on another stream while the head kernels of step k + 1
// (noise, schedules) already read the incremented counter
__global__ void counter_snapshot_increment_kernel(int* c, int* snap) {
    const int v = c[0];
    snap[0] = v;
    c[0] = v + 1;
}

extern "C" int pm_counter_snapshot_increment(pm_stream_t stream, int* count_dev, int* snapshot_dev) {
    if (!count_dev || !snapshot_dev) return PM_EINVAL;
    hipLaunchKernelGGL(counter_snapshot_increment_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, count_dev, snapshot_dev);
    return pm_check_launch("pm_counter_snapshot_increment");
}

extern "C" int pm_counter_increment(pm_stream_t stream, int* count_dev) {
    if (!count_dev) return PM_EINVAL;
    hipLaunchKernelGGL(counter_increment_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, count_dev);
    return pm_check_launch("pm_counter_increment");
}

extern "C" int pm_normal_fill(pm_stream_t stream, float* out, long long n, unsigned long long seed,
                              const int* step_dev, int stream_id) {
    if (!out || n <= 0) return PM_EINVAL;
    long long blocks = ((n + 3) / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(normal_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, n, seed,
                       step_dev, stream_id);
    return pm_check_launch("pm_normal_fill");
}

namespace {
__global__ __launch_bounds__(256) void zero_kernel(unsigned* __restrict__ p, size_t nwords) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < nwords; i += stride) {
        if (i + 4 <= nwords && (reinterpret_cast<uintptr_t>(p + i) & 15) == 0) {
            *reinterpret_cast<uint4*>(p + i) = uint4{0u, 0u, 0u, 0u};
        } else {
            for (size_t j = i; j < nwords && j < i + 4; ++j) p[j] = 0u;
        }
    }
}
}  // namespace

int pm_zero_async(hipStream_t stream, void* ptr, size_t nbytes) {
    if (!ptr || nbytes == 0 || (nbytes & 3) || (reinterpret_cast<uintptr_t>(ptr) & 3)) return PM_EINVAL;
    const size_t nwords = nbytes / 4;
    size_t blocks = (nwords / 4 + 255) / 256 + 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<unsigned*>(ptr),
                       nwords);
    return pm_check_launch("pm_zero_async");
}

extern "C" int pm_gumbel_fill(pm_stream_t stream, float* out, long long n, unsigned long long seed, const int* step_dev,
                              int stream_id) {
    if (!out || n <= 0) return PM_EINVAL;
    long long blocks = ((n + 3) / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gumbel_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, n, seed,
                       step_dev, stream_id);
    return pm_check_launch("pm_gumbel_fill");
}

extern "C" int pm_dropout_mask(pm_stream_t stream, float* out, long long n, float rate, unsigned long long seed,
                               const int* step_dev, int stream_id) {
    if (!out || n <= 0 || !(rate >= 0.f) || !(rate < 1.f)) return PM_EINVAL;
    long long blocks = ((n + 3) / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, n, rate, seed,
                       step_dev, stream_id);
    return pm_check_launch("pm_dropout_mask");
}

extern "C" int pm_fill_zero(pm_stream_t stream, void* ptr, long long nbytes) {
    if (!ptr || nbytes <= 0) return PM_EINVAL;
    return pm_zero_async((hipStream_t)stream, ptr, (size_t)nbytes);
}

extern "C" int pm_colsum(pm_stream_t stream, const float* x, float* out, long long M, int N) {
    if (!x || !out || M <= 0 || N <= 0 || N > 8192) return PM_EINVAL;
    // every workgroup ends with N same-address atomics: with ~800 workgroups on 32 columns the adds serialise in the L2 atomic
    // unit (19 us for 25.7 MB = 1.3 TB/s); ~256 - 512 workgroups of >= 64 KB with 8 loads in flight per thread instead
    int rows = 256;
    while (rows < 8192 && ((M + rows - 1) / rows > 512 || (long long)rows * N * 4 < 65536)) rows *= 2;
    long long blocks = (M + rows - 1) / rows;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)blocks), dim3(256), (size_t)N * sizeof(float), (hipStream_t)stream,
                       x, out, M, N, rows);
    return pm_check_launch("pm_colsum");
}

static int colsum_rows(long long M, int N) {         // rows per workgroup: ~256 - 512 workgroups of >= 64 KB (see pm_colsum)
    int rows = 256;
    while (rows < 8192 && ((M + rows - 1) / rows > 512 || (long long)rows * N * 4 < 65536)) rows *= 2;
    return rows;
}

extern "C" int pm_colsum_part_slots(long long M, int N, int* nslots) {
    if (!nslots || M <= 0 || N <= 0 || N > 8192) return PM_EINVAL;
    const int rows = colsum_rows(M, N);
    *nslots = (int)((M + rows - 1) / rows);
    return PM_OK;
}

extern "C" int pm_colsum_part(pm_stream_t stream, const float* x, long long M, int N, float* part, long long part_stride,
                              int nslots) {
    int need = 0;
    if (!x || !part || part_stride < N || pm_colsum_part_slots(M, N, &need) != PM_OK || need != nslots) return PM_EINVAL;
    if ((reinterpret_cast<size_t>(x) & 15) && N % 4 == 0 && 1024 % N == 0) return PM_EINVAL;     // the vector form's loads
    PM_KTAG("colsum_part_kernel");
    hipLaunchKernelGGL(colsum_part_kernel, dim3((unsigned)need), dim3(256), 0, (hipStream_t)stream, x, part, part_stride, M, N,
                       colsum_rows(M, N));
    return pm_check_launch("pm_colsum_part");
}

extern "C" int pm_colsum_part_multi(pm_stream_t stream, const pm_colsum_job* jobs, int njobs) {
    if (!jobs || njobs <= 0 || njobs > CSM_MAX) return PM_EINVAL;
    ColsumMulti a;
    int total = 0;
    for (int j = 0; j < CSM_MAX; ++j) {
        const pm_colsum_job& q = jobs[j < njobs ? j : 0];
        int need = 0;
        if (!q.x || !q.part || q.part_stride < q.N || pm_colsum_part_slots(q.M, q.N, &need) != PM_OK || need != q.nslots)
            return PM_EINVAL;
        if ((reinterpret_cast<size_t>(q.x) & 15) && q.N % 4 == 0 && 1024 % q.N == 0) return PM_EINVAL;
        a.x[j] = q.x; a.part[j] = q.part; a.stride[j] = q.part_stride; a.M[j] = q.M; a.N[j] = q.N;
        a.rows[j] = colsum_rows(q.M, q.N);
        a.first[j] = total;
        if (j < njobs) total += need;
    }
    a.first[CSM_MAX] = total;
    a.njobs = njobs;
    PM_KTAG("colsum_part_multi_kernel");
    hipLaunchKernelGGL(colsum_part_multi_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, a);
    return pm_check_launch("pm_colsum_part_multi");
}

extern "C" int pm_reduce_partials(pm_stream_t stream, const pm_reduce_job* jobs_dev, int njobs, float* g) {
    if (!jobs_dev || !g || njobs <= 0) return PM_EINVAL;
    PM_KTAG("reduce_partials_kernel");
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev, g);
    return pm_check_launch("pm_reduce_partials");
}

extern "C" int pm_axpy1(pm_stream_t stream, const float* x, float* y, long long n) {
    if (!x || !y || n <= 0) return PM_EINVAL;
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(axpy1_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return pm_check_launch("pm_axpy1");
}
