// Vector quantisation with EMA codebook updates: hk.nets.VectorQuantizerEMA as the reference's
// VQVAE uses it (posterior_matching/models/vqvae.py:66-72,80; third party dm-haiku 0.0.5, SURVEY.md
// Appendix A5).  The distance GEMM z @ emb runs in the gather-GEMM engine (pm_conv.hip, f32: the
// argmin must see exact f32 products); everything here is HBM-bound row work:
//   select : dist = |z|^2 - 2 z.e + |e|^2 -> first argmin -> lookup, commitment gradient, counts, dw
//   ema    : the two zero-debiased EMAs, Laplace smoothing and the new codebook
//   loss   : reconstruction / commitment loss, perplexity
#include "pm_common.h"

namespace {

// e2[k] = sum_d emb[d,k]^2 ; zero counts[K] and dw[D*K]
__global__ __launch_bounds__(256) void vq_prep_kernel(const float* __restrict__ emb, float* __restrict__ e2,
                                                       float* __restrict__ counts, float* __restrict__ dw, int D, int K) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < K) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) {
            float e = emb[(size_t)d * K + i];
            s = fmaf(e, e, s);
        }
        e2[i] = s;
        counts[i] = 0.f;
    }
    if (dw)
        for (int j = i; j < D * K; j += gridDim.x * 256) dw[j] = 0.f;
}

// A 16-lane group per row of z (4 rows in flight per wave), VQ_ITERS rows per group.  The EMA
// statistics (dw[:, idx] += z, counts[idx] += 1) are accumulated in an LDS copy of the [D, K] table
// first: early in training a handful of codes takes every row (perplexity 2-5), and 800 k global
// atomics onto ~300 addresses serialise in L2 (measured: 135 us for N = 12544).  Only the columns a
// workgroup touched are flushed to global memory.
constexpr int VQ_WAVES = 8;
constexpr int VQ_ITERS = 2;   // rows per 16-lane group -> 8 waves x 4 groups x 2 = 64 rows per workgroup
constexpr int VQ_ROWS_PER_WG = VQ_WAVES * 4 * VQ_ITERS;

__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <bool LDS_STATS>
__global__ __launch_bounds__(64 * VQ_WAVES) void vq_select_kernel(
    const float* __restrict__ z, const float* __restrict__ emb, const float* __restrict__ dots,
    const float* __restrict__ e2, int* __restrict__ idx_out, float* __restrict__ quant,
    float* __restrict__ commit_grad, float* __restrict__ sqerr, float* __restrict__ counts, float* __restrict__ dw,
    int N, int D, int K, float commit_coef) {
    extern __shared__ float vq_smem[];
    float* dwl = vq_smem;                                      // [D][K]   (LDS_STATS && dw)
    float* cntl = vq_smem + (LDS_STATS && dw ? D * K : 0);     // [K]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int sl = lane & 15;                                  // lane within the row's group
    if (LDS_STATS) {
        if (dw)
            for (int j = threadIdx.x; j < D * K; j += 64 * VQ_WAVES) dwl[j] = 0.f;
        for (int k = threadIdx.x; k < K; k += 64 * VQ_WAVES) cntl[k] = 0.f;
        __syncthreads();
    }
    const int row0 = blockIdx.x * VQ_ROWS_PER_WG + (wave * 4 + (lane >> 4)) * VQ_ITERS;
    for (int rr = 0; rr < VQ_ITERS; ++rr) {
        const int row = row0 + rr;
        const bool live = row < N;                             // uniform within the 16-lane group
        const int rrow = live ? row : 0;                       // dead groups redo row 0 and store nothing
        const float* zr = z + (size_t)rrow * D;
        float x2 = 0.f;
        for (int d = sl; d < D; d += 16) x2 = fmaf(zr[d], zr[d], x2);
        x2 = group16_sum(x2);
        // nearest code; ties -> lowest index (argmax(-dist) returns the first maximum)
        float best = INFINITY;
        int bi = 0x7fffffff;
        const float* dr = dots + (size_t)rrow * K;
        for (int k = sl; k < K; k += 16) {
            const float dist = x2 - 2.f * dr[k] + e2[k];
            if (dist < best) { best = dist; bi = k; }          // k ascends within a lane: strict < keeps the first
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (bi >= K) bi = 0;                                   // all-NaN row: argmax of NaNs is index 0
        float se = 0.f;
        for (int d = sl; d < D; d += 16) {
            const float q = emb[(size_t)d * K + bi];
            const float zv = zr[d];
            const float diff = q - zv;
            se = fmaf(diff, diff, se);
            if (live) {
                quant[(size_t)row * D + d] = q;
                if (commit_grad) commit_grad[(size_t)row * D + d] = -commit_coef * diff;
                if (dw) {
                    if (LDS_STATS) atomicAdd(dwl + d * K + bi, zv);
                    else atomicAdd(dw + (size_t)d * K + bi, zv);
                }
            }
        }
        se = group16_sum(se);
        if (sl == 0 && live) {
            idx_out[row] = bi;
            sqerr[row] = se;
            if (LDS_STATS) atomicAdd(cntl + bi, 1.f);
            else atomicAdd(counts + bi, 1.f);
        }
    }
    if (LDS_STATS) {
        __syncthreads();
        for (int k = threadIdx.x; k < K; k += 64 * VQ_WAVES)
            if (cntl[k] > 0.f) atomicAdd(counts + k, cntl[k]);
        if (dw)
            for (int j = threadIdx.x; j < D * K; j += 64 * VQ_WAVES)
                if (cntl[j % K] > 0.f) atomicAdd(dw + j, dwl[j]);
    }
}

// dw[d, k] = sum over the rows assigned to code k of z[row, d], WITHOUT atomics.  Workgroup (k, s) owns code k on row segment s
// (<= VQ_SEG rows): its 1024 threads read the segment's indices once (coalesced), compact the matching rows IN ROW ORDER into an
// LDS list (wave ballots + a prefix over the 16 waves), then RL row lanes x D dimension lanes add the listed rows - lane l takes
// list entries l, l + RL, ... in order, the RL lane sums are added in lane order.  Segment sums go to part[s][d][k] (plain
// stores) and vq_dw_sum_kernel adds the segments in order.  The result is an exact function of (z, idx) - the same bits on
// every run - where the float atomics of vq_select_kernel add in arrival order.  Segments keep the work balanced when a
// handful of codes takes every row (perplexity 2 - 5 early in training): the first form of this kernel, one workgroup per
// code scanning all N rows, took 160 us for N = 12544 against ~15 us for the atomics.
#ifndef PM_VQ_SEG
#define PM_VQ_SEG 2048
#endif
constexpr int VQ_SEG = PM_VQ_SEG;
constexpr int VQ_LPT = (VQ_SEG + 1023) / 1024;                  // index loads per thread

// KMAJOR: segment sums laid out [segment][K][D] (an embedding table's gradient, pm_embed_bwd_sorted) instead of [segment][D][K]
template <bool KMAJOR>
__global__ __launch_bounds__(1024) void vq_dw_exact_kernel(const float* __restrict__ z, const int* __restrict__ idx,
                                                            float* __restrict__ out, int N, int D, int K, int RL, int seg) {
    __shared__ int list[VQ_SEG];
    __shared__ int wcount[VQ_LPT][16];
    __shared__ float part[1024];
    const int k = blockIdx.x, sgm = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = sgm * seg, r1 = min(N, r0 + seg);
    // the segment's indices: VQ_LPT per thread, all loads in flight
    bool mt[VQ_LPT];
    unsigned long long bt[VQ_LPT];
#pragma unroll
    for (int i = 0; i < VQ_LPT; ++i) {
        const int r = r0 + i * 1024 + (int)threadIdx.x;
        mt[i] = r < r1 && idx[r] == k;
    }
#pragma unroll
    for (int i = 0; i < VQ_LPT; ++i) {
        bt[i] = __ballot(mt[i]);
        if (lane == 0) wcount[i][wave] = __popcll(bt[i]);
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    int nm = 0;                                                  // uniform over the workgroup
#pragma unroll
    for (int i = 0; i < VQ_LPT; ++i) {
        int off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int c = wcount[i][w];
            if (w < wave) off += c;
            tot += c;
        }
        if (mt[i]) list[nm + off + __popcll(bt[i] & below)] = r0 + i * 1024 + (int)threadIdx.x;
        nm += tot;
    }
    __syncthreads();
    const int l = threadIdx.x / D, d = threadIdx.x - l * D;
    float s = 0.f;
    if (l < RL) {
        int m = l;
        for (; m + 3 * RL < nm; m += 4 * RL) {                  // four independent row loads in flight, added in list order
            const float v0 = z[(size_t)list[m] * D + d], v1 = z[(size_t)list[m + RL] * D + d];
            const float v2 = z[(size_t)list[m + 2 * RL] * D + d], v3 = z[(size_t)list[m + 3 * RL] * D + d];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; m < nm; m += RL) s += z[(size_t)list[m] * D + d];
        part[threadIdx.x] = s;
    }
    __syncthreads();
    if (l == 0) {
        float t = 0.f;
        for (int j = 0; j < RL; ++j) t += part[j * D + d];
        if (KMAJOR) out[((size_t)sgm * K + k) * D + d] = t;
        else out[((size_t)sgm * D + d) * K + k] = t;
    }
}

template <bool ACCUMULATE>
__global__ __launch_bounds__(256) void vq_dw_sum_kernel(const float* __restrict__ part, float* __restrict__ dw, int DK, int S) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= DK) return;
    float t = 0.f;
    for (int s = 0; s < S; ++s) t += part[(size_t)s * DK + j];
    dw[j] = ACCUMULATE ? dw[j] + t : t;
}

__global__ __launch_bounds__(256) void vq_lookup_kernel(const int* __restrict__ idx, const float* __restrict__ emb,
                                                         float* __restrict__ quant, long long total, int D, int K) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long n = i / D;
    const int d = (int)(i - n * D);
    quant[i] = emb[(size_t)d * K + idx[n]];
}

__device__ __forceinline__ float block_sum_1024(float v, float* red) {
    v = pm_wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < 16; ++w) s += red[w];
    return s;
}

// single workgroup: K and D*K are a few thousand elements
__global__ __launch_bounds__(1024) void vq_ema_kernel(const float* __restrict__ counts, const float* __restrict__ dw,
                                                       float* __restrict__ cs_hidden, float* __restrict__ cs_avg,
                                                       float* __restrict__ dw_hidden, float* __restrict__ dw_avg,
                                                       float* __restrict__ emb, int* __restrict__ counter, int D, int K,
                                                       float decay, float epsilon) {
    __shared__ float red[16];
    const int t = counter[0] + 1;
    const float debias = 1.f / (1.f - powf(decay, (float)t));
    float part = 0.f;
    for (int k = threadIdx.x; k < K; k += 1024) {
        const float h = cs_hidden[k] * decay + counts[k] * (1.f - decay);
        cs_hidden[k] = h;
        const float a = h * debias;
        cs_avg[k] = a;
        part += a;
    }
    const float n = block_sum_1024(part, red);
    for (int j = threadIdx.x; j < D * K; j += 1024) {
        const int k = j % K;
        const float h = dw_hidden[j] * decay + dw[j] * (1.f - decay);
        dw_hidden[j] = h;
        const float a = h * debias;
        dw_avg[j] = a;
        const float cs = (cs_avg[k] + epsilon) / (n + (float)K * epsilon) * n;   // cs_avg: visible after the block sum's barriers
        emb[j] = a / cs;
    }
    if (threadIdx.x == 0) counter[0] = t;
}

__global__ __launch_bounds__(1024) void vqvae_loss_kernel(const float* __restrict__ ll, const float* __restrict__ sqerr,
                                                           const float* __restrict__ counts, int B, int N, int D, int K,
                                                           float commitment_cost, float grad_scale,
                                                           float* __restrict__ out, float* __restrict__ g_ll) {
    __shared__ float red[16];
    float a = 0.f, b = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < B; i += 1024) {
        a += ll[i];
        if (g_ll) g_ll[i] = -grad_scale;
    }
    for (int i = threadIdx.x; i < N; i += 1024) b += sqerr[i];
    for (int k = threadIdx.x; k < K; k += 1024) {
        const float p = counts[k] / (float)N;
        c += p * logf(p + 1e-10f);
    }
    a = block_sum_1024(a, red);
    b = block_sum_1024(b, red);
    c = block_sum_1024(c, red);
    if (threadIdx.x == 0) {
        const float rec = -a / (float)B;
        const float vq = commitment_cost * b / ((float)N * (float)D);
        out[0] = rec + vq;
        out[1] = rec;
        out[2] = vq;
        out[3] = expf(-c);
    }
}

}  // namespace

extern "C" int pm_vq_select(pm_stream_t stream, const float* z, const float* emb, const float* dots, float* e2,
                            int* idx, float* quant, float* commit_grad, float* sqerr, float* counts, float* dw, int N,
                            int D, int K, float commit_coef) {
    if (!z || !emb || !dots || !e2 || !idx || !quant || !sqerr || !counts || N <= 0 || D <= 0 || K <= 0)
        return PM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int pb = dw ? (D * K + 255) / 256 : (K + 255) / 256;
    hipLaunchKernelGGL(vq_prep_kernel, dim3(pb), dim3(256), 0, s, emb, e2, counts, dw, D, K);
    const int rows_per_wg = VQ_ROWS_PER_WG;
    const dim3 grid((N + rows_per_wg - 1) / rows_per_wg), block(64 * VQ_WAVES);
    const size_t lds = ((dw ? (size_t)D * K : 0) + K) * sizeof(float);
    if (lds <= 150 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {   // > 64 KB of dynamic LDS needs the opt-in (K = 512, D = 64 is 130 KB)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vq_select_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            attr_set = true;
        }
        PM_KTAG("vq_select_kernel<true>");
        hipLaunchKernelGGL(vq_select_kernel<true>, grid, block, lds, s, z, emb, dots, e2, idx, quant, commit_grad, sqerr,
                           counts, dw, N, D, K, commit_coef);
    } else {
        PM_KTAG("vq_select_kernel<false>");
        hipLaunchKernelGGL(vq_select_kernel<false>, grid, block, 0, s, z, emb, dots, e2, idx, quant, commit_grad, sqerr,
                           counts, dw, N, D, K, commit_coef);
    }
    return pm_check_launch("pm_vq_select");
}

static int vq_dw_segments(int N) { return (N + VQ_SEG - 1) / VQ_SEG; }

extern "C" int pm_vq_dw_exact_floats(int N, int D, int K, long long* floats) {
    if (!floats || N <= 0 || D <= 0 || D > 1024 || K <= 0) return PM_EINVAL;
    const int S = vq_dw_segments(N);
    *floats = S > 1 ? (long long)S * D * K : 0;
    return PM_OK;
}

extern "C" int pm_vq_dw_exact(pm_stream_t stream, const float* z, const int* idx, float* dw, int N, int D, int K,
                              float* scratch, long long scratch_floats) {
    if (!z || !idx || !dw || N <= 0 || D <= 0 || D > 1024 || K <= 0) return PM_EINVAL;
    const int S = vq_dw_segments(N);
    if (S > 65535 || (S > 1 && (!scratch || scratch_floats < (long long)S * D * K))) return PM_EINVAL;
    const int RL = 1024 / D;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(vq_dw_exact_kernel<false>, dim3(K, S), dim3(1024), 0, s, z, idx, S > 1 ? scratch : dw, N, D, K, RL, VQ_SEG);
    if (S > 1)
        hipLaunchKernelGGL(vq_dw_sum_kernel<false>, dim3((D * K + 255) / 256), dim3(256), 0, s, scratch, dw, D * K, S);
    return pm_check_launch("pm_vq_dw_exact");
}

// dtable[idx[r], :] += dout[r, :] (the gradient of an embedding lookup: PixelCNN input embedding, reference pixel_cnn.py:372-380
// under jax.grad) in a FIXED order, two launches: the segment kernel above with [K][F] sums, then dtable += the segments in
// order.  Replaces the four launches of pm_embed_bwd_exact (64-bit fixed-point atomics: 175 us at the CelebA PixelCNN's 4096 rows
// x 128 features x 512 codes, at the very end of the backward pass) where F <= 1024.  scratch: >= ceil(rows / 2048) * K * F floats
// (ceil(rows / 512) * K * F lets it use the shortest segments).
extern "C" int pm_embed_bwd_sorted(pm_stream_t stream, const int* idx, const float* dout, float* dtable, long long rows, int F,
                                   int K, float* scratch, long long scratch_floats) {
    if (!idx || !dout || !dtable || !scratch || rows <= 0 || rows > 0x7fffffffLL || F <= 0 || F > 1024 || K <= 0) return PM_EINVAL;
    // Shorter segments bound the work of a code that takes most rows (a random-init codebook: perplexity 2 - 5; the workgroup
    // of such a code adds seg / RL rows one after the other - 84 us at 4096 rows x 512 codes with 2048-row segments), as long as
    // the grid stays <= 4096 workgroups; never more segments than the caller's scratch holds.
    const int N = (int)rows;
    int seg = VQ_SEG;
    for (int cand : {512, 1024}) {
        const long long S2 = (N + cand - 1) / cand;
        if ((long long)K * S2 <= 4096 && scratch_floats >= S2 * F * K) { seg = cand; break; }
    }
    const int S = (N + seg - 1) / seg;
    if (S > 65535 || scratch_floats < (long long)S * F * K) return PM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    PM_KTAG("vq_dw_exact_kernel<true>");
    hipLaunchKernelGGL(vq_dw_exact_kernel<true>, dim3(K, S), dim3(1024), 0, s, dout, idx, scratch, N, F, K, 1024 / F, seg);
    hipLaunchKernelGGL(vq_dw_sum_kernel<true>, dim3((F * K + 255) / 256), dim3(256), 0, s, scratch, dtable, F * K, S);
    return pm_check_launch("pm_embed_bwd_sorted");
}

extern "C" int pm_vq_ema_update(pm_stream_t stream, const float* counts, const float* dw, float* cs_hidden,
                                float* cs_avg, float* dw_hidden, float* dw_avg, float* emb, int* counter, int D, int K,
                                float decay, float epsilon) {
    if (!counts || !dw || !cs_hidden || !cs_avg || !dw_hidden || !dw_avg || !emb || !counter || D <= 0 || K <= 0)
        return PM_EINVAL;
    hipLaunchKernelGGL(vq_ema_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, counts, dw, cs_hidden, cs_avg,
                       dw_hidden, dw_avg, emb, counter, D, K, decay, epsilon);
    return pm_check_launch("pm_vq_ema_update");
}

extern "C" int pm_vq_lookup(pm_stream_t stream, const int* idx, const float* emb, float* quant, int N, int D, int K) {
    if (!idx || !emb || !quant || N <= 0 || D <= 0 || K <= 0) return PM_EINVAL;
    const long long total = (long long)N * D;
    hipLaunchKernelGGL(vq_lookup_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx,
                       emb, quant, total, D, K);
    return pm_check_launch("pm_vq_lookup");
}

extern "C" int pm_vqvae_loss(pm_stream_t stream, const float* ll, const float* sqerr, const float* counts, int B, int N,
                             int D, int K, float commitment_cost, float grad_scale, float* out, float* g_ll) {
    if (!ll || !sqerr || !counts || !out || B <= 0 || N <= 0 || D <= 0 || K <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(vqvae_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ll, sqerr, counts, B, N, D, K,
                       commitment_cost, grad_scale, out, g_ll);
    return pm_check_launch("pm_vqvae_loss");
}
