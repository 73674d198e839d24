// Two dense layers of a ResidualMLP block in ONE launch (reference networks.py:111-135), forward or data gradient:
//   forward :  u  = relu(h) W1 + b1                 hn    = relu(u) W2 + b2 + h
//   backward:  du = (dh W2^T) * relu'(u)            dprev = (du W1^T) * relu'(h) + dh
// i.e.  out1 = epi1(act_in(X) B1),  out2 = epi2(act_mid(out1) B2)  with both results stored (u / du are needed later).
// A workgroup owns 64 rows: X is staged once as hi / lo bf16 in LDS, every wave owns 64 of the H = 256 output columns
// (two 32-row x two 32-column accumulator tiles), the B fragments come straight from the pre-split, K-contiguous
// weights in L2 (16 bytes per lane, prefetched one k-step ahead) and out1 goes back into the same LDS buffer as the
// second product's A operand.  Same bf16x3 arithmetic as pm_gather_gemm_bf16 (three bf16 MFMA products, f32 accumulate).
// Two launches of 256 / 512 workgroups and a round trip through HBM become one launch of R / 64 workgroups.
#include "pm_common.h"
#include <type_traits>

namespace {

typedef __bf16 m_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 m_bf16x2 __attribute__((ext_vector_type(2)));
typedef float m_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned m_u32x2 __attribute__((ext_vector_type(2)));

constexpr int H = 256;          // hidden width (columns of every operand)
constexpr int LDA = H + 8;      // bf16 per LDS row: 528 bytes = 33 x 16 (conflict-free 16-byte reads down a column)
#ifndef PM_MLP_ROWS
#define PM_MLP_ROWS 64
#endif
constexpr int ROWS = PM_MLP_ROWS;
constexpr int MS = ROWS / 32;   // 32-row accumulator tiles per wave

__device__ __forceinline__ void m_split4(const f32x4& x, m_u32x2& hi, m_u32x2& lo) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float a0 = x[2 * j], a1 = x[2 * j + 1];
        const unsigned hh = __builtin_bit_cast(unsigned, __builtin_convertvector(m_f32x2{a0, a1}, m_bf16x2));
        const float f0 = __builtin_bit_cast(float, hh << 16);
        const float f1 = __builtin_bit_cast(float, hh & 0xffff0000u);
        hi[j] = hh;
        lo[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(m_f32x2{a0 - f0, a1 - f1}, m_bf16x2));
    }
}

struct PairArgs {
    const float* x;          // [R, H]  first operand (also the residual of the second product)
    const __bf16* w1;        // split weights of the first product: [2 planes][H/32][H][32]
    const __bf16* w2;
    const float* b1;         // may be NULL
    const float* b2;
    const float* aux1;       // act'(aux1) multiplies out1 (may be NULL)
    const float* aux2;
    float* out1;
    float* out2;
    int R;
    int in_act, mid_act, aux_act1, aux_act2;
    float slope;
};

__global__ __launch_bounds__(256) void mlp_pair_bf16_kernel(PairArgs p) {
    __shared__ __attribute__((aligned(16))) __bf16 Ah[ROWS * LDA];
    __shared__ __attribute__((aligned(16))) __bf16 Al[ROWS * LDA];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int r0 = blockIdx.x * ROWS;
    constexpr long long PLANE = (long long)H * H;

    // stage act_in(X) as hi / lo planes: 64 x 256 floats, 16 float4 per thread, coalesced rows
    for (int e = tid; e < ROWS * (H / 4); e += 256) {
        const int row = e / (H / 4), c4 = e - row * (H / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + row < p.R) v = *reinterpret_cast<const f32x4*>(p.x + (size_t)(r0 + row) * H + 4 * c4);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = pm_act(v[k], p.in_act, p.slope);
        m_u32x2 h2, l2;
        m_split4(v, h2, l2);
        *reinterpret_cast<m_u32x2*>(Ah + row * LDA + 4 * c4) = h2;
        *reinterpret_cast<m_u32x2*>(Al + row * LDA + 4 * c4) = l2;
    }
    __syncthreads();

    const int ncol0 = 64 * wave;     // this wave's 64 output columns
    for (int phase = 0; phase < 2; ++phase) {
        const __bf16* w = phase == 0 ? p.w1 : p.w2;
        f32x16 acc[MS][2];
#pragma unroll
        for (int m = 0; m < MS; ++m)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][t][e] = 0.f;
        // B fragments of k-step s: [kk][t][plane]
        m_bf16x8 bcur[2][2][2], bnxt[2][2][2];
        auto load_b = [&](int s, m_bf16x8 (&b)[2][2][2]) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const size_t o = ((size_t)s * H + ncol0 + 32 * t + i) * 32 + 16 * kk + 8 * h;
                    b[kk][t][0] = *reinterpret_cast<const m_bf16x8*>(w + o);
                    b[kk][t][1] = *reinterpret_cast<const m_bf16x8*>(w + PLANE + o);
                }
        };
        load_b(0, bcur);
#pragma unroll 1
        for (int s = 0; s < H / 32; ++s) {
            if (s + 1 < H / 32) load_b(s + 1, bnxt);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                m_bf16x8 ah[MS], al[MS];
#pragma unroll
                for (int m = 0; m < MS; ++m) {
                    const int o = (32 * m + i) * LDA + 32 * s + 16 * kk + 8 * h;
                    ah[m] = *reinterpret_cast<const m_bf16x8*>(Ah + o);
                    al[m] = *reinterpret_cast<const m_bf16x8*>(Al + o);
                }
#pragma unroll
                for (int m = 0; m < MS; ++m)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bcur[kk][t][0], acc[m][t], 0, 0, 0);
                        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bcur[kk][t][1], acc[m][t], 0, 0, 0);
                        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bcur[kk][t][0], acc[m][t], 0, 0, 0);
                    }
            }
            if (s + 1 < H / 32) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        bcur[kk][t][0] = bnxt[kk][t][0];
                        bcur[kk][t][1] = bnxt[kk][t][1];
                    }
            }
        }
        // epilogue: C/D layout col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * h
        const float* bias = phase == 0 ? p.b1 : p.b2;
        const float* aux = phase == 0 ? p.aux1 : p.aux2;
        const int aux_act = phase == 0 ? p.aux_act1 : p.aux_act2;
        const float* res = phase == 0 ? nullptr : p.x;
        float* out = phase == 0 ? p.out1 : p.out2;
        if (phase == 0) __syncthreads();          // every wave is done reading X before act_mid(out1) overwrites it
#pragma unroll
        for (int m = 0; m < MS; ++m)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int n = ncol0 + 32 * t + i;
                const float bv = bias ? bias[n] : 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rl = 32 * m + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const int row = r0 + rl;
                    float v = 0.f;
                    if (row < p.R) {
                        const size_t o = (size_t)row * H + n;
                        v = pm_epilogue(acc[m][t][e] + bv, aux, res, o, aux_act, PM_ACT_NONE, p.slope);
                        out[o] = v;
                    }
                    if (phase == 0) {
                        const float a = pm_act(v, p.mid_act, p.slope);
                        const __bf16 hi = (__bf16)a;
                        Ah[rl * LDA + n] = hi;
                        Al[rl * LDA + n] = (__bf16)(a - (float)hi);
                    }
                }
            }
        if (phase == 0) __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// A chain of up to four 256 -> 256 layers of a ResidualMLP in ONE launch (both blocks of the AR-GMM network forward, or all
// four of their data gradients):  for layer j = 0 .. L-1
//     out_j = (A_j W_j + b_j) * act'(aux_j)  [+ r_j for odd j],     A_0 = in_act(x),  A_{j+1} = mid_act(out_j),
//     r_1 = x,  r_j = out_{j-2}:  the block's residual  h_{k+1} = ... + h_k   /   dh_k = ... + dh_{k+1}.
// Every out_j is stored (u, h / du, dh are needed by the other pass and by the weight gradients).
// Why a new kernel instead of two mlp_pair launches: that kernel (above) kept one k-step of weights in flight, moved its
// prefetch registers by copies and ran a per-element epilogue (load aux, wait, store - 64 serial round trips per wave):
// 60 - 110 us for 0.3 us of MFMA work per k-step.  Here
//  * a workgroup owns 64 rows, 8 waves; a wave owns ALL 64 rows x 32 columns, so its B fragments (the weights) are read by
//    no other wave: they go straight from L2 to registers, FOUR k-steps ahead (4 static register sets; the stream runs on
//    across layer boundaries), and LDS only carries the A operand (64 KB per k-step and workgroup instead of 96);
//  * activations stay in LDS as hi / lo bf16 planes from layer to layer, the residual stays in registers (a wave owns the
//    same (rows, columns) in every layer);
//  * act'(aux) of relu / leaky-relu is two-valued, so a layer's aux tile is requested during the epilogue of the layer
//    before it (into the then dead A registers) and kept as ONE register of sign bits; the epilogue itself is arithmetic,
//    then stores.
// 12.9 G bf16 MACs*2 per 8192-row launch of four layers = 10 us of matrix time on 128 CUs.
constexpr int CROWS = 64;
struct ChainArgs {
    const float* x;
    const __bf16 *w0, *w1, *w2, *w3;
    const float *b0, *b1, *b2, *b3;
    const float *a0, *a1, *a2, *a3;
    float *o0, *o1, *o2, *o3;
    int R, L;
    int in_act, mid_act, aux_act;
    float slope;
};

template <int I> using ic = std::integral_constant<int, I>;

// none / relu / leaky-relu without branches: v >= 0 ? v : (relu ? 0 : v * ns), ns = 1 (none) or the slope (leaky)
__device__ __forceinline__ float chain_act(float v, bool relu, float ns) {
    const float neg = relu ? 0.f : v * ns;
    return v >= 0.f ? v : neg;
}

template <bool AUX>
__global__ __launch_bounds__(512) void mlp_chain_bf16_kernel(ChainArgs p) {
    __shared__ __attribute__((aligned(16))) __bf16 Ah[CROWS * LDA];
    __shared__ __attribute__((aligned(16))) __bf16 Al[CROWS * LDA];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int r0 = blockIdx.x * CROWS;
    const int ncol = 32 * wave + i;                      // this lane's output column in every layer
    constexpr long long PLANE = (long long)H * H;
    const int aux_act = p.aux_act, L = p.L;
    const float slope = p.slope;
    const bool in_relu = p.in_act == PM_ACT_RELU, mid_relu = p.mid_act == PM_ACT_RELU;
    const float in_ns = p.in_act == PM_ACT_LEAKY ? slope : 1.f, mid_ns = p.mid_act == PM_ACT_LEAKY ? slope : 1.f;

    // weights: B fragment of (k-step s, half kk, plane) = 16 bytes at ((s*H + n)*32 + 16 kk + 8 h); q = 2 kk + plane
    m_bf16x8 bq[4][4];
    auto load_b = [&](const __bf16* w, int s, m_bf16x8 (&b)[4]) {
        const __bf16* src = w + ((size_t)s * H + ncol) * 32 + 8 * h;
        b[0] = *reinterpret_cast<const m_bf16x8*>(src);
        b[1] = *reinterpret_cast<const m_bf16x8*>(src + PLANE);
        b[2] = *reinterpret_cast<const m_bf16x8*>(src + 16);
        b[3] = *reinterpret_cast<const m_bf16x8*>(src + PLANE + 16);
    };
    load_b(p.w0, 0, bq[0]);
    load_b(p.w0, 1, bq[1]);
    load_b(p.w0, 2, bq[2]);
    load_b(p.w0, 3, bq[3]);

    // the block residual of layer 1 (= x) in the accumulator layout: row = 32 m + (e & 3) + 8 (e >> 2) + 4 h, column ncol
    float res[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            res[m][e] = p.x[(size_t)(r0 + 32 * m + (e & 3) + 8 * (e >> 2) + 4 * h) * H + ncol];
    // bit (16 m + e) of the mask: act'(aux) = 1 at that element (else `dneg`: 0 for relu, slope for leaky-relu)
    const float dneg = aux_act == PM_ACT_LEAKY ? slope : 0.f;
    auto aux_mask = [&](const float* ap) -> unsigned {
        float t[2][16];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                t[m][e] = ap[(size_t)(r0 + 32 * m + (e & 3) + 8 * (e >> 2) + 4 * h) * H + ncol];
        unsigned mask = 0u;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const bool one = aux_act == PM_ACT_LEAKY ? t[m][e] >= 0.f : t[m][e] > 0.f;
                mask |= one ? 1u << (16 * m + e) : 0u;
            }
        return mask;
    };
    unsigned amask = 0u;
    if constexpr (AUX) amask = aux_mask(p.a0);

    {   // stage in_act(x) as hi / lo planes: 64 x 256 floats, 8 float4 per thread, all loads first
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 512 * u;
            v[u] = *reinterpret_cast<const f32x4*>(p.x + (size_t)(r0 + (e >> 6)) * H + 4 * (e & 63));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 512 * u;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[u][k] = chain_act(v[u][k], in_relu, in_ns);
            m_u32x2 h2, l2;
            m_split4(v[u], h2, l2);
            *reinterpret_cast<m_u32x2*>(Ah + (e >> 6) * LDA + 4 * (e & 63)) = h2;
            *reinterpret_cast<m_u32x2*>(Al + (e >> 6) * LDA + 4 * (e & 63)) = l2;
        }
    }
    __syncthreads();

    const int abase = i * LDA + 8 * h;                   // A fragment of row tile m, k-step s, half kk: + 32 m LDA + 32 s + 16 kk
#pragma unroll 1
    for (int j = 0; j < L; ++j) {
        const __bf16* wcur = j == 0 ? p.w0 : j == 1 ? p.w1 : j == 2 ? p.w2 : p.w3;
        const int jn = j + 1 < L ? j + 1 : j;
        const __bf16* wnxt = jn == 0 ? p.w0 : jn == 1 ? p.w1 : jn == 2 ? p.w2 : p.w3;
        const float* bj = j == 0 ? p.b0 : j == 1 ? p.b1 : j == 2 ? p.b2 : p.b3;
        const float* ajn = jn == 0 ? p.a0 : jn == 1 ? p.a1 : jn == 2 ? p.a2 : p.a3;    // aux of the NEXT layer
        float* oj = j == 0 ? p.o0 : j == 1 ? p.o1 : j == 2 ? p.o2 : p.o3;
        const float bv = bj ? bj[ncol] : 0.f;

        f32x16 acc[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;

        m_bf16x8 a0[4], a1[4];                            // [2 m + plane] for the k16 halves 0 / 1
        auto read_a = [&](int s, int kk, m_bf16x8 (&a)[4]) {
            const int o = abase + 32 * s + 16 * kk;
            a[0] = *reinterpret_cast<const m_bf16x8*>(Ah + o);
            a[1] = *reinterpret_cast<const m_bf16x8*>(Al + o);
            a[2] = *reinterpret_cast<const m_bf16x8*>(Ah + o + 32 * LDA);
            a[3] = *reinterpret_cast<const m_bf16x8*>(Al + o + 32 * LDA);
        };
        auto mma = [&](const m_bf16x8 (&a)[4], const m_bf16x8& bh, const m_bf16x8& bl) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 * m], bh, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 * m], bl, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 * m + 1], bh, acc[m], 0, 0, 0);
            }
        };
        auto kstep = [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int set = s & 3;
            mma(a0, bq[set][0], bq[set][1]);
            if (s + 1 < 8) read_a(s + 1, 0, a0);
            mma(a1, bq[set][2], bq[set][3]);
            if (s + 1 < 8) read_a(s + 1, 1, a1);
            if (s < 4) load_b(wcur, s + 4, bq[set]);      // four k-steps ahead, on into the next layer's weights
            else load_b(wnxt, s - 4, bq[set]);
            // left alone the scheduler sinks these loads behind the next k-steps' MFMAs until the prefetch distance is gone
            __builtin_amdgcn_sched_barrier(0);
        };
        read_a(0, 0, a0);
        read_a(0, 1, a1);
        kstep(ic<0>{});
        kstep(ic<1>{});
        kstep(ic<2>{});
        kstep(ic<3>{});
        kstep(ic<4>{});
        kstep(ic<5>{});
        kstep(ic<6>{});
        kstep(ic<7>{});
        __syncthreads();                                  // every wave is done reading A_j

        const bool odd = (j & 1) != 0;
        float v[2][16];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float t = acc[m][e] + bv;
                if constexpr (AUX) t = (amask >> (16 * m + e)) & 1u ? t : t * dneg;
                if (odd) t += res[m][e];
                asm volatile("" : "+v"(t));
                v[m][e] = t;
            }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                oj[(size_t)(r0 + 32 * m + (e & 3) + 8 * (e >> 2) + 4 * h) * H + ncol] = v[m][e];
        if constexpr (AUX) amask = aux_mask(ajn);
        if (odd) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int e = 0; e < 16; ++e) res[m][e] = v[m][e];
        }
        if (j + 1 < L) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rl = 32 * m + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const float a = chain_act(v[m][e], mid_relu, mid_ns);
                    const __bf16 hi = (__bf16)a;
                    Ah[rl * LDA + ncol] = hi;
                    Al[rl * LDA + ncol] = (__bf16)(a - (float)hi);
                }
        }
        __syncthreads();
    }
}

// ---- hk.LayerNorm(axis=-1, create_scale=False, create_offset=False) of a ResidualMLP (networks.py:116-131) and the
// relu -> hk.dropout pair between a block's two linears (:124-125).  One wave per row; wave-shuffle reductions.
//   y = (x - mean(x)) * rsqrt(var(x) + eps), biased variance (jnp.var), eps 1e-5;  out = y + res (the block's h += res)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                             float* __restrict__ y, float* __restrict__ out,
                                                             float* __restrict__ rstd, long long R, int Hd, float eps) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + (size_t)r * Hd;
    float s = 0.f;
    for (int i = lane; i < Hd; i += 64) s += xr[i];
    const float mean = pm_wave_sum(s) / (float)Hd;
    float q = 0.f;
    for (int i = lane; i < Hd; i += 64) {
        const float d = xr[i] - mean;
        q += d * d;
    }
    const float rs = rsqrtf(pm_wave_sum(q) / (float)Hd + eps);
    if (lane == 0) rstd[r] = rs;
    for (int i = lane; i < Hd; i += 64) {
        const float v = (xr[i] - mean) * rs;
        y[(size_t)r * Hd + i] = v;
        if (out) out[(size_t)r * Hd + i] = v + (res ? res[(size_t)r * Hd + i] : 0.f);
    }
}

//   dx = rstd * (dy - mean(dy) - y * mean(dy * y))
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ rstd,
                                                             const float* __restrict__ dy, float* __restrict__ dx, long long R,
                                                             int Hd) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int lane = threadIdx.x & 63;
    const float* yr = y + (size_t)r * Hd;
    const float* gr = dy + (size_t)r * Hd;
    float a = 0.f, b = 0.f;
    for (int i = lane; i < Hd; i += 64) {
        a += gr[i];
        b += gr[i] * yr[i];
    }
    a = pm_wave_sum(a) / (float)Hd;
    b = pm_wave_sum(b) / (float)Hd;
    const float rs = rstd[r];
    for (int i = lane; i < Hd; i += 64) dx[(size_t)r * Hd + i] = rs * (gr[i] - a - yr[i] * b);
}

//   out = relu(x) * mask   (mask = keep / (1 - rate), hk.dropout)   and its gradient dx = dout * mask * [x > 0]
__global__ __launch_bounds__(256) void relu_mask_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                             float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = fmaxf(x[i], 0.f) * mask[i];
}
__global__ __launch_bounds__(256) void relu_mask_bwd_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                             const float* __restrict__ dout, float* __restrict__ dx,
                                                             long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dx[i] = x[i] > 0.f ? dout[i] * mask[i] : 0.f;
}

}  // namespace

extern "C" int pm_mlp_pair_bf16(pm_stream_t stream, const float* x, const void* w1_split, const void* w2_split, const float* b1,
                                const float* b2, const float* aux1, const float* aux2, float* out1, float* out2, long long R,
                                int hidden, int in_act, int mid_act, int aux_act1, int aux_act2, float slope) {
    if (!x || !w1_split || !w2_split || !out1 || !out2 || R <= 0 || R > (1LL << 30) || hidden != H) return PM_EINVAL;
    if ((reinterpret_cast<size_t>(x) & 15) || (reinterpret_cast<size_t>(w1_split) & 15) ||
        (reinterpret_cast<size_t>(w2_split) & 15))
        return PM_EINVAL;
    PairArgs p;
    p.x = x; p.w1 = reinterpret_cast<const __bf16*>(w1_split); p.w2 = reinterpret_cast<const __bf16*>(w2_split);
    p.b1 = b1; p.b2 = b2; p.aux1 = aux1; p.aux2 = aux2; p.out1 = out1; p.out2 = out2; p.R = (int)R;
    p.in_act = in_act; p.mid_act = mid_act; p.aux_act1 = aux_act1; p.aux_act2 = aux_act2; p.slope = slope;
    hipLaunchKernelGGL(mlp_pair_bf16_kernel, dim3((unsigned)((R + ROWS - 1) / ROWS)), dim3(256), 0, (hipStream_t)stream, p);
    return pm_check_launch("pm_mlp_pair_bf16");
}

extern "C" int pm_mlp_chain_bf16(pm_stream_t stream, const float* x, int layers, const void* const* w_split,
                                 const float* const* bias, const float* const* aux, float* const* out, long long R, int hidden,
                                 int in_act, int mid_act, int aux_act, float slope) {
    if (!x || !w_split || !out || layers < 1 || layers > 4 || R <= 0 || R > (1LL << 30) || hidden != H) return PM_EINVAL;
    if (R % CROWS != 0 || (reinterpret_cast<size_t>(x) & 15)) return PM_EINVAL;     // whole 64-row tiles only (no row guards)
    for (int a : {in_act, mid_act, aux_act})
        if (a != PM_ACT_NONE && a != PM_ACT_RELU && a != PM_ACT_LEAKY) return PM_EINVAL;   // two-valued act' / branch-free act only
    ChainArgs p{};
    p.x = x;
    const __bf16* w[4] = {nullptr, nullptr, nullptr, nullptr};
    const float* b[4] = {nullptr, nullptr, nullptr, nullptr};
    const float* a[4] = {nullptr, nullptr, nullptr, nullptr};
    float* o[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int j = 0; j < layers; ++j) {
        if (!w_split[j] || !out[j] || (reinterpret_cast<size_t>(w_split[j]) & 15)) return PM_EINVAL;
        w[j] = reinterpret_cast<const __bf16*>(w_split[j]);
        b[j] = bias ? bias[j] : nullptr;
        a[j] = aux ? aux[j] : nullptr;
        o[j] = out[j];
        if (aux && (a[j] == nullptr) != (a[0] == nullptr)) return PM_EINVAL;       // aux for every layer or for none
    }
    const bool has_aux = a[0] != nullptr && aux_act != PM_ACT_NONE;
    p.w0 = w[0]; p.w1 = w[1]; p.w2 = w[2]; p.w3 = w[3];
    p.b0 = b[0]; p.b1 = b[1]; p.b2 = b[2]; p.b3 = b[3];
    p.a0 = a[0]; p.a1 = a[1]; p.a2 = a[2]; p.a3 = a[3];
    p.o0 = o[0]; p.o1 = o[1]; p.o2 = o[2]; p.o3 = o[3];
    p.R = (int)R; p.L = layers; p.in_act = in_act; p.mid_act = mid_act; p.aux_act = aux_act; p.slope = slope;
    PM_KTAG("mlp_chain_bf16_kernel<%d>", (int)has_aux);
    if (has_aux) hipLaunchKernelGGL(mlp_chain_bf16_kernel<true>, dim3((unsigned)(R / CROWS)), dim3(512), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(mlp_chain_bf16_kernel<false>, dim3((unsigned)(R / CROWS)), dim3(512), 0, (hipStream_t)stream, p);
    return pm_check_launch("pm_mlp_chain_bf16");
}

extern "C" int pm_layernorm_fwd(pm_stream_t stream, const float* x, const float* res, float* y, float* out, float* rstd,
                                long long R, int H_, float eps) {
    if (!x || !y || !rstd || R <= 0 || H_ <= 0 || (res && !out)) return PM_EINVAL;
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, res, y, out,
                       rstd, R, H_, eps);
    return pm_check_launch("pm_layernorm_fwd");
}

extern "C" int pm_layernorm_bwd(pm_stream_t stream, const float* y, const float* rstd, const float* dy, float* dx, long long R,
                                int H_) {
    if (!y || !rstd || !dy || !dx || R <= 0 || H_ <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, y, rstd, dy, dx,
                       R, H_);
    return pm_check_launch("pm_layernorm_bwd");
}

extern "C" int pm_relu_mask_fwd(pm_stream_t stream, const float* x, const float* mask, float* out, long long n) {
    if (!x || !mask || !out || n <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(relu_mask_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, mask, out, n);
    return pm_check_launch("pm_relu_mask_fwd");
}

extern "C" int pm_relu_mask_bwd(pm_stream_t stream, const float* x, const float* mask, const float* dout, float* dx,
                                long long n) {
    if (!x || !mask || !dout || !dx || n <= 0) return PM_EINVAL;
    hipLaunchKernelGGL(relu_mask_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, mask, dout,
                       dx, n);
    return pm_check_launch("pm_relu_mask_bwd");
}
