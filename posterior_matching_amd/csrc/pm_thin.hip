// "Thin" convolutions of the PM-VAE: the layers that touch the 1- or 2-channel image.
//   encoder conv_0 (1 -> 32), partial-encoder conv_0 (2 -> 32)       reference networks.py:30-36
//   decoder conv_t_5 (32 -> 1) forward / data-gradient                     networks.py:62-68
// With K = 25..50 or N = 1 these are no matrix problems: padded to MFMA tiles they waste 32x the
// work.  They are HBM / VALU-bound and get plain f32 FMA kernels (coalesced NHWC rows, weights in
// LDS), using the same index rule  src*d = dst*a + tap*cs + off  (d = 1 only) as the GEMM engine.
#include <cstdint>
#include "pm_common.h"

namespace {

struct ThinArgs {
    int B, IH, IW, C, OH, OW, N, KH, KW;
    int a, cs, off;
    int wts, wcs, wns;
    int M;
    int in_act, out_act, aux_act;
    float slope;
};

bool fill_thin(const pm_gather_desc* d, ThinArgs& t) {
    if (!d || d->groups != 1 || d->d != 1 || d->B <= 0 || d->a <= 0) return false;
    if (d->cs != 1 && d->cs != -1) return false;
    if (d->off_x != d->off || d->kws != d->KW) return false;   // no sub-kernel form here
    long long M = (long long)d->B * d->OH * d->OW;
    if (M >= 0x7fffffffLL / 64 || (long long)d->B * d->IH * d->IW * d->C >= 0x7fffffffLL) return false;
    t.B = d->B; t.IH = d->IH; t.IW = d->IW; t.C = d->C; t.OH = d->OH; t.OW = d->OW; t.N = d->N;
    t.KH = d->KH; t.KW = d->KW; t.a = d->a; t.cs = d->cs; t.off = d->off;
    t.wts = d->wts; t.wcs = d->wcs; t.wns = d->wns; t.M = (int)M;
    t.in_act = d->in_act; t.out_act = d->out_act; t.aux_act = d->aux_act; t.slope = d->slope;
    return true;
}

constexpr int THIN_MAXKK = 64;  // taps * C of the thin operand

// out[m][n] = epi( sum_kk in[src(m, tap)][c] * w[kk][n] ),  N in {8,16,24,32}, taps*C <= 64.
// 256 threads = 64 output positions x 4 groups of 8 channels; weights staged in LDS.
__global__ __launch_bounds__(256) void thin_conv_kernel(ThinArgs t, const float* __restrict__ in,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          const float* __restrict__ aux, const float* __restrict__ res,
                                                          float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float Ws[THIN_MAXKK * 32];
    const int KK = t.KH * t.KW * t.C;
    for (int e = threadIdx.x; e < KK * 32; e += 256) {
        int kk = e >> 5, n = e & 31;
        int tap = kk / t.C, c = kk - tap * t.C;
        Ws[e] = n < t.N ? w[tap * t.wts + c * t.wcs + n * t.wns] : 0.f;
    }
    __syncthreads();
    const int ng = threadIdx.x & 3;
    const int m = blockIdx.x * 64 + (threadIdx.x >> 2);
    if (m >= t.M || ng * 8 >= t.N) return;
    const int q = m % t.OW;
    const int r = m / t.OW;
    const int p = r % t.OH;
    const int b = r / t.OH;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const float* ib = in + (size_t)b * t.IH * t.IW * t.C;
    for (int ky = 0; ky < t.KH; ++ky) {
        const int sy = p * t.a + ky * t.cs + t.off;
        if ((unsigned)sy >= (unsigned)t.IH) continue;
        for (int kx = 0; kx < t.KW; ++kx) {
            const int sx = q * t.a + kx * t.cs + t.off;
            if ((unsigned)sx >= (unsigned)t.IW) continue;
            const float* ip = ib + (sy * t.IW + sx) * t.C;
            const float* wp = Ws + ((ky * t.KW + kx) * t.C) * 32 + ng * 8;
            for (int c = 0; c < t.C; ++c) {
                const float v = pm_act(ip[c], t.in_act, t.slope);
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp + c * 32);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(wp + c * 32 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[e] = fmaf(v, w0[e], acc[e]);
                    acc[4 + e] = fmaf(v, w1[e], acc[4 + e]);
                }
            }
        }
    }
    const size_t o = (size_t)m * t.N + ng * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = acc[e] + (bias ? bias[ng * 8 + e] : 0.f);
        out[o + e] = pm_epilogue(v, aux, res, o + e, t.aux_act, t.out_act, t.slope);
    }
}

// out[b,p,q] = act( bias + sum_tap T[b, src(p,q,tap), tap] ): the second half of a wide -> 1-channel
// transposed conv whose per-tap dot products T = in . w[tap] were produced by one GEMM (N = taps).
__global__ __launch_bounds__(256) void tap_shift_add_kernel(ThinArgs t, const float* __restrict__ T, int ldt,
                                                              const float* __restrict__ bias, float* __restrict__ out) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= t.M) return;
    const int q = m % t.OW;
    const int r = m / t.OW;
    const int p = r % t.OH;
    const int b = r / t.OH;
    const float* tb = T + (size_t)b * t.IH * t.IW * ldt;
    float s = bias ? bias[0] : 0.f;
    for (int ky = 0; ky < t.KH; ++ky) {
        const int sy = p * t.a + ky * t.cs + t.off;
        if ((unsigned)sy >= (unsigned)t.IH) continue;
        for (int kx = 0; kx < t.KW; ++kx) {
            const int sx = q * t.a + kx * t.cs + t.off;
            if ((unsigned)sx >= (unsigned)t.IW) continue;
            s += tb[(sy * t.IW + sx) * ldt + ky * t.KW + kx];
        }
    }
    out[m] = pm_act(s, t.out_act, t.slope);
}


// ------------------------------ lane = output channel forms ------------------------------------
// Stride-1 thin layers with C in {1, 2} and a KS x KS kernel, KS in {3, 5} (the 28x28 image side of the PM-VAE
// encoders, and the data gradient of the decoder's last layer).  A workgroup owns TH rows of ONE image:
//   * the (TH+KS-1) x (OW+KS-1) x C source patch is staged ONCE in LDS (planar per channel, pending input
//     activation applied, zero outside the image);
//   * lane n = tid & 31 owns output channel n and keeps its KS*KS*C weights in REGISTERS;
//   * slot = tid >> 5 walks work items of 4 adjacent output positions: per kernel row two 16-byte LDS
//     broadcast reads feed 4*KS FMAs, and every output row is one coalesced 128-byte store.
// VALU-bound by construction (B*OH*OW*KS*KS*C*32 lane-FMAs), no bounds checks in the FMA loop.
template <int CC, int KS>
__global__ __launch_bounds__(256) void thin_conv_lane_kernel(ThinArgs t, int TH, const float* __restrict__ in,
                                                               const float* __restrict__ w,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ aux,
                                                               const float* __restrict__ res,
                                                               float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float thin_lds[];
    float* P = thin_lds;
    const int tid = threadIdx.x;
    const int n = tid & 31, slot = tid >> 5;
    const int tiles_y = (t.OH + TH - 1) / TH;
    const int b = blockIdx.x / tiles_y;
    const int y0 = (blockIdx.x - b * tiles_y) * TH;
    const int PH = TH + KS - 1;
    const int QX = (t.OW + 3) >> 2;
    const int PW = 4 * QX + 4;                       // item q reads patch columns 4q .. 4q+7
    const int sy0 = y0 + t.off + (t.cs < 0 ? -(KS - 1) : 0);
    const int sx0 = t.off + (t.cs < 0 ? -(KS - 1) : 0);
    const float* img = in + (size_t)b * t.IH * t.IW * CC;
    for (int e = tid; e < CC * PH * PW; e += 256) {
        const int c = e / (PH * PW);
        const int r = e - c * PH * PW;
        const int py = r / PW, px = r - py * PW;
        const int gy = sy0 + py, gx = sx0 + px;
        float v = 0.f;
        if ((unsigned)gy < (unsigned)t.IH && (unsigned)gx < (unsigned)t.IW)
            v = pm_act(img[(gy * t.IW + gx) * CC + c], t.in_act, t.slope);
        P[e] = v;
    }
    // weights of this lane's channel, indexed by patch-relative shift (jy, jx)
    float wr[CC][KS][KS];
#pragma unroll
    for (int c = 0; c < CC; ++c)
#pragma unroll
        for (int jy = 0; jy < KS; ++jy)
#pragma unroll
            for (int jx = 0; jx < KS; ++jx) {
                const int ky = t.cs > 0 ? jy : KS - 1 - jy;
                const int kx = t.cs > 0 ? jx : KS - 1 - jx;
                wr[c][jy][jx] = n < t.N ? w[(ky * KS + kx) * t.wts + c * t.wcs + n * t.wns] : 0.f;
            }
    const float bv = (bias && n < t.N) ? bias[n] : 0.f;
    __syncthreads();
    const int rows = t.OH - y0 < TH ? t.OH - y0 : TH;
    for (int item = slot; item < rows * QX; item += 8) {
        const int y = item / QX, qx = item - y * QX;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < CC; ++c)
#pragma unroll
            for (int jy = 0; jy < KS; ++jy) {
                const float* row = P + (c * PH + y + jy) * PW + 4 * qx;
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(row);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(row + 4);
                const float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
                for (int jx = 0; jx < KS; ++jx)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaf(xv[j + jx], wr[c][jy][jx], acc[j]);
            }
        if (n < t.N) {
            // loads, arithmetic, stores in three phases (element by element every aux / res load waits for the store in
            // front of it: vmcnt counts both)
            const size_t o0 = ((size_t)(b * t.OH + y0 + y) * t.OW + 4 * qx) * t.N + n;
            const int nv = t.OW - 4 * qx < 4 ? t.OW - 4 * qx : 4;         // valid positions of the item
            float av[4], rv[4], v[4];
            if (aux) {
#pragma unroll
                for (int j = 0; j < 4; ++j) av[j] = aux[o0 + (size_t)(j < nv ? j : 0) * t.N];
            }
            if (res) {
#pragma unroll
                for (int j = 0; j < 4; ++j) rv[j] = res[o0 + (size_t)(j < nv ? j : 0) * t.N];
            }
            const bool after = (t.aux_act & PM_AUX_AFTER_RES) != 0;
            const int dact = t.aux_act & (PM_AUX_AFTER_RES - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float xx = acc[j] + bv;
                if (res && after) xx += rv[j];
                if (aux) xx *= pm_dact(av[j], dact, t.slope);
                if (res && !after) xx += rv[j];
                xx = pm_act(xx, t.out_act, t.slope);
                asm volatile("" : "+v"(xx));
                v[j] = xx;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nv) out[o0 + (size_t)j * t.N] = v[j];
        }
    }
}

// Weight gradient of the same layers:  dw[tap][c][n] += sum_m G[src(m, tap)][c] * D[m][n],  db[n] += sum_m D[m][n]
// with G the thin (C in {1,2}) operand and D the wide one (N <= 32).  One workgroup of NS slots x 32 lanes walks
// whole images: G's padded image lives in LDS, lane n reads its D row coalesced (4 adjacent positions per item,
// the next item's values prefetched), and keeps all KS*KS*C partial sums of its column in registers.  One LDS
// reduction over the slots and ONE atomic per weight per workgroup (gridDim <= 256) end the kernel - the generic
// kernel's V1 loader spent 44..96 us on these layers gathering scalars.
template <int CC, int KS, int NS>
__global__ __launch_bounds__(32 * NS) void thin_wgrad_lane_kernel(ThinArgs t, const float* __restrict__ gathered,
                                                                    const float* __restrict__ dense,
                                                                    float* __restrict__ dw, float* __restrict__ db,
                                                                    float* __restrict__ dbg, pm_wgrad_part part) {
    // part.w != NULL (partial-sum mode): this workgroup STORES its sums into slot blockIdx.x of the arenas instead of adding
    // them to dw / db / dbg with atomics (one writer per slot element; pm_reduce_partials sums the slots in a fixed order)
    const bool pmode = part.w != nullptr;
    if (pmode) {
        dw = part.w + (size_t)blockIdx.x * part.w_stride;
        db = part.b ? part.b + (size_t)blockIdx.x * part.b_stride : nullptr;
        dbg = part.bg ? part.bg + (size_t)blockIdx.x * part.bg_stride : nullptr;
    }
    extern __shared__ __attribute__((aligned(16))) float thin_lds[];
    constexpr int NT = 32 * NS;
    constexpr int KK = CC * KS * KS;
    float* P = thin_lds;
    const int tid = threadIdx.x;
    const int n = tid & 31, slot = tid >> 5;
    const int PH = t.OH + KS - 1;
    const int QX = (t.OW + 3) >> 2;
    const int PW = 4 * QX + 4;
    const int sy0 = t.off + (t.cs < 0 ? -(KS - 1) : 0);
    const int sx0 = sy0;
    const bool nok = n < t.N;
    float acc[CC][KS][KS];
#pragma unroll
    for (int c = 0; c < CC; ++c)
#pragma unroll
        for (int jy = 0; jy < KS; ++jy)
#pragma unroll
            for (int jx = 0; jx < KS; ++jx) acc[c][jy][jx] = 0.f;
    float accb = 0.f;
    float sg[CC];                                     // column sums of the gathered operand (dbg)
#pragma unroll
    for (int c = 0; c < CC; ++c) sg[c] = 0.f;
    const int items = t.OH * QX;
    auto load_d = [&](const float* dimg, int item, float (&d)[4]) {
        const int y = item / QX, qx = item - y * QX;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = 4 * qx + j;
            d[j] = (nok && item < items && x < t.OW) ? dimg[(size_t)(y * t.OW + x) * t.N + n] : 0.f;
        }
    };
    for (int b = blockIdx.x; b < t.B; b += gridDim.x) {
        const float* img = gathered + (size_t)b * t.IH * t.IW * CC;
        const float* dimg = dense + (size_t)b * t.OH * t.OW * t.N;
        float dn[4];
        load_d(dimg, slot, dn);                       // in flight while the patch is staged
        __syncthreads();                              // previous image's readers are done with P
        for (int e = tid; e < CC * PH * PW; e += NT) {
            const int c = e / (PH * PW);
            const int r = e - c * PH * PW;
            const int py = r / PW, px = r - py * PW;
            const int gy = sy0 + py, gx = sx0 + px;
            float v = 0.f;
            if ((unsigned)gy < (unsigned)t.IH && (unsigned)gx < (unsigned)t.IW) {
                const float raw = img[(gy * t.IW + gx) * CC + c];
#pragma unroll
                for (int k = 0; k < CC; ++k) sg[k] += k == c ? raw : 0.f;
                v = pm_act(raw, t.in_act, t.slope);
            }
            P[e] = v;
        }
        __syncthreads();
        for (int item = slot; item < items; item += NS) {
            float d[4] = {dn[0], dn[1], dn[2], dn[3]};
            load_d(dimg, item + NS, dn);
            const int y = item / QX, qx = item - y * QX;
            accb += (d[0] + d[1]) + (d[2] + d[3]);
#pragma unroll
            for (int c = 0; c < CC; ++c)
#pragma unroll
                for (int jy = 0; jy < KS; ++jy) {
                    const float* row = P + (c * PH + y + jy) * PW + 4 * qx;
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(row);
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(row + 4);
                    const float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                    // Scalar v_fmac through inline asm ON PURPOSE.  Written as fmaf() the compiler emits packed
                    // v_pk_fma_f32 pairs, and that build returned run-to-run different sums - one 16-lane group of
                    // one accumulator off by about one item's contribution - but only while an MFMA-heavy kernel
                    // (patch_conv_bf16 / gather_wgrad_bf16_sub) ran beside it on another stream; alone, or beside
                    // light kernels, it was bit-stable, and inputs and outputs were never touched by the neighbour
                    // (tools/debug_thin_wgrad2.py bisects it: not the dense loads, not LDS, not the atomics).
                    // Root cause not established (ROCm 7.2 hazard handling of packed FP32?); this form is stable.
#pragma unroll
                    for (int jx = 0; jx < KS; ++jx)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[c][jy][jx]) : "v"(xv[j + jx]), "v"(d[j]));
                }
        }
    }
    // slots -> one sum per (tap, c, n): the two slots of a wave by shuffle, the waves through LDS
    __syncthreads();
    float* red = thin_lds;                            // [NS/2][KK + 1][32]
    const int wave = tid >> 6;
#pragma unroll
    for (int c = 0; c < CC; ++c)
#pragma unroll
        for (int jy = 0; jy < KS; ++jy)
#pragma unroll
            for (int jx = 0; jx < KS; ++jx) {
                float v = acc[c][jy][jx];
                v += __shfl_xor(v, 32, 64);
                if ((tid & 32) == 0) red[(wave * (KK + 1) + (c * KS + jy) * KS + jx) * 32 + n] = v;
            }
    {
        float v = accb;
        v += __shfl_xor(v, 32, 64);
        if ((tid & 32) == 0) red[(wave * (KK + 1) + KK) * 32 + n] = v;
    }
    float* redg = red + (NS / 2) * (KK + 1) * 32;     // [NS/2][CC]
    if (dbg) {
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const float v = pm_wave_sum(sg[c]);
            if ((tid & 63) == 0) redg[wave * CC + c] = v;
        }
    }
    __syncthreads();
    if (dbg && tid < CC) {
        float s = 0.f;
        for (int wv = 0; wv < NS / 2; ++wv) s += redg[wv * CC + tid];
        if (pmode) dbg[tid] = s;
        else atomicAdd(dbg + tid, s);
    }
    for (int e = tid; e < (KK + 1) * 32; e += NT) {
        const int idx = e >> 5, nn = e & 31;
        if (nn >= t.N) continue;
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < NS / 2; ++wv) s += red[(wv * (KK + 1) + idx) * 32 + nn];
        if (idx == KK) {
            if (db) {
                if (pmode) db[nn] = s;
                else atomicAdd(db + nn, s);
            }
            continue;
        }
        const int c = idx / (KS * KS);
        const int jj = idx - c * KS * KS;
        const int jy = jj / KS, jx = jj - jy * KS;
        const int ky = t.cs > 0 ? jy : KS - 1 - jy;
        const int kx = t.cs > 0 ? jx : KS - 1 - jx;
        float* wd = dw + (size_t)(ky * KS + kx) * t.wts + (size_t)c * t.wcs + (size_t)nn * t.wns;
        if (pmode) *wd = s;
        else atomicAdd(wd, s);
    }
}

// Wide -> 1 channel stride-1 layer (the decoder's last transposed conv, forward):
//   out[b,p,q] = epi( bias + sum_tap sum_c act_in(in[b, src(p,q,tap), c]) * w[tap][c] )
// One thread per output position of a TH x OW tile of one image; the source patch sits in LDS with a 16-byte
// pad per position (conflict-free 16-byte reads at a 144-byte lane stride for C = 32), the weights in LDS as
// [shift][C] and are read as wave-uniform broadcasts.  Replaces a padded GEMM (N = taps) + a shifted sum.
__global__ __launch_bounds__(256) void thin_to1_kernel(ThinArgs t, int TH, const float* __restrict__ in,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         const float* __restrict__ aux, const float* __restrict__ res,
                                                         float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float thin_lds[];
    const int tid = threadIdx.x;
    const int tiles_y = (t.OH + TH - 1) / TH;
    const int b = blockIdx.x / tiles_y;
    const int y0 = (blockIdx.x - b * tiles_y) * TH;
    const int PH = TH + t.KH - 1, PW = t.OW + t.KW - 1;
    const int PS = t.C + 4;
    const int c4n = t.C >> 2;
    float* Wl = thin_lds;                              // [KH*KW][C]
    float* P = thin_lds + t.KH * t.KW * t.C;           // [PH][PW][PS]
    const int sy0 = y0 + t.off + (t.cs < 0 ? -(t.KH - 1) : 0);
    const int sx0 = t.off + (t.cs < 0 ? -(t.KW - 1) : 0);
    for (int e = tid; e < t.KH * t.KW * t.C; e += 256) {
        const int jj = e / t.C, c = e - jj * t.C;
        const int jy = jj / t.KW, jx = jj - jy * t.KW;
        const int ky = t.cs > 0 ? jy : t.KH - 1 - jy;
        const int kx = t.cs > 0 ? jx : t.KW - 1 - jx;
        Wl[e] = w[(ky * t.KW + kx) * t.wts + c * t.wcs];
    }
    const float* img = in + (size_t)b * t.IH * t.IW * t.C;
    for (int e = tid; e < PH * PW * c4n; e += 256) {
        const int pos = e / c4n, c4 = e - pos * c4n;
        const int py = pos / PW, px = pos - py * PW;
        const int gy = sy0 + py, gx = sx0 + px;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)gy < (unsigned)t.IH && (unsigned)gx < (unsigned)t.IW) {
            v = *reinterpret_cast<const f32x4*>(img + ((size_t)gy * t.IW + gx) * t.C + 4 * c4);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = pm_act(v[k], t.in_act, t.slope);
        }
        *reinterpret_cast<f32x4*>(P + pos * PS + 4 * c4) = v;
    }
    __syncthreads();
    const int y = tid / t.OW, x = tid - y * t.OW;
    if (y >= TH || y0 + y >= t.OH) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int jy = 0; jy < t.KH; ++jy)
        for (int jx = 0; jx < t.KW; ++jx) {
            const float* px = P + ((y + jy) * PW + x + jx) * PS;
            const float* wp = Wl + (jy * t.KW + jx) * t.C;
            for (int c4 = 0; c4 < c4n; ++c4) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(px + 4 * c4);
                const f32x4 ww = *reinterpret_cast<const f32x4*>(wp + 4 * c4);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = fmaf(a[k], ww[k], acc[k]);
            }
        }
    const size_t o = (size_t)(b * t.OH + y0 + y) * t.OW + x;
    const float v = (acc[0] + acc[1]) + (acc[2] + acc[3]) + (bias ? bias[0] : 0.f);
    out[o] = pm_epilogue(v, aux, res, o, t.aux_act, t.out_act, t.slope);
}

// The same wide -> 1 layer on the bf16 matrix cores (bf16x3, f32-grade like every other MFMA path here):
//   T[p][tap] = in[p] . w[tap]   for the source positions p of the strip (one 32x32 MFMA tile per 32 consecutive
//                                NHWC positions: the A operand is a CONTIGUOUS 4 KB read, no gather at all)
//   out[y][x] = epi( bias + sum_tap T[src(y, x, tap)][tap] )   from LDS
// i.e. the GEMM + pm_tap_shift_add pair fused, with T never leaving the workgroup.  C % 32 == 0, KH*KW <= 32.
typedef __bf16 t_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 t_bf16x2 __attribute__((ext_vector_type(2)));
typedef float t_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned t_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void t_split8(const float (&x)[8], t_bf16x8& hi, t_bf16x8& lo) {
    t_u32x4 hp, lp;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a0 = x[2 * j], a1 = x[2 * j + 1];
        const unsigned hh = __builtin_bit_cast(unsigned, __builtin_convertvector(t_f32x2{a0, a1}, t_bf16x2));
        const float f0 = __builtin_bit_cast(float, hh << 16);
        const float f1 = __builtin_bit_cast(float, hh & 0xffff0000u);
        hp[j] = hh;
        lp[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(t_f32x2{a0 - f0, a1 - f1}, t_bf16x2));
    }
    hi = __builtin_bit_cast(t_bf16x8, hp);
    lo = __builtin_bit_cast(t_bf16x8, lp);
}

template <int CCH>   // CCH = C / 32
__global__ __launch_bounds__(256) void thin_to1_bf16_kernel(ThinArgs t, int TH, const float* __restrict__ in,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ aux,
                                                              const float* __restrict__ res,
                                                              float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float thin_lds[];
    float* T = thin_lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int tiles_y = (t.OH + TH - 1) / TH;
    const int b = blockIdx.x / tiles_y;
    const int y0 = (blockIdx.x - b * tiles_y) * TH;
    const int taps = t.KH * t.KW;
    const int TP = taps | 1;
    // source rows this strip touches, clipped to the image
    int r_lo = y0 + t.off + (t.cs < 0 ? -(t.KH - 1) : 0);
    int r_hi = y0 + TH - 1 + t.off + (t.cs > 0 ? t.KH - 1 : 0);
    if (r_lo < 0) r_lo = 0;
    if (r_hi > t.IH - 1) r_hi = t.IH - 1;
    const int np = (r_hi - r_lo + 1) * t.IW;
    const float* src = in + ((size_t)b * t.IH + r_lo) * t.IW * t.C;

    // B operand: lane (i, h) holds w[tap = i][c0 + 16kk + 8h .. +8), split once
    t_bf16x8 bh[CCH][2], bl[CCH][2];
#pragma unroll
    for (int cc = 0; cc < CCH; ++cc)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = cc * 32 + 16 * kk + 8 * h + e;
                v[e] = i < taps ? w[i * t.wts + c * t.wcs] : 0.f;
            }
            t_split8(v, bh[cc][kk], bl[cc][kk]);
        }
    const int ntiles = (np + 31) >> 5;
    for (int tile = wave; tile < ntiles; tile += 4) {
        const int p = tile * 32 + i;
        const bool ok = p < np;
        const float* row = src + (size_t)(ok ? p : 0) * t.C + 8 * h;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int cc = 0; cc < CCH; ++cc)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(row + cc * 32 + 16 * kk);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(row + cc * 32 + 16 * kk + 4);
                float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = ok ? pm_act(v[e], t.in_act, t.slope) : 0.f;
                t_bf16x8 ah, al;
                t_split8(v, ah, al);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[cc][kk], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[cc][kk], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[cc][kk], acc, 0, 0, 0);
            }
        if (i < taps) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int pr = tile * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;     // C/D layout: row of register e
                if (pr < np) T[pr * TP + i] = acc[e];
            }
        }
    }
    __syncthreads();
    const int y = tid / t.OW, x = tid - y * t.OW;
    if (y >= TH || y0 + y >= t.OH) return;
    float s = bias ? bias[0] : 0.f;
    for (int ky = 0; ky < t.KH; ++ky) {
        const int sy = y0 + y + ky * t.cs + t.off;
        if (sy < r_lo || sy > r_hi) continue;
        for (int kx = 0; kx < t.KW; ++kx) {
            const int sx = x + kx * t.cs + t.off;
            if ((unsigned)sx >= (unsigned)t.IW) continue;
            s += T[((sy - r_lo) * t.IW + sx) * TP + ky * t.KW + kx];
        }
    }
    const size_t o = (size_t)(b * t.OH + y0 + y) * t.OW + x;
    out[o] = pm_epilogue(s, aux, res, o, t.aux_act, t.out_act, t.slope);
}

// rows per workgroup of the lane forms: the largest divisor-friendly strip that still gives >= ~2 workgroups per CU
int lane_rows(const ThinArgs& t) {
    int th = t.OH;
    while (th > 7 && (long long)t.B * ((t.OH + th - 1) / th) < 512) th = (th + 1) / 2;
    return th;
}
bool lane_form_ok(const ThinArgs& t) {
    return t.a == 1 && t.KH == t.KW && (t.KH == 3 || t.KH == 5) && (t.C == 1 || t.C == 2) && t.N <= 32 &&
           t.OW <= 128 && t.OH <= 128;
}

}  // namespace

extern "C" int pm_thin_conv(pm_stream_t stream, const pm_gather_desc* d, const float* in, const float* w,
                            const float* bias, const float* aux, const float* res, float* out) {
    ThinArgs t;
    if (!fill_thin(d, t) || !in || !w || !out) return PM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (t.N == 1) {   // wide -> 1 channel
        if (t.a != 1 || t.C % 4 != 0 || t.OW > 256 || (reinterpret_cast<uintptr_t>(in) & 15)) return PM_EINVAL;
        int TH = 256 / t.OW;
        if (TH > t.OH) TH = t.OH;
        size_t lds = 0;
        for (; TH >= 1; --TH) {
            const int tiles = (t.OH + TH - 1) / TH;
            TH = (t.OH + tiles - 1) / tiles;              // even strips
            lds = ((size_t)t.KH * t.KW * t.C + (size_t)(TH + t.KH - 1) * (t.OW + t.KW - 1) * (t.C + 4)) * 4;
            if (lds <= 64 * 1024) break;
        }
        if (TH < 1) return PM_EINVAL;
        const int tiles = (t.OH + TH - 1) / TH;
        PM_KTAG("thin_to1_kernel");
        hipLaunchKernelGGL(thin_to1_kernel, dim3(t.B * tiles), dim3(256), lds, s, t, TH, in, w, bias, aux, res, out);
        return pm_check_launch("pm_thin_conv");
    }
    if (t.N > 32) return PM_EINVAL;
    if (lane_form_ok(t)) {
        const int TH = lane_rows(t);
        const int tiles = (t.OH + TH - 1) / TH;
        const size_t lds = (size_t)t.C * (TH + t.KH - 1) * (4 * ((t.OW + 3) / 4) + 4) * 4;
        const dim3 grid(t.B * tiles);
#define PM_TL(CCv, KSv) hipLaunchKernelGGL((thin_conv_lane_kernel<CCv, KSv>), grid, dim3(256), lds, s, t, TH, in, w, bias, aux, res, out)
        PM_KTAG("thin_conv_lane_kernel<%d, %d>", t.C == 1 ? 1 : 2, t.KH == 5 ? 5 : 3);
        if (t.C == 1 && t.KH == 5) PM_TL(1, 5);
        else if (t.C == 2 && t.KH == 5) PM_TL(2, 5);
        else if (t.C == 1) PM_TL(1, 3);
        else PM_TL(2, 3);
#undef PM_TL
        return pm_check_launch("pm_thin_conv");
    }
    if (t.KH * t.KW * t.C > THIN_MAXKK || t.N % 8 != 0) return PM_EINVAL;
    PM_KTAG("thin_conv_kernel");
    hipLaunchKernelGGL(thin_conv_kernel, dim3((t.M + 63) / 64), dim3(256), 0, s, t, in, w, bias, aux, res, out);
    return pm_check_launch("pm_thin_conv");
}

extern "C" int pm_thin_to1_bf16(pm_stream_t stream, const pm_gather_desc* d, const float* in, const float* w,
                                const float* bias, const float* aux, const float* res, float* out) {
    ThinArgs t;
    if (!fill_thin(d, t) || !in || !w || !out) return PM_EINVAL;
    if (t.N != 1 || t.a != 1 || (t.C != 32 && t.C != 64) || t.KH * t.KW > 32 || t.OW > 256 ||
        (reinterpret_cast<uintptr_t>(in) & 15))
        return PM_EINVAL;
    int TH = 256 / t.OW;
    if (TH > t.OH) TH = t.OH;
    const int TP = (t.KH * t.KW) | 1;
    size_t lds = 0;
    for (; TH >= 1; --TH) {
        const int tiles = (t.OH + TH - 1) / TH;
        TH = (t.OH + tiles - 1) / tiles;
        int rows = TH + t.KH - 1;
        if (rows > t.IH) rows = t.IH;
        lds = (size_t)rows * t.IW * TP * 4;
        if (lds <= 64 * 1024) break;
    }
    if (TH < 1) return PM_EINVAL;
    const int tiles = (t.OH + TH - 1) / TH;
    hipStream_t s = (hipStream_t)stream;
    PM_KTAG("thin_to1_bf16_kernel<%d>", t.C == 32 ? 1 : 2);
    if (t.C == 32) hipLaunchKernelGGL(thin_to1_bf16_kernel<1>, dim3(t.B * tiles), dim3(256), lds, s, t, TH, in, w, bias, aux, res, out);
    else hipLaunchKernelGGL(thin_to1_bf16_kernel<2>, dim3(t.B * tiles), dim3(256), lds, s, t, TH, in, w, bias, aux, res, out);
    return pm_check_launch("pm_thin_to1_bf16");
}

static int thin_wgrad_impl(pm_stream_t stream, const pm_gather_desc* d, const float* gathered, const float* dense,
                           float* dw, float* db, float* db_gathered, const pm_wgrad_part* partp, int* slots_out) {
    ThinArgs t;
    pm_wgrad_part part = {};
    if (partp) {
        part = *partp;
        dw = part.w; db = part.b; db_gathered = part.bg;
    }
    if (!fill_thin(d, t) || !gathered || !dense || (!dw && !slots_out)) return PM_EINVAL;
    if (!lane_form_ok(t)) return PM_EINVAL;
    {
        const int nslots = t.B < 256 ? t.B : 256;            // = the grid below: one slot per persistent workgroup
        if (slots_out) { *slots_out = nslots; return PM_OK; }
        if (partp && part.nslots != nslots) return PM_EINVAL;
    }
    constexpr int NS = 32;
    const int KK = t.C * t.KH * t.KW;
    const size_t patch = (size_t)t.C * (t.OH + t.KH - 1) * (4 * ((t.OW + 3) / 4) + 4);
    const size_t red = (size_t)(NS / 2) * (KK + 1) * 32 + (NS / 2) * 2;
    const size_t lds = (patch > red ? patch : red) * 4;
    if (lds > 150 * 1024) return PM_EINVAL;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_wgrad_lane_kernel<1, 5, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_wgrad_lane_kernel<2, 5, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_wgrad_lane_kernel<1, 3, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_wgrad_lane_kernel<2, 3, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set = true;
    }
    const dim3 grid(t.B < 256 ? t.B : 256);
    hipStream_t s = (hipStream_t)stream;
#define PM_TW(CCv, KSv) hipLaunchKernelGGL((thin_wgrad_lane_kernel<CCv, KSv, NS>), grid, dim3(32 * NS), lds, s, t, gathered, dense, dw, db, db_gathered, part)
    PM_KTAG("thin_wgrad_lane_kernel<%d, %d, 32>", t.C == 1 ? 1 : 2, t.KH == 5 ? 5 : 3);
    if (t.C == 1 && t.KH == 5) PM_TW(1, 5);
    else if (t.C == 2 && t.KH == 5) PM_TW(2, 5);
    else if (t.C == 1) PM_TW(1, 3);
    else PM_TW(2, 3);
#undef PM_TW
    return pm_check_launch("pm_thin_wgrad");
}

extern "C" int pm_thin_wgrad(pm_stream_t stream, const pm_gather_desc* d, const float* gathered, const float* dense,
                             float* dw, float* db, float* db_gathered) {
    return thin_wgrad_impl(stream, d, gathered, dense, dw, db, db_gathered, nullptr, nullptr);
}
extern "C" int pm_thin_wgrad_part(pm_stream_t stream, const pm_gather_desc* d, const float* gathered, const float* dense,
                                  const pm_wgrad_part* part) {
    if (!part || !part->w) return PM_EINVAL;
    return thin_wgrad_impl(stream, d, gathered, dense, nullptr, nullptr, nullptr, part, nullptr);
}
extern "C" int pm_thin_wgrad_part_slots(const pm_gather_desc* d, const float* gathered, const float* dense, int* nslots) {
    if (!nslots) return PM_EINVAL;
    return thin_wgrad_impl(nullptr, d, gathered, dense, nullptr, nullptr, nullptr, nullptr, nslots);
}

extern "C" int pm_tap_shift_add(pm_stream_t stream, const pm_gather_desc* d, const float* T, int ldt,
                                const float* bias, float* out) {
    ThinArgs t;
    if (!fill_thin(d, t) || !T || !out || d->N != 1 || ldt < d->KH * d->KW) return PM_EINVAL;
    hipLaunchKernelGGL(tap_shift_add_kernel, dim3((t.M + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, T, ldt,
                       bias, out);
    return pm_check_launch("pm_tap_shift_add");
}
