// Gather-GEMM engine: conv / transposed-conv / dense forward, data-gradient and weight-gradient
// on the f32 MFMA (v_mfma_f32_32x32x2_f32) of gfx950.
//
// Replaces the XLA kernels behind hk.Conv2D / hk.Conv2DTranspose / hk.Linear and their jax.grad
// (reference networks.py:30-36,62-68,116-129).  One index rule
//     src*d = dst*a + tap*cs + off
// expresses all four conv flavours (see include/pmhip.h), so one kernel serves every layer.
//
// Tiling: a 256-thread workgroup (4 waves) owns a BM x BN output tile and walks K = taps*C in
// chunks of 32.  The gathered operand is loaded along its contiguous channel axis (16 B per lane
// when C % 4 == 0), staged in LDS with a 36-float row stride (conflict-free ds_read_b128) and
// consumed as 32x32x2 MFMA fragments; register prefetch of chunk t+1 overlaps the MFMAs of t.
//
// Structural zeros are never multiplied: output rows are enumerated class-major, a class being
// (p mod PY, q mod PX) with PY = PX = d for the zero-dilated forms (stride-2 transposed conv and
// stride-2 conv data-gradient) and PY x PX = the whole output grid when the gathered grid is 1x1
// (7x7 VALID layers).  Every tile then sees one class, and taps that no row of the tile can reach
// (dilation holes, padding) are dropped from its K loop.
//
// The f32 MFMA retires 32x32x2 in 64 cycles, so a 32-deep chunk is only 16 MFMAs per wave: the
// loaders are written to cost a few VALU instructions per row (tap decode and weight base are
// wave-uniform scalars, per-row state is precomputed), otherwise address arithmetic - not the
// matrix pipe - bounds the kernel.
#include <cstdlib>
#include <type_traits>
#include "pm_common.h"

namespace {

struct Geom {
    int B, IH, IW, C, OH, OW, N, KH, KW;
    int a, cs, off, offx, d;   // off: row (y) offset of the index rule, offx: column (x) offset
    int wts, wcs, wns, kws;    // kws: taps per kernel ROW in the weight layout (>= KW; sub-kernels of masked convs)
    int M, K;
    int in_act, out_act, aux_act;
    float slope;
    int PY, PX, OHc, OWc, Mc;  // class-major row order: Mc rows per class
};

struct GemmArgs {
    Geom g;
    const float* in;
    const float* w;
    const float* bias;
    const float* aux;
    const float* res;
    float* out;
    long long in_gs, w_gs, out_gs, bias_gs;
    int ksplit;
    float* out2;   // optional second store: out2 = act2(out) (a layer's pre-activation AND its activation in one launch)
    int act2;
    float* in_colsum;   // image-resident form only: in_colsum[c] += sum over every position of in[.., c] (a bias gradient)
    // split-K slabs (round 4): sk != NULL -> K-slice ks STORES its partial tile into sk + ks * sk_stride (one writer per
    // element) and the epilogue kernel adds the slices in order - no zero-fill, no f32 atomics, run-to-run identical outputs
    float* sk;
    long long sk_stride;
};

struct WgradArgs {
    Geom g;
    const float* gathered;
    const float* dense;
    float* dw;
    float* db;
    long long in_gs, w_gs, out_gs, bias_gs;
    int chunks_per_split;
    int step_b, step_p, step_q;  // 128 rows = step_b images + step_p rows + step_q pixels
    int ntiles, nsplits, xcd_map;  // bf16 kernel: (k-block, n-block) tiles x m-splits, see the block-id remap there
    int cpad;                      // sub kernel: 32 < C < 64 - a k-block is ONE tap, its channels zero-padded to 64
    // optional per-group element offsets (gathered, dense, dw, db) relative to the four base pointers, in DEVICE memory:
    // groups whose operands are separately allocated buffers (the weight gradients of many same-shaped layers in one launch)
    const long long* gtab;
    // partial-sum destination (pm_wgrad_part): part_w != NULL -> a workgroup STORES its sums into slot s of the arena
    // (part_w + s * part_ws + the offsets it would add to dw) instead of adding them to dw with f32 atomics; every element of
    // a slot has exactly one writer per launch, pm_reduce_partials sums the slots in a fixed order.
    float* part_w;
    float* part_b;
    long long part_ws, part_bs;
};

// element offset of group `grp`'s operand `which` (0 gathered, 1 dense, 2 dw, 3 db)
__device__ __forceinline__ long long wg_off(const WgradArgs& p, int grp, int which, long long stride) {
    return p.gtab ? p.gtab[4 * grp + which] : (long long)grp * stride;
}

// where a workgroup of m-split / persistent slot `slot` sends its weight (bias) sums, and how
__device__ __forceinline__ float* wg_dw(const WgradArgs& p, int slot) {
    return p.part_w ? p.part_w + (size_t)slot * p.part_ws : p.dw;
}
__device__ __forceinline__ float* wg_db(const WgradArgs& p, int slot) {
    return p.part_w ? p.part_b + (size_t)slot * p.part_bs : p.db;
}
__device__ __forceinline__ void wg_put(bool part, float* addr, float v) {
    if (part) *addr = v;
    else atomicAdd(addr, v);
}

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;  // 144-byte rows: 16-B aligned, ds_read_b128 conflict-free
constexpr int MAXTAP = 16;      // per-axis kernel extent supported by the tap lists
constexpr int ROW_INVALID = -(1 << 28);

// loader modes
constexpr int MODE_TU1 = 0;  // C % 32 == 0 (every chunk inside one tap), d == 1
constexpr int MODE_TU2 = 1;  // C % 32 == 0, d == 2
constexpr int MODE_V4 = 2;   // C % 4 == 0, anything else
constexpr int MODE_V1 = 3;   // scalar gathers (C = 1, 2, ...)

template <int DD>
__device__ __forceinline__ bool coord_ok(int t, int d, int lim, int& s) {
    if (DD == 1) {
        s = t;
        return (unsigned)t < (unsigned)lim;
    } else if (DD == 2) {
        s = t >> 1;
        return t >= 0 && !(t & 1) && s < lim;
    } else {
        if (t < 0) return false;
        int q = t / d;
        s = q;
        return q * d == t && q < lim;
    }
}

// row m of the class-major enumeration -> (image b, output position p, q)
__device__ __forceinline__ void decode_row(const Geom& g, int m, int& b, int& p, int& q) {
    int cls = m / g.Mc;
    int r = m - cls * g.Mc;
    int cyi = cls / g.PX;
    int cxi = cls - cyi * g.PX;
    int hw = g.OHc * g.OWc;
    b = r / hw;
    int rem = r - b * hw;
    int p2 = rem / g.OWc;
    int q2 = rem - p2 * g.OWc;
    p = p2 * g.PY + cyi;
    q = q2 * g.PX + cxi;
}

struct TapList {  // lives in LDS: the taps a tile iterates, as a product of a ky list and a kx list
    int ky[MAXTAP];
    int kx[MAXTAP];
    int nvy, nvx;
    unsigned masky, maskx;
};

// hi/lo bf16 split of a float4 -> two packed 8-byte values
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4(const f32x4& x, u32x2& hi, u32x2& lo) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float a0 = x[2 * j], a1 = x[2 * j + 1];
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a0, a1}, bf16x2_t));
        const float f0 = __builtin_bit_cast(float, h << 16);
        const float f1 = __builtin_bit_cast(float, h & 0xffff0000u);
        hi[j] = h;
        lo[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a0 - f0, a1 - f1}, bf16x2_t));
    }
}

// ---- gathered-operand tile loaders: tile[row][k], rows = output positions, k = (tap, channel) ----
template <int BM, int BKT, int DD>
struct LoaderV4 {  // C % 4 == 0: one 16-byte load per (row, 4 channels)
    static constexpr int SLOTS = BKT / 4;
    static constexpr int RPP = 256 / SLOTS;
    static constexpr int NP = BM / RPP;
    int slot4, r0;
    int rbase[NP], rpy[NP], rqx[NP];
    f32x4 regs[NP];   // raw loaded data: first touched in store(), so the loads stay in flight
    unsigned okmask;  // bit j: row j of the pending chunk hit a real pixel

    __device__ __forceinline__ void init(int tid) {
        slot4 = 4 * (tid % SLOTS);
        r0 = tid / SLOTS;
    }
    __device__ __forceinline__ void set_row(const Geom& g, int j, int m) {
        if (m < g.M) {
            int b, p, q;
            decode_row(g, m, b, p, q);
            rbase[j] = b * g.IH * g.IW * g.C;
            rpy[j] = p * g.a + g.off;
            rqx[j] = q * g.a + g.offx;
        } else {
            rbase[j] = 0;
            rpy[j] = ROW_INVALID;
            rqx[j] = ROW_INVALID;
        }
    }
    __device__ __forceinline__ void set_rows(const Geom& g, int m0) {
#pragma unroll
        for (int j = 0; j < NP; ++j) set_row(g, j, m0 + r0 + j * RPP);
    }
    // natural row order only (weight gradient): advance every row by 128 = sb images + sp rows + sq pixels
    __device__ __forceinline__ void advance_rows(const Geom& g, int m_next0, int sb, int sp, int sq) {
        const int img = g.IH * g.IW * g.C;
        const int wlim = g.OW * g.a + g.offx, hlim = g.OH * g.a + g.off;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            if (m_next0 + r0 + j * RPP >= g.M) {
                rpy[j] = ROW_INVALID;
                rqx[j] = ROW_INVALID;
                continue;
            }
            int x = rqx[j] + sq * g.a;
            int y = rpy[j] + sp * g.a;
            int bb = rbase[j] + sb * img;
            if (x >= wlim) {
                x -= g.OW * g.a;
                y += g.a;
            }
            if (y >= hlim) {
                y -= g.OH * g.a;
                bb += img;
            }
            rqx[j] = x;
            rpy[j] = y;
            rbase[j] = bb;
        }
    }
    // which ky / kx can reach a valid source pixel from any of this thread's rows
    __device__ __forceinline__ void tap_masks(const Geom& g, unsigned& my, unsigned& mx) const {
        my = mx = 0u;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            int s;
            for (int t = 0; t < g.KH; ++t)
                if (coord_ok<DD>(rpy[j] + t * g.cs, g.d, g.IH, s)) my |= 1u << t;
            for (int t = 0; t < g.KW; ++t)
                if (coord_ok<DD>(rqx[j] + t * g.cs, g.d, g.IW, s)) mx |= 1u << t;
        }
    }
    // one chunk: tap (ky, kx) and first channel cbeg are what THIS thread loads (uniform in TU modes)
    __device__ __forceinline__ void load_tap(const Geom& g, const float* __restrict__ in, int ky, int kx, int cbeg,
                                             bool kok) {
        const int dyv = ky * g.cs, dxv = kx * g.cs;
        const int c = cbeg + slot4;
        unsigned okm = 0u;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            int sy, sx;
            const bool oky = coord_ok<DD>(rpy[j] + dyv, g.d, g.IH, sy);
            const bool okx = coord_ok<DD>(rqx[j] + dxv, g.d, g.IW, sx);
            const bool ok = (int(oky) & int(okx) & int(kok)) != 0;
            // branch-free: always load (offset c of image 0 when the tap misses), then select.  A
            // conditional load would put an s_waitcnt behind every row and serialise the gather.
            const int o = ok ? rbase[j] + (sy * g.IW + sx) * g.C + c : c;
            regs[j] = *reinterpret_cast<const f32x4*>(in + o);
            okm |= (ok ? 1u : 0u) << j;
        }
        okmask = okm;
    }
    // generic chunk: per-thread tap decode from the flattened k index
    __device__ __forceinline__ void load_flat(const Geom& g, const float* __restrict__ in, int kk0, int keff,
                                              const TapList* tl) {
        int kk = kk0 + slot4;
        bool kok = kk < keff;
        int tj = kk / g.C;
        int c = kk - tj * g.C;
        int jy = tj / tl->nvx;
        int ky = kok ? tl->ky[jy] : 0;
        int kx = kok ? tl->kx[tj - jy * tl->nvx] : 0;
        load_tap(g, in, ky, kx, c - slot4, kok);
    }
    // zero the misses, apply the pending input activation (one uniform branch), write the tile
    __device__ __forceinline__ void store(const Geom& g, float* tile, int ld) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const bool ok = (okmask >> j) & 1u;
#pragma unroll
            for (int e = 0; e < 4; ++e) regs[j][e] = ok ? regs[j][e] : 0.f;
        }
        if (g.in_act == PM_ACT_RELU) {
#pragma unroll
            for (int j = 0; j < NP; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) regs[j][e] = fmaxf(regs[j][e], 0.f);
        } else if (g.in_act == PM_ACT_LEAKY) {
#pragma unroll
            for (int j = 0; j < NP; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) regs[j][e] = regs[j][e] >= 0.f ? regs[j][e] : g.slope * regs[j][e];
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) *reinterpret_cast<f32x4*>(tile + (r0 + j * RPP) * ld + slot4) = regs[j];
    }
    // same, but the tile is written as two bf16 planes (hi, lo) with row stride ld elements
    __device__ __forceinline__ void store_split(const Geom& g, short* hi, short* lo, int ld) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const bool ok = (okmask >> j) & 1u;
            f32x4 v = regs[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = ok ? v[e] : 0.f;
                if (g.in_act == PM_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
                else if (g.in_act == PM_ACT_LEAKY) v[e] = v[e] >= 0.f ? v[e] : g.slope * v[e];
            }
            u32x2 h2, l2;
            split4(v, h2, l2);
            *reinterpret_cast<u32x2*>(hi + (r0 + j * RPP) * ld + slot4) = h2;
            *reinterpret_cast<u32x2*>(lo + (r0 + j * RPP) * ld + slot4) = l2;
        }
    }
};

template <int BM, int BKT>
struct LoaderV1 {  // any C (used for C = 1, 2): scalar gathers, k flattened over (tap, c); no tap culling
    static_assert(BKT == 32, "scalar loader walks 32 k per chunk");
    static constexpr int RPT = BM / 8;  // consecutive rows per thread
    int kslot, rg;
    int m_first;
    float regs[RPT];

    __device__ __forceinline__ void init(int tid) {
        kslot = tid & 31;
        rg = tid >> 5;
    }
    __device__ __forceinline__ void set_rows(const Geom&, int m0) { m_first = m0 + rg * RPT; }
    __device__ __forceinline__ void advance_rows(const Geom&, int m_next0, int, int, int) { m_first = m_next0 + rg * RPT; }
    __device__ __forceinline__ void tap_masks(const Geom& g, unsigned& my, unsigned& mx) const {
        my = (1u << g.KH) - 1u;
        mx = (1u << g.KW) - 1u;
    }
    __device__ __forceinline__ void load_tap(const Geom&, const float*, int, int, int, bool) {}
    __device__ __forceinline__ void load_flat(const Geom& g, const float* __restrict__ in, int kk0, int keff,
                                              const TapList* tl) {
        int kk = kk0 + kslot;
        bool kok = kk < keff;
        int tj = kk / g.C;
        int c = kk - tj * g.C;
        int jy = tj / tl->nvx;
        int ky = kok ? tl->ky[jy] : 0;
        int kx = kok ? tl->kx[tj - jy * tl->nvx] : 0;
        int b, p, q;
        decode_row(g, m_first < g.M ? m_first : 0, b, p, q);
        const bool natural = (g.PY == 1 && g.PX == 1);
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            float v = 0.f;
            int m = m_first + j;
            if (kok && m < g.M) {
                if (!natural) decode_row(g, m, b, p, q);
                int sy, sx;
                if (coord_ok<0>(p * g.a + g.off + ky * g.cs, g.d, g.IH, sy) &&
                    coord_ok<0>(q * g.a + g.offx + kx * g.cs, g.d, g.IW, sx)) {
                    v = pm_act(in[((b * g.IH + sy) * g.IW + sx) * g.C + c], g.in_act, g.slope);
                }
            }
            regs[j] = v;
            if (natural) {  // next row in (b, p, q) order
                if (++q == g.OW) {
                    q = 0;
                    if (++p == g.OH) {
                        p = 0;
                        ++b;
                    }
                }
            }
        }
    }
    __device__ __forceinline__ void store(const Geom&, float* tile, int ld) {
#pragma unroll
        for (int j = 0; j < RPT; ++j) tile[(rg * RPT + j) * ld + kslot] = regs[j];
    }
};

template <int BM, int BKT, int MODE>
struct LoaderSel {
    typedef LoaderV4<BM, BKT, (MODE == MODE_TU1 ? 1 : (MODE == MODE_TU2 ? 2 : 0))> type;
};
template <int BM, int BKT>
struct LoaderSel<BM, BKT, MODE_V1> {
    typedef LoaderV1<BM, BKT> type;
};

// Builds the tile's tap list from the per-thread reachability masks (all threads must call).
__device__ __forceinline__ void build_tap_list(const Geom& g, TapList* tl, unsigned my, unsigned mx, int tid) {
    if (tid == 0) {
        tl->masky = 0u;
        tl->maskx = 0u;
    }
    __syncthreads();
    if (my) atomicOr(&tl->masky, my);
    if (mx) atomicOr(&tl->maskx, mx);
    __syncthreads();
    if (tid == 0) {
        int ny = 0, nx = 0;
        for (int t = 0; t < g.KH; ++t)
            if (tl->masky >> t & 1u) tl->ky[ny++] = t;
        for (int t = 0; t < g.KW; ++t)
            if (tl->maskx >> t & 1u) tl->kx[nx++] = t;
        tl->nvy = ny;
        tl->nvx = nx > 0 ? nx : 1;
        if (nx == 0) tl->nvy = 0;
    }
    __syncthreads();
}

// -------------------------------- forward / data-gradient ------------------------------------
template <int BM, int BN, int MODE>
__global__ __launch_bounds__(256) void gather_gemm_kernel(GemmArgs p) {
    constexpr bool TU = (MODE == MODE_TU1 || MODE == MODE_TU2);
    constexpr int WM = BM / 32;
    constexpr int WN = 4 / WM;
    constexpr int RN = BN / (32 * WN);
    constexpr int NBE = BK * BN / 256;
    static_assert(RN >= 1, "tile too narrow for the wave layout");
    constexpr int TILE_F = (BM + BN) * LDS_LD;
    constexpr int TL_F = (sizeof(TapList) + 3) / 4;
    __shared__ __attribute__((aligned(16))) float smem[TILE_F + BM + TL_F];
    float* As = smem;
    float* Bs = smem + BM * LDS_LD;
    int* rowoff32 = reinterpret_cast<int*>(smem + TILE_F);  // B*OH*OW*N < 2^31 is checked on the host
    TapList* tl = reinterpret_cast<TapList*>(smem + TILE_F + BM);

    const Geom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 31;
    const int h = lane >> 5;
    const int wm = wave % WM;
    const int wn = wave / WM;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int grp = blockIdx.z / p.ksplit;
    const int ks = blockIdx.z - grp * p.ksplit;
    const float* in = p.in + (size_t)grp * p.in_gs;
    const float* w = p.w + (size_t)grp * p.w_gs;

    typename LoaderSel<BM, BK, MODE>::type la;
    la.init(tid);
    la.set_rows(g, m0);
    {
        unsigned my, mx;
        la.tap_masks(g, my, mx);
        build_tap_list(g, tl, my, mx, tid);
    }
    // output offsets of the tile's rows (class-major row order -> NHWC address), -1 = no such row
    if (tid < BM) {
        int m = m0 + tid;
        int o = -1;
        if (m < g.M) {
            int b, pp, q;
            decode_row(g, m, b, pp, q);
            o = ((b * g.OH + pp) * g.OW + q) * g.N;
        }
        rowoff32[tid] = o;
    }
    __syncthreads();
    const int nvx = tl->nvx;
    const int keff = tl->nvy * nvx * g.C;

    // weight tile: Bs[n][k] = w[tap][c0 + k][n0 + n]; lanes run along the contiguous weight axis
    const bool ncontig = (g.wns == 1);
    int boff[NBE];
    int bkl[NBE];
    float breg[NBE];
#pragma unroll
    for (int j = 0; j < NBE; ++j) {
        int kl, nl;
        if (ncontig) {
            nl = tid % BN;
            kl = tid / BN + (256 / BN) * j;
        } else {
            kl = tid & 31;
            nl = (tid >> 5) + 8 * j;
        }
        bkl[j] = kl | (nl << 8);
        boff[j] = (n0 + nl < g.N) ? kl * g.wcs + (n0 + nl) * g.wns : -1;
    }
    auto load_b_tap = [&](int tap, int cbeg) {  // uniform tap: one scalar base per chunk
        const float* wb = w + ((size_t)tap * g.wts + (size_t)cbeg * g.wcs);
#pragma unroll
        for (int j = 0; j < NBE; ++j) breg[j] = wb[boff[j] >= 0 ? boff[j] : 0];  // masked in store_b
    };
    auto load_b_flat = [&](int kk0) {
#pragma unroll
        for (int j = 0; j < NBE; ++j) {
            int kl = bkl[j] & 255;
            int kk = kk0 + kl;
            float v = 0.f;
            if (kk < keff && boff[j] >= 0) {
                int tj = kk / g.C;
                int c = kk - tj * g.C;
                int jy = tj / nvx;
                int tap = tl->ky[jy] * g.kws + tl->kx[tj - jy * nvx];
                v = w[(size_t)tap * g.wts + (size_t)c * g.wcs + (size_t)(boff[j] - kl * g.wcs)];
            }
            breg[j] = v;
        }
    };
    auto store_b = [&]() {
#pragma unroll
        for (int j = 0; j < NBE; ++j) Bs[(bkl[j] >> 8) * LDS_LD + (bkl[j] & 255)] = boff[j] >= 0 ? breg[j] : 0.f;
    };

    f32x16 acc[RN];
#pragma unroll
    for (int r = 0; r < RN; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

    const int nchunks = (keff + BK - 1) / BK;
    const int cb = (int)(((long long)nchunks * ks) / p.ksplit);
    const int ce = (int)(((long long)nchunks * (ks + 1)) / p.ksplit);
    // chunk cursor of the tap-uniform modes: (jy, jx) index the tap lists, c0 is the channel offset
    int jy = 0, jx = 0, c0 = 0;
    if (TU) {
        int tj = (cb * BK) / g.C;
        c0 = cb * BK - tj * g.C;
        jy = tj / nvx;
        jx = tj - jy * nvx;
    }
    auto load_chunk = [&](int ch) {
        if constexpr (TU) {
            const int ky = tl->ky[jy], kx = tl->kx[jx];
            la.load_tap(g, in, ky, kx, c0, true);
            load_b_tap(ky * g.kws + kx, c0);
            c0 += BK;
            if (c0 >= g.C) {
                c0 = 0;
                if (++jx == nvx) {
                    jx = 0;
                    ++jy;
                }
            }
        } else {
            la.load_flat(g, in, ch * BK, keff, tl);
            load_b_flat(ch * BK);
        }
    };
    if (cb < ce) load_chunk(cb);
    for (int ch = cb; ch < ce; ++ch) {
        la.store(g, As, LDS_LD);
        store_b();
        __syncthreads();
        if (ch + 1 < ce) load_chunk(ch + 1);
        const float* arow = As + (wm * 32 + i) * LDS_LD + 4 * h;
        const float* brow = Bs + (wn * RN * 32 + i) * LDS_LD + 4 * h;
#pragma unroll
        for (int u = 0; u < BK / 8; ++u) {
            f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 8 * u);
#pragma unroll
            for (int r = 0; r < RN; ++r) {
                f32x4 b4 = *reinterpret_cast<const f32x4*>(brow + r * 32 * LDS_LD + 8 * u);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc[r], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const float* bias = p.bias ? p.bias + (size_t)grp * p.bias_gs : nullptr;
    const float* aux = p.aux ? p.aux + (size_t)grp * p.out_gs : nullptr;
    const float* res = p.res ? p.res + (size_t)grp * p.out_gs : nullptr;
    float* out = p.out + (size_t)grp * p.out_gs;
    int ro[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) ro[e] = rowoff32[wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h];
#pragma unroll
    for (int r = 0; r < RN; ++r) {
        int n = n0 + (wn * RN + r) * 32 + i;
        if (n >= g.N) continue;
        float bv = bias ? bias[n] : 0.f;
        if (p.ksplit > 1) {  // partial sum of a K slice: the epilogue kernel finishes the tile
            float* slab = p.sk ? p.sk + (size_t)ks * p.sk_stride : nullptr;
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (ro[e] >= 0) {
                    if (slab) slab[(size_t)ro[e] + n] = acc[r][e];
                    else atomicAdd(out + (size_t)ro[e] + n, acc[r][e]);
                }
            continue;
        }
        pm_epilogue_tile(acc[r], ro, n, bv, aux, res, out, nullptr, PM_ACT_NONE, g.aux_act, g.out_act, g.slope);
    }
}

// epilogue of the split-K form, over the finished sums: out = act((out + bias) * act'(aux) + res)
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(GemmArgs p, long long total_per_group) {
    const Geom& g = p.g;
    const int grp = blockIdx.y;
    const float* bias = p.bias ? p.bias + (size_t)grp * p.bias_gs : nullptr;
    const float* aux = p.aux ? p.aux + (size_t)grp * p.out_gs : nullptr;
    const float* res = p.res ? p.res + (size_t)grp * p.out_gs : nullptr;
    float* out = p.out + (size_t)grp * p.out_gs;
    const long long stride = (long long)gridDim.x * 256;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total_per_group; o += stride) {
        int n = (int)(o % g.N);
        float sum;
        if (p.sk) {                                    // K slices in slabs: added in slice order (groups == 1 with split-K)
            sum = 0.f;
            for (int ks = 0; ks < p.ksplit; ++ks) sum += p.sk[(size_t)ks * p.sk_stride + o];
        } else {
            sum = out[o];
        }
        float v = sum + (bias ? bias[n] : 0.f);
        v = pm_epilogue(v, aux, res, o, g.aux_act, g.out_act, g.slope);
        out[o] = v;
        if (p.out2) p.out2[(size_t)grp * p.out_gs + o] = pm_act(v, p.act2, g.slope);
    }
}

// ----------------------- forward / data-gradient, direct-fragment form ------------------------
// On gfx950 the f32 MFMA runs at the f32 VALU rate and (measured: kernel time = MFMA time + time
// of everything else) does not overlap with VALU work, so every address / select / LDS-staging
// instruction of the gathered operand is paid in full.  This form therefore never stages the
// gathered operand: each lane loads its MFMA A fragment (row = its output position, 4 consecutive
// channels) straight from global memory with a bounds-checked buffer load - a tap that misses the
// image gets an out-of-range offset and the hardware returns zeros - at a cost of ~10 VALU
// instructions per 16 MFMAs.  Weights are the only LDS traffic: GS k-steps (one k-step = one tap x
// 32 channels) are staged per barrier, double-buffered.  Needs C % 32 == 0.
constexpr int DGS = 4;          // k-steps staged per barrier
constexpr int DMAXSTEPS = 256;  // k-step descriptors kept in LDS

struct KStep {
    int dy, dx;  // tap displacement in the index rule (ky*cs, kx*cs)
    int c0;      // first channel of the step
    int woff;    // weight offset tap*wts + c0*wcs
};

template <int RN, int DD, int IN_ACT>
__global__ __launch_bounds__(256) void direct_gemm_kernel(GemmArgs p) {
    constexpr int NB = 32 * RN;
    constexpr int BTILE = DGS * NB * LDS_LD;            // floats of one weight stage
    constexpr int EPS = NB * BK / 256;                  // weight elements per thread per k-step
    constexpr int TL_F = (sizeof(TapList) + 3) / 4;
    constexpr int KD_F = (DMAXSTEPS + 8) * (sizeof(KStep) / 4);   // + zero entries the prefetches run into (written up to nsteps + DGS)
    __shared__ __attribute__((aligned(16))) float smem[2 * BTILE + KD_F + TL_F];
    float* Bs = smem;
    KStep* kd = reinterpret_cast<KStep*>(smem + 2 * BTILE);
    TapList* tl = reinterpret_cast<TapList*>(smem + 2 * BTILE + KD_F);

    const Geom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 31;
    const int h = lane >> 5;
    const int n0 = blockIdx.y * NB;
    const int grp = blockIdx.z / p.ksplit;
    const int ks = blockIdx.z - grp * p.ksplit;
    const float* in = p.in + (size_t)grp * p.in_gs;
    const float* w = p.w + (size_t)grp * p.w_gs;

    // this lane's output row (class-major) and its source-coordinate bases
    const int m = blockIdx.x * 128 + wave * 32 + i;
    int rbase = 0, rpy = ROW_INVALID, rqx = ROW_INVALID, rowoff = -1;
    if (m < g.M) {
        int b, pp, q;
        decode_row(g, m, b, pp, q);
        rbase = b * g.IH * g.IW * g.C + 4 * h;
        rpy = pp * g.a + g.off;
        rqx = q * g.a + g.offx;
        rowoff = ((b * g.OH + pp) * g.OW + q) * g.N;
    }
    {
        unsigned my = 0u, mx = 0u;
        int s;
        for (int t = 0; t < g.KH; ++t)
            if (coord_ok<DD>(rpy + t * g.cs, g.d, g.IH, s)) my |= 1u << t;
        for (int t = 0; t < g.KW; ++t)
            if (coord_ok<DD>(rqx + t * g.cs, g.d, g.IW, s)) mx |= 1u << t;
        build_tap_list(g, tl, my, mx, tid);
    }
    const int cchunks = g.C / BK;
    const int nsteps_all = tl->nvy * tl->nvx * cchunks;
    for (int s = tid; s < nsteps_all + DGS; s += 256) {  // DGS padding entries: safe reads past the end
        KStep k{0, 0, 0, 0};
        if (s < nsteps_all) {
            int tj = s / cchunks;
            int cc = s - tj * cchunks;
            int jy = tj / tl->nvx;
            int ky = tl->ky[jy], kx = tl->kx[tj - jy * tl->nvx];
            k.dy = ky * g.cs;
            k.dx = kx * g.cs;
            k.c0 = cc * BK;
            k.woff = (ky * g.kws + kx) * g.wts + cc * BK * g.wcs;
        }
        kd[s] = k;
    }
    __syncthreads();
    const int sb = (int)(((long long)nsteps_all * ks) / p.ksplit);
    const int se = (int)(((long long)nsteps_all * (ks + 1)) / p.ksplit);

    // bounds-checked view of the gathered tensor: offsets past the end read as zero
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(in), 0, (int)((long long)g.B * g.IH * g.IW * g.C * 4), 0x00020000);
    constexpr int OOB = 0x7ffffff0;
    f32x4 af[4][4];
    auto issue_a = [&](int s, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        const KStep k = kd[s];
        int sy, sx;
        const bool oky = coord_ok<DD>(rpy + k.dy, g.d, g.IH, sy);
        const bool okx = coord_ok<DD>(rqx + k.dx, g.d, g.IW, sx);
        const int off = (rbase + (sy * g.IW + sx) * g.C + k.c0) * 4;
        const int voff = (int(oky) & int(okx)) ? off : OOB;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            af[SET][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 32 * u, 0));
    };

    // weight stage loader: Bs[stage][step][n][k] <- w[woff(step) + k*wcs + (n0+n)*wns].  Columns
    // past N and steps past the end load a safe address instead of being masked: their products
    // land in accumulator columns / steps that are never stored or never executed.
    const bool ncontig = (g.wns == 1);
    int eo[EPS];     // element offset inside a k-step's weight block
    int lo[EPS];     // LDS offset inside a step's Bs block
#pragma unroll
    for (int j = 0; j < EPS; ++j) {
        int e = tid + 256 * j;
        int nl, kl;
        if (ncontig) {
            nl = e % NB;
            kl = e / NB;
        } else {
            kl = e % BK;
            nl = e / BK;
        }
        eo[j] = kl * g.wcs + (n0 + nl < g.N ? n0 + nl : 0) * g.wns;
        lo[j] = nl * LDS_LD + kl;
    }
    float breg[DGS][EPS];
    auto load_b = [&](int s0) {
#pragma unroll
        for (int st = 0; st < DGS; ++st) {
            const float* wb = w + kd[s0 + st].woff;   // kd is padded: s0 + st < nsteps_all + DGS
#pragma unroll
            for (int j = 0; j < EPS; ++j) breg[st][j] = wb[eo[j]];
        }
    };
    auto store_b = [&](float* dst) {
#pragma unroll
        for (int st = 0; st < DGS; ++st)
#pragma unroll
            for (int j = 0; j < EPS; ++j) dst[st * NB * LDS_LD + lo[j]] = breg[st][j];
    };

    f32x16 acc[RN];
#pragma unroll
    for (int r = 0; r < RN; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

    auto mma_step = [&](const float* bstep, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        if constexpr (IN_ACT == PM_ACT_RELU) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) af[SET][u][e] = fmaxf(af[SET][u][e], 0.f);
        } else if constexpr (IN_ACT == PM_ACT_LEAKY) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    af[SET][u][e] = af[SET][u][e] >= 0.f ? af[SET][u][e] : g.slope * af[SET][u][e];
        }
        const float* brow = bstep + i * LDS_LD + 4 * h;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int r = 0; r < RN; ++r) {
                f32x4 b4 = *reinterpret_cast<const f32x4*>(brow + r * 32 * LDS_LD + 8 * u);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[SET][u][e], b4[e], acc[r], 0, 0, 0);
            }
        }
    };

    // A fragments run PD = 2 k-steps ahead of the MFMAs in 4 register sets (set = step index in its
    // stage, so every access is a compile-time register): one step of MFMAs (1024 cycles) does not
    // cover an L2 round trip when the whole chip is streaming.
    if (sb < se) {
        load_b(sb);
        store_b(Bs);
        issue_a(sb, std::integral_constant<int, 0>{});
        if (sb + 1 < se) issue_a(sb + 1, std::integral_constant<int, 1>{});
    }
    __syncthreads();
    int stage = 0;
    for (int s0 = sb; s0 < se; s0 += DGS) {
        const float* bcur = Bs + stage * BTILE;
        const bool more = s0 + DGS < se;
        if (more) load_b(s0 + DGS);  // stays in flight during the MFMAs below
        if (s0 + 2 < se) issue_a(s0 + 2, std::integral_constant<int, 2>{});
        mma_step(bcur, std::integral_constant<int, 0>{});
        if (s0 + 1 < se) {
            if (s0 + 3 < se) issue_a(s0 + 3, std::integral_constant<int, 3>{});
            mma_step(bcur + NB * LDS_LD, std::integral_constant<int, 1>{});
        }
        if (s0 + 2 < se) {
            if (s0 + 4 < se) issue_a(s0 + 4, std::integral_constant<int, 0>{});
            mma_step(bcur + 2 * NB * LDS_LD, std::integral_constant<int, 2>{});
        }
        if (s0 + 3 < se) {
            if (s0 + 5 < se) issue_a(s0 + 5, std::integral_constant<int, 1>{});
            mma_step(bcur + 3 * NB * LDS_LD, std::integral_constant<int, 3>{});
        }
        if (more) store_b(Bs + (stage ^ 1) * BTILE);
        __syncthreads();
        stage ^= 1;
    }

    // epilogue (C/D layout: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)); the row a
    // register belongs to lives in another lane: fetch its output offset with a wave shuffle
    const float* bias = p.bias ? p.bias + (size_t)grp * p.bias_gs : nullptr;
    const float* aux = p.aux ? p.aux + (size_t)grp * p.out_gs : nullptr;
    const float* res = p.res ? p.res + (size_t)grp * p.out_gs : nullptr;
    float* out = p.out + (size_t)grp * p.out_gs;
    int ro[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) ro[e] = __shfl(rowoff, (e & 3) + 8 * (e >> 2) + 4 * h, 64);
#pragma unroll
    for (int r = 0; r < RN; ++r) {
        int n = n0 + r * 32 + i;
        if (n >= g.N) continue;
        float bv = bias ? bias[n] : 0.f;
        if (p.ksplit > 1) {
            float* slab = p.sk ? p.sk + (size_t)ks * p.sk_stride : nullptr;
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (ro[e] >= 0) {
                    if (slab) slab[(size_t)ro[e] + n] = acc[r][e];
                    else atomicAdd(out + (size_t)ro[e] + n, acc[r][e]);
                }
            continue;
        }
        pm_epilogue_tile(acc[r], ro, n, bv, aux, res, out, nullptr, PM_ACT_NONE, g.aux_act, g.out_act, g.slope);
    }
}

template <int RN, int DD>
void launch_direct(hipStream_t s, const GemmArgs& a, dim3 grid) {
    PM_KTAG("direct_gemm_kernel<%d, %d, %d>", RN, DD, a.g.in_act == PM_ACT_RELU || a.g.in_act == PM_ACT_LEAKY ? a.g.in_act : 0);
    switch (a.g.in_act) {
        case PM_ACT_RELU: hipLaunchKernelGGL((direct_gemm_kernel<RN, DD, PM_ACT_RELU>), grid, dim3(256), 0, s, a); break;
        case PM_ACT_LEAKY: hipLaunchKernelGGL((direct_gemm_kernel<RN, DD, PM_ACT_LEAKY>), grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((direct_gemm_kernel<RN, DD, PM_ACT_NONE>), grid, dim3(256), 0, s, a); break;
    }
}

// ------------------ forward / data-gradient, direct form on the bf16 matrix cores -------------
// The f32 "MFMA" of gfx950 runs on the VALU pipeline (measured: 16 MFMAs + N v_fma take the SUM of
// their times, on one wave or across waves), so it can never beat ~64 FLOP/clk/SIMD and every
// address instruction is stolen from it.  The bf16 MFMA is a separate pipe, 16x faster.  To keep
// float32-grade results each operand is split into two bf16 terms, x = hi + lo with
// hi = bf16(x), lo = bf16(x - hi) (16 mantissa bits kept), and three products are accumulated in
// f32:  a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  (the dropped a_lo*b_lo term is ~2^-18 |ab|).
// Relative error of a dot product ~1e-5 instead of ~1e-7: far inside the 1e-3 parity bar.
// Weights arrive pre-split and K-contiguous (pm_split_weights, once per optimizer step); the
// gathered operand is split in registers right after its buffer load.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BROW = 40;  // LDS row of a weight stage: 32 bf16 (64 B) + 16 B pad, in 2-byte units -> 80 B

// split 8 floats into packed bf16 hi and lo vectors (2 elements per dword)
__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, bf16x8& hi, bf16x8& lo) {
    u32x4 hp, lp;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a0 = j < 2 ? x0[2 * j] : x1[2 * j - 4];
        const float a1 = j < 2 ? x0[2 * j + 1] : x1[2 * j - 3];
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a0, a1}, bf16x2));
        const float f0 = __builtin_bit_cast(float, h << 16);
        const float f1 = __builtin_bit_cast(float, h & 0xffff0000u);
        hp[j] = h;
        lp[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a0 - f0, a1 - f1}, bf16x2));
    }
    hi = __builtin_bit_cast(bf16x8, hp);
    lo = __builtin_bit_cast(bf16x8, lp);
}

struct KStepB {
    int dy, dx;
    int c0;
    int woff;  // offset (in bf16 elements) of this step's [Npad][32] hi block inside the split weights
};

// DENSE: 1x1 problem on a 1x1 grid (hk.Linear): no taps, no coordinates - rows are plain matrix rows.
template <int RN, int DD, int IN_ACT, bool DENSE>
__global__ __launch_bounds__(256) void direct_gemm_bf16_kernel(GemmArgs p, const __bf16* __restrict__ wsplit,
                                                               int npad, long long plane) {
    constexpr int NB = 32 * RN;
    constexpr int STEP_E = 2 * NB * BROW;               // bf16 elements of one k-step in LDS (hi rows, lo rows)
    constexpr int BTILE = DGS * STEP_E;                 // one stage
    constexpr int TL_F = (sizeof(TapList) + 3) / 4;
    constexpr int KD_F = (DMAXSTEPS + 8) * (sizeof(KStepB) / 4);   // + zero entries the prefetches run into
    __shared__ __attribute__((aligned(16))) float smem[BTILE + KD_F + TL_F];   // 2 stages of bf16 = BTILE floats
    __bf16* Bs = reinterpret_cast<__bf16*>(smem);
    KStepB* kd = reinterpret_cast<KStepB*>(smem + BTILE);
    TapList* tl = reinterpret_cast<TapList*>(smem + BTILE + KD_F);

    const Geom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 31;
    const int h = lane >> 5;
    const int n0 = blockIdx.y * NB;
    const int grp = blockIdx.z / p.ksplit;
    const int ks = blockIdx.z - grp * p.ksplit;
    const float* in = p.in + (size_t)grp * p.in_gs;

    const int m = blockIdx.x * 128 + wave * 32 + i;
    int rbase = 0, rpy = ROW_INVALID, rqx = ROW_INVALID, rowoff = -1;
    // channels are walked in chunks of 32; when C % 32 != 0 the last chunk of a tap reads past the row's
    // channels (the next pixel, or zeros past the tensor end) against zero-padded weights
    const int cchunks = (g.C + BK - 1) / BK;
    int nsteps_all;
    if constexpr (DENSE) {
        if (m < g.M) {
            rbase = m * g.C + 8 * h;
            rpy = rqx = 0;
            rowoff = m * g.N;
        }
        nsteps_all = cchunks;
    } else {
        if (m < g.M) {
            int b, pp, q;
            decode_row(g, m, b, pp, q);
            rbase = b * g.IH * g.IW * g.C + 8 * h;
            rpy = pp * g.a + g.off;
            rqx = q * g.a + g.offx;
            rowoff = ((b * g.OH + pp) * g.OW + q) * g.N;
        }
        unsigned my = 0u, mx = 0u;
        int s;
        for (int t = 0; t < g.KH; ++t)
            if (coord_ok<DD>(rpy + t * g.cs, g.d, g.IH, s)) my |= 1u << t;
        for (int t = 0; t < g.KW; ++t)
            if (coord_ok<DD>(rqx + t * g.cs, g.d, g.IW, s)) mx |= 1u << t;
        build_tap_list(g, tl, my, mx, tid);
        nsteps_all = tl->nvy * tl->nvx * cchunks;
    }
    for (int s = tid; s < nsteps_all + DGS; s += 256) {
        KStepB k{0, 0, 0, 0};
        if (DENSE) {
            if (s < nsteps_all) {
                k.c0 = s * BK;
                k.woff = s * npad * BK;
            }
        } else if (s < nsteps_all) {
            int tj = s / cchunks;
            int cc = s - tj * cchunks;
            int jy = tj / tl->nvx;
            int ky = tl->ky[jy], kx = tl->kx[tj - jy * tl->nvx];
            k.dy = ky * g.cs;
            k.dx = kx * g.cs;
            k.c0 = cc * BK;
            k.woff = ((ky * g.KW + kx) * cchunks + cc) * npad * BK;
        }
        kd[s] = k;
    }
    __syncthreads();
    const int sb = (int)(((long long)nsteps_all * ks) / p.ksplit);
    const int se = (int)(((long long)nsteps_all * (ks + 1)) / p.ksplit);

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(in), 0, (int)((long long)g.B * g.IH * g.IW * g.C * 4), 0x00020000);
    constexpr int OOB = 0x7ffffff0;
    // lane (r, h) needs channels c0 + 16*kk + 8h + 0..7 for the two k16 MFMA steps kk = 0, 1
    f32x4 af[4][4];
    auto issue_a = [&](int s, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        const KStepB k = kd[s];
        int sy, sx;
        const bool oky = coord_ok<DD>(rpy + k.dy, g.d, g.IH, sy);
        const bool okx = coord_ok<DD>(rqx + k.dx, g.d, g.IW, sx);
        const int off = (rbase + (sy * g.IW + sx) * g.C + k.c0) * 4;
        const int voff = (int(oky) & int(okx)) ? off : OOB;
        af[SET][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
        af[SET][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 16, 0));
        af[SET][2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 64, 0));
        af[SET][3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 80, 0));
    };

    // weight stage: per k-step 2*NB rows (hi then lo) of 64 bytes; 16 bytes per thread per piece
    constexpr int PIECES = 2 * NB * 4;            // 16-byte pieces per k-step
    constexpr int PPT = (PIECES + 255) / 256;     // pieces per thread per k-step (1 for NB=32, 2 for NB=64)
    const __bf16* wg = wsplit + (size_t)grp * p.w_gs;
    u32x4 breg[DGS][PPT];
    auto load_b = [&](int s0) {
#pragma unroll
        for (int st = 0; st < DGS; ++st) {
            const int wo = kd[s0 + st].woff;
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                const int pc = tid + 256 * j;          // piece id: plane, row, quarter
                const int pl = pc / (NB * 4);
                const int row = (pc / 4) % NB;
                const int qtr = pc & 3;
                const int nrow = n0 + row < npad ? n0 + row : 0;
                const __bf16* src = wg + (size_t)pl * plane + wo + nrow * BK + qtr * 8;
                breg[st][j] = *reinterpret_cast<const u32x4*>(src);
            }
        }
    };
    auto store_b = [&](__bf16* dst) {
#pragma unroll
        for (int st = 0; st < DGS; ++st)
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                const int pc = tid + 256 * j;
                const int pl = pc / (NB * 4);
                const int row = (pc / 4) % NB;
                const int qtr = pc & 3;
                *reinterpret_cast<u32x4*>(dst + st * STEP_E + (pl * NB + row) * BROW + qtr * 8) = breg[st][j];
            }
    };

    f32x16 acc[RN];
#pragma unroll
    for (int r = 0; r < RN; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

    auto mma_step = [&](const __bf16* bstep, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        if constexpr (IN_ACT == PM_ACT_RELU) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) af[SET][u][e] = fmaxf(af[SET][u][e], 0.f);
        } else if constexpr (IN_ACT == PM_ACT_LEAKY) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    af[SET][u][e] = af[SET][u][e] >= 0.f ? af[SET][u][e] : g.slope * af[SET][u][e];
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 ah, al;
            split8(af[SET][2 * kk], af[SET][2 * kk + 1], ah, al);
#pragma unroll
            for (int r = 0; r < RN; ++r) {
                const __bf16* brow = bstep + (r * 32 + i) * BROW + 16 * kk + 8 * h;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(brow);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(brow + NB * BROW);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[r], 0, 0, 0);
            }
        }
    };

    // A fragments run 3 k-steps ahead of the MFMAs in 4 register sets (set = step index % 4: the set a step
    // prefetches into was consumed by the step before it)
    if (sb < se) {
        load_b(sb);
        store_b(Bs);
        issue_a(sb, std::integral_constant<int, 0>{});
        if (sb + 1 < se) issue_a(sb + 1, std::integral_constant<int, 1>{});
        if (sb + 2 < se) issue_a(sb + 2, std::integral_constant<int, 2>{});
    }
    __syncthreads();
    int stage = 0;
    for (int s0 = sb; s0 < se; s0 += DGS) {
        const __bf16* bcur = Bs + stage * BTILE;
        const bool more = s0 + DGS < se;
        if (more) load_b(s0 + DGS);
        if (s0 + 3 < se) issue_a(s0 + 3, std::integral_constant<int, 3>{});
        mma_step(bcur, std::integral_constant<int, 0>{});
        if (s0 + 1 < se) {
            if (s0 + 4 < se) issue_a(s0 + 4, std::integral_constant<int, 0>{});
            mma_step(bcur + STEP_E, std::integral_constant<int, 1>{});
        }
        if (s0 + 2 < se) {
            if (s0 + 5 < se) issue_a(s0 + 5, std::integral_constant<int, 1>{});
            mma_step(bcur + 2 * STEP_E, std::integral_constant<int, 2>{});
        }
        if (s0 + 3 < se) {
            if (s0 + 6 < se) issue_a(s0 + 6, std::integral_constant<int, 2>{});
            mma_step(bcur + 3 * STEP_E, std::integral_constant<int, 3>{});
        }
        if (more) store_b(Bs + (stage ^ 1) * BTILE);
        __syncthreads();
        stage ^= 1;
    }

    const float* bias = p.bias ? p.bias + (size_t)grp * p.bias_gs : nullptr;
    const float* aux = p.aux ? p.aux + (size_t)grp * p.out_gs : nullptr;
    const float* res = p.res ? p.res + (size_t)grp * p.out_gs : nullptr;
    float* out = p.out + (size_t)grp * p.out_gs;
    int ro[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) ro[e] = __shfl(rowoff, (e & 3) + 8 * (e >> 2) + 4 * h, 64);
#pragma unroll
    for (int r = 0; r < RN; ++r) {
        int n = n0 + r * 32 + i;
        if (n >= g.N) continue;
        float bv = bias ? bias[n] : 0.f;
        if (p.ksplit > 1) {
            float* slab = p.sk ? p.sk + (size_t)ks * p.sk_stride : nullptr;
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (ro[e] >= 0) {
                    if (slab) slab[(size_t)ro[e] + n] = acc[r][e];
                    else atomicAdd(out + (size_t)ro[e] + n, acc[r][e]);
                }
            continue;
        }
        pm_epilogue_tile(acc[r], ro, n, bv, aux, res, out, p.out2 ? p.out2 + (size_t)grp * p.out_gs : nullptr, p.act2, g.aux_act,
                         g.out_act, g.slope);
    }
}

// ----------------------- whole-image layers with ONE output position and few rows (bf16x3) -----------------------
// The 7x7 layers at either end of the PM-VAE's convolution stacks (7x7x64 -> 1x1x128 forward, 7x7x64 -> 1x1x32 as a data
// gradient; reference networks.py conv stacks at configs/pm_vae_mnist.py sizes) are GEMMs of 256 rows with K = 3136: the
// direct form has 4 tiles, so it ran split 24 ways over K with a zero-fill in front and an epilogue launch behind (three
// dependent launches, 24 us of a dependent chain).  Here one launch does it: a workgroup owns a 16 x 16 output tile
// (16 x 8 = 128 workgroups for 256 x 128), its 8 waves take the k-steps (32 channels of one tap) round-robin - 12 or 13
// each, all of them in flight at once (13 register sets: 233 VGPRs, two waves per SIMD), A rows and pre-split weights straight from L2 into MFMA fragments - and the 8
// partial tiles meet in 8 KB of LDS; the shared epilogue (bias, act'(aux), residual, activation, second store) follows.
constexpr int SK_NW = 8;      // waves per workgroup = interleaved K slices
#ifndef PM_SK_PD
#define PM_SK_PD 13
#endif
constexpr int SK_PD = PM_SK_PD;   // k-steps in flight per wave (16 registers each)
template <class F, int... U>
__device__ __forceinline__ void sk_for(F&& f, std::integer_sequence<int, U...>) { (f(std::integral_constant<int, U>{}), ...); }

__global__ __launch_bounds__(64 * SK_NW) void skinny_gemm_bf16_kernel(GemmArgs p, const __bf16* __restrict__ wsplit,
                                                                      int npad, long long plane) {
    __shared__ float red[SK_NW][4][64];
    const Geom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kg = lane >> 4;        // A: row r, B: column r; both: k = 8 kg .. 8 kg + 7 of the 32-deep step
    const int m0 = blockIdx.x * 16, n0 = blockIdx.y * 16;
    const int cchunks = (g.C + BK - 1) / BK;       // C % 32 != 0: the last chunk of a tap is part zero-padded weights
    const int nsteps = g.KH * g.KW * cchunks;
    const int mrow = m0 + r < g.M ? m0 + r : g.M - 1;
    const int ncol = n0 + r < npad ? n0 + r : npad - 1;
    const float* arow = p.in + (size_t)mrow * g.IH * g.IW * g.C + 8 * kg;
    const __bf16* bcol = wsplit + (size_t)ncol * BK + 8 * kg;

    f32x4 a0[SK_PD], a1[SK_PD];
    bf16x8 bh[SK_PD], bl[SK_PD];
    auto issue = [&](int s, auto set_tag) {          // this wave's s-th step = global step wave + SK_NW * s (clamped)
        constexpr int U = decltype(set_tag)::value;
        int st = wave + SK_NW * s;
        st = st < nsteps ? st : nsteps - 1;
        const int tap = st / cchunks, cc = st - tap * cchunks;
        const int ky = tap / g.KW, kx = tap - ky * g.KW;
        const int pix = (g.off + ky * g.cs) * g.IW + g.offx + kx * g.cs;
        const bool kok = cc * BK + 8 * kg < g.C;       // C % 8 == 0: a lane's 8 channels are all inside the row or all outside
        const float* ap = arow + (size_t)pix * g.C + (kok ? cc * BK : 0);
        a0[U] = *reinterpret_cast<const f32x4*>(ap);
        a1[U] = *reinterpret_cast<const f32x4*>(ap + 4);
        if (!kok) a0[U] = a1[U] = f32x4{0.f, 0.f, 0.f, 0.f};
        const __bf16* bp = bcol + (size_t)st * npad * BK;
        bh[U] = *reinterpret_cast<const bf16x8*>(bp);
        bl[U] = *reinterpret_cast<const bf16x8*>(bp + plane);
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto mma = [&](auto set_tag, bool valid) {
        constexpr int U = decltype(set_tag)::value;
        bf16x8 ah, al;
        split8(a0[U], a1[U], ah, al);
        if (!valid) {                                 // a clamped step past this wave's share: zero operand, finite weights
            const u32x4 z = {0u, 0u, 0u, 0u};
            ah = __builtin_bit_cast(bf16x8, z);
            al = ah;
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[U], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[U], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[U], acc, 0, 0, 0);
    };
    const int my = wave < nsteps ? (nsteps - wave + SK_NW - 1) / SK_NW : 0;       // steps of this wave
    constexpr auto sets = std::make_integer_sequence<int, SK_PD>{};
    sk_for([&](auto u) { issue(decltype(u)::value, u); }, sets);
    int s0 = 0;
    for (; s0 + SK_PD < my; s0 += SK_PD)              // full groups: every step is this wave's, the next group's loads follow
        sk_for([&](auto u) {
            mma(u, true);
            issue(s0 + SK_PD + decltype(u)::value, u);
            __builtin_amdgcn_sched_barrier(0);        // or the scheduler sinks every load of the group behind its last MFMA
        }, sets);
    sk_for([&](auto u) { mma(u, s0 + decltype(u)::value < my); }, sets);

#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][e][lane] = acc[e];
    __syncthreads();
    if (tid < 256) {                                  // C/D layout: column = lane & 15, row = 4 (lane >> 4) + e
        const int e = tid >> 6;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < SK_NW; ++w) v += red[w][e][lane];
        const int row = m0 + 4 * (lane >> 4) + e, n = n0 + (lane & 15);
        if (row < g.M && n < g.N) {
            const size_t o = (size_t)row * g.N + n;
            if (p.bias) v += p.bias[n];
            v = pm_epilogue(v, p.aux, p.res, o, g.aux_act, g.out_act, g.slope);
            p.out[o] = v;
            if (p.out2) p.out2[o] = pm_act(v, p.act2, g.slope);
        }
    }
}

// one output position per image, every tap inside the image, few rows, long K
bool plan_skinny(const Geom& g, int groups) {
    static const bool off = getenv("PM_NO_SKINNY") != nullptr;         // A/B switch for measurements
    if (off || groups != 1 || g.d != 1 || g.OH != 1 || g.OW != 1 || g.C % 8 != 0 || g.in_act != PM_ACT_NONE) return false;
    if (g.M > 512 || g.KH * g.KW * g.C < 512) return false;
    // wide outputs (the partial encoder's 512 -> 32768 data gradient at the CelebA size: 2048 workgroups that each re-read the
    // 16 rows and meet in LDS for two k-steps per wave - 99 us for 67 MB of weights) stream better through the direct form
    static const int max_n = getenv("PM_SKINNY_MAXN") ? atoi(getenv("PM_SKINNY_MAXN")) : 2048;      // A/B knob
    if (g.N > max_n) return false;
    const int y0 = g.off, y1 = g.off + (g.KH - 1) * g.cs, x0 = g.offx, x1 = g.offx + (g.KW - 1) * g.cs;
    if (y0 < 0 || y1 < 0 || y0 >= g.IH || y1 >= g.IH || x0 < 0 || x1 < 0 || x0 >= g.IW || x1 >= g.IW) return false;
    return true;
}

// ----------------------- stride-1 convolutions, patch-staged form (bf16x3) -----------------------
// The direct form above re-reads every input element once per tap from L2 (25x for a 5x5 kernel: the
// measured limiter, FETCH_SIZE 1.6x the algorithmic bytes and 40 % of the kernel time in the gather).
// For a = d = 1 problems (stride-1 conv / transposed conv, forward and data-gradient) a workgroup owns a
// TH x TW tile of ONE image: the (TH+KH-1) x (TW+KW-1) x C input patch is loaded ONCE, split into hi / lo
// bf16 planes ONCE (zero outside the image), and every tap's MFMA A fragment is a shifted 16-byte LDS
// read; the k-loop has no global loads of the gathered operand, no bounds checks and no conversions.
// Weights: the same pre-split K-contiguous copy and double-buffered LDS stages as the direct form.
constexpr int PDGS = 2;   // k-steps per weight stage of the patch form: LDS is what limits its workgroups per CU

template <int RN>
__global__ __launch_bounds__(256) void patch_conv_bf16_kernel(GemmArgs p, const __bf16* __restrict__ wsplit, int npad,
                                                              long long plane, int tw_log2) {
    constexpr int NB = 32 * RN;
    constexpr int STEP_E = 2 * NB * BROW;
    constexpr int BTILE = PDGS * STEP_E;
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    __bf16* Bs = reinterpret_cast<__bf16*>(dsm);                        // 2 stages = BTILE floats
    KStepB* kd = reinterpret_cast<KStepB*>(dsm + BTILE);

    const Geom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 31;
    const int h = lane >> 5;
    const int n0 = blockIdx.y * NB;
    const int TW = 1 << tw_log2, TH = 128 >> tw_log2;
    const int PH = TH + g.KH - 1, PW = TW + g.KW - 1;
    const int PS = g.C + 8;                                             // bf16 per patch position (16 B pad)
    const int cch = g.C / BK;
    const int nsteps = g.KH * g.KW * cch;
    __bf16* Ph = reinterpret_cast<__bf16*>(dsm + BTILE) + (size_t)(nsteps + PDGS) * (sizeof(KStepB) / 2);
    __bf16* Pl = Ph + (size_t)PH * PW * PS;

    const int tiles_x = (g.OW + TW - 1) >> tw_log2;
    const int tiles_y = (g.OH + TH - 1) / TH;
    int t = blockIdx.x;
    const int txi = t % tiles_x;
    t /= tiles_x;
    const int tyi = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = tyi * TH, x0 = txi << tw_log2;
    // source coordinate of patch position (0,0); patch row of tap ky = ky (cs > 0) or KH-1-ky (cs < 0)
    const int sy0 = y0 + g.off + (g.cs < 0 ? -(g.KH - 1) : 0);
    const int sx0 = x0 + g.offx + (g.cs < 0 ? -(g.KW - 1) : 0);

    for (int s = tid; s < nsteps + PDGS; s += 256) {
        KStepB k{0, 0, 0, 0};
        if (s < nsteps) {
            const int tap = s / cch, cc = s - tap * cch;
            const int ky = tap / g.KW, kx = tap - ky * g.KW;
            k.dy = g.cs > 0 ? ky : g.KH - 1 - ky;
            k.dx = g.cs > 0 ? kx : g.KW - 1 - kx;
            k.c0 = cc * BK;
            k.woff = (tap * cch + cc) * npad * BK;
        }
        kd[s] = k;
    }
    // weight stages: the pre-split K-contiguous copy, PDGS k-steps per LDS stage, two LDS stages.  A stage's pieces are
    // loaded into registers THREE stages before it is consumed (4 rotating register sets, one 16-byte piece per thread and
    // k-step for 32 columns): a stage is 0.16 us of MFMA work and an L2 round trip 0.5 - 0.8 us, so with one stage in flight
    // (the first form of this kernel) every workgroup waited for the weights once per stage.  Loads are unconditional with a
    // clamped stage index - a load under a branch makes the compiler drain vmcnt at the join.
    constexpr int PIECES = 2 * NB * 4;
    constexpr int PPT = (PIECES + 255) / 256;
    constexpr int NSET = 4;
    const __bf16* wg = wsplit;
    const int nstages = (nsteps + PDGS - 1) / PDGS;
    u32x4 breg[NSET][PDGS][PPT];
    auto load_b = [&](int stg, u32x4 (&r)[PDGS][PPT]) {
        const int s0 = (stg < nstages ? stg : nstages - 1) * PDGS;
#pragma unroll
        for (int st = 0; st < PDGS; ++st) {
            const int sc = s0 + st < nsteps ? s0 + st : nsteps - 1;
            const size_t wo = (size_t)sc * npad * BK;                   // = kd[sc].woff
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                const int pc = tid + 256 * j;
                const int pl = pc / (NB * 4);
                const int row = (pc / 4) % NB;
                const int qtr = pc & 3;
                const int nrow = n0 + row < npad ? n0 + row : 0;
                r[st][j] = *reinterpret_cast<const u32x4*>(wg + (size_t)pl * plane + wo + nrow * BK + qtr * 8);
            }
        }
    };
    auto store_b = [&](__bf16* dst, const u32x4 (&r)[PDGS][PPT]) {
#pragma unroll
        for (int st = 0; st < PDGS; ++st)
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                const int pc = tid + 256 * j;
                const int pl = pc / (NB * 4);
                const int row = (pc / 4) % NB;
                const int qtr = pc & 3;
                *reinterpret_cast<u32x4*>(dst + st * STEP_E + (pl * NB + row) * BROW + qtr * 8) = r[st][j];
            }
    };

    // this lane's tile position and its A-fragment base inside the patch
    const int ml = wave * 32 + i;
    const int ty = ml >> tw_log2, tx = ml & (TW - 1);
    const int abase = (ty * PW + tx) * PS + 8 * h;
    const int gy = y0 + ty, gx = x0 + tx;
    const int rowoff = (gy < g.OH && gx < g.OW) ? ((b * g.OH + gy) * g.OW + gx) * g.N : -1;

    f32x16 acc[RN];
#pragma unroll
    for (int r = 0; r < RN; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

    load_b(0, breg[0]);
    load_b(1, breg[1]);
    load_b(2, breg[2]);
    // (the first three weight stages are in flight while the patch is staged)
    {   // the patch: f32 -> hi / lo bf16, zero outside the image.  Loads are issued in batches of 8 per thread so
        // that a workgroup pays ~2 global latencies for its patch instead of one per 16 bytes
        const float* img = p.in + (size_t)b * g.IH * g.IW * g.C;
        const int c4n = g.C >> 2;
        const int total = PH * PW * c4n;
        constexpr int PB = 8;
        for (int e0 = tid; e0 < total; e0 += 256 * PB) {
            f32x4 v[PB];
            int dst[PB];
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int e = e0 + 256 * j;
                const int ee = e < total ? e : total - 1;
                const int pos = ee / c4n, c4 = ee - pos * c4n;
                const int py = pos / PW, px = pos - py * PW;
                const int gy = sy0 + py, gx = sx0 + px;
                const bool ok = e < total && (unsigned)gy < (unsigned)g.IH && (unsigned)gx < (unsigned)g.IW;
                const size_t so = ok ? ((size_t)gy * g.IW + gx) * g.C + 4 * c4 : 0;
                v[j] = *reinterpret_cast<const f32x4*>(img + so);
                if (!ok) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                dst[j] = e < total ? pos * PS + 4 * c4 : -1;
            }
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                if (dst[j] < 0) continue;
                u32x2 h2, l2;
                split4(v[j], h2, l2);
                *reinterpret_cast<u32x2*>(Ph + dst[j]) = h2;
                *reinterpret_cast<u32x2*>(Pl + dst[j]) = l2;
            }
        }
    }
    __syncthreads();                 // kd table visible
    store_b(Bs, breg[0]);
    __syncthreads();                 // patch + first weight stage visible

    bf16x8 ah[2][2], al[2][2];       // [register set][k16 half]
    auto read_a = [&](int s, int set) {
        const KStepB k = kd[s];
        const int o = abase + (k.dy * PW + k.dx) * PS + k.c0;
        ah[set][0] = *reinterpret_cast<const bf16x8*>(Ph + o);
        ah[set][1] = *reinterpret_cast<const bf16x8*>(Ph + o + 16);
        al[set][0] = *reinterpret_cast<const bf16x8*>(Pl + o);
        al[set][1] = *reinterpret_cast<const bf16x8*>(Pl + o + 16);
    };
    auto mma = [&](const __bf16* bstep, int set) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int r = 0; r < RN; ++r) {
                const __bf16* brow = bstep + (r * 32 + i) * BROW + 16 * kk + 8 * h;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(brow);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(brow + NB * BROW);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[set][kk], bh, acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[set][kk], bl, acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[set][kk], bh, acc[r], 0, 0, 0);
            }
    };

    static_assert(PDGS == 2, "the A-fragment register sets below alternate per k-step of a 2-step stage");
    // The patch loads above sit under per-lane branches, so the compiler cannot prove them complete and would drain vmcnt
    // at the top of the k-loop on every trip; the patch is needed here anyway: one explicit vmcnt(0) settles its bookkeeping.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    read_a(0, 0);                    // kd[] is padded by PDGS zero entries: reads one step past the end stay inside it
    // one stage; u = stage index mod NSET is a compile-time constant at every call site so that the register sets are static
    auto stage = [&](int t, auto uc) {
        constexpr int u = decltype(uc)::value;
        const int s0 = t * PDGS;
        const __bf16* bcur = Bs + (u & 1) * BTILE;
        load_b(t + 3, breg[(u + 3) & 3]);                   // consumed three stages from now
        read_a(s0 + 1, 1);
        mma(bcur, 0);
        read_a(s0 + 2, 0);                                  // first step of the next stage (a padding entry after the last)
        if (s0 + 1 < nsteps) mma(bcur + STEP_E, 1);
        store_b(Bs + ((u + 1) & 1) * BTILE, breg[(u + 1) & 3]);       // stage t+1, loaded two stages ago
        __syncthreads();
    };
    using U0 = std::integral_constant<int, 0>;
    using U1 = std::integral_constant<int, 1>;
    using U2 = std::integral_constant<int, 2>;
    using U3 = std::integral_constant<int, 3>;
    // full groups of NSET stages without exits (an early exit inside the group lands in the loop latch, whose merged
    // wait-count state drains vmcnt at the top of every trip), then the 0 - 3 remaining stages as straight-line code
    int t0 = 0;
    for (; t0 + NSET <= nstages; t0 += NSET) {
        stage(t0, U0{});
        stage(t0 + 1, U1{});
        stage(t0 + 2, U2{});
        stage(t0 + 3, U3{});
    }
    if (t0 < nstages) {
        stage(t0, U0{});
        if (t0 + 1 < nstages) {
            stage(t0 + 1, U1{});
            if (t0 + 2 < nstages) stage(t0 + 2, U2{});
        }
    }

    const float* bias = p.bias;
    const float* aux = p.aux;
    const float* res = p.res;
    float* out = p.out;
    int ro[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) ro[e] = __shfl(rowoff, (e & 3) + 8 * (e >> 2) + 4 * h, 64);
#pragma unroll
    for (int r = 0; r < RN; ++r) {
        int n = n0 + r * 32 + i;
        if (n >= g.N) continue;
        float bv = bias ? bias[n] : 0.f;
        pm_epilogue_tile(acc[r], ro, n, bv, aux, res, out, p.out2, p.act2, g.aux_act, g.out_act, g.slope);
    }
}

// The same tile, weights straight to registers.  Dissection of the LDS-staged form above on the 28x28 32->32 5x5 layer
// (64 us; -DPM_EXP builds): patch staging 14 us, epilogue 9 us, k-loop 36 us of which 12 us are the weight stages (global
// -> registers -> LDS -> barrier every two k-steps) - for an operand that every wave reads in full anyway (a wave's B
// fragment of a k-step IS the whole 32-column stage).  Here each wave loads its own B fragments from the pre-split copy
// (16 bytes per lane and fragment, served by L1 after the first wave), NSET k-steps ahead in static register sets: no LDS
// stores, no barriers in the k-loop, waves drift freely; LDS holds only the patch, so three workgroups fit a CU.
template <int RN>
__global__ __launch_bounds__(256, RN == 1 ? 3 : 2) void patch_conv_bd_bf16_kernel(GemmArgs p, const __bf16* __restrict__ wsplit, int npad,
                                                                 long long plane, int tw_log2) {
    constexpr int NSET = 4;
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    KStepB* kd = reinterpret_cast<KStepB*>(dsm);

    const Geom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 31;
    const int h = lane >> 5;
    const int n0 = blockIdx.y * 32 * RN;
    const int TW = 1 << tw_log2, TH = 128 >> tw_log2;
    const int PH = TH + g.KH - 1, PW = TW + g.KW - 1;
    const int PS = g.C + 8;                                             // bf16 per patch position (16 B pad)
    const int cch = g.C / BK;
    const int nsteps = g.KH * g.KW * cch;
    __bf16* Ph = reinterpret_cast<__bf16*>(dsm) + (size_t)(nsteps + 2) * (sizeof(KStepB) / 2);
    __bf16* Pl = Ph + (size_t)PH * PW * PS;

    const int tiles_x = (g.OW + TW - 1) >> tw_log2;
    const int tiles_y = (g.OH + TH - 1) / TH;
    int t = blockIdx.x;
    const int txi = t % tiles_x;
    t /= tiles_x;
    const int tyi = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = tyi * TH, x0 = txi << tw_log2;
    const int sy0 = y0 + g.off + (g.cs < 0 ? -(g.KH - 1) : 0);
    const int sx0 = x0 + g.offx + (g.cs < 0 ? -(g.KW - 1) : 0);

    // B fragment of (k-step s, column tile r, half kk, plane): 16 bytes at ((s * npad + n) * 32 + 16 kk + 8 h); q = 2 kk + plane
    bf16x8 bq[NSET][RN][4];
    int ncl[RN];
#pragma unroll
    for (int r = 0; r < RN; ++r) ncl[r] = (n0 + 32 * r + i < npad ? n0 + 32 * r + i : 0) * BK + 8 * h;
    auto load_b = [&](int s, bf16x8 (&bb)[RN][4]) {
        const int sc = s < nsteps ? s : nsteps - 1;
        const __bf16* base = wsplit + (size_t)sc * npad * BK;
#pragma unroll
        for (int r = 0; r < RN; ++r) {
            const __bf16* src = base + ncl[r];
            bb[r][0] = *reinterpret_cast<const bf16x8*>(src);
            bb[r][1] = *reinterpret_cast<const bf16x8*>(src + plane);
            bb[r][2] = *reinterpret_cast<const bf16x8*>(src + 16);
            bb[r][3] = *reinterpret_cast<const bf16x8*>(src + plane + 16);
        }
    };
    load_b(0, bq[0]);
    load_b(1, bq[1]);
    load_b(2, bq[2]);
    load_b(3, bq[3]);
    float bvr[RN];
#pragma unroll
    for (int r = 0; r < RN; ++r) bvr[r] = p.bias ? p.bias[n0 + 32 * r + i < g.N ? n0 + 32 * r + i : 0] : 0.f;

    for (int s = tid; s < nsteps + 2; s += 256) {
        KStepB k{0, 0, 0, 0};
        if (s < nsteps) {
            const int tap = s / cch, cc = s - tap * cch;
            const int ky = tap / g.KW, kx = tap - ky * g.KW;
            k.dy = g.cs > 0 ? ky : g.KH - 1 - ky;
            k.dx = g.cs > 0 ? kx : g.KW - 1 - kx;
            k.c0 = cc * BK;
        }
        kd[s] = k;
    }
    {   // the patch: f32 -> hi / lo bf16, zero outside the image; 12 loads per thread in flight: one round trip for the
        // patches of the 28x28 layers (9 per thread), two for 64-channel ones
        const float* img = p.in + (size_t)b * g.IH * g.IW * g.C;
        const int c4n = g.C >> 2;
        const int total = PH * PW * c4n;
        constexpr int PB = 12;
        for (int e0 = tid; e0 < total; e0 += 256 * PB) {
            f32x4 v[PB];
            int dst[PB];
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int e = e0 + 256 * j;
                const int ee = e < total ? e : total - 1;
                const int pos = ee / c4n, c4 = ee - pos * c4n;
                const int py = pos / PW, px = pos - py * PW;
                const int gy = sy0 + py, gx = sx0 + px;
                const bool ok = e < total && (unsigned)gy < (unsigned)g.IH && (unsigned)gx < (unsigned)g.IW;
                const size_t so = ok ? ((size_t)gy * g.IW + gx) * g.C + 4 * c4 : 0;
                v[j] = *reinterpret_cast<const f32x4*>(img + so);
                if (!ok) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                dst[j] = e < total ? pos * PS + 4 * c4 : -1;
            }
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                if (dst[j] < 0) continue;
                u32x2 h2, l2;
                split4(v[j], h2, l2);
                *reinterpret_cast<u32x2*>(Ph + dst[j]) = h2;
                *reinterpret_cast<u32x2*>(Pl + dst[j]) = l2;
            }
        }
    }

    const int ml = wave * 32 + i;
    const int ty = ml >> tw_log2, tx = ml & (TW - 1);
    const int abase = (ty * PW + tx) * PS + 8 * h;
    const int gy = y0 + ty, gx = x0 + tx;
    const int rowoff = (gy < g.OH && gx < g.OW) ? ((b * g.OH + gy) * g.OW + gx) * g.N : -1;

    f32x16 acc[RN];
#pragma unroll
    for (int r = 0; r < RN; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

    // the patch loads sit under per-lane branches: one explicit vmcnt(0) (the patch is needed here anyway) keeps the
    // compiler from draining vmcnt inside the k-loop
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();                 // patch + kd visible; the only barrier of the kernel

    bf16x8 ah[2][2], al[2][2];       // [register set][k16 half]
    auto read_a = [&](int s, int set) {
        const KStepB k = kd[s];
        const int o = abase + (k.dy * PW + k.dx) * PS + k.c0;
        ah[set][0] = *reinterpret_cast<const bf16x8*>(Ph + o);
        ah[set][1] = *reinterpret_cast<const bf16x8*>(Ph + o + 16);
        al[set][0] = *reinterpret_cast<const bf16x8*>(Pl + o);
        al[set][1] = *reinterpret_cast<const bf16x8*>(Pl + o + 16);
    };
    auto step = [&](int s, auto uc) {
        constexpr int u = decltype(uc)::value;
        read_a(s + 1, (u + 1) & 1);                         // kd[] has two zero entries past the end
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int r = 0; r < RN; ++r) {
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u & 1][kk], bq[u][r][2 * kk], acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u & 1][kk], bq[u][r][2 * kk + 1], acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[u & 1][kk], bq[u][r][2 * kk], acc[r], 0, 0, 0);
            }
        load_b(s + NSET, bq[u]);                            // NSET k-steps ahead (clamped at the end)
        __builtin_amdgcn_sched_barrier(0);                  // or the scheduler sinks the loads until the distance is gone
    };
    using U0 = std::integral_constant<int, 0>;
    using U1 = std::integral_constant<int, 1>;
    using U2 = std::integral_constant<int, 2>;
    using U3 = std::integral_constant<int, 3>;
    read_a(0, 0);
    int s0 = 0;
    for (; s0 + NSET <= nsteps; s0 += NSET) {               // exit-free groups, then 0 - 3 steps as straight-line code
        step(s0, U0{});
        step(s0 + 1, U1{});
        step(s0 + 2, U2{});
        step(s0 + 3, U3{});
    }
    if (s0 < nsteps) {
        step(s0, U0{});
        if (s0 + 1 < nsteps) {
            step(s0 + 1, U1{});
            if (s0 + 2 < nsteps) step(s0 + 2, U2{});
        }
    }

    const float* aux = p.aux;
    const float* res = p.res;
    float* out = p.out;
    int ro[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) ro[e] = __shfl(rowoff, (e & 3) + 8 * (e >> 2) + 4 * h, 64);
#pragma unroll
    for (int r = 0; r < RN; ++r) {
        int n = n0 + r * 32 + i;
        if (n >= g.N) continue;
        pm_epilogue_tile(acc[r], ro, n, bvr[r], aux, res, out, p.out2, p.act2, g.aux_act, g.out_act, g.slope);
    }
}

// Channel-pass form of the tile above for layers whose patch does not fit LDS at full depth (the CelebA PixelCNN's 16 x 16 x 256
// grids: a 9 x 18 patch of 256 channels is 171 KB as hi / lo planes).  The direct form these layers fell back to re-gathers
// every tap's rows from L2 for every 32-column tile: 16 x 16 x 16 x 256 floats (4 MB) read 6 taps x 8 column tiles = 48 times
// = 201 MB per launch, ~5 TB/s for the 40 us it takes whatever N is - an L2-bandwidth-bound GEMM at 17 % of the matrix cores.
// Here the K loop runs CHANNEL CHUNK outermost: the patch of CC channels (64 or 128) is staged once (f32 -> hi / lo bf16), all
// taps walk it (k-steps of 32 channels, weights straight to registers NSET steps ahead, exactly as above), then the next chunk
// replaces it; the accumulators stay in registers across the passes.  Operand traffic: every workgroup reads its patch once
// (9 x 18 x 256 floats = 166 KB; 42 MB per launch).  The chunk of pass c + 1 is requested from memory BEFORE the k-loop of pass c
// (registers), so its round trip hides behind that pass's MFMAs; two barriers per pass.
// The k-steps of a pass are STRAIGHT-LINE code (SPC = steps per pass = 2 x taps is a template parameter): with the pass switch as a
// branch inside the unrolled steps the compiler could not count the loads in flight at the join and drained them all
// (s_waitcnt vmcnt(0)) at the head of every group of steps - weights requested NSET steps ahead arrived "just in time" by
// stalling.  Here every load of the loop body is unconditional (the last pass requests a patch it never stores), the
// prologue issues its loads in the body's order, and the waits the compiler emits are exact.
// (Tried and dropped: two or four wave groups dealing the passes between them - 27.7 / 88 us against 24.6 us per launch.)
template <int RN, int SPC, int NSET>
__global__ __launch_bounds__(256, 1) void patch_conv_cp_bf16_kernel(GemmArgs p, const __bf16* __restrict__ wsplit, int npad,
                                                                    long long plane, int tw_log2) {
    constexpr int CC = 64, NPF = 12;
    static_assert(SPC % NSET == 0, "static register-set rotation");
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const Geom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 31;
    const int h = lane >> 5;
    const int n0 = blockIdx.y * 32 * RN;
    const int TW = 1 << tw_log2, TH = 128 >> tw_log2;
    const int PH = TH + g.KH - 1, PW = TW + g.KW - 1;
    constexpr int PS = CC + 8;                                          // bf16 per patch position (16 B pad)
    constexpr int cpp = CC / BK;                                        // 32-channel chunks per pass
    const int cch = g.C / BK;                                           // ... per tap in the weight layout
    const int npass = g.C / CC;
    const int nsteps = npass * SPC;
    // LDS: the step table (nsteps + NSET + 2 entries), then the patch (hi plane, lo plane)
    KStepB* kd = reinterpret_cast<KStepB*>(dsm);
    const size_t pelems = (size_t)PH * PW * PS;
    __bf16* Ph = reinterpret_cast<__bf16*>(kd + (nsteps + NSET + 2));
    __bf16* Pl = Ph + pelems;

    const int tiles_x = (g.OW + TW - 1) >> tw_log2;
    const int tiles_y = (g.OH + TH - 1) / TH;
    int t = blockIdx.x;
    const int txi = t % tiles_x;
    t /= tiles_x;
    const int tyi = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = tyi * TH, x0 = txi << tw_log2;
    const int sy0 = y0 + g.off + (g.cs < 0 ? -(g.KH - 1) : 0);
    const int sx0 = x0 + g.offx + (g.cs < 0 ? -(g.KW - 1) : 0);

    // k-step table in pass order: step s = (pass, tap, chunk of the pass); dy = the tap's patch offset (elements, the chunk's
    // channel offset included), woff = offset of the step's [npad][32] hi block in the weights
    for (int s = tid; s < nsteps + NSET + 2; s += 256) {
        KStepB k{0, 0, 0, 0};
        const int sc = s < nsteps ? s : nsteps - 1;
        const int ps = sc / SPC, r = sc - ps * SPC;
        const int tap = r / cpp, cl = r - tap * cpp;
        const int ky = tap / g.KW, kx = tap - ky * g.KW;
        const int dy = g.cs > 0 ? ky : g.KH - 1 - ky;
        const int dx = g.cs > 0 ? kx : g.KW - 1 - kx;
        k.dy = (dy * PW + dx) * PS + cl * BK;
        k.woff = (tap * cch + ps * cpp + cl) * npad * BK;
        kd[s] = k;
    }

    // the patch of one pass: NPF f32x4 pieces per thread (positions x CC / 4), zero outside the image
    const float* img = p.in + (size_t)b * g.IH * g.IW * g.C;
    constexpr int c4n = CC >> 2;
    const int total = PH * PW * c4n;
    f32x4 pv[NPF];
    int pdst[NPF];
    int psrc[NPF];                                       // element offset inside the image (B * IH * IW * C * 4 < 2^31: host check)
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
        const int e = tid + 256 * j;
        const int ee = e < total ? e : total - 1;
        const int pos = ee / c4n, c4 = ee - pos * c4n;
        const int py = pos / PW, px = pos - py * PW;
        const int gy = sy0 + py, gx = sx0 + px;
        const bool ok = e < total && (unsigned)gy < (unsigned)g.IH && (unsigned)gx < (unsigned)g.IW;
        psrc[j] = ok ? (gy * g.IW + gx) * g.C + 4 * c4 : -1;
        pdst[j] = e < total ? pos * PS + 4 * c4 : -1;
    }
    auto fetch_patch = [&](int ps) {                     // unconditional loads (clamped address): all NPF in flight
        const int c0 = ps * CC;
#pragma unroll
        for (int q = 0; q < NPF; ++q) pv[q] = *reinterpret_cast<const f32x4*>(img + (psrc[q] >= 0 ? psrc[q] + c0 : 0));
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            f32x4 v = pv[q];
            if (psrc[q] < 0) v = f32x4{0.f, 0.f, 0.f, 0.f};
            u32x2 h2, l2;
            split4(v, h2, l2);
            if (pdst[q] >= 0) {
                *reinterpret_cast<u32x2*>(Ph + pdst[q]) = h2;
                *reinterpret_cast<u32x2*>(Pl + pdst[q]) = l2;
            }
        }
    };

    // B fragment of (k-step s, column tile r, half kk, plane): 16 bytes at (woff(s) + n * 32 + 16 kk + 8 h); q = 2 kk + plane
    bf16x8 bq[NSET][RN][4];
    int ncl[RN];
#pragma unroll
    for (int r = 0; r < RN; ++r) ncl[r] = (n0 + 32 * r + i < npad ? n0 + 32 * r + i : 0) * BK + 8 * h;
    auto load_b = [&](int s, bf16x8 (&bb)[RN][4]) {       // s <= nsteps + NSET - 1: the table's padding repeats the last step
        const __bf16* base = wsplit + kd[s].woff;
#pragma unroll
        for (int r = 0; r < RN; ++r) {
            const __bf16* src = base + ncl[r];
            bb[r][0] = *reinterpret_cast<const bf16x8*>(src);
            bb[r][1] = *reinterpret_cast<const bf16x8*>(src + plane);
            bb[r][2] = *reinterpret_cast<const bf16x8*>(src + 16);
            bb[r][3] = *reinterpret_cast<const bf16x8*>(src + plane + 16);
        }
    };
    float bvr[RN];
#pragma unroll
    for (int r = 0; r < RN; ++r) bvr[r] = p.bias ? p.bias[n0 + 32 * r + i < g.N ? n0 + 32 * r + i : 0] : 0.f;

    const int ml = wave * 32 + i;
    const int ty = ml >> tw_log2, tx = ml & (TW - 1);
    const int abase = (ty * PW + tx) * PS + 8 * h;
    const int gy = y0 + ty, gx = x0 + tx;
    const int rowoff = (gy < g.OH && gx < g.OW) ? ((b * g.OH + gy) * g.OW + gx) * g.N : -1;

    f32x16 acc[RN];
#pragma unroll
    for (int r = 0; r < RN; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

    __syncthreads();                                     // step table visible
    fetch_patch(0);                                      // the prologue's loads in the loop body's order: patch, then weights
#pragma unroll
    for (int u = 0; u < NSET; ++u) load_b(u, bq[u]);

    bf16x8 ah[2][2], al[2][2];       // [register set][k16 half]
    auto read_a = [&](int o, int set) {
        ah[set][0] = *reinterpret_cast<const bf16x8*>(Ph + o);
        ah[set][1] = *reinterpret_cast<const bf16x8*>(Ph + o + 16);
        al[set][0] = *reinterpret_cast<const bf16x8*>(Pl + o);
        al[set][1] = *reinterpret_cast<const bf16x8*>(Pl + o + 16);
    };
    for (int ps = 0; ps < npass; ++ps) {
        const int sb = ps * SPC;
        if (ps) __syncthreads();                            // every wave is done with the old patch
        store_patch();                                      // this pass's pieces (requested one pass ago)
        __syncthreads();
        fetch_patch(ps + 1 < npass ? ps + 1 : 0);           // in flight during this pass's MFMAs (the last one is never stored)
        int ao[2];                                          // A offsets, read from the table one step before they are needed
        read_a(abase + kd[sb].dy, 0);
        ao[1] = abase + kd[sb + 1].dy;
#pragma unroll
        for (int q = 0; q < SPC; ++q) {                     // straight-line: q, the register sets and the waits are static
            const int s = sb + q;
            const int u = q % NSET;
            if (q + 1 < SPC) read_a(ao[(q + 1) & 1], (q + 1) & 1);          // A of step s + 1
            if (q + 2 < SPC) ao[q & 1] = abase + kd[s + 2].dy;              // the table read completes behind this step's MFMAs
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int r = 0; r < RN; ++r) {
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[q & 1][kk], bq[u][r][2 * kk], acc[r], 0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[q & 1][kk], bq[u][r][2 * kk + 1], acc[r], 0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[q & 1][kk], bq[u][r][2 * kk], acc[r], 0, 0, 0);
                }
            load_b(s + NSET, bq[u]);                        // NSET k-steps ahead (the table's padding covers the end)
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    const float* aux = p.aux;
    const float* res = p.res;
    float* out = p.out;
    int ro[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) ro[e] = __shfl(rowoff, (e & 3) + 8 * (e >> 2) + 4 * h, 64);
#pragma unroll
    for (int r = 0; r < RN; ++r) {
        int n = n0 + r * 32 + i;
        if (n >= g.N) continue;
        pm_epilogue_tile(acc[r], ro, n, bvr[r], aux, res, out, p.out2, p.act2, g.aux_act, g.out_act, g.slope);
    }
}

// host-side plan of the patch form: tile shape and dynamic LDS size; returns false when the problem does not qualify
struct PatchPlan { int tw_log2; size_t lds, lds_bd; dim3 grid; };
bool plan_patch(const Geom& g, int groups, int rn, PatchPlan& pp) {
    if (groups != 1 || g.a != 1 || g.d != 1 || g.C % BK != 0 || g.in_act != PM_ACT_NONE) return false;
    if (g.KH * g.KW < 4 || g.OW < 12 || g.OH < 4) return false;        // 1x1 / tiny grids: the direct form is fine
    pp.tw_log2 = g.OW > 16 ? 5 : 4;
    const int TW = 1 << pp.tw_log2, TH = 128 >> pp.tw_log2;
    const int tiles_x = (g.OW + TW - 1) / TW, tiles_y = (g.OH + TH - 1) / TH;
    if ((long long)tiles_x * TW * tiles_y * TH * 2 > 3LL * g.OH * g.OW) return false;   // > 50 % padded slots
    const int NB = 32 * rn;
    const size_t bt = (size_t)PDGS * 2 * NB * BROW * 2 * 2;             // two stages, bytes
    const int nsteps = g.KH * g.KW * (g.C / BK);
    if (nsteps > 4096) return false;
    const size_t kdb = (size_t)(nsteps + PDGS) * sizeof(KStepB);
    const size_t patch = (size_t)(TH + g.KH - 1) * (TW + g.KW - 1) * (g.C + 8) * 2 * 2;
    pp.lds = bt + kdb + patch;
    pp.lds_bd = (size_t)(nsteps + 2) * sizeof(KStepB) + patch;          // weights-to-registers form: no weight stages
    if (pp.lds > 150 * 1024) return false;
    pp.grid = dim3((unsigned)(g.B * tiles_y * tiles_x), (g.N + NB - 1) / NB, 1);
    return true;
}

// channel-pass plan: the patch form's conditions except that the patch only has to fit LDS CC channels at a time
struct PatchCpPlan { int tw_log2, spc, nset; size_t lds; dim3 grid; };
bool plan_patch_cp(const Geom& g, int groups, int rn, PatchCpPlan& pp) {
    constexpr int CC = 64;
    if (groups != 1 || g.a != 1 || g.d != 1 || g.C % CC != 0 || g.C <= CC || g.in_act != PM_ACT_NONE) return false;
    if (g.KH * g.KW < 4 || g.OW < 12 || g.OH < 4) return false;
    if ((long long)g.B * g.IH * g.IW * g.C * 4 >= 0x7ffffff0LL) return false;
    pp.tw_log2 = g.OW > 16 ? 5 : 4;
    const int TW = 1 << pp.tw_log2, TH = 128 >> pp.tw_log2;
    const int tiles_x = (g.OW + TW - 1) / TW, tiles_y = (g.OH + TH - 1) / TH;
    if ((long long)tiles_x * TW * tiles_y * TH * 2 > 3LL * g.OH * g.OW) return false;
    const int taps = g.KH * g.KW;
    if (taps != 4 && taps != 6 && taps != 9) return false;               // the instantiated pass lengths: 8 / 12 / 18 k-steps
    pp.spc = taps * (CC / BK);
    pp.nset = pp.spc == 8 ? 4 : 6;
    const int nsteps = taps * (g.C / BK);
    if (nsteps > 4096) return false;
    const int NB = 32 * rn;
    if ((TH + g.KH - 1) * (TW + g.KW - 1) * (CC / 4) > 12 * 256) return false;   // 12 register-staged pieces per thread
    pp.lds = (size_t)(nsteps + pp.nset + 2) * sizeof(KStepB) + (size_t)(TH + g.KH - 1) * (TW + g.KW - 1) * (CC + 8) * 2 * 2;
    if (pp.lds > 150 * 1024) return false;
    pp.grid = dim3((unsigned)(g.B * tiles_y * tiles_x), (g.N + NB - 1) / NB, 1);
    return true;
}

// ------------------- d = 1 convolutions with the WHOLE input image of a workgroup resident in LDS (bf16x3) -------------------
// The patch form above re-stages halos (a 4 x 32 output tile of a 5x5 layer loads an 8 x 36 patch: 2.25x the input), cuts
// the batch into 1792 tiles for 512 - 768 workgroup slots (a 3.5-round grid runs 4 rounds) and has no stride-2 form at all
// (those layers re-gather their rows from L2 once per tap in the direct form: 2.9x the operand bytes).  The images of this
// path are small - 28 x 28 x 32 floats are 125 KB as hi / lo bf16 planes with the 16-byte position pad - so ONE workgroup
// stages ONE image once (a plain linear, fully coalesced read), and B = 256 images are one round on 256 CUs:
//  * any stride a (src = dst * a + tap * cs + off): an A fragment is a 16-byte LDS read at the tap's input position; taps that
//    fall outside the image read a zero slot (one select on the address), so there is no padded border in LDS;
//  * output positions are walked in flat order (32 consecutive positions = one MFMA row tile - no 2-D tile shape is needed
//    when the whole image is resident); a wave owns one 32-column tile and up to T row tiles, and a B fragment - loaded
//    straight from the pre-split weights, four k-steps ahead, like patch_conv_bd - feeds all T of them;
//  * one barrier (after staging); waves run the k-loop and their epilogues independently.
// TR: the MFMAs take the weights as A and the activations as B, so a lane's accumulators are ONE output position x 16
// columns and the epilogue moves 16-byte vectors (pm_epilogue_tile_t); N % 4 == 0.  TR = false: rows in registers, dword epilogue.
template <int NW, int T, bool TR>
__global__ __launch_bounds__(64 * NW, 2) void image_conv_bf16_kernel(GemmArgs p, const __bf16* __restrict__ wsplit, int npad,
                                                                     long long plane, int nct) {
    constexpr int NSET = 4;
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const Geom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 31;
    const int h = lane >> 5;
    const int b = blockIdx.x;
    const int PS = g.C + 8;                                             // bf16 per input position (16 B pad)
    const int cch = g.C / BK;
    const int nsteps = g.KH * g.KW * cch;
    const int npos = g.IH * g.IW;
    __bf16* Ph = reinterpret_cast<__bf16*>(dsm);
    __bf16* Pl = Ph + (size_t)npos * PS + 64;                           // 64 zero elements (128 B) behind each plane
    const int zoff = npos * PS;                                         // the zero slot of a plane

    // this wave: column tile ct, row tiles rt0 + j * wct
    const int wct = NW / nct;                                           // waves per column tile
    const int ct = wave / wct, rt0 = wave - ct * wct;
    const int Mi = g.OH * g.OW;
    const int n = ct * 32 + i;
    const int ncl = (n < npad ? n : 0) * BK + 8 * h;

    bf16x8 bq[NSET][4];                                                 // [set][2 kk + plane]
    auto load_b = [&](int s, bf16x8 (&bb)[4]) {
        const int sc = s < nsteps ? s : nsteps - 1;
        const __bf16* src = wsplit + (size_t)sc * npad * BK + ncl;
        bb[0] = *reinterpret_cast<const bf16x8*>(src);
        bb[1] = *reinterpret_cast<const bf16x8*>(src + plane);
        bb[2] = *reinterpret_cast<const bf16x8*>(src + 16);
        bb[3] = *reinterpret_cast<const bf16x8*>(src + plane + 16);
    };
    load_b(0, bq[0]);
    load_b(1, bq[1]);
    load_b(2, bq[2]);
    load_b(3, bq[3]);
    const float bv = p.bias ? p.bias[n < g.N ? n : 0] : 0.f;

    if (tid < 32) {                                                     // the zero slots
        reinterpret_cast<unsigned*>(Ph + zoff)[tid] = 0u;
        reinterpret_cast<unsigned*>(Pl + zoff)[tid] = 0u;
    }
    {   // the image: f32 -> in_act -> hi / lo bf16; a linear read, up to 13 float4 per thread in flight
        const float* img = p.in + (size_t)b * npos * g.C;
        const int c4n = g.C >> 2;
        const int total = npos * c4n;
        const bool in_relu = g.in_act == PM_ACT_RELU;
        const float in_ns = g.in_act == PM_ACT_LEAKY ? g.slope : 1.f;
        constexpr int PB = NW == 8 ? 13 : 8;                             // 28 x 28 x 32 on 512 threads: 12.25 per thread
        [[maybe_unused]] f32x4 csum = {0.f, 0.f, 0.f, 0.f};              // in_colsum: this thread's channel quad never changes
        for (int e0 = tid; e0 < total; e0 += 64 * NW * PB) {             // (c4n divides 64: the launcher checks)
            f32x4 v[PB];
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int e = e0 + 64 * NW * j;
                v[j] = *reinterpret_cast<const f32x4*>(img + 4 * (size_t)(e < total ? e : total - 1));
            }
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int e = e0 + 64 * NW * j;
                const int ee = e < total ? e : total - 1;               // the overhang rewrites the last piece: harmless
                const int pos = ee / c4n, c4 = ee - pos * c4n;
#ifdef PM_IMAGE_INSUM
                if (e < total) csum += v[j];
#endif
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float x = v[j][q];
                    const float neg = in_relu ? 0.f : x * in_ns;
                    v[j][q] = x >= 0.f ? x : neg;
                }
                u32x2 h2, l2;
                split4(v[j], h2, l2);
                *reinterpret_cast<u32x2*>(Ph + pos * PS + 4 * c4) = h2;
                *reinterpret_cast<u32x2*>(Pl + pos * PS + 4 * c4) = l2;
            }
        }
#ifdef PM_IMAGE_INSUM
        if (p.in_colsum) {           // lanes l, l + c4n, ... of a wave hold the same channel quad: one row of sums per wave
            for (int o = c4n; o < 64; o <<= 1)
#pragma unroll
                for (int q = 0; q < 4; ++q) csum[q] += __shfl_xor(csum[q], o, 64);
            float* cs = reinterpret_cast<float*>(Pl + (size_t)npos * PS + 64);
            if (lane < c4n) *reinterpret_cast<f32x4*>(cs + wave * g.C + 4 * lane) = csum;
        }
#endif
    }

    // per row tile: this lane's A row = output position m -> top-left input coordinate of its receptive field
    int py[T], px[T];
#pragma unroll
    for (int j = 0; j < T; ++j) {
        const int m = 32 * (rt0 + j * wct) + i;
        const int oy = m / g.OW, ox = m - oy * g.OW;
        py[j] = m < Mi ? oy * g.a + g.off : ROW_INVALID;
        px[j] = ox * g.a + g.offx;
    }

    f32x16 acc[T];
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    __syncthreads();                 // image visible; the only barrier of the kernel
#ifdef PM_IMAGE_INSUM
    if (p.in_colsum && tid < g.C) {  // the image's channel sums (the NW waves' rows) -> one atomic per channel and image
        const float* cs = reinterpret_cast<const float*>(Pl + (size_t)npos * PS + 64);
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += cs[w * g.C + tid];
        atomicAdd(p.in_colsum + tid, t);
    }
#endif

    // A fragments run TWO (k-step, row tile) items ahead of the MFMAs in four static register sets (item index mod 4); the
    // tap walk (ky, kx, channel chunk) of the three k-steps in flight is kept in scalar registers - with the table lookup
    // -> address -> fragment chain of the first form (LDS, VALU, LDS: ~300 clocks) one item ahead, every item waited
    bf16x8 a[4][4];                  // [register set][hi kk0, hi kk1, lo kk0, lo kk1]
    struct Tap { int dy, dx, c0; };
    int wky = 0, wkx = 0, wcc = 0;   // the walk's position = the furthest k-step in flight
    auto advance = [&]() -> Tap {
        const Tap t{wky * g.cs, wkx * g.cs, wcc * BK};
        if (++wcc == cch) {
            wcc = 0;
            if (++wkx == g.KW) {
                wkx = 0;
                ++wky;                                                   // past the last tap: coordinates nobody uses
            }
        }
        return t;
    };
    auto read_a = [&](const Tap& k, int j, bf16x8 (&aa)[4]) {
        const int iy = py[j] + k.dy, ix = px[j] + k.dx;
        const bool ok = (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW;
        const int o = (ok ? (iy * g.IW + ix) * PS + k.c0 : zoff) + 8 * h;
        aa[0] = *reinterpret_cast<const bf16x8*>(Ph + o);
        aa[1] = *reinterpret_cast<const bf16x8*>(Ph + o + 16);
        aa[2] = *reinterpret_cast<const bf16x8*>(Pl + o);
        aa[3] = *reinterpret_cast<const bf16x8*>(Pl + o + 16);
    };
    Tap t0, t1, t2;                  // taps of k-steps s, s+1, s+2
    auto step = [&](int s, auto uc) {
        constexpr int u = decltype(uc)::value;                          // s mod NSET
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const int cur = (u * T + j) & 3;
            const int jn = j + 2;                                       // the item two ahead: (s + jn / T, jn % T)
            read_a(jn / T == 0 ? t0 : jn / T == 1 ? t1 : t2, jn % T, a[(cur + 2) & 3]);
            if (TR) {           // D^T = W^T X^T: same products, the accumulator tile comes out transposed
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[u][0], a[cur][0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[u][1], a[cur][0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[u][0], a[cur][2], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[u][2], a[cur][1], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[u][3], a[cur][1], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[u][2], a[cur][3], acc[j], 0, 0, 0);
            } else {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][0], bq[u][0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][0], bq[u][1], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][2], bq[u][0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][1], bq[u][2], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][1], bq[u][3], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][3], bq[u][2], acc[j], 0, 0, 0);
            }
        }
        t0 = t1;
        t1 = t2;
        t2 = advance();
        load_b(s + NSET, bq[u]);                                        // NSET k-steps ahead (clamped at the end)
        __builtin_amdgcn_sched_barrier(0);                              // or the scheduler sinks the loads behind later MFMAs
    };
    using U0 = std::integral_constant<int, 0>;
    using U1 = std::integral_constant<int, 1>;
    using U2 = std::integral_constant<int, 2>;
    using U3 = std::integral_constant<int, 3>;
    static_assert(T == 1 || T == 2 || T == 4, "item index mod 4 must be static inside a group of NSET k-steps");
    if (rt0 * 32 < Mi) {                                                // wave-uniform: waves without a row tile skip it all
        t0 = advance();
        t1 = advance();
        t2 = advance();
        read_a(t0, 0, a[0]);                                            // items 0 and 1
        read_a(T == 1 ? t1 : t0, T == 1 ? 0 : 1, a[1]);
        int s0 = 0;
        for (; s0 + NSET <= nsteps; s0 += NSET) {                       // exit-free groups, then 0 - 3 steps straight-line
            step(s0, U0{});
            step(s0 + 1, U1{});
            step(s0 + 2, U2{});
            step(s0 + 3, U3{});
        }
        if (s0 < nsteps) {
            step(s0, U0{});
            if (s0 + 1 < nsteps) {
                step(s0 + 1, U1{});
                if (s0 + 2 < nsteps) step(s0 + 2, U2{});
            }
        }
        if (TR) {
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const int m = 32 * (rt0 + j * wct) + i;                 // this lane's output position
                const long long ro = m < Mi ? ((long long)b * Mi + m) * g.N : -1;
                pm_epilogue_tile_t(acc[j], ro, ct * 32, h, g.N, p.bias, p.aux, p.res, p.out, p.out2, p.act2, g.aux_act,
                                   g.out_act, g.slope);
            }
        } else if (n < g.N) {
#pragma unroll
            for (int j = 0; j < T; ++j) {
                int ro[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = 32 * (rt0 + j * wct) + (e & 3) + 8 * (e >> 2) + 4 * h;
                    ro[e] = m < Mi ? (b * Mi + m) * g.N : -1;
                }
                pm_epilogue_tile(acc[j], ro, n, bv, p.aux, p.res, p.out, p.out2, p.act2, g.aux_act, g.out_act, g.slope);
            }
        }
    }
}

struct ImagePlan { int nw, t, nct; size_t lds; };
// qualifies: plain d = 1 problems (any stride a) whose input image fits LDS as hi / lo planes, batches that fill the chip
bool plan_image(const Geom& g, int groups, ImagePlan& ip) {
    if (groups != 1 || g.d != 1 || g.C % BK != 0 || g.B < 128) return false;
    if (g.in_act != PM_ACT_NONE && g.in_act != PM_ACT_RELU && g.in_act != PM_ACT_LEAKY) return false;
    // 1x1: the dense form.  Masked sub-kernels (kws > KW: the PixelCNN's horizontal stack, 2 x 2 of a 3 x 3 kernel) qualify like
    // any other kernel - the pre-split weight copy is compact in the walked taps (PM_NO_IMAGE_MASKED=1: back on the direct form)
    static const bool masked_off = getenv("PM_NO_IMAGE_MASKED") != nullptr;
    if (g.KH * g.KW < 4 || (masked_off && g.kws != g.KW)) return false;
    // every tap is walked for every position: grids smaller than the kernel (the 7x7 <-> 1x1 layers: 1 valid tap of 49)
    // stay on the direct form and its tap lists
    if (g.IH < g.KH || g.IW < g.KW || g.OH * g.OW < 32) return false;
    const int nsteps = g.KH * g.KW * (g.C / BK);
    if (nsteps > 2048) return false;
    ip.lds = 2 * ((size_t)g.IH * g.IW * (g.C + 8) + 64) * 2;
    if (ip.lds > 158 * 1024) return false;
    ip.nct = (g.N + 31) / 32;
    if (ip.nct > 8 || (8 % ip.nct) != 0) return false;
    const int rt = (g.OH * g.OW + 31) / 32;
    ip.nw = rt * ip.nct <= 4 ? 4 : 8;
    if (ip.nw % ip.nct != 0) return false;
    const int wct = ip.nw / ip.nct;
    int t = (rt + wct - 1) / wct;
    if (t > 4) return false;
    ip.t = t == 3 ? 4 : t;
    return true;
}

template <int NW, int T, bool TR>
void launch_image_tr(const ImagePlan& ip, hipStream_t s, const GemmArgs& a, const __bf16* ws, int npad, long long plane) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&image_conv_bf16_kernel<NW, T, TR>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    PM_KTAG("image_conv_bf16_kernel<%d, %d, %s>", NW, T, TR ? "true" : "false");
    if (a.in_colsum) PM_KVAR("insum");
    const size_t lds = ip.lds + (a.in_colsum ? (size_t)NW * a.g.C * sizeof(float) : 0);
    hipLaunchKernelGGL((image_conv_bf16_kernel<NW, T, TR>), dim3((unsigned)a.g.B), dim3(64 * NW), lds, s, a, ws, npad, plane,
                       ip.nct);
}
template <int NW, int T>
void launch_image(const ImagePlan& ip, hipStream_t s, const GemmArgs& a, const __bf16* ws, int npad, long long plane) {
    static const bool tr_off = getenv("PM_IMAGE_NO_TR") != nullptr;      // A/B switch for measurements
    const bool al = !((reinterpret_cast<size_t>(a.out) | reinterpret_cast<size_t>(a.aux) | reinterpret_cast<size_t>(a.res) |
                       reinterpret_cast<size_t>(a.out2) | reinterpret_cast<size_t>(a.bias)) & 15);
    if (!tr_off && a.g.N % 4 == 0 && al) launch_image_tr<NW, T, true>(ip, s, a, ws, npad, plane);
    else launch_image_tr<NW, T, false>(ip, s, a, ws, npad, plane);
}

// ------------- the same residence for the zero-dilated (d = 2, a = 1) problems: stride-2 transposed convolutions forward and
// the data gradients of stride-2 convolutions,  src * 2 = dst + tap * cs + off ---------------------------------------------
// In flat position order half the lanes of a row tile would miss every tap, so positions are walked CLASS-major: class
// (ry, rx) = (dst_y & 1, dst_x & 1) only meets the taps ky = k0y, k0y + 2, ... with (ry + k0y * cs + off) even (likewise
// in x): four small stride-1 convolutions over the same LDS-resident source image, each with its own tap list.  A wave owns
// one class, one 32-column tile and up to T row tiles of that class (B fragments shared by the T tiles); the classes carry
// 9 / 6 / 6 / 4 taps of a 5x5 kernel, so waves finish at different times - there is no barrier to hold them.
template <int T>
__global__ __launch_bounds__(512, 2) void image_d2_bf16_kernel(GemmArgs p, const __bf16* __restrict__ wsplit, int npad,
                                                               long long plane, int nct) {
    constexpr int NSET = 4, NW = 8;
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const Geom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform for the compiler too: the class and its tap
                                                                        // walk stay in scalar registers
    const int i = lane & 31;
    const int h = lane >> 5;
    const int b = blockIdx.x;
    const int PS = g.C + 8;
    const int cch = g.C / BK;
    const int npos = g.IH * g.IW;
    __bf16* Ph = reinterpret_cast<__bf16*>(dsm);
    __bf16* Pl = Ph + (size_t)npos * PS + 64;
    const int zoff = npos * PS;

    // this wave: class (ry, rx), column tile ct, row tiles rt0 + j * wct of the class
    const int wct = NW / (4 * nct);                                     // waves per (class, column tile)
    const int grp = wave / wct, rt0 = wave - grp * wct;
    const int cls = grp / nct, ct = grp - cls * nct;
    const int ry = cls >> 1, rx = cls & 1;
    const int OHc = g.OH >> 1, OWc = g.OW >> 1, Mc = OHc * OWc;
    const int k0y = (ry + g.off) & 1, k0x = (rx + g.offx) & 1;         // cs = +-1: parity of ky * cs is that of ky
    const int nty = (g.KH - k0y + 1) >> 1, ntx = (g.KW - k0x + 1) >> 1;
    const int nsteps = nty * ntx * cch;                                 // this wave's k-steps
    const int nw_all = g.KH * g.KW * cch;
    const int n = ct * 32 + i;
    const int ncl = (n < npad ? n : 0) * BK + 8 * h;

    // a walker yields the k-steps of the class in order: (ky, kx, channel chunk) -> weight step index and tap offsets
    struct Walk { int ky, kx, cc; };
    struct Tap { int dy, dx, c0, widx; };
    auto advance = [&](Walk& w) -> Tap {
        int widx = (w.ky * g.KW + w.kx) * cch + w.cc;
        widx = widx < nw_all ? widx : nw_all - 1;                       // past the last tap: anything valid (never used)
        const Tap t{__builtin_amdgcn_readfirstlane(w.ky * g.cs), __builtin_amdgcn_readfirstlane(w.kx * g.cs),
                    __builtin_amdgcn_readfirstlane(w.cc * BK), __builtin_amdgcn_readfirstlane(widx)};
        if (++w.cc == cch) {
            w.cc = 0;
            w.kx += 2;
            if (w.kx >= g.KW) {
                w.kx = k0x;
                w.ky += 2;
            }
        }
        return t;
    };
    Walk wa{k0y, k0x, 0}, wb{k0y, k0x, 0};                              // A side (3 steps in flight), B side (4 ahead)

    bf16x8 bq[NSET][4];
    auto load_b = [&](const Tap& t, bf16x8 (&bb)[4]) {
        const __bf16* src = wsplit + (size_t)t.widx * npad * BK + ncl;
        bb[0] = *reinterpret_cast<const bf16x8*>(src);
        bb[1] = *reinterpret_cast<const bf16x8*>(src + plane);
        bb[2] = *reinterpret_cast<const bf16x8*>(src + 16);
        bb[3] = *reinterpret_cast<const bf16x8*>(src + plane + 16);
    };
    load_b(advance(wb), bq[0]);
    load_b(advance(wb), bq[1]);
    load_b(advance(wb), bq[2]);
    load_b(advance(wb), bq[3]);
    const float bv = p.bias ? p.bias[n < g.N ? n : 0] : 0.f;

    if (tid < 32) {
        reinterpret_cast<unsigned*>(Ph + zoff)[tid] = 0u;
        reinterpret_cast<unsigned*>(Pl + zoff)[tid] = 0u;
    }
    {   // the source image: f32 -> in_act -> hi / lo bf16, a linear read
        const float* img = p.in + (size_t)b * npos * g.C;
        const int c4n = g.C >> 2;
        const int total = npos * c4n;
        const bool in_relu = g.in_act == PM_ACT_RELU;
        const float in_ns = g.in_act == PM_ACT_LEAKY ? g.slope : 1.f;
        constexpr int PB = 8;
        for (int e0 = tid; e0 < total; e0 += 64 * NW * PB) {
            f32x4 v[PB];
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int e = e0 + 64 * NW * j;
                v[j] = *reinterpret_cast<const f32x4*>(img + 4 * (size_t)(e < total ? e : total - 1));
            }
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int e = e0 + 64 * NW * j;
                const int ee = e < total ? e : total - 1;
                const int pos = ee / c4n, c4 = ee - pos * c4n;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float x = v[j][q];
                    const float neg = in_relu ? 0.f : x * in_ns;
                    v[j][q] = x >= 0.f ? x : neg;
                }
                u32x2 h2, l2;
                split4(v[j], h2, l2);
                *reinterpret_cast<u32x2*>(Ph + pos * PS + 4 * c4) = h2;
                *reinterpret_cast<u32x2*>(Pl + pos * PS + 4 * c4) = l2;
            }
        }
    }

    // per row tile: this lane's A row = class position m -> dst (oy, ox); py = oy + off, px = ox + offx (source coordinate
    // times two before the tap is added)
    int py[T], px[T];
#pragma unroll
    for (int j = 0; j < T; ++j) {
        const int m = 32 * (rt0 + j * wct) + i;
        const int cy = m / OWc, cx = m - cy * OWc;
        py[j] = m < Mc ? 2 * cy + ry + g.off : ROW_INVALID;
        px[j] = 2 * cx + rx + g.offx;
    }
    f32x16 acc[T];
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    __syncthreads();                 // image visible; the only barrier of the kernel

    bf16x8 a[4][4];
    auto read_a = [&](const Tap& k, int j, bf16x8 (&aa)[4]) {
        const int iy = (py[j] + k.dy) >> 1, ix = (px[j] + k.dx) >> 1;    // even by construction of the class's tap list
        const bool ok = (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW;
        const int o = (ok ? (iy * g.IW + ix) * PS + k.c0 : zoff) + 8 * h;
        aa[0] = *reinterpret_cast<const bf16x8*>(Ph + o);
        aa[1] = *reinterpret_cast<const bf16x8*>(Ph + o + 16);
        aa[2] = *reinterpret_cast<const bf16x8*>(Pl + o);
        aa[3] = *reinterpret_cast<const bf16x8*>(Pl + o + 16);
    };
    Tap t0, t1, t2;
    auto step = [&](auto uc) {
        constexpr int u = decltype(uc)::value;
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const int cur = (u * T + j) & 3;
            const int jn = j + 2;
            read_a(jn / T == 0 ? t0 : jn / T == 1 ? t1 : t2, jn % T, a[(cur + 2) & 3]);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][0], bq[u][0], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][0], bq[u][1], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][2], bq[u][0], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][1], bq[u][2], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][1], bq[u][3], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][3], bq[u][2], acc[j], 0, 0, 0);
        }
        t0 = t1;
        t1 = t2;
        t2 = advance(wa);
        load_b(advance(wb), bq[u]);
        __builtin_amdgcn_sched_barrier(0);
    };
    using U0 = std::integral_constant<int, 0>;
    using U1 = std::integral_constant<int, 1>;
    using U2 = std::integral_constant<int, 2>;
    using U3 = std::integral_constant<int, 3>;
    static_assert(T == 1 || T == 2 || T == 4, "item index mod 4 must be static inside a group of NSET k-steps");
    if (rt0 * 32 < Mc && nsteps > 0) {
        t0 = advance(wa);
        t1 = advance(wa);
        t2 = advance(wa);
        read_a(t0, 0, a[0]);
        read_a(T == 1 ? t1 : t0, T == 1 ? 0 : 1, a[1]);
        int s0 = 0;
        for (; s0 + NSET <= nsteps; s0 += NSET) {
            step(U0{});
            step(U1{});
            step(U2{});
            step(U3{});
        }
        if (s0 < nsteps) {
            step(U0{});
            if (s0 + 1 < nsteps) {
                step(U1{});
                if (s0 + 2 < nsteps) step(U2{});
            }
        }
    }
    if (rt0 * 32 < Mc && n < g.N) {                                      // a class without taps (1x1 kernels) still owes bias
#pragma unroll
        for (int j = 0; j < T; ++j) {
            int ro[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = 32 * (rt0 + j * wct) + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int cy = m / OWc, cx = m - cy * OWc;
                ro[e] = m < Mc ? ((b * g.OH + 2 * cy + ry) * g.OW + 2 * cx + rx) * g.N : -1;
            }
            pm_epilogue_tile(acc[j], ro, n, bv, p.aux, p.res, p.out, p.out2, p.act2, g.aux_act, g.out_act, g.slope);
        }
    }
}

bool plan_image_d2(const Geom& g, int groups, ImagePlan& ip) {
    if (groups != 1 || g.d != 2 || g.a != 1 || g.C % BK != 0 || g.B < 128) return false;
    if (g.in_act != PM_ACT_NONE && g.in_act != PM_ACT_RELU && g.in_act != PM_ACT_LEAKY) return false;
    if (g.KH * g.KW < 4 || g.kws != g.KW || (g.OH & 1) || (g.OW & 1)) return false;
    if (g.IH * 2 < g.KH || g.IW * 2 < g.KW) return false;
    ip.lds = 2 * ((size_t)g.IH * g.IW * (g.C + 8) + 64) * 2;
    if (ip.lds > 158 * 1024) return false;
    ip.nct = (g.N + 31) / 32;
    if (ip.nct != 1 && ip.nct != 2) return false;
    ip.nw = 8;
    const int wct = 8 / (4 * ip.nct);
    const int rt = ((g.OH >> 1) * (g.OW >> 1) + 31) / 32;
    int t = (rt + wct - 1) / wct;
    if (t > 4) return false;
    ip.t = t == 3 ? 4 : t;
    return true;
}

template <int T>
void launch_image_d2(const ImagePlan& ip, hipStream_t s, const GemmArgs& a, const __bf16* ws, int npad, long long plane) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&image_d2_bf16_kernel<T>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    PM_KTAG("image_d2_bf16_kernel<%d>", T);
    hipLaunchKernelGGL((image_d2_bf16_kernel<T>), dim3((unsigned)a.g.B), dim3(512), ip.lds, s, a, ws, npad, plane, ip.nct);
}

// ------------- zero-dilated (d = 2) problems, patch-staged with the four residue classes fused -------------
// Stride-2 transposed convolutions (forward) and the data gradients of stride-2 convolutions:  src * 2 = dst + tap * cs + off.
// An output position of residue class (ry, rx) = (dst_y & 1, dst_x & 1) only meets the taps with
// (ry + ky * cs + off) even (likewise in x), at source row i + (ry + ky * cs + off) / 2 with i = dst_y >> 1: four small
// stride-1 convolutions over the SAME source neighbourhood.  A workgroup therefore owns a tile of class coordinates (i, j)
// (8 x 16 of one image, or 8 x 8 of two images when the class grid is at most 8 wide), stages that neighbourhood ONCE
// (f32 -> hi / lo bf16, zeros outside the image; 14 - 58 KB of LDS), and walks the four classes one after the other:
// per class the valid taps only, every A fragment a 16-byte LDS read, weights through the usual double-buffered stages.
// The direct form re-gathers its rows from L2 once per tap and runs at 36 - 46 TFLOP/s on these layers.
struct KStepD2 {
    int aoff;   // patch offset (bf16 elements) of this step's source shift and channel chunk
    int woff;   // offset of this step's [Npad][32] hi block inside the split weights
};

template <int RN>
__global__ __launch_bounds__(256) void patch_d2_bf16_kernel(GemmArgs p, const __bf16* __restrict__ wsplit, int npad,
                                                             long long plane, int tw_log2, int ni) {
    constexpr int NB = 32 * RN;
    constexpr int STEP_E = 2 * NB * BROW;
    constexpr int BTILE = PDGS * STEP_E;
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    __bf16* Bs = reinterpret_cast<__bf16*>(dsm);                        // 2 stages = BTILE floats
    int* meta = reinterpret_cast<int*>(dsm + BTILE);                    // [0..4]: first step of each class, [5..8]: shifts
    KStepD2* kd = reinterpret_cast<KStepD2*>(meta + 16);

    const Geom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 31;
    const int h = lane >> 5;
    const int n0 = blockIdx.y * NB;
    const int TW = 1 << tw_log2;
    const int TH = 128 / (TW * ni);
    const int cch = g.C / BK;
    const int CHg = g.OH >> 1, CWg = g.OW >> 1;                         // class grid (OH, OW even)
    const int PS = g.C + 8;

    // shifts delta(r, t) = (r + t * cs + off) / 2 of the valid taps: range over both residues, per axis
    auto shift = [&](int r, int t, int off, int& dl) {
        const int v = r + t * g.cs + off;
        dl = v >> 1;                                                    // arithmetic shift: exact when v is even
        return (v & 1) == 0;
    };
    int dymin = 1 << 20, dymax = -(1 << 20), dxmin = 1 << 20, dxmax = -(1 << 20);
    for (int r = 0; r < 2; ++r) {
        for (int t = 0; t < g.KH; ++t) {
            int dl;
            if (shift(r, t, g.off, dl)) { dymin = dl < dymin ? dl : dymin; dymax = dl > dymax ? dl : dymax; }
        }
        for (int t = 0; t < g.KW; ++t) {
            int dl;
            if (shift(r, t, g.offx, dl)) { dxmin = dl < dxmin ? dl : dxmin; dxmax = dl > dxmax ? dl : dxmax; }
        }
    }
    const int PH = TH + dymax - dymin, PW = TW + dxmax - dxmin;
    const int nsteps_max = g.KH * g.KW * cch;
    __bf16* Ph = reinterpret_cast<__bf16*>(kd + ((nsteps_max + PDGS + 1) & ~1));   // 16-byte aligned
    __bf16* Pl = Ph + (size_t)ni * PH * PW * PS;

    // tile -> images and class-grid origin
    const int tiles_x = (CWg + TW - 1) >> tw_log2;
    const int tiles_y = (CHg + TH - 1) / TH;
    int t = blockIdx.x;
    const int txi = t % tiles_x;
    t /= tiles_x;
    const int tyi = t % tiles_y;
    const int b0 = (t / tiles_y) * ni;
    const int i0 = tyi * TH, j0 = txi << tw_log2;

    // step table: class c = ry * 2 + rx, its taps in (ky, kx) order, channel chunks innermost
    if (tid < 4) {
        const int ry = tid >> 1, rx = tid & 1;
        int before = 0;
        for (int c = 0; c < tid; ++c) {
            int ny = 0, nx = 0, dl;
            for (int tt = 0; tt < g.KH; ++tt) ny += shift(c >> 1, tt, g.off, dl) ? 1 : 0;
            for (int tt = 0; tt < g.KW; ++tt) nx += shift(c & 1, tt, g.offx, dl) ? 1 : 0;
            before += ny * nx * cch;
        }
        int s = before;
        for (int ky = 0; ky < g.KH; ++ky) {
            int dy;
            if (!shift(ry, ky, g.off, dy)) continue;
            for (int kx = 0; kx < g.KW; ++kx) {
                int dx;
                if (!shift(rx, kx, g.offx, dx)) continue;
                for (int cc = 0; cc < cch; ++cc) {
                    kd[s].aoff = ((dy - dymin) * PW + (dx - dxmin)) * PS + cc * BK;
                    kd[s].woff = ((ky * g.KW + kx) * cch + cc) * npad * BK;
                    ++s;
                }
            }
        }
        meta[tid] = before;
        if (tid == 3) {
            meta[4] = s;
            for (int e = 0; e < PDGS; ++e) kd[s + e] = KStepD2{0, 0};
        }
    }
    {   // the source neighbourhood of the tile, per image: f32 -> hi / lo bf16, zero outside the image
        const int c4n = g.C >> 2;
        const int per_img = PH * PW * c4n;
        const int total = ni * per_img;
        constexpr int PB = 8;
        for (int e0 = tid; e0 < total; e0 += 256 * PB) {
            f32x4 v[PB];
            int dst[PB];
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int e = e0 + 256 * j;
                const int ee = e < total ? e : total - 1;
                const int im = ee / per_img, r = ee - im * per_img;
                const int pos = r / c4n, c4 = r - pos * c4n;
                const int py = pos / PW, px = pos - py * PW;
                const int gy = i0 + dymin + py, gx = j0 + dxmin + px;
                const bool ok = e < total && b0 + im < g.B && (unsigned)gy < (unsigned)g.IH && (unsigned)gx < (unsigned)g.IW;
                const size_t so = ok ? (((size_t)(b0 + im) * g.IH + gy) * g.IW + gx) * g.C + 4 * c4 : 0;
                v[j] = *reinterpret_cast<const f32x4*>(p.in + so);
                if (!ok) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                dst[j] = e < total ? (im * PH * PW + pos) * PS + 4 * c4 : -1;
            }
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                if (dst[j] < 0) continue;
                u32x2 h2, l2;
                split4(v[j], h2, l2);
                *reinterpret_cast<u32x2*>(Ph + dst[j]) = h2;
                *reinterpret_cast<u32x2*>(Pl + dst[j]) = l2;
            }
        }
    }

    constexpr int PIECES = 2 * NB * 4;
    constexpr int PPT = (PIECES + 255) / 256;
    u32x4 breg[PDGS][PPT];
    auto load_b = [&](int s0) {
#pragma unroll
        for (int st = 0; st < PDGS; ++st) {
            const int wo = kd[s0 + st].woff;
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                const int pc = tid + 256 * j;
                const int pl = pc / (NB * 4);
                const int row = (pc / 4) % NB;
                const int qtr = pc & 3;
                const int nrow = n0 + row < npad ? n0 + row : 0;
                breg[st][j] = *reinterpret_cast<const u32x4*>(wsplit + (size_t)pl * plane + wo + nrow * BK + qtr * 8);
            }
        }
    };
    auto store_b = [&](__bf16* dstp) {
#pragma unroll
        for (int st = 0; st < PDGS; ++st)
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                const int pc = tid + 256 * j;
                const int pl = pc / (NB * 4);
                const int row = (pc / 4) % NB;
                const int qtr = pc & 3;
                *reinterpret_cast<u32x4*>(dstp + st * STEP_E + (pl * NB + row) * BROW + qtr * 8) = breg[st][j];
            }
    };

    // this lane's class coordinate and its A-fragment base inside the patch
    const int ml = wave * 32 + i;
    const int im = ml / (TH * TW);
    const int rem = ml - im * TH * TW;
    const int ty = rem >> tw_log2, tx = rem & (TW - 1);
    const int abase = ((im * PH + ty) * PW + tx) * PS + 8 * h;
    const int ci = i0 + ty, cj = j0 + tx;
    const bool row_ok = ci < CHg && cj < CWg && b0 + im < g.B;

    const float* bias = p.bias;
    const float* aux = p.aux;
    const float* res = p.res;
    float* out = p.out;
    __syncthreads();                 // step table + patch visible

    // short grids (7 x 7 class grids at B = 256: 128 tiles) spread the four classes over blockIdx.z instead
    const int cls_begin = gridDim.z == 4 ? (int)blockIdx.z : 0;
    const int cls_end = gridDim.z == 4 ? cls_begin + 1 : 4;
    for (int cls = cls_begin; cls < cls_end; ++cls) {
        const int sb = meta[cls], se = meta[cls + 1];
        f32x16 acc[RN];
#pragma unroll
        for (int r = 0; r < RN; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
        if (sb < se) {
            load_b(sb);
            store_b(Bs);
        }
        __syncthreads();
        bf16x8 ah[2][2], al[2][2];
        auto read_a = [&](int s, int set) {
            const int o = abase + kd[s].aoff;
            ah[set][0] = *reinterpret_cast<const bf16x8*>(Ph + o);
            ah[set][1] = *reinterpret_cast<const bf16x8*>(Ph + o + 16);
            al[set][0] = *reinterpret_cast<const bf16x8*>(Pl + o);
            al[set][1] = *reinterpret_cast<const bf16x8*>(Pl + o + 16);
        };
        auto mma = [&](const __bf16* bstep, int set) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int r = 0; r < RN; ++r) {
                    const __bf16* brow = bstep + (r * 32 + i) * BROW + 16 * kk + 8 * h;
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(brow);
                    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(brow + NB * BROW);
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[set][kk], bh, acc[r], 0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[set][kk], bl, acc[r], 0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[set][kk], bh, acc[r], 0, 0, 0);
                }
        };
        if (sb < se) read_a(sb, 0);
        int stage = 0;
        for (int s0 = sb; s0 < se; s0 += PDGS) {
            const __bf16* bcur = Bs + stage * BTILE;
            const bool more = s0 + PDGS < se;
            if (more) load_b(s0 + PDGS);
#pragma unroll
            for (int st = 0; st < PDGS; ++st) {
                if (s0 + st < se) {
                    if (s0 + st + 1 < se) read_a(s0 + st + 1, (st + 1) & 1);
                    mma(bcur + st * STEP_E, st & 1);
                }
            }
            if (more) store_b(Bs + (stage ^ 1) * BTILE);
            __syncthreads();
            stage ^= 1;
        }
        // epilogue of this class: output position (2 ci + ry, 2 cj + rx)
        const int ry = cls >> 1, rx = cls & 1;
        const int rowoff = row_ok ? (((b0 + im) * g.OH + 2 * ci + ry) * g.OW + 2 * cj + rx) * g.N : -1;
        int ro[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) ro[e] = __shfl(rowoff, (e & 3) + 8 * (e >> 2) + 4 * h, 64);
#pragma unroll
        for (int r = 0; r < RN; ++r) {
            const int n = n0 + r * 32 + i;
            if (n >= g.N) continue;
            const float bv = bias ? bias[n] : 0.f;
            pm_epilogue_tile(acc[r], ro, n, bv, aux, res, out, p.out2, p.act2, g.aux_act, g.out_act, g.slope);
        }
        __syncthreads();             // stage buffers are rewritten by the next class
    }
}

struct PatchD2Plan { int tw_log2, ni; size_t lds; dim3 grid; };
bool plan_patch_d2(const Geom& g, int groups, int rn, PatchD2Plan& pp) {
    if (groups != 1 || g.a != 1 || g.d != 2 || g.C % BK != 0 || g.in_act != PM_ACT_NONE) return false;
    if ((g.cs != 1 && g.cs != -1) || (g.OH & 1) || (g.OW & 1) || g.KH > 8 || g.KW > 8 || g.KH * g.KW < 4) return false;
    const int CHg = g.OH / 2, CWg = g.OW / 2;
    if (CHg > g.IH + 4 || CWg > g.IW + 4) return false;
    // Class grids at most 8 wide (two images per tile, optionally one class per workgroup) are implemented and
    // parity-tested but measured SLOWER than the direct form (7x7 -> 14x14, 64 -> 64 at B = 256: 110 / 76 us against 57 us:
    // 128 tiles cannot fill the chip, and a class per workgroup re-stages the patch four times): only wider grids qualify.
    static const bool small_on = getenv("PM_PATCH_D2_SMALL") != nullptr;
    if (CWg > 8) { pp.tw_log2 = 4; pp.ni = 1; }
    else if (CHg <= 8 && small_on) { pp.tw_log2 = 3; pp.ni = 2; }
    else return false;
    const int TW = 1 << pp.tw_log2, TH = 128 / (TW * pp.ni);
    const int tiles_x = (CWg + TW - 1) / TW, tiles_y = (CHg + TH - 1) / TH;
    if ((long long)tiles_x * TW * tiles_y * TH * 2 > 3LL * CHg * CWg) return false;     // > 50 % padded slots
    // shift range: at most (K + 1) / 2 distinct shifts per axis
    const int spanh = (g.KH + 1) / 2 + 1, spanw = (g.KW + 1) / 2 + 1;
    const int NB = 32 * rn;
    const size_t bt = (size_t)PDGS * 2 * NB * BROW * 2 * 2;
    const int nsteps = g.KH * g.KW * (g.C / BK);
    if (nsteps > 2048) return false;
    const size_t kdb = 64 + (size_t)((nsteps + PDGS + 1) & ~1) * sizeof(KStepD2);
    const size_t patch = (size_t)pp.ni * (TH + spanh) * (TW + spanw) * (g.C + 8) * 2 * 2;
    pp.lds = bt + kdb + patch;
    if (pp.lds > 150 * 1024) return false;
    const int nimg_tiles = (g.B + pp.ni - 1) / pp.ni;
    pp.grid = dim3((unsigned)(nimg_tiles * tiles_y * tiles_x), (g.N + NB - 1) / NB, 1);
    if ((long long)pp.grid.x * pp.grid.y < 384) pp.grid.z = 4;       // one residue class per workgroup
    return true;
}

template <int RN, int DD>
void launch_direct_bf16(hipStream_t s, const GemmArgs& a, dim3 grid, const __bf16* ws, int npad, long long plane) {
    const Geom& g = a.g;
    // plain matrix rows: hk.Linear, and every 1x1 stride-1 unpadded convolution (position m of the output IS position m
    // of the input in NHWC order) - no taps, no coordinates, no tap lists: 6 us less per launch on the VDVAE's 1x1 layers
    const bool dense = DD == 1 && g.KH == 1 && g.KW == 1 && g.IH == g.OH && g.IW == g.OW && g.a == 1 && g.off == 0 &&
                       g.offx == 0 && g.PY == 1 && g.PX == 1;
    PM_KTAG("direct_gemm_bf16_kernel<%d, %d, %d, %s>", RN, dense ? 1 : DD,
            g.in_act == PM_ACT_RELU || g.in_act == PM_ACT_LEAKY ? g.in_act : 0, dense ? "true" : "false");
#define PM_LB(ACT)                                                                                                   \
    do {                                                                                                              \
        if (dense) hipLaunchKernelGGL((direct_gemm_bf16_kernel<RN, 1, ACT, true>), grid, dim3(256), 0, s, a, ws, npad, plane); \
        else hipLaunchKernelGGL((direct_gemm_bf16_kernel<RN, DD, ACT, false>), grid, dim3(256), 0, s, a, ws, npad, plane);     \
    } while (0)
    switch (g.in_act) {
        case PM_ACT_RELU: PM_LB(PM_ACT_RELU); break;
        case PM_ACT_LEAKY: PM_LB(PM_ACT_LEAKY); break;
        default: PM_LB(PM_ACT_NONE); break;
    }
#undef PM_LB
}

// Pre-split, K-contiguous copy of one layer's weights for one direction:
//   dst[plane][tap][cchunk][n < npad][32]  with plane 0 = hi, 1 = lo,  value = w[tap*wts + c*wcs + n*wns]
struct SplitJob {
    long long src_off;   // element offset of the layer's weights in the flat f32 parameter buffer
    long long dst_off;   // element offset of the job's output in the bf16 buffer
    long long plane;     // elements between the hi and lo planes
    int taps, C, N, npad;
    int wts, wcs, wns;
    int first_block, num_blocks;
    int kw, kws;         // compact tap t reads weight-layout tap (t / kw) * kws + t % kw
    int dense_k;         // 1: K runs over (tap, c) WITHOUT per-tap padding: dst[plane][ceil(taps*C/32)][npad][32], k = tap * C + c
};

// One workgroup = SPLIT_TPB consecutive (tap, 32-channel chunk, 32-column tile) tiles of 1024 elements each, usually of one
// job.  A source tile is read along whichever of its axes is contiguous in the parameter buffer (columns n for HWIO forward
// weights, channels c for the data-gradient direction) and written k-contiguous; a 32 x 33 LDS tile does the transpose.
// All 4 x SPLIT_TPB loads of a thread are issued before the first is used, the job of the first tile is found with one
// parallel pass over the job table (not a 9-deep chain of dependent loads per 1024 elements), and a thread stores 4 bf16 =
// 8 bytes per plane and tile: ~99 -> 63 us per launch at the 35 M-parameter PixelCNN (steady state; rocprofv3 averages of
// this kernel carry one ~15 ms first-touch call).
// SPLIT_TPB = 8 for large stores, 2 for small ones (the PM-VAE's 4 k tiles are 2 k workgroups instead of 500).

struct SplitTile {
    long long base, dst0, plane;
    int cc, nt, wcs, wns, C, N;
    int dense, taps, kw, kws, wts;           // dense_k jobs: the tap of an element depends on its k (base = src_off)
    bool n_contig;
};

__device__ __forceinline__ SplitTile split_tile_of(const SplitJob& job, int t) {
    SplitTile r;
    // channel chunks per tap, the last one zero-padded; dense_k: chunks of the whole (tap, c) range, only the very last padded
    const int cch = job.dense_k ? (job.taps * job.C + BK - 1) / BK : (job.C + BK - 1) / BK;
    const int ntiles = job.npad / 32;
    r.nt = t % ntiles;
    t /= ntiles;
    r.cc = t % cch;
    const int tap = t / cch;                 // 0 for dense_k jobs
    r.base = job.src_off + (long long)((tap / job.kw) * job.kws + tap % job.kw) * job.wts;
    r.dst0 = job.dst_off + ((long long)(tap * cch + r.cc) * job.npad + 32 * r.nt) * BK;
    r.plane = job.plane;
    r.wcs = job.wcs, r.wns = job.wns, r.C = job.C, r.N = job.N;
    r.dense = job.dense_k, r.taps = job.taps, r.kw = job.kw, r.kws = job.kws, r.wts = job.wts;
    r.n_contig = job.wns <= job.wcs;         // which source axis is the faster one
    return r;
}

// source offset (without the column term) of element k of a tile's 32-deep chunk, or -1 past the end of K
__device__ __forceinline__ long long split_src_k(const SplitTile& t, int k) {
    const int kk = t.cc * BK + k;
    if (!t.dense) return kk < t.C ? t.base + (long long)kk * t.wcs : -1;
    if (kk >= t.taps * t.C) return -1;
    const int tap = kk / t.C, c = kk - tap * t.C;
    return t.base + (long long)((tap / t.kw) * t.kws + tap % t.kw) * t.wts + (long long)c * t.wcs;
}

template <int SPLIT_TPB>
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ params, __bf16* __restrict__ out,
                                                              const SplitJob* __restrict__ jobs, int njobs,
                                                              int total_tiles) {
    __shared__ float tile[SPLIT_TPB][32][33];
    __shared__ int s_job;
    const int vb0 = (int)blockIdx.x * SPLIT_TPB;
    if (vb0 >= total_tiles) return;
    if (threadIdx.x == 0) s_job = -1;
    __syncthreads();
    // jobs are sorted by first_block and cover [0, total_tiles) without gaps: exactly one of them holds tile vb0
    for (int i = threadIdx.x; i < njobs; i += 256) {
        const int fb = jobs[i].first_block;
        if (vb0 >= fb && vb0 < fb + jobs[i].num_blocks) s_job = i;
    }
    __syncthreads();
    int ji = s_job;
    if (ji < 0) return;                                              // a job table with gaps: nothing to do for this range
    SplitJob job = jobs[ji];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8 threads, 4 passes per tile
    const int ntl = min(SPLIT_TPB, total_tiles - vb0);
    SplitTile tl[SPLIT_TPB];
    float v[SPLIT_TPB][4];
#pragma unroll
    for (int q = 0; q < SPLIT_TPB; ++q) {
        if (q < ntl) {
            while (vb0 + q >= job.first_block + job.num_blocks) job = jobs[++ji];   // workgroup-uniform
            tl[q] = split_tile_of(job, vb0 + q - job.first_block);
#pragma unroll
            for (int pss = 0; pss < 4; ++pss) {
                const int a = ty + 8 * pss;                                  // slow index of this pass
                const int k = tl[q].n_contig ? a : tx;                       // channel within the chunk
                const int n = (tl[q].n_contig ? tx : a) + 32 * tl[q].nt;     // output column
                v[q][pss] = 0.f;
                const long long so = split_src_k(tl[q], k);
                if (n < tl[q].N && so >= 0) v[q][pss] = params[so + (long long)n * tl[q].wns];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < SPLIT_TPB; ++q) {
        if (q < ntl) {
#pragma unroll
            for (int pss = 0; pss < 4; ++pss) {
                const int a = ty + 8 * pss;
                tile[q][tl[q].n_contig ? a : tx][tl[q].n_contig ? tx : a] = v[q][pss];   // tile[k][n - 32 nt]
            }
        }
    }
    __syncthreads();
    const int k4 = (threadIdx.x & 7) * 4, nl = threadIdx.x >> 3;     // 4 consecutive k of one column: 8-byte stores
#pragma unroll
    for (int q = 0; q < SPLIT_TPB; ++q) {
        if (q < ntl) {
            typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
            bf4 h, l;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = tile[q][k4 + j][nl];
                h[j] = (__bf16)x;
                l[j] = (__bf16)(x - (float)h[j]);
            }
            __bf16* d = out + tl[q].dst0 + (long long)nl * BK + k4;
            *reinterpret_cast<bf4*>(d) = h;
            *reinterpret_cast<bf4*>(d + tl[q].plane) = l;
        }
    }
}

// ------------------------------------ weight gradient ----------------------------------------
// dw[kk][n] += sum_m G[m][kk] * D[m][n].  A workgroup owns a (32*RC) x (32*RN) block of dw and a
// range of 128-row chunks of m; its 4 waves each take 32 rows of a chunk (MFMA k = 2 rows per
// instruction), are summed through LDS at the end and added to global memory with f32 atomics.
// In the tap-uniform modes the block's tap and channel range are fixed for the whole kernel and
// the per-row state advances incrementally from chunk to chunk.
template <int RC, int RN, int MODE, int DVEC>
__global__ __launch_bounds__(256) void gather_wgrad_kernel(WgradArgs p) {
    constexpr bool TU = (MODE == MODE_TU1 || MODE == MODE_TU2);
    constexpr int CB = 32 * RC;
    constexpr int NB = 32 * RN;
    constexpr int BMC = 128;
    constexpr int TILE = BMC * (CB + NB);
    constexpr int RED = 4 * CB * NB;
    constexpr int SMEM = TILE > RED ? TILE : RED;
    constexpr int TL_F = (sizeof(TapList) + 3) / 4;
    __shared__ __attribute__((aligned(16))) float smem[SMEM + TL_F];
    float* Gs = smem;
    float* Ds = smem + BMC * CB;
    TapList* tl = reinterpret_cast<TapList*>(smem + SMEM);  // identity: every tap, natural order
    if (threadIdx.x < MAXTAP) {
        tl->ky[threadIdx.x] = threadIdx.x;
        tl->kx[threadIdx.x] = threadIdx.x;
    }
    if (threadIdx.x == 0) {
        tl->nvy = p.g.KH;
        tl->nvx = p.g.KW;
    }
    __syncthreads();

    const Geom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 31;
    const int h = lane >> 5;
    const int nkb = (g.K + CB - 1) / CB;
    const int kkb = blockIdx.x % nkb;
    const int nb = blockIdx.x / nkb;
    const int kk0 = kkb * CB;
    const int n0 = nb * NB;
    const int grp = blockIdx.z;
    const float* gin = p.gathered + wg_off(p, grp, 0, p.in_gs);
    const float* din = p.dense + wg_off(p, grp, 1, p.out_gs);
    const bool do_bias = (p.db != nullptr) && (kkb == 0);
    // tap-uniform modes: this block's tap and first channel
    const int tap_u = kk0 / g.C;
    const int c_u = kk0 - tap_u * g.C;
    const int ky_u = tap_u / g.KW;
    const int kx_u = tap_u - ky_u * g.KW;

    const int total_chunks = (g.M + BMC - 1) / BMC;
    const int c_begin = blockIdx.y * p.chunks_per_split;
    int c_end = c_begin + p.chunks_per_split;
    if (c_end > total_chunks) c_end = total_chunks;

    typename LoaderSel<BMC, CB, MODE>::type lg;
    lg.init(tid);

    // dense tile loader: Ds[row][n]
    constexpr int DSLOTS = NB / DVEC;
    constexpr int DRPP = 256 / DSLOTS;
    constexpr int DNP = BMC / DRPP;
    const int dslot = tid % DSLOTS;
    const int dr0 = tid / DSLOTS;
    const int dn = n0 + dslot * DVEC;
    const bool dn_ok = dn < g.N;
    float dreg[DNP][DVEC];
    unsigned dmask = 0u;
    auto load_d = [&](int m0) {
        dmask = 0u;
#pragma unroll
        for (int j = 0; j < DNP; ++j) {
            int m = m0 + dr0 + j * DRPP;
            const bool ok = m < g.M && dn_ok;
            const size_t o = ok ? (size_t)m * g.N + dn : 0;   // branch-free load + select
            dmask |= (ok ? 1u : 0u) << j;
            if (DVEC == 4) {
                f32x4 v = *reinterpret_cast<const f32x4*>(din + o);
#pragma unroll
                for (int e = 0; e < DVEC; ++e) dreg[j][e] = v[e];
            } else {
                dreg[j][0] = din[o];
            }
        }
    };
    auto store_d = [&]() {
#pragma unroll
        for (int j = 0; j < DNP; ++j) {
            const bool ok = (dmask >> j) & 1u;
            if (DVEC == 4) {
                f32x4 v = {dreg[j][0], dreg[j][1 % DVEC], dreg[j][2 % DVEC], dreg[j][3 % DVEC]};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
                *reinterpret_cast<f32x4*>(Ds + (dr0 + j * DRPP) * NB + dslot * DVEC) = v;
            } else {
                Ds[(dr0 + j * DRPP) * NB + dslot] = ok ? dreg[j][0] : 0.f;
            }
        }
    };
    auto load_g = [&]() {
        if constexpr (TU)
            lg.load_tap(g, gin, ky_u, kx_u, c_u, true);
        else
            lg.load_flat(g, gin, kk0, g.K, tl);
    };

    f32x16 acc[RC][RN];
    f32x16 accb[RN];
#pragma unroll
    for (int a = 0; a < RC; ++a)
#pragma unroll
        for (int b = 0; b < RN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
#pragma unroll
    for (int b = 0; b < RN; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) accb[b][e] = 0.f;

    if (c_begin < c_end) {
        lg.set_rows(g, c_begin * BMC);
        load_g();
        load_d(c_begin * BMC);
    }
    for (int ch = c_begin; ch < c_end; ++ch) {
        lg.store(g, Gs, CB);
        store_d();
        __syncthreads();
        if (ch + 1 < c_end) {
            lg.advance_rows(g, (ch + 1) * BMC, p.step_b, p.step_p, p.step_q);
            load_g();
            load_d((ch + 1) * BMC);
        }
        const float* grow = Gs + (wave * 32 + h) * CB + i;
        const float* drow = Ds + (wave * 32 + h) * NB + i;
        // two copies of the MFMA loop, selected by one uniform branch: a bias MFMA under a branch
        // INSIDE the loop makes hipcc shuttle the bias accumulators between AGPRs and VGPRs
        auto mma = [&](auto with_bias) {
#pragma unroll 4
            for (int t = 0; t < 16; ++t) {
                float av[RC], bv[RN];
#pragma unroll
                for (int a = 0; a < RC; ++a) av[a] = grow[2 * t * CB + 32 * a];
#pragma unroll
                for (int b = 0; b < RN; ++b) bv[b] = drow[2 * t * NB + 32 * b];
#pragma unroll
                for (int a = 0; a < RC; ++a)
#pragma unroll
                    for (int b = 0; b < RN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
                if constexpr (decltype(with_bias)::value) {
#pragma unroll
                    for (int b = 0; b < RN; ++b)
                        accb[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, bv[b], accb[b], 0, 0, 0);
                }
            }
        };
        if (do_bias) mma(std::true_type{});
        else mma(std::false_type{});
        __syncthreads();
    }

    // cross-wave reduction through LDS, then one atomic per element per workgroup
    float* red = smem;
#pragma unroll
    for (int a = 0; a < RC; ++a)
#pragma unroll
        for (int b = 0; b < RN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int cl = a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                int nl = b * 32 + i;
                red[wave * CB * NB + cl * NB + nl] = acc[a][b][e];
            }
    __syncthreads();
    const bool part = p.part_w != nullptr;
    float* dw = wg_dw(p, (int)blockIdx.y) + wg_off(p, grp, 2, p.w_gs);
    for (int e = tid; e < CB * NB; e += 256) {
        int cl = e / NB;
        int nl = e - cl * NB;
        int kk = kk0 + cl;
        int n = n0 + nl;
        if (kk < g.K && n < g.N) {
            float s = red[e] + red[CB * NB + e] + red[2 * CB * NB + e] + red[3 * CB * NB + e];
            int tap = kk / g.C;
            int c = kk - tap * g.C;
            const int wtap = (tap / g.KW) * g.kws + tap % g.KW;   // compact tap -> tap of the weight layout
            wg_put(part, dw + (size_t)wtap * g.wts + (size_t)c * g.wcs + (size_t)n * g.wns, s);
        }
    }
    if (do_bias && !part && h == 0) {
        float* db = p.db + wg_off(p, grp, 3, p.bias_gs);
#pragma unroll
        for (int b = 0; b < RN; ++b) {
            int n = n0 + b * 32 + i;
            if (n < g.N) atomicAdd(db + n, accb[b][0]);
        }
    }
    if (do_bias && part) {       // one writer per slot element: the four waves' row shares meet in LDS (workgroup-uniform branch)
        __syncthreads();
        if (h == 0) {
#pragma unroll
            for (int b = 0; b < RN; ++b) red[wave * NB + b * 32 + i] = accb[b][0];
        }
        __syncthreads();
        float* db = wg_db(p, (int)blockIdx.y) + wg_off(p, grp, 3, p.bias_gs);
        if (tid < NB && n0 + tid < g.N) db[n0 + tid] = (red[tid] + red[NB + tid]) + (red[2 * NB + tid] + red[3 * NB + tid]);
    }
}

// ----------------------- weight gradient on the bf16 matrix cores (bf16x3) ---------------------
// Same decomposition as gather_wgrad_kernel, but the two 128-row tiles are converted ONCE at
// staging time into hi / lo bf16 planes and every MFMA operand is a transposed LDS read
// (ds_read_b64_tr_b16: 4 rows x 16 columns delivered column-major), because both operands of
// dw = G^T D run along the reduction axis m, which is the slow axis of the NHWC tiles.
//   A[c][k = m] = G[m][c]   B[k = m][n] = D[m][n]   dw[c][n] += A_hi B_hi + A_hi B_lo + A_lo B_hi
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 tr_frag(const short* base0, int row_stride_elems) {
    // two transposed reads: rows +0..3 and +4..7 of this lane group's 8-row slab
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base0 + 4 * row_stride_elems));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// the same fragment from two per-lane addresses: rows +0..3 at base0, rows +4..7 at base1 (positions that do not sit a
// fixed pitch apart: consecutive output positions of a linearised grid)
__device__ __forceinline__ bf16x8 tr_frag2(const short* base0, const short* base1) {
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base1));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int RC, int RN, int DD>
__global__ __launch_bounds__(256) void gather_wgrad_bf16_kernel(WgradArgs p) {
    constexpr int CB = 32 * RC;
    constexpr int NB = 32 * RN;
    constexpr int BMC = 128;
    constexpr int GS = CB + 8;   // bf16 elements per LDS row (16 B pad)
    constexpr int DS = NB + 8;
    constexpr int TILE_S = 2 * BMC * (GS + DS);          // shorts: hi+lo planes of both tiles
    constexpr int RED_F = 4 * CB * NB;                   // floats of the cross-wave reduction
    constexpr int SMEM_F = (TILE_S / 2 > RED_F) ? TILE_S / 2 : RED_F;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_F];
    short* Gh = reinterpret_cast<short*>(smem);
    short* Gl = Gh + BMC * GS;
    short* Dh = Gl + BMC * GS;
    short* Dl = Dh + BMC * DS;

    const Geom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 31;
    const int h = lane >> 5;
    // Block id -> (tile, m-split).  All tiles of one m-split read the SAME rows of both operands (every n-block re-reads
    // the gathered rows, every k-block the dense rows, every tap re-gathers them), and blocks b, b + 8 share an XCD's
    // L2 (round-robin placement: a speed assumption only).  When the splits divide evenly over the 8 XCDs a split's
    // tiles are kept on one XCD: the grouped 8192x256x256 launch (1024 workgroups, fabric-bound) fetches 16 MB per
    // group instead of 48 MB.  Uneven splits are left alone (20 splits: 75 vs 50 workgroups per XCD cost 40 %).
    int tile, split;
    if (p.xcd_map) {
        const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
        tile = r % p.ntiles;
        split = xcd + 8 * (r / p.ntiles);
    } else {
        tile = blockIdx.x % p.ntiles;
        split = blockIdx.x / p.ntiles;
    }
    const int nkb = (g.K + CB - 1) / CB;
    const int kkb = tile % nkb;
    const int nb = tile / nkb;
    const int kk0 = kkb * CB;
    const int n0 = nb * NB;
    const int grp = blockIdx.z;
    const float* gin = p.gathered + wg_off(p, grp, 0, p.in_gs);
    const float* din = p.dense + wg_off(p, grp, 1, p.out_gs);
    const bool do_bias = (p.db != nullptr) && (kkb == 0);
    const int tap_u = kk0 / g.C;
    const int c_u = kk0 - tap_u * g.C;
    const int ky_u = tap_u / g.KW;
    const int kx_u = tap_u - ky_u * g.KW;

    const int total_chunks = (g.M + BMC - 1) / BMC;
    const int c_begin = split * p.chunks_per_split;
    int c_end = c_begin + p.chunks_per_split;
    if (c_end > total_chunks) c_end = total_chunks;

    LoaderV4<BMC, CB, DD> lg;
    lg.init(tid);

    constexpr int DSLOTS = NB / 4;
    constexpr int DRPP = 256 / DSLOTS;
    constexpr int DNP = BMC / DRPP;
    const int dslot = tid % DSLOTS;
    const int dr0 = tid / DSLOTS;
    const int dn = n0 + dslot * 4;
    const bool dn_ok = dn < g.N;
    f32x4 dreg[DNP];
    unsigned dmask = 0u;
    auto load_d = [&](int m0) {
        dmask = 0u;
#pragma unroll
        for (int j = 0; j < DNP; ++j) {
            int m = m0 + dr0 + j * DRPP;
            const bool ok = m < g.M && dn_ok;
            const size_t o = ok ? (size_t)m * g.N + dn : 0;
            dmask |= (ok ? 1u : 0u) << j;
            dreg[j] = *reinterpret_cast<const f32x4*>(din + o);
        }
    };
    auto store_d = [&]() {
#pragma unroll
        for (int j = 0; j < DNP; ++j) {
            const bool ok = (dmask >> j) & 1u;
            f32x4 v = dreg[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
            u32x2 h2, l2;
            split4(v, h2, l2);
            *reinterpret_cast<u32x2*>(Dh + (dr0 + j * DRPP) * DS + dslot * 4) = h2;
            *reinterpret_cast<u32x2*>(Dl + (dr0 + j * DRPP) * DS + dslot * 4) = l2;
        }
    };

    f32x16 acc[RC][RN];
    f32x16 accb[RN];
#pragma unroll
    for (int a = 0; a < RC; ++a)
#pragma unroll
        for (int b = 0; b < RN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
#pragma unroll
    for (int b = 0; b < RN; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) accb[b][e] = 0.f;

    // transposed-read lane geometry: 16-lane group gq reads rows +q, columns 16*(gq&1) + 4*pq
    const int gq = lane >> 4;
    const int q = (lane & 15) >> 2;
    const int pq = lane & 3;
    const int tr_row = 8 * (gq >> 1) + q;          // + 16*ks + 32*wave
    const int tr_col = 16 * (gq & 1) + 4 * pq;     // + 32*block
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    if (c_begin < c_end) {
        lg.set_rows(g, c_begin * BMC);
        lg.load_tap(g, gin, ky_u, kx_u, c_u, true);
        load_d(c_begin * BMC);
    }
    for (int ch = c_begin; ch < c_end; ++ch) {
        lg.store_split(g, Gh, Gl, GS);
        store_d();
        __syncthreads();
        if (ch + 1 < c_end) {
            lg.advance_rows(g, (ch + 1) * BMC, p.step_b, p.step_p, p.step_q);
            lg.load_tap(g, gin, ky_u, kx_u, c_u, true);
            load_d((ch + 1) * BMC);
        }
        auto mma = [&](auto with_bias) {
#pragma unroll
            for (int ksx = 0; ksx < 2; ++ksx) {
                const int row = wave * 32 + 16 * ksx + tr_row;
                bf16x8 ah[RC], al[RC], bh[RN], bl[RN];
#pragma unroll
                for (int a = 0; a < RC; ++a) {
                    ah[a] = tr_frag(Gh + row * GS + 32 * a + tr_col, GS);
                    al[a] = tr_frag(Gl + row * GS + 32 * a + tr_col, GS);
                }
#pragma unroll
                for (int b = 0; b < RN; ++b) {
                    bh[b] = tr_frag(Dh + row * DS + 32 * b + tr_col, DS);
                    bl[b] = tr_frag(Dl + row * DS + 32 * b + tr_col, DS);
                }
#pragma unroll
                for (int a = 0; a < RC; ++a)
#pragma unroll
                    for (int b = 0; b < RN; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
                    }
                if constexpr (decltype(with_bias)::value) {
#pragma unroll
                    for (int b = 0; b < RN; ++b) {
                        accb[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bh[b], accb[b], 0, 0, 0);
                        accb[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bl[b], accb[b], 0, 0, 0);
                    }
                }
            }
        };
        if (do_bias) mma(std::true_type{});
        else mma(std::false_type{});
        __syncthreads();
    }

    float* red = smem;
#pragma unroll
    for (int a = 0; a < RC; ++a)
#pragma unroll
        for (int b = 0; b < RN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int cl = a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                int nl = b * 32 + i;
                red[wave * CB * NB + cl * NB + nl] = acc[a][b][e];
            }
    __syncthreads();
    const bool part = p.part_w != nullptr;
    float* dw = wg_dw(p, split) + wg_off(p, grp, 2, p.w_gs);
    for (int e = tid; e < CB * NB; e += 256) {
        int cl = e / NB;
        int nl = e - cl * NB;
        int kk = kk0 + cl;
        int n = n0 + nl;
        if (kk < g.K && n < g.N) {
            float s = red[e] + red[CB * NB + e] + red[2 * CB * NB + e] + red[3 * CB * NB + e];
            int tap = kk / g.C;
            int c = kk - tap * g.C;
            const int wtap = (tap / g.KW) * g.kws + tap % g.KW;   // compact tap -> tap of the weight layout
            wg_put(part, dw + (size_t)wtap * g.wts + (size_t)c * g.wcs + (size_t)n * g.wns, s);
        }
    }
    if (do_bias && !part && h == 0) {
        float* db = p.db + wg_off(p, grp, 3, p.bias_gs);
#pragma unroll
        for (int b = 0; b < RN; ++b) {
            int n = n0 + b * 32 + i;
            if (n < g.N) atomicAdd(db + n, accb[b][0]);
        }
    }
    if (do_bias && part) {       // one writer per slot element: the four waves' row shares meet in LDS (workgroup-uniform branch)
        __syncthreads();
        if (h == 0) {
#pragma unroll
            for (int b = 0; b < RN; ++b) red[wave * NB + b * 32 + i] = accb[b][0];
        }
        __syncthreads();
        float* db = wg_db(p, split) + wg_off(p, grp, 3, p.bias_gs);
        if (tid < NB && n0 + tid < g.N) db[n0 + tid] = (red[tid] + red[NB + tid]) + (red[2 * NB + tid] + red[3 * NB + tid]);
    }
}

// The 64 x 64-tile case again, organised for OCCUPANCY.  Measured on the kernel above (SQ counters, 1024 workgroups):
// 0.9 waves per SIMD, 41 % of the wave cycles issuing VALU (f32 -> hi/lo splits, row bookkeeping), 5 % MFMA - its 74 KB
// of LDS (two 128-row tiles, then a 64 KB cross-wave reduction) leave one workgroup per CU, so load latency, split,
// LDS round trip and MFMAs run one after the other.  Here every wave owns ONE 32 x 32 block of the tile and walks all
// rows of a 64-row chunk: no cross-wave reduction (accumulators 16 registers instead of 64, flushed straight from the
// C layout as two 128-byte segments per atomic instruction), 37 KB of LDS -> four workgroups per CU whose phases overlap.
template <int DD>
__global__ __launch_bounds__(256) void gather_wgrad_bf16_sub_kernel(WgradArgs p) {
    constexpr int CB = 64, NB = 64, BMC = 64;
    constexpr int GS = CB + 8, DS = NB + 8;
    __shared__ __attribute__((aligned(16))) short smem_s[2 * BMC * (GS + DS)];
    short* Gh = smem_s;
    short* Gl = Gh + BMC * GS;
    short* Dh = Gl + BMC * GS;
    short* Dl = Dh + BMC * DS;

    const Geom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wc = wave >> 1, wn = wave & 1;      // this wave's 32 x 32 block of the 64 x 64 tile
    const int i = lane & 31;
    const int h = lane >> 5;
    // all tiles of an m-split read the same rows of both operands: keep them on one XCD when the splits divide evenly over
    // the eight (blocks b and b + 8 share an XCD's L2 under round-robin placement - speed only, see gather_wgrad_bf16_kernel)
    int tile, split;
    if (p.xcd_map) {
        const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
        tile = r % p.ntiles;
        split = xcd + 8 * (r / p.ntiles);
    } else {
        tile = blockIdx.x % p.ntiles;
        split = blockIdx.x / p.ntiles;
    }
    // cpad (32 < C < 64, the 48-channel layers of the VDVAE): a k-block is one tap with its C channels zero-padded to 64 -
    // 56 % of the MFMA work is real, on a pipe 16x faster than the f32 form these shapes ran on before
    const bool cpad = p.cpad != 0;
    const int nkb = cpad ? g.KH * g.KW : (g.K + CB - 1) / CB;
    const int kkb = tile % nkb;
    const int nb = tile / nkb;
    const int kk0 = kkb * CB;
    const int n0 = nb * NB;
    const int grp = blockIdx.z;
    const float* gin = p.gathered + wg_off(p, grp, 0, p.in_gs);
    const float* din = p.dense + wg_off(p, grp, 1, p.out_gs);
    const bool do_bias = (p.db != nullptr) && (kkb == 0) && (wc == 0);
    const int tap_u = cpad ? kkb : kk0 / g.C;
    const int c_u = cpad ? 0 : kk0 - tap_u * g.C;
    const int ky_u = tap_u / g.KW;
    const int kx_u = tap_u - ky_u * g.KW;

    const int total_chunks = (g.M + BMC - 1) / BMC;
    const int c_begin = split * p.chunks_per_split;
    int c_end = c_begin + p.chunks_per_split;
    if (c_end > total_chunks) c_end = total_chunks;

    LoaderV4<BMC, CB, DD> lg;
    lg.init(tid);
    const bool kok_t = !cpad || lg.slot4 < g.C;       // this thread's 4 channels exist
    constexpr int DSLOTS = NB / 4;
    constexpr int DRPP = 256 / DSLOTS;
    constexpr int DNP = BMC / DRPP;
    const int dslot = tid % DSLOTS;
    const int dr0 = tid / DSLOTS;
    const int dn = n0 + dslot * 4;
    const bool dn_ok = dn < g.N;
    f32x4 dreg[DNP];
    unsigned dmask = 0u;
    auto load_d = [&](int m0) {
        dmask = 0u;
#pragma unroll
        for (int j = 0; j < DNP; ++j) {
            int m = m0 + dr0 + j * DRPP;
            const bool ok = m < g.M && dn_ok;
            const size_t o = ok ? (size_t)m * g.N + dn : 0;
            dmask |= (ok ? 1u : 0u) << j;
            dreg[j] = *reinterpret_cast<const f32x4*>(din + o);
        }
    };
    auto store_d = [&]() {
#pragma unroll
        for (int j = 0; j < DNP; ++j) {
            const bool ok = (dmask >> j) & 1u;
            f32x4 v = dreg[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
            u32x2 h2, l2;
            split4(v, h2, l2);
            *reinterpret_cast<u32x2*>(Dh + (dr0 + j * DRPP) * DS + dslot * 4) = h2;
            *reinterpret_cast<u32x2*>(Dl + (dr0 + j * DRPP) * DS + dslot * 4) = l2;
        }
    };

    f32x16 acc, accb;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = accb[e] = 0.f;
    const int gq = lane >> 4;
    const int q = (lane & 15) >> 2;
    const int pq = lane & 3;
    const int tr_row = 8 * (gq >> 1) + q;
    const int tr_col = 16 * (gq & 1) + 4 * pq;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    if (c_begin < c_end) {
        lg.set_rows(g, c_begin * BMC);
        lg.load_tap(g, gin, ky_u, kx_u, c_u, kok_t);
        load_d(c_begin * BMC);
    }
    for (int ch = c_begin; ch < c_end; ++ch) {
        lg.store_split(g, Gh, Gl, GS);
        store_d();
        __syncthreads();
        if (ch + 1 < c_end) {
            lg.advance_rows(g, (ch + 1) * BMC, p.step_b, p.step_p, p.step_q);
            lg.load_tap(g, gin, ky_u, kx_u, c_u, kok_t);
            load_d((ch + 1) * BMC);
        }
#pragma unroll
        for (int ks = 0; ks < BMC / 16; ++ks) {
            const int row = 16 * ks + tr_row;
            const bf16x8 ah = tr_frag(Gh + row * GS + 32 * wc + tr_col, GS);
            const bf16x8 al = tr_frag(Gl + row * GS + 32 * wc + tr_col, GS);
            const bf16x8 bh = tr_frag(Dh + row * DS + 32 * wn + tr_col, DS);
            const bf16x8 bl = tr_frag(Dl + row * DS + 32 * wn + tr_col, DS);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            if (do_bias) {
                accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bh, accb, 0, 0, 0);
                accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bl, accb, 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // flush straight from the C/D layout: col = lane & 31 (n), row = (e & 3) + 8 * (e >> 2) + 4 * h (k index)
    const bool part = p.part_w != nullptr;
    float* dw = wg_dw(p, split) + wg_off(p, grp, 2, p.w_gs);
    const int n = n0 + 32 * wn + i;
    if (n < g.N) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int cl = 32 * wc + (e & 3) + 8 * (e >> 2) + 4 * h;
            const int kk = kk0 + cl;
            if (cpad ? cl >= g.C : kk >= g.K) continue;
            const int tap = cpad ? kkb : kk / g.C;
            const int c = cpad ? cl : kk - tap * g.C;
            const int wtap = (tap / g.KW) * g.kws + tap % g.KW;
            wg_put(part, dw + (size_t)wtap * g.wts + (size_t)c * g.wcs + (size_t)n * g.wns, acc[e]);
        }
        if (do_bias && h == 0) wg_put(part, wg_db(p, split) + wg_off(p, grp, 3, p.bias_gs) + n, accb[0]);
    }
}

// ----------------------- stride-1 weight gradients, patch-staged form (bf16x3) -----------------------
// gather_wgrad_bf16_kernel gives every (tap x 32 channels) block its own workgroups: each of them re-gathers the
// input rows and re-reads the dense operand (measured on the 28x28 5x5 layers: 212 MB of HBM-side traffic per
// launch against 51 MB of operands).  Here a persistent workgroup walks 128-position tiles of the images; per tile
// the input patch (with halo) and the dense tile are loaded ONCE, split to hi / lo bf16 planes in LDS, and every
// tap's operand is a shifted transposed LDS read.  The 4 waves own disjoint tap subsets (accumulators for
// TPW taps x 32 x 32*RN stay in registers across all tiles of the workgroup), so there is no cross-wave reduction;
// one atomic flush per workgroup at the end.  Needs a = d = 1 and C == 32.
template <int RN, int TPW>
__global__ __launch_bounds__(256) void patch_wgrad_bf16_kernel(WgradArgs p, int tw_log2, int ntiles) {
    constexpr int NB = 32 * RN;
    constexpr int DS = NB + 8;                     // bf16 per dense-tile row (16 B pad)
    constexpr int NPL = 10;                        // 16-byte patch loads per thread and tile (<= 2560 / 256)
    constexpr int NDL = 4 * RN;                    // 16-byte dense-tile loads per thread and tile
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const Geom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 31;
    const int h = lane >> 5;
    const int TW = 1 << tw_log2, TH = 128 >> tw_log2;
    const int PH = TH + g.KH - 1, PW = TW + g.KW - 1;
    const int PS = g.C + 8;
    short* Ph = reinterpret_cast<short*>(dsm);
    short* Pl = Ph + PH * PW * PS;
    short* Dh = Pl + PH * PW * PS;
    short* Dl = Dh + 128 * DS;
    const int tiles_x = (g.OW + TW - 1) >> tw_log2;
    const int tiles_y = (g.OH + TH - 1) / TH;
    const int ntaps = g.KH * g.KW;
    const int c4n = g.C >> 2;
    const int ptotal = PH * PW * c4n;

    // this wave's taps: unit u = wave + 4 j  ->  patch offset of the tap
    int toff[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int tap = wave + 4 * j;
        const int tt = tap < ntaps ? tap : 0;
        const int ky = tt / g.KW, kx = tt - ky * g.KW;
        const int pdy = g.cs > 0 ? ky : g.KH - 1 - ky, pdx = g.cs > 0 ? kx : g.KW - 1 - kx;
        toff[j] = (pdy * PW + pdx) * PS;
    }
    // transposed-read lane geometry (see gather_wgrad_bf16_kernel)
    const int gq = lane >> 4;
    const int q = (lane & 15) >> 2;
    const int pq = lane & 3;
    const int tr_row = 8 * (gq >> 1) + q;
    const int tr_col = 16 * (gq & 1) + 4 * pq;

    f32x16 acc[TPW][RN];
    f32x16 accb[RN];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int b = 0; b < RN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][b][e] = 0.f;
#pragma unroll
    for (int b = 0; b < RN; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) accb[b][e] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    const bool do_bias = p.db != nullptr && wave == 0;

    f32x4 pv[NPL], dv[NDL];
    auto issue = [&](int tile) {                   // global loads of one tile into registers
        int t = tile;
        const int txi = t % tiles_x;
        t /= tiles_x;
        const int tyi = t % tiles_y;
        const int b = t / tiles_y;
        const int y0 = tyi * TH, x0 = txi << tw_log2;
        const int sy0 = y0 + g.off + (g.cs < 0 ? -(g.KH - 1) : 0);
        const int sx0 = x0 + g.offx + (g.cs < 0 ? -(g.KW - 1) : 0);
        const float* img = p.gathered + (size_t)b * g.IH * g.IW * g.C;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int e = tid + 256 * j;
            const int ee = e < ptotal ? e : ptotal - 1;
            const int pos = ee / c4n, c4 = ee - pos * c4n;
            const int py = pos / PW, px = pos - py * PW;
            const int gy = sy0 + py, gx = sx0 + px;
            const bool ok = e < ptotal && (unsigned)gy < (unsigned)g.IH && (unsigned)gx < (unsigned)g.IW;
            const size_t so = ok ? ((size_t)gy * g.IW + gx) * g.C + 4 * c4 : 0;
            pv[j] = *reinterpret_cast<const f32x4*>(img + so);
            if (!ok) pv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const float* dimg = p.dense + (size_t)b * g.OH * g.OW * g.N;
#pragma unroll
        for (int j = 0; j < NDL; ++j) {
            const int e = tid + 256 * j;                      // (position, 4-column group) of the 128 x NB tile
            const int pos = e / (NB / 4), n4 = e - pos * (NB / 4);
            const int ty = pos >> tw_log2, tx = pos & (TW - 1);
            const int gy = y0 + ty, gx = x0 + tx;
            const bool ok = gy < g.OH && gx < g.OW && 4 * n4 < g.N;
            const size_t so = ok ? ((size_t)gy * g.OW + gx) * g.N + 4 * n4 : 0;
            dv[j] = *reinterpret_cast<const f32x4*>(dimg + so);
            if (!ok) dv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage = [&]() {                           // registers -> hi / lo bf16 planes in LDS
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int e = tid + 256 * j;
            if (e < ptotal) {
                const int pos = e / c4n, c4 = e - pos * c4n;
                u32x2 h2, l2;
                split4(pv[j], h2, l2);
                *reinterpret_cast<u32x2*>(Ph + pos * PS + 4 * c4) = h2;
                *reinterpret_cast<u32x2*>(Pl + pos * PS + 4 * c4) = l2;
            }
        }
#pragma unroll
        for (int j = 0; j < NDL; ++j) {
            const int e = tid + 256 * j;
            const int pos = e / (NB / 4), n4 = e - pos * (NB / 4);
            u32x2 h2, l2;
            split4(dv[j], h2, l2);
            *reinterpret_cast<u32x2*>(Dh + pos * DS + 4 * n4) = h2;
            *reinterpret_cast<u32x2*>(Dl + pos * DS + 4 * n4) = l2;
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        stage();
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);      // next tile's loads fly during the MFMAs
#pragma unroll 2
        for (int ks = 0; ks < 8; ++ks) {           // 16 consecutive positions of one tile row per k16 step
            const int pos0 = 16 * ks;
            const int ty = pos0 >> tw_log2, tx0 = pos0 & (TW - 1);
            bf16x8 bh[RN], bl[RN];
#pragma unroll
            for (int b = 0; b < RN; ++b) {
                bh[b] = tr_frag(Dh + (pos0 + tr_row) * DS + 32 * b + tr_col, DS);
                bl[b] = tr_frag(Dl + (pos0 + tr_row) * DS + 32 * b + tr_col, DS);
            }
            const int abase = (ty * PW + tx0 + tr_row) * PS + tr_col;
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                const bf16x8 ah = tr_frag(Ph + abase + toff[j], PS);
                const bf16x8 al = tr_frag(Pl + abase + toff[j], PS);
#pragma unroll
                for (int b = 0; b < RN; ++b) {
                    acc[j][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[b], acc[j][b], 0, 0, 0);
                    acc[j][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[b], acc[j][b], 0, 0, 0);
                    acc[j][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[b], acc[j][b], 0, 0, 0);
                }
            }
            if (do_bias) {
#pragma unroll
                for (int b = 0; b < RN; ++b) {
                    accb[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bh[b], accb[b], 0, 0, 0);
                    accb[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bl[b], accb[b], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // flush: C/D layout row (= channel) = (e&3) + 8*(e>>2) + 4*h, column (= n) = lane & 31
    // (partial-sum mode: slot = this workgroup; its four waves own disjoint taps, so every slot element has one writer)
    const bool part = p.part_w != nullptr;
    float* dwp = wg_dw(p, blockIdx.x);
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int tap = wave + 4 * j;
        if (tap >= ntaps) continue;
        const int wtap = (tap / g.KW) * g.kws + tap % g.KW;
#pragma unroll
        for (int b = 0; b < RN; ++b) {
            const int n = 32 * b + i;
            if (n >= g.N) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = (e & 3) + 8 * (e >> 2) + 4 * h;
                wg_put(part, dwp + (size_t)wtap * g.wts + (size_t)c * g.wcs + (size_t)n * g.wns, acc[j][b][e]);
            }
        }
    }
    if (do_bias && h == 0) {
#pragma unroll
        for (int b = 0; b < RN; ++b) {
            const int n = 32 * b + i;
            if (n < g.N) wg_put(part, wg_db(p, blockIdx.x) + n, accb[b][0]);
        }
    }
}

// ----------------------- weight gradients with BOTH images of a sample resident in LDS (bf16x3) -----------------------
// Stride-2 layers had no patch-staged weight gradient (gather_wgrad_bf16_kernel re-gathers its rows and re-reads the dense
// operand once per tap: 25 x through L2, 46 TFLOP/s on the 28x28 -> 14x14 layers), and the patch-staged form walks 14x14
// grids as two 8 x 16 tiles per image (77 % of the positions real).  Here a persistent workgroup takes whole samples: the
// gathered image ([IH][IW][32] f32 -> hi / lo bf16 planes, one extra all-zero pixel for the taps that fall outside) and
// the dense image ([OH*OW][N] -> hi / lo, plus one all-zero row) are staged once, every k16 step is one output row
// (9 <= OW <= 16: 14 of 16 positions real on 14x14), and a tap's A operand is a transposed LDS read at per-lane pixel
// addresses (any stride a).  Taps are owned by the 4 waves as in patch_wgrad_bf16_kernel; accumulators
// stay in registers over all samples of the workgroup; one atomic flush at the end.  Needs d = 1 and C == 32.
template <int RN, int NPI, int NPD, int TPW>      // NPI / NPD: 16-byte pieces per thread of the gathered / dense image
__global__ __launch_bounds__(256) void image_wgrad_bf16_kernel(WgradArgs p, int nsub) {
    constexpr int NB = 32 * RN;
    constexpr int DS = NB + 8;
    constexpr int PS = 40;                         // 32 channels + 16 B pad
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const Geom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31;
    const int h = lane >> 5;
    const int npix = g.IH * g.IW, npos = g.OH * g.OW;
    short* Ph = reinterpret_cast<short*>(dsm);
    short* Pl = Ph + (npix + 1) * PS;
    short* Dh = Pl + (npix + 1) * PS;
    short* Dl = Dh + (npos + 1) * DS;
    const int ntaps = g.KH * g.KW;

    // the taps are dealt to nsub workgroup classes (sub = blockIdx.x % nsub), inside a class to its 4 waves: TPW each
    // (the nsub classes of a sample slot on ONE XCD when the slots divide over the eight: they stage the same samples)
    const int b_step = gridDim.x / nsub;
    int sub, b_first;
    if (b_step % 8 == 0) {
        const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
        sub = r % nsub;
        b_first = xcd + 8 * (r / nsub);
    } else {
        sub = blockIdx.x % nsub;
        b_first = blockIdx.x / nsub;
    }
    int tdy[TPW], tdx[TPW];                        // pixel offsets of this wave's taps
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int tap = sub + nsub * (wave + 4 * j);
        const int tt = tap < ntaps ? tap : 0;
        const int ky = tt / g.KW, kx = tt - ky * g.KW;
        tdy[j] = ky * g.cs;
        tdx[j] = kx * g.cs;
    }
    const int gq = lane >> 4;                      // transposed-read lane geometry (see gather_wgrad_bf16_kernel)
    const int q = (lane & 15) >> 2;
    const int pq = lane & 3;
    const int tr_row = 8 * (gq >> 1) + q;
    const int tr_col = 16 * (gq & 1) + 4 * pq;

    f32x16 acc[TPW][RN];
    f32x16 accb[RN];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int b = 0; b < RN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][b][e] = 0.f;
#pragma unroll
    for (int b = 0; b < RN; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) accb[b][e] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    const bool do_bias = p.db != nullptr && wave == 0 && sub == 0;

    // the zero pixel and the zero dense row are written once: staging never touches them
    if (tid < PS / 2) {
        reinterpret_cast<unsigned*>(Ph + npix * PS)[tid] = 0u;
        reinterpret_cast<unsigned*>(Pl + npix * PS)[tid] = 0u;
    }
    if (tid < DS / 2) {
        reinterpret_cast<unsigned*>(Dh + npos * DS)[tid] = 0u;
        reinterpret_cast<unsigned*>(Dl + npos * DS)[tid] = 0u;
    }

    // A k16 step is ONE output row (9 <= OW <= 16): whether a tap's source row exists is then the same for the whole wave, and
    // whether its source column exists depends on the lane and the tap only - both leave the loop.  Per lane and tap: the column
    // part of the two pixel addresses (positions tr_row and tr_row + 4 of the row) and their validity; per step and tap what is
    // left is one scalar row base, an add and a select per address.
    const int oxa = tr_row, oxb = tr_row + 4;
    int xoa[TPW], xob[TPW];
    bool oka[TPW], okb[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int xa = oxa * g.a + g.offx + tdx[j], xb = oxb * g.a + g.offx + tdx[j];
        oka[j] = oxa < g.OW && (unsigned)xa < (unsigned)g.IW;
        okb[j] = oxb < g.OW && (unsigned)xb < (unsigned)g.IW;
        xoa[j] = xa * PS + tr_col;
        xob[j] = xb * PS + tr_col;
    }
    const int zpix = npix * PS + tr_col;
    const bool okda = oxa < g.OW, okdb = oxb < g.OW;
    const int dxa = oxa * DS + tr_col, dxb = oxb * DS + tr_col, zrow = npos * DS + tr_col;

    const int np4 = npix * 8;                      // 16-byte pieces of the gathered image (C = 32)
    const int nd4 = npos * (NB / 4);               // ... of the dense image (columns >= N are zero)
    f32x4 pv[NPI], dv[NPD];
    auto issue = [&](int b) {                      // global loads of one sample into registers
        const float* img = p.gathered + (size_t)b * npix * g.C;
        const float* dimg = p.dense + (size_t)b * npos * g.N;
#pragma unroll
        for (int j = 0; j < NPI; ++j) {
            const int e = tid + 256 * j;
            pv[j] = *reinterpret_cast<const f32x4*>(img + 4 * (size_t)(e < np4 ? e : 0));
        }
#pragma unroll
        for (int j = 0; j < NPD; ++j) {
            const int e = tid + 256 * j;
            const int ee = e < nd4 ? e : 0;
            const int pos = ee / (NB / 4), n4 = ee - pos * (NB / 4);
            const bool ok = 4 * n4 < g.N;
            dv[j] = *reinterpret_cast<const f32x4*>(dimg + (ok ? (size_t)pos * g.N + 4 * n4 : 0));
            if (!ok) dv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage = [&]() {                           // registers -> hi / lo bf16 planes in LDS
#pragma unroll
        for (int j = 0; j < NPI; ++j) {
            const int e = tid + 256 * j;
            if (e < np4) {
                u32x2 h2, l2;
                split4(pv[j], h2, l2);
                *reinterpret_cast<u32x2*>(Ph + (e >> 3) * PS + 4 * (e & 7)) = h2;
                *reinterpret_cast<u32x2*>(Pl + (e >> 3) * PS + 4 * (e & 7)) = l2;
            }
        }
#pragma unroll
        for (int j = 0; j < NPD; ++j) {
            const int e = tid + 256 * j;
            if (e < nd4) {
                const int pos = e / (NB / 4), n4 = e - pos * (NB / 4);
                u32x2 h2, l2;
                split4(dv[j], h2, l2);
                *reinterpret_cast<u32x2*>(Dh + pos * DS + 4 * n4) = h2;
                *reinterpret_cast<u32x2*>(Dl + pos * DS + 4 * n4) = l2;
            }
        }
    };
    issue(b_first);
    for (int b = b_first; b < g.B; b += b_step) {
        stage();
        __syncthreads();
        // the next sample's loads fly during the MFMAs (the last round re-reads its own sample: no loads under a branch)
        issue(b + b_step < g.B ? b + b_step : b);
        for (int oy = 0; oy < g.OH; ++oy) {
            const int drow = oy * g.OW * DS;
            const int da = okda ? drow + dxa : zrow, db2 = okdb ? drow + dxb : zrow;
            bf16x8 bh[RN], bl[RN];
#pragma unroll
            for (int b2 = 0; b2 < RN; ++b2) {
                bh[b2] = tr_frag2(Dh + da + 32 * b2, Dh + db2 + 32 * b2);
                bl[b2] = tr_frag2(Dl + da + 32 * b2, Dl + db2 + 32 * b2);
            }
            const int ybase = oy * g.a + g.off;
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                const int y = ybase + tdy[j];                       // wave-uniform
                const bool oky = (unsigned)y < (unsigned)g.IH;
                const int rbase = y * g.IW * PS;
                const int o0 = (oky && oka[j]) ? rbase + xoa[j] : zpix;
                const int o1 = (oky && okb[j]) ? rbase + xob[j] : zpix;
                const bf16x8 ah = tr_frag2(Ph + o0, Ph + o1);
                const bf16x8 al = tr_frag2(Pl + o0, Pl + o1);
#pragma unroll
                for (int b2 = 0; b2 < RN; ++b2) {
                    acc[j][b2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[b2], acc[j][b2], 0, 0, 0);
                    acc[j][b2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[b2], acc[j][b2], 0, 0, 0);
                    acc[j][b2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[b2], acc[j][b2], 0, 0, 0);
                }
            }
            if (do_bias) {
#pragma unroll
                for (int b2 = 0; b2 < RN; ++b2) {
                    accb[b2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bh[b2], accb[b2], 0, 0, 0);
                    accb[b2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bl[b2], accb[b2], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // flush: C/D layout row (= channel) = (e&3) + 8*(e>>2) + 4*h, column (= n) = lane & 31
    // (partial-sum mode: slot = b_first, shared by the nsub workgroups whose tap classes tile the kernel)
    const bool part = p.part_w != nullptr;
    float* dwp = wg_dw(p, b_first);
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int tap = sub + nsub * (wave + 4 * j);
        if (tap >= ntaps) continue;
        const int wtap = (tap / g.KW) * g.kws + tap % g.KW;
#pragma unroll
        for (int b2 = 0; b2 < RN; ++b2) {
            const int n = 32 * b2 + i;
            if (n >= g.N) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = (e & 3) + 8 * (e >> 2) + 4 * h;
                wg_put(part, dwp + (size_t)wtap * g.wts + (size_t)c * g.wcs + (size_t)n * g.wns, acc[j][b2][e]);
            }
        }
    }
    if (do_bias && h == 0) {
#pragma unroll
        for (int b2 = 0; b2 < RN; ++b2) {
            const int n = 32 * b2 + i;
            if (n < g.N) wg_put(part, wg_db(p, b_first) + n, accb[b2][0]);
        }
    }
}

// ----------------------- weight gradients of 64-channel layers on SMALL grids, one TAP ROW per workgroup class (bf16x3) ------
// The 14x14x64 <-> 7x7x64 stride-2 5x5 layers of the PM-VAE (three launches per step) ran on gather_wgrad_bf16_sub_kernel: every
// (tap, 64 channels) k-block is a workgroup column of its own that re-gathers its input rows and re-reads the dense rows - 25 taps
// x 24 m-splits = 600 workgroups, 160 MB through L2 for 16 MB of operands, 35 us (58 algorithmic TFLOP/s).  image_wgrad_bf16 cannot
// take them: the weights (25 x 64 x 64 floats = 410 KB) are 6 x the operands of a sample, so accumulating all taps per workgroup
// makes the flush, not the MFMAs, the kernel.  Here a workgroup class owns ONE ROW of taps (ky: KW taps x 64 x 64 = 80 KB of
// accumulators, 16 x KW registers per lane: wave w owns the 32 x 32 block (w >> 1, w & 1) of every tap) and walks whole samples:
// the OH input rows that tap row meets (y = oy * a + off + ky * cs) and the dense image are staged once per sample as hi / lo bf16
// planes (47 KB of LDS: two workgroups per CU), a k16 step is TWO output rows of up to 8 positions (dense pads are zero in LDS, a
// gathered position outside the image reads the zero pixel), every operand a transposed LDS read.  Slots of the partial-sum arena =
// sample groups.  Needs d = 1, C = N = 64, OW <= 8, OH <= 8, IW <= 16.
template <int KW>
__global__ __launch_bounds__(256, 2) void rowtap_wgrad_bf16_kernel(WgradArgs p, int ngroups) {
    constexpr int PS = 72, DS = 72;                // bf16 per pixel / dense position: 64 + 16 B pad
    constexpr int NPI = 8, NPD = 4;                // 16-byte pieces per thread: gathered rows (<= 8 x 16 x 16), dense image (<= 64 x 16)
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const Geom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave >> 1, nb = wave & 1;       // this wave's 32 x 32 block of every tap's 64 x 64 tile
    const int i = lane & 31, h = lane >> 5;
    // the KH classes of a sample group stage the same samples: keep them on one XCD (blocks b, b + 8, .. share an L2 under
    // round-robin placement - speed only) when the groups divide over the eight XCDs; measured without: every sample crosses
    // the fabric once per class (profiles/r04_final_pmc_traffic.txt: 62 MB for 16 MB of operands)
    int ky, grp;
    if (ngroups % 8 == 0) {
        const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
        ky = r % g.KH;
        grp = xcd + 8 * (r / g.KH);
    } else {
        ky = blockIdx.x % g.KH;
        grp = blockIdx.x / g.KH;
    }
    const int npix = g.OH * g.IW;                  // staged gathered pixels: OH rows of IW
    short* Ph = reinterpret_cast<short*>(dsm);
    short* Pl = Ph + (npix + 1) * PS;
    short* Dh = Pl + (npix + 1) * PS;
    short* Dl = Dh + 64 * DS;

    // zero pixel + dense pads: written once (the dense buffer is [8][8] positions; staging only touches real ones)
    for (int e = tid; e < 64 * DS / 2; e += 256) {
        reinterpret_cast<unsigned*>(Dh)[e] = 0u;
        reinterpret_cast<unsigned*>(Dl)[e] = 0u;
    }
    if (tid < PS / 2) {
        reinterpret_cast<unsigned*>(Ph + npix * PS)[tid] = 0u;
        reinterpret_cast<unsigned*>(Pl + npix * PS)[tid] = 0u;
    }

    const int gq = lane >> 4;                      // transposed-read lane geometry (see gather_wgrad_bf16_kernel)
    const int q = (lane & 15) >> 2;
    const int pq = lane & 3;
    const int half = gq >> 1;                      // which of the step's two output rows this lane's positions sit in
    const int tr_col = 16 * (gq & 1) + 4 * pq;
    const int oxa = q, oxb = q + 4;                // this lane's two positions of the row: columns q and q + 4
    int xoa[KW], xob[KW];
    bool oka[KW], okb[KW];
#pragma unroll
    for (int kx = 0; kx < KW; ++kx) {
        const int xa = oxa * g.a + g.offx + kx * g.cs, xb = oxb * g.a + g.offx + kx * g.cs;
        oka[kx] = kx < g.KW && oxa < g.OW && (unsigned)xa < (unsigned)g.IW;
        okb[kx] = kx < g.KW && oxb < g.OW && (unsigned)xb < (unsigned)g.IW;
        xoa[kx] = xa * PS + tr_col + 32 * cb;
        xob[kx] = xb * PS + tr_col + 32 * cb;
    }
    const int zpix = npix * PS + tr_col + 32 * cb;
    // which staged rows exist in the image: row r holds input row r * a + off + ky * cs
    unsigned rowmask = 0u;
    for (int r = 0; r < g.OH; ++r) rowmask |= ((unsigned)(r * g.a + g.off + ky * g.cs) < (unsigned)g.IH ? 1u : 0u) << r;

    f32x16 acc[KW], accb;
#pragma unroll
    for (int kx = 0; kx < KW; ++kx)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[kx][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) accb[e] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    const bool do_bias = p.db != nullptr && ky == 0 && cb == 0;       // wave-uniform

    // staging addresses do not depend on the sample: per piece the global float offset inside the sample's image (low 16 bits,
    // 0xffff: not loaded - the piece is past the end or its input row is outside the image) and the LDS offset in shorts (high
    // 16 bits) are worked out ONCE (the divisions by IW / OW cost more than the MFMAs of a sample when left in the loop)
    const int np4 = npix * 16, nd4 = g.OH * g.OW * 16;
    unsigned paddr[NPI], daddr[NPD];
#pragma unroll
    for (int j = 0; j < NPI; ++j) {
        const int e = tid + 256 * j;
        const int ee = e < np4 ? e : 0;
        const int pos = ee >> 4, c4 = ee & 15;
        const int r = pos / g.IW, col = pos - r * g.IW;
        const int y = r * g.a + g.off + ky * g.cs;
        const bool ok = e < np4 && (unsigned)y < (unsigned)g.IH;
        const unsigned go = ok ? (unsigned)((y * g.IW + col) * 16 + c4) : 0xffffu;        // in 16-byte units (< 16 * 16 * 16)
        const unsigned lo = e < np4 ? (unsigned)(pos * PS + 4 * c4) : 0xffffu;            // < 129 * 72
        paddr[j] = go | (lo << 16);
    }
#pragma unroll
    for (int j = 0; j < NPD; ++j) {
        const int e = tid + 256 * j;
        const int ee = e < nd4 ? e : 0;
        const int pos = ee >> 4, c4 = ee & 15;
        const int oy = pos / g.OW, ox = pos - oy * g.OW;
        daddr[j] = (e < nd4 ? (unsigned)ee : 0u) | ((e < nd4 ? (unsigned)((oy * 8 + ox) * DS + 4 * c4) : 0xffffu) << 16);
    }
    f32x4 pv[NPI], dv[NPD];
    auto issue = [&](int b) {
        const f32x4* img = reinterpret_cast<const f32x4*>(p.gathered + (size_t)b * g.IH * g.IW * 64);
        const f32x4* dimg = reinterpret_cast<const f32x4*>(p.dense + (size_t)b * g.OH * g.OW * 64);
#pragma unroll
        for (int j = 0; j < NPI; ++j) {
            const unsigned go = paddr[j] & 0xffffu;
            pv[j] = img[go == 0xffffu ? 0u : go];
            if (go == 0xffffu) pv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < NPD; ++j) dv[j] = dimg[daddr[j] & 0xffffu];
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < NPI; ++j) {
            const unsigned lo = paddr[j] >> 16;
            if (lo != 0xffffu) {
                u32x2 h2, l2;
                split4(pv[j], h2, l2);
                *reinterpret_cast<u32x2*>(Ph + lo) = h2;
                *reinterpret_cast<u32x2*>(Pl + lo) = l2;
            }
        }
#pragma unroll
        for (int j = 0; j < NPD; ++j) {
            const unsigned lo = daddr[j] >> 16;
            if (lo != 0xffffu) {
                u32x2 h2, l2;
                split4(dv[j], h2, l2);
                *reinterpret_cast<u32x2*>(Dh + lo) = h2;
                *reinterpret_cast<u32x2*>(Dl + lo) = l2;
            }
        }
    };
    const int nsteps = (g.OH + 1) >> 1;
    const int dcol = tr_col + 32 * nb;
    if (grp < g.B) issue(grp);
    __syncthreads();                                                   // the zero fills above
    for (int b = grp; b < g.B; b += ngroups) {
        stage();
        __syncthreads();
        issue(b + ngroups < g.B ? b + ngroups : b);                    // next sample's loads fly during the MFMAs
        for (int st = 0; st < nsteps; ++st) {
            const int oy = 2 * st + half;                              // <= 7: inside the [8][8] dense buffer
            const bool rv = (rowmask >> oy) & 1u;
            const int da = (oy * 8 + oxa) * DS + dcol, db2 = (oy * 8 + oxb) * DS + dcol;
            const bf16x8 bh = tr_frag2(Dh + da, Dh + db2);
            const bf16x8 bl = tr_frag2(Dl + da, Dl + db2);
            const int rbase = oy * g.IW * PS;
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
                const int o0 = (rv && oka[kx]) ? rbase + xoa[kx] : zpix;
                const int o1 = (rv && okb[kx]) ? rbase + xob[kx] : zpix;
                const bf16x8 ah = tr_frag2(Ph + o0, Ph + o1);
                const bf16x8 al = tr_frag2(Pl + o0, Pl + o1);
                acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[kx], 0, 0, 0);
                acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[kx], 0, 0, 0);
                acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[kx], 0, 0, 0);
            }
            if (do_bias) {
                accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bh, accb, 0, 0, 0);
                accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bl, accb, 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // flush: C/D layout row (= channel) = (e&3) + 8*(e>>2) + 4*h, column (= n) = lane & 31; slot = sample group
    const bool part = p.part_w != nullptr;
    float* dwp = wg_dw(p, grp);
    const int n = 32 * nb + i;
#pragma unroll
    for (int kx = 0; kx < KW; ++kx) {
        if (kx >= g.KW) continue;
        const int wtap = ky * g.kws + kx;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = 32 * cb + (e & 3) + 8 * (e >> 2) + 4 * h;
            wg_put(part, dwp + (size_t)wtap * g.wts + (size_t)c * g.wcs + (size_t)n * g.wns, acc[kx][e]);
        }
    }
    if (do_bias && h == 0) wg_put(part, wg_db(p, grp) + n, accb[0]);
}

bool fill_geom(const pm_gather_desc* d, Geom& g, bool class_major) {
    if (!d || d->B <= 0 || d->C <= 0 || d->N <= 0 || d->KH <= 0 || d->KW <= 0 || d->d <= 0 || d->groups <= 0)
        return false;
    if (d->IH <= 0 || d->IW <= 0 || d->OH <= 0 || d->OW <= 0 || d->a <= 0) return false;
    if (d->cs != 1 && d->cs != -1) return false;
    if (d->KH > MAXTAP || d->KW > MAXTAP) return false;
    long long M = (long long)d->B * d->OH * d->OW;
    long long K = (long long)d->KH * d->KW * d->C;
    if (M > 0x7fffffffLL / 4 || K > 0x7fffffffLL / 4) return false;
    // 32-bit element offsets inside one group's tensors and weights
    if (M * d->N >= 0x7fffffffLL || (long long)d->B * d->IH * d->IW * d->C >= 0x7fffffffLL) return false;
    if (d->wts < 0 || d->wcs < 0 || d->wns < 0) return false;
    if (d->kws < d->KW) return false;
    long long wmax = (long long)((d->KH - 1) * d->kws + d->KW - 1) * d->wts + (long long)(d->C - 1) * d->wcs +
                     (long long)(d->N - 1) * d->wns;
    if (wmax >= 0x7fffffffLL) return false;
    if ((long long)(d->OH + d->OW + 2 * MAXTAP) * d->a + (d->off < 0 ? -d->off : d->off) +
            (d->off_x < 0 ? -d->off_x : d->off_x) >= (1 << 27))
        return false;
    g.B = d->B; g.IH = d->IH; g.IW = d->IW; g.C = d->C; g.OH = d->OH; g.OW = d->OW; g.N = d->N;
    g.KH = d->KH; g.KW = d->KW; g.a = d->a; g.cs = d->cs; g.off = d->off; g.offx = d->off_x; g.d = d->d;
    g.wts = d->wts; g.wcs = d->wcs; g.wns = d->wns; g.kws = d->kws; g.M = (int)M; g.K = (int)K;
    g.in_act = d->in_act; g.out_act = d->out_act; g.aux_act = d->aux_act; g.slope = d->slope;
    g.PY = g.PX = 1;
    if (d->kws != d->KW) PM_KVAR("masked");
    if (class_major) {
        if (d->d > 1 && d->OH % d->d == 0 && d->OW % d->d == 0) {
            g.PY = g.PX = d->d;                      // zero-dilated forms: one class per residue
        } else if (d->IH == 1 && d->IW == 1 && d->OH * d->OW > 1) {
            g.PY = d->OH; g.PX = d->OW;              // 1x1 gathered grid: one class per output position
        }
    }
    g.OHc = g.OH / g.PY; g.OWc = g.OW / g.PX;
    g.Mc = g.B * g.OHc * g.OWc;
    return true;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

int pick_mode(const Geom& g, bool vec4, int kblock) {
    if (!vec4) return MODE_V1;
    if (g.C % kblock == 0 && g.d == 1) return MODE_TU1;
    if (g.C % kblock == 0 && g.d == 2) return MODE_TU2;
    return MODE_V4;
}

template <int BM, int BN>
void launch_gemm(hipStream_t s, const GemmArgs& a, int groups, int mode) {
    dim3 grid((a.g.M + BM - 1) / BM, (a.g.N + BN - 1) / BN, groups * a.ksplit);
    PM_KTAG("gather_gemm_kernel<%d, %d, %d>", BM, BN, mode);
    switch (mode) {
        case MODE_TU1: hipLaunchKernelGGL((gather_gemm_kernel<BM, BN, MODE_TU1>), grid, dim3(256), 0, s, a); break;
        case MODE_TU2: hipLaunchKernelGGL((gather_gemm_kernel<BM, BN, MODE_TU2>), grid, dim3(256), 0, s, a); break;
        case MODE_V4: hipLaunchKernelGGL((gather_gemm_kernel<BM, BN, MODE_V4>), grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((gather_gemm_kernel<BM, BN, MODE_V1>), grid, dim3(256), 0, s, a); break;
    }
}

template <int RC, int RN, int DVEC>
void launch_wgrad_mode(hipStream_t s, const WgradArgs& a, dim3 grid, int mode) {
    PM_KTAG("gather_wgrad_kernel<%d, %d, %d, %d>", RC, RN, mode, DVEC);
    switch (mode) {
        case MODE_TU1: hipLaunchKernelGGL((gather_wgrad_kernel<RC, RN, MODE_TU1, DVEC>), grid, dim3(256), 0, s, a); break;
        case MODE_TU2: hipLaunchKernelGGL((gather_wgrad_kernel<RC, RN, MODE_TU2, DVEC>), grid, dim3(256), 0, s, a); break;
        case MODE_V4: hipLaunchKernelGGL((gather_wgrad_kernel<RC, RN, MODE_V4, DVEC>), grid, dim3(256), 0, s, a); break;
        default:
            if constexpr (RC == 1)  // the scalar gather loader only exists for 32-wide k blocks
                hipLaunchKernelGGL((gather_wgrad_kernel<RC, RN, MODE_V1, DVEC>), grid, dim3(256), 0, s, a);
            break;
    }
}

template <int RC, int RN>
void launch_wgrad(hipStream_t s, const WgradArgs& a, dim3 grid, int mode, bool dvec4) {
    if (dvec4) launch_wgrad_mode<RC, RN, 4>(s, a, grid, mode);
    else launch_wgrad_mode<RC, RN, 1>(s, a, grid, mode);
}

struct GemmPlan { int bm, bn, mode, ksplit; bool direct; };
struct WgradPlan { int rc, rn, mode, dvec, nkb, nnb, splits, chunks_per_split; };

GemmPlan plan_gemm(const Geom& g, int groups, bool vec4) {
    const int M = g.M, N = g.N;
    auto nwg = [&](int bm, int bn) { return (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * groups; };
    GemmPlan p{128, 32, pick_mode(g, vec4, BK), 1, false};
    if (N <= 32) {
        p.bm = 128; p.bn = 32;
    } else if (N <= 64) {
        p.bn = 64;
        p.bm = nwg(128, 64) >= 256 ? 128 : 64;
    } else {
        if (nwg(128, 128) >= 256) { p.bm = 128; p.bn = 128; }
        else if (nwg(64, 128) >= 256) { p.bm = 64; p.bn = 128; }
        else if (nwg(64, 64) >= 2 * nwg(32, 128)) { p.bm = 64; p.bn = 64; }
        else { p.bm = 32; p.bn = 128; }
    }
    // split K over workgroups when the tile grid cannot fill the chip and K is long
    int taps = g.KH * g.KW;
    if (g.PY > 1 || g.PX > 1)
        taps = (g.IH == 1 && g.IW == 1) ? 1 : ((g.KH + g.PY - 1) / g.PY) * ((g.KW + g.PX - 1) / g.PX);
    const int chunks = (taps * g.C + BK - 1) / BK;
    p.direct = (p.mode == MODE_TU1 || p.mode == MODE_TU2) && g.KH * g.KW * (g.C / BK) <= DMAXSTEPS && g.KH <= 15 &&
               g.KW <= 15 && (long long)g.B * g.IH * g.IW * g.C * 4 < 0x7ffffff0LL;
    if (p.direct) {
        p.bm = 128;
        p.bn = N > 32 ? 64 : 32;
    }
    const long long tiles = nwg(p.bm, p.bn);
    // split K over workgroups only when the tile grid is far from filling the chip AND K is long: a split costs a
    // zero-fill and an epilogue launch plus f32 atomics (measured on the PM-VAE step: splitting the 14x14 / 7x7
    // layers 3-8 ways to reach 1024+ workgroups LOSES 8-11 % end to end)
    // (tiles < 64: the CelebA PixelCNN's 64-tile 256 -> 128 layers at per-GPU 16 run 2 % faster unsplit; 128 measured neutral elsewhere)
    static const int ksplit_tiles = getenv("PM_KSPLIT_TILES") ? atoi(getenv("PM_KSPLIT_TILES")) : 64;      // A/B knob
    if (groups == 1 && tiles < ksplit_tiles && chunks >= 32) {
        int ks = (int)((256 + tiles - 1) / tiles);
        if (ks > chunks / 4) ks = chunks / 4;
        if (ks > 32) ks = 32;
        if (ks > 1) p.ksplit = ks;
    }
    return p;
}

WgradPlan plan_wgrad(const Geom& g, int groups, bool vec4, bool dvec4) {
    WgradPlan p;
    p.dvec = dvec4 ? 4 : 1;
    p.rc = (vec4 && g.K >= 64 && g.C % 64 == 0) ? 2 : 1;
    p.mode = pick_mode(g, vec4, 32 * p.rc);
    p.rn = g.N > 32 ? 2 : 1;
    p.nkb = (g.K + 32 * p.rc - 1) / (32 * p.rc);
    p.nnb = (g.N + 32 * p.rn - 1) / (32 * p.rn);
    const int total_chunks = (g.M + 127) / 128;
    // m-splits: enough workgroups to fill the chip (~4 per CU), but every workgroup should walk
    // several 128-row chunks - its LDS reduction and atomics are a fixed cost per workgroup
    long long blocks = (long long)p.nkb * p.nnb * groups;
    static const int wg_target = getenv("PM_WG_TARGET") ? atoi(getenv("PM_WG_TARGET")) : 1024;     // A/B knob
    int splits = (int)((wg_target + blocks - 1) / blocks);
    int max_splits = total_chunks / 4;      // measured flat optimum: 2..8 chunks per workgroup within 1 % (PM-VAE, PM-VQVAE)
    if (max_splits < 1) max_splits = 1;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    p.chunks_per_split = (total_chunks + splits - 1) / splits;
    p.splits = (total_chunks + p.chunks_per_split - 1) / p.chunks_per_split;
    return p;
}

const char* mode_name(int mode) {
    switch (mode) {
        case MODE_TU1: return "TU1";
        case MODE_TU2: return "TU2";
        case MODE_V4: return "V4";
        default: return "V1";
    }
}

// (placed behind every other kernel of this file: inserting it in the middle moved the hot PM-VAE kernels in the code object and
// cost that step 2 % - the step is sensitive to where its concurrently running kernels sit in the instruction cache)
// The 128 x 128-tile case for LARGE weight gradients (the PixelCNN's 256- / 512-channel masked convolutions over 12 544 rows,
// grouped eight to sixteen layers to a launch: 2.9 ms of chip-filling launches at the end of a pm_vqvae_mnist step).  With
// 64 x 64 tiles a workgroup moves 32 KB of operands from L2 per 64-row chunk for 64 x 64 x 64 MACs - every element of the
// gathered operand is fetched N / 64 times and every element of the dense one K / 64 times (conv2 of a vertical block:
// 8 and 48 times, 19 GB per launch through L2); a 128 x 128 tile halves both.  Four waves, each a 64 x 64 block (four 32 x 32
// accumulators: one A / B fragment pair feeds 12 MFMAs instead of 3), 64-row chunks, 70 KB of LDS (two workgroups per CU).
// Needs C % 128 == 0 (a 128-wide k-block inside one tap); flush with atomics like the 64 x 64 form.
template <int DD, int BMC>
__global__ __launch_bounds__(256, BMC == 32 ? 3 : 2) void gather_wgrad_bf16_big_kernel(WgradArgs p) {
    constexpr int CB = 128, NB = 128;
    constexpr int GS = CB + 8, DS = NB + 8;
    extern __shared__ __attribute__((aligned(16))) short smem_big[];
    short* Gh = smem_big;
    short* Gl = Gh + BMC * GS;
    short* Dh = Gl + BMC * GS;
    short* Dl = Dh + BMC * DS;

    const Geom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wc = wave >> 1, wn = wave & 1;      // this wave's 64 x 64 block of the 128 x 128 tile
    const int i = lane & 31;
    const int h = lane >> 5;
    const int tile = blockIdx.x % p.ntiles;
    const int split = blockIdx.x / p.ntiles;
    const int nkb = g.K / CB;
    const int kkb = tile % nkb;
    const int nb = tile / nkb;
    const int kk0 = kkb * CB;
    const int n0 = nb * NB;
    const int grp = blockIdx.z;
    const float* gin = p.gathered + wg_off(p, grp, 0, p.in_gs);
    const float* din = p.dense + wg_off(p, grp, 1, p.out_gs);
    const bool do_bias = (p.db != nullptr) && (kkb == 0);          // workgroup-uniform: the first k-block of every column block
    const int tap_u = kk0 / g.C;
    const int c_u = kk0 - tap_u * g.C;
    const int ky_u = tap_u / g.KW;
    const int kx_u = tap_u - ky_u * g.KW;

    const int total_chunks = (g.M + BMC - 1) / BMC;
    const int c_begin = split * p.chunks_per_split;
    int c_end = c_begin + p.chunks_per_split;
    if (c_end > total_chunks) c_end = total_chunks;

    // one chunk of operands in flight per thread (two - a second loader and register set - was measured: 350 registers, one wave
    // per SIMD, 22.5 k img/s against 25.8 k)
    LoaderV4<BMC, CB, DD> lg0;
    lg0.init(tid);
    constexpr int DSLOTS = NB / 4;
    constexpr int DRPP = 256 / DSLOTS;
    constexpr int DNP = BMC / DRPP;
    const int dslot = tid % DSLOTS;
    const int dr0 = tid / DSLOTS;
    const int dn = n0 + dslot * 4;
    const bool dn_ok = dn < g.N;
    f32x4 dreg0[DNP];
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};              // bias gradient: this thread's 4 columns over the rows it stages
    auto load_d = [&](int m0, f32x4 (&dreg)[DNP]) {
#pragma unroll
        for (int j = 0; j < DNP; ++j) {
            const int m = m0 + dr0 + j * DRPP;
            const bool ok = m < g.M && dn_ok;
            dreg[j] = *reinterpret_cast<const f32x4*>(din + (ok ? (size_t)m * g.N + dn : 0));
        }
    };
    auto store_d = [&](int m0, f32x4 (&dreg)[DNP]) {
#pragma unroll
        for (int j = 0; j < DNP; ++j) {
            const bool ok = (m0 + dr0 + j * DRPP) < g.M && dn_ok;
            f32x4 v = dreg[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
            bsum += v;
            u32x2 h2, l2;
            split4(v, h2, l2);
            *reinterpret_cast<u32x2*>(Dh + (dr0 + j * DRPP) * DS + dslot * 4) = h2;
            *reinterpret_cast<u32x2*>(Dl + (dr0 + j * DRPP) * DS + dslot * 4) = l2;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    const int gq = lane >> 4;
    const int q = (lane & 15) >> 2;
    const int pq = lane & 3;
    const int tr_row = 8 * (gq >> 1) + q;
    const int tr_col = 16 * (gq & 1) + 4 * pq;

    if (c_begin < c_end) {
        lg0.set_rows(g, c_begin * BMC);
        lg0.load_tap(g, gin, ky_u, kx_u, c_u, true);
        load_d(c_begin * BMC, dreg0);
    }
    auto mma_chunk = [&]() {
#pragma unroll
        for (int ks = 0; ks < BMC / 16; ++ks) {
            const int row = 16 * ks + tr_row;
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                ah[a] = tr_frag(Gh + row * GS + 64 * wc + 32 * a + tr_col, GS);
                al[a] = tr_frag(Gl + row * GS + 64 * wc + 32 * a + tr_col, GS);
                bh[a] = tr_frag(Dh + row * DS + 64 * wn + 32 * a + tr_col, DS);
                bl[a] = tr_frag(Dl + row * DS + 64 * wn + 32 * a + tr_col, DS);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
                }
        }
    };
    for (int ch = c_begin; ch < c_end; ++ch) {
        lg0.store_split(g, Gh, Gl, GS);
        store_d(ch * BMC, dreg0);
        __syncthreads();
        if (ch + 1 < c_end) {
            lg0.advance_rows(g, (ch + 1) * BMC, p.step_b, p.step_p, p.step_q);
            lg0.load_tap(g, gin, ky_u, kx_u, c_u, true);
            load_d((ch + 1) * BMC, dreg0);
        }
        mma_chunk();
        __syncthreads();
    }

    // flush straight from the C/D layout: col = lane & 31 (n), row = (e & 3) + 8 * (e >> 2) + 4 * h (k index)
    const bool part = p.part_w != nullptr;
    float* dw = wg_dw(p, split) + wg_off(p, grp, 2, p.w_gs);
    const int wtap = (tap_u / g.KW) * g.kws + tap_u % g.KW;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = n0 + 64 * wn + 32 * b + i;
        if (n >= g.N) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = c_u + 64 * wc + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * h;
                wg_put(part, dw + (size_t)wtap * g.wts + (size_t)c * g.wcs + (size_t)n * g.wns, acc[a][b][e]);
            }
    }
    if (do_bias) {               // the 8 threads of a column quad (dr0 = 0 .. 7) meet in LDS, one atomic per column
        float* red = reinterpret_cast<float*>(smem_big);                  // the tiles are dead (barrier at the end of the loop)
        *reinterpret_cast<f32x4*>(red + (dr0 * DSLOTS + dslot) * 4) = bsum;
        __syncthreads();
        if (tid < NB && n0 + tid < g.N) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < DRPP; ++r) t += red[(r * DSLOTS + tid / 4) * 4 + (tid & 3)];
            wg_put(part, wg_db(p, split) + wg_off(p, grp, 3, p.bias_gs) + n0 + tid, t);
        }
    }
}

}  // namespace

// sk / sk_floats: caller's slab scratch for the split-K form (NULL: f32 atomics into the zero-filled output, as before round 4);
// query != NULL: no launch, *query = floats of scratch this problem's split-K form wants (0: it does not split K)
static int gather_gemm_impl(pm_stream_t stream, const pm_gather_desc* d, const float* in, const float* w, const float* bias,
                            const float* aux, const float* res, float* out, float* sk, long long sk_floats,
                            long long* query) {
    GemmArgs a;
    if (!fill_geom(d, a.g, true) || (!query && (!in || !w || !out))) return PM_EINVAL;
    a.in = in; a.w = w; a.bias = bias; a.aux = aux; a.res = res; a.out = out;
    a.out2 = nullptr; a.act2 = PM_ACT_NONE; a.in_colsum = nullptr;
    a.in_gs = d->in_gs; a.w_gs = d->w_gs; a.out_gs = d->out_gs; a.bias_gs = d->bias_gs;
    const bool vec4 = (d->C % 4 == 0) && aligned16(in) && (d->in_gs % 4 == 0);
    hipStream_t s = (hipStream_t)stream;
    const int G = d->groups;
    const GemmPlan p = plan_gemm(a.g, G, vec4);
    a.ksplit = p.ksplit;
    a.sk = nullptr; a.sk_stride = (long long)a.g.M * a.g.N;
    if (query) {
        *query = p.ksplit > 1 ? a.sk_stride * p.ksplit : 0;
        return PM_OK;
    }
    if (p.ksplit > 1) {
        if (sk) {
            if (sk_floats < a.sk_stride * p.ksplit || !aligned16(sk)) return PM_EINVAL;
            a.sk = sk;
        } else if (pm_zero_async(s, out, (size_t)a.g.M * a.g.N * sizeof(float))) return PM_ELAUNCH;
    }
    if (p.direct) {
        const int rn = a.g.N > 32 ? 2 : 1;
        dim3 grid((a.g.M + 127) / 128, (a.g.N + 32 * rn - 1) / (32 * rn), G * a.ksplit);
        if (rn == 1 && p.mode == MODE_TU1) launch_direct<1, 1>(s, a, grid);
        else if (rn == 1) launch_direct<1, 2>(s, a, grid);
        else if (p.mode == MODE_TU1) launch_direct<2, 1>(s, a, grid);
        else launch_direct<2, 2>(s, a, grid);
    } else if (p.bm == 128 && p.bn == 32) launch_gemm<128, 32>(s, a, G, p.mode);
    else if (p.bm == 128 && p.bn == 64) launch_gemm<128, 64>(s, a, G, p.mode);
    else if (p.bm == 64 && p.bn == 64) launch_gemm<64, 64>(s, a, G, p.mode);
    else if (p.bm == 128 && p.bn == 128) launch_gemm<128, 128>(s, a, G, p.mode);
    else if (p.bm == 64 && p.bn == 128) launch_gemm<64, 128>(s, a, G, p.mode);
    else launch_gemm<32, 128>(s, a, G, p.mode);
    if (p.ksplit > 1) {
        long long total = (long long)a.g.M * a.g.N;
        long long blocks = (total + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)blocks, G), dim3(256), 0, s, a, total);
    }
    return pm_check_launch("pm_gather_gemm");
}

extern "C" int pm_gather_gemm(pm_stream_t stream, const pm_gather_desc* d, const float* in, const float* w,
                              const float* bias, const float* aux, const float* res, float* out) {
    return gather_gemm_impl(stream, d, in, w, bias, aux, res, out, nullptr, 0, nullptr);
}

extern "C" int pm_gather_gemm_sk(pm_stream_t stream, const pm_gather_desc* d, const float* in, const float* w,
                                 const float* bias, const float* aux, const float* res, float* out, float* scratch,
                                 long long scratch_floats) {
    if (!scratch) return PM_EINVAL;
    return gather_gemm_impl(stream, d, in, w, bias, aux, res, out, scratch, scratch_floats, nullptr);
}

// Partial-sum mode of the weight-gradient entry points.  `part` != NULL: dw / db are ignored, the launch STORES into the arenas
// of `part` (which must hold exactly the number of slots the launch plan writes: pm_wgrad_part_slots).  `slots_out` != NULL: nothing
// is launched, the slot count of the plan is returned (one planning code path for the query and the launch).
static bool set_part(WgradArgs& a, const pm_wgrad_part* part, bool want_bias) {
    a.part_w = a.part_b = nullptr;
    a.part_ws = a.part_bs = 0;
    if (!part) return true;
    if (!part->w || part->nslots < 1 || part->w_stride < 0 || (want_bias && (!part->b || part->b_stride < 0))) return false;
    a.part_w = part->w; a.part_ws = part->w_stride;
    a.part_b = part->b; a.part_bs = part->b_stride;
    a.dw = part->w;                               // non-NULL for the argument checks; never written in this mode
    return true;
}
#define PM_PART_SLOTS(n)                                              \
    do {                                                              \
        if (slots_out) { *slots_out = (n); return PM_OK; }            \
        if (part && part->nslots != (n)) {                            \
            snprintf(pm_err_text, sizeof(pm_err_text), "pm_wgrad_part: the launch writes %d slots, the arena holds %d", (int)(n), part->nslots); \
            return PM_EINVAL;                                         \
        }                                                             \
    } while (0)

static int gather_wgrad_impl(pm_stream_t stream, const pm_gather_desc* d, const float* gathered, const float* dense,
                             float* dw, float* db, const long long* gtab, bool tab_aligned,
                             const pm_wgrad_part* part = nullptr, int* slots_out = nullptr) {
    WgradArgs a;
    if (part) { dw = part->w; db = part->b; }
    if (!fill_geom(d, a.g, false) || !gathered || !dense || (!dw && !slots_out)) return PM_EINVAL;
    a.gathered = gathered; a.dense = dense; a.dw = dw; a.db = db; a.gtab = gtab;
    a.in_gs = d->in_gs; a.w_gs = d->w_gs; a.out_gs = d->out_gs; a.bias_gs = d->bias_gs;
    if (!set_part(a, part, db != nullptr)) return PM_EINVAL;
    const bool vec4 = (d->C % 4 == 0) && aligned16(gathered) && (gtab ? tab_aligned : d->in_gs % 4 == 0);
    const bool dvec4 = (d->N % 4 == 0) && aligned16(dense) && (gtab ? tab_aligned : d->out_gs % 4 == 0);
    const WgradPlan p = plan_wgrad(a.g, d->groups, vec4, dvec4);
    PM_PART_SLOTS(p.splits);
    a.chunks_per_split = p.chunks_per_split;
    const int hw = a.g.OH * a.g.OW;
    a.step_b = 128 / hw;
    a.step_p = (128 - a.step_b * hw) / a.g.OW;
    a.step_q = 128 - a.step_b * hw - a.step_p * a.g.OW;
    dim3 grid(p.nkb * p.nnb, p.splits, d->groups);
    hipStream_t s = (hipStream_t)stream;
    if (p.rc == 2 && p.rn == 2) launch_wgrad<2, 2>(s, a, grid, p.mode, dvec4);
    else if (p.rc == 2) launch_wgrad<2, 1>(s, a, grid, p.mode, dvec4);
    else if (p.rn == 2) launch_wgrad<1, 2>(s, a, grid, p.mode, dvec4);
    else launch_wgrad<1, 1>(s, a, grid, p.mode, dvec4);
    return pm_check_launch("pm_gather_wgrad");
}

extern "C" int pm_gather_wgrad(pm_stream_t stream, const pm_gather_desc* d, const float* gathered,
                               const float* dense, float* dw, float* db) {
    return gather_wgrad_impl(stream, d, gathered, dense, dw, db, nullptr, false);
}

// Which kernel instantiation a problem dispatches to (bench.py names its roofline row with it).
extern "C" int pm_query_gemm_plan(const pm_gather_desc* d, int in_aligned16, int* bm, int* bn, int* mode) {
    Geom g;
    if (!fill_geom(d, g, true) || !bm || !bn || !mode) return PM_EINVAL;
    const bool vec4 = (d->C % 4 == 0) && in_aligned16 && (d->in_gs % 4 == 0);
    const GemmPlan p = plan_gemm(g, d->groups, vec4);
    *bm = p.bm; *bn = p.bn; *mode = p.mode + (p.direct ? 16 : 0);
    return PM_OK;
}

extern "C" int pm_query_wgrad_plan(const pm_gather_desc* d, int gathered_aligned16, int dense_aligned16, int* rc,
                                   int* rn, int* mode, int* dvec, int* workgroups) {
    Geom g;
    if (!fill_geom(d, g, false) || !rc || !rn || !mode || !dvec || !workgroups) return PM_EINVAL;
    const bool vec4 = (d->C % 4 == 0) && gathered_aligned16 && (d->in_gs % 4 == 0);
    const bool dvec4 = (d->N % 4 == 0) && dense_aligned16 && (d->out_gs % 4 == 0);
    const WgradPlan p = plan_wgrad(g, d->groups, vec4, dvec4);
    *rc = p.rc; *rn = p.rn; *mode = p.mode; *dvec = p.dvec;
    *workgroups = p.nkb * p.nnb * p.splits * d->groups;
    return PM_OK;
}

// ---- bf16x3 ("split bf16") direct path ----
// in_colsum != NULL: only the image-resident form can produce it (pm_image_conv_applies); PM_EINVAL otherwise
static bool image_insum_ok(const Geom& g, const ImagePlan& ip) {
    const int c4n = g.C / 4;
    return g.C % 4 == 0 && c4n <= 64 && 64 % c4n == 0 && g.in_act == PM_ACT_NONE &&
           ip.lds + (size_t)ip.nw * g.C * sizeof(float) <= 160 * 1024;
}

static int gather_gemm_bf16_impl(pm_stream_t stream, const pm_gather_desc* d, const float* in, const void* wsplit,
                                 const float* bias, const float* aux, const float* res, float* out, float* out2, int act2,
                                 float* in_colsum = nullptr, float* sk = nullptr, long long sk_floats = 0,
                                 long long* query = nullptr) {
    GemmArgs a;
    a.out2 = out2; a.act2 = act2; a.in_colsum = in_colsum;
    a.sk = nullptr; a.sk_stride = 0;
    if (query) {                                       // pm_gemm_splitk_floats: the dispatch below without its launches
        *query = 0;
        static float dummy_in[4] __attribute__((aligned(16)));
        in = dummy_in; wsplit = dummy_in; out = dummy_in;
    }
    if (!fill_geom(d, a.g, true) || !in || !wsplit || !out) return PM_EINVAL;
    if (d->C % 8 != 0 || (d->groups != 1 && d->w_gs % 8 != 0)) return PM_EINVAL;   // C % 32 != 0: zero-padded weight chunks
    if (!aligned16(in) || !aligned16(wsplit) || (d->in_gs % 4) != 0) return PM_EINVAL;
    if (a.g.KH * a.g.KW * ((a.g.C + BK - 1) / BK) > DMAXSTEPS || a.g.KH > 15 || a.g.KW > 15) return PM_EINVAL;
    if ((long long)a.g.B * a.g.IH * a.g.IW * a.g.C * 4 >= 0x7ffffff0LL) return PM_EINVAL;
    if (d->d != 1 && d->d != 2) return PM_EINVAL;
    a.in = in; a.w = nullptr; a.bias = bias; a.aux = aux; a.res = res; a.out = out;
    a.in_gs = d->in_gs; a.w_gs = d->w_gs; a.out_gs = d->out_gs; a.bias_gs = d->bias_gs;
    hipStream_t s = (hipStream_t)stream;
    const int G = d->groups;
    const int npad = (a.g.N + 31) / 32 * 32;
    const long long plane = (long long)a.g.KH * a.g.KW * ((a.g.C + BK - 1) / BK) * BK * npad;
    int rn = a.g.N > 32 ? 2 : 1;
    {   // 64-column workgroups need 80 KB of LDS (one per CU); when their grid would leave most CUs with a single
        // workgroup anyway, 32-column workgroups (40 KB, twice as many) hide each other's latencies: measured
        // 14x14x64 -> 7x7x64 forward 55 -> 44 us, its data gradient 73 -> 56 us, 8192x256x256 data gradient 29 -> 25 us
        const long long wgs64 = (long long)((a.g.M + 127) / 128) * ((a.g.N + 63) / 64) * d->groups;
        static const long long rn2_min = getenv("PM_RN2_MIN_WGS") ? atoll(getenv("PM_RN2_MIN_WGS")) : 512;   // A/B knob
        if (rn == 2 && wgs64 < rn2_min) rn = 1;
    }
    const __bf16* ws = reinterpret_cast<const __bf16*>(wsplit);
#ifndef PM_IMAGE_INSUM
    if (in_colsum) return PM_EINVAL;                   // compiled out of the default build (see pm_image_conv_insum_applies)
#endif
    if (in_colsum) {
        ImagePlan ipc;
        if (plan_skinny(a.g, G) || getenv("PM_NO_IMAGE_CONV") || !plan_image(a.g, G, ipc) || !image_insum_ok(a.g, ipc))
            return PM_EINVAL;
    }
    if (plan_skinny(a.g, G)) {                         // one output position, few rows, long K: one launch, K over the waves
        a.ksplit = 1;
        if (query) return PM_OK;
        PM_KTAG("skinny_gemm_bf16_kernel");
        hipLaunchKernelGGL(skinny_gemm_bf16_kernel, dim3((a.g.M + 15) / 16, (a.g.N + 15) / 16), dim3(64 * SK_NW), 0, s, a, ws,
                           npad, plane);
        return pm_check_launch("pm_gather_gemm_bf16(skinny)");
    }
    ImagePlan ip;
    static const bool image_off = getenv("PM_NO_IMAGE_CONV") != nullptr;   // A/B switch for measurements
    if (!image_off && plan_image(a.g, G, ip)) {       // whole input image of a workgroup resident in LDS
        a.ksplit = 1;
        if (query) return PM_OK;
        if (ip.nw == 8 && ip.t == 4) launch_image<8, 4>(ip, s, a, ws, npad, plane);
        else if (ip.nw == 8 && ip.t == 2) launch_image<8, 2>(ip, s, a, ws, npad, plane);
        else if (ip.nw == 8) launch_image<8, 1>(ip, s, a, ws, npad, plane);
        else if (ip.t == 4) launch_image<4, 4>(ip, s, a, ws, npad, plane);
        else if (ip.t == 2) launch_image<4, 2>(ip, s, a, ws, npad, plane);
        else launch_image<4, 1>(ip, s, a, ws, npad, plane);
        return pm_check_launch("pm_gather_gemm_bf16(image)");
    }
    if (!image_off && plan_image_d2(a.g, G, ip)) {    // zero-dilated problems: class-major walk over the resident source image
        a.ksplit = 1;
        if (query) return PM_OK;
        if (ip.t == 4) launch_image_d2<4>(ip, s, a, ws, npad, plane);
        else if (ip.t == 2) launch_image_d2<2>(ip, s, a, ws, npad, plane);
        else launch_image_d2<1>(ip, s, a, ws, npad, plane);
        return pm_check_launch("pm_gather_gemm_bf16(image_d2)");
    }
    PatchPlan pp;
    static const bool patch_off = getenv("PM_NO_PATCH") != nullptr;      // A/B switch for measurements
    if (!patch_off && plan_patch(a.g, G, rn, pp)) {   // stride-1 convs on grids >= 12 wide: patch-staged form
        a.ksplit = 1;
        if (query) return PM_OK;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_conv_bf16_kernel<1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_conv_bf16_kernel<2>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            attr_set = true;
        }
        static const bool bd_off = getenv("PM_NO_PATCH_BD") != nullptr;     // A/B switch for measurements
        if (!bd_off) {
            static bool attr_bd = false;
            if (!attr_bd) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_conv_bd_bf16_kernel<1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_conv_bd_bf16_kernel<2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
                attr_bd = true;
            }
            PM_KTAG("patch_conv_bd_bf16_kernel<%d>", rn);
            if (rn == 1) hipLaunchKernelGGL(patch_conv_bd_bf16_kernel<1>, pp.grid, dim3(256), pp.lds_bd, s, a, ws, npad, plane, pp.tw_log2);
            else hipLaunchKernelGGL(patch_conv_bd_bf16_kernel<2>, pp.grid, dim3(256), pp.lds_bd, s, a, ws, npad, plane, pp.tw_log2);
            return pm_check_launch("pm_gather_gemm_bf16(patch_bd)");
        }
        PM_KTAG("patch_conv_bf16_kernel<%d>", rn);
        if (rn == 1) hipLaunchKernelGGL(patch_conv_bf16_kernel<1>, pp.grid, dim3(256), pp.lds, s, a, ws, npad, plane, pp.tw_log2);
        else hipLaunchKernelGGL(patch_conv_bf16_kernel<2>, pp.grid, dim3(256), pp.lds, s, a, ws, npad, plane, pp.tw_log2);
        return pm_check_launch("pm_gather_gemm_bf16(patch)");
    }
    PatchCpPlan pc;
    static const bool cp_off = getenv("PM_NO_PATCH_CP") != nullptr;      // A/B switch for measurements
    if (!patch_off && !cp_off && plan_patch_cp(a.g, G, rn, pc)) {   // deep stride-1 convs: the patch CC channels at a time
        a.ksplit = 1;
        if (query) return PM_OK;
        PM_KTAG("patch_conv_cp_bf16_kernel<%d, %d>", rn, pc.spc);
#define PM_CP(RNv, SPCv, NSv)                                                                                          \
    do {                                                                                                               \
        static bool attr = false;                                                                                      \
        if (!attr) {                                                                                                   \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_conv_cp_bf16_kernel<RNv, SPCv, NSv>),       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);                         \
            attr = true;                                                                                               \
        }                                                                                                              \
        hipLaunchKernelGGL((patch_conv_cp_bf16_kernel<RNv, SPCv, NSv>), pc.grid, dim3(256), pc.lds, s, a, ws, npad, plane, \
                           pc.tw_log2);                                                                                \
    } while (0)
        if (rn == 1) {
            if (pc.spc == 8) PM_CP(1, 8, 4);
            else if (pc.spc == 12) PM_CP(1, 12, 6);
            else PM_CP(1, 18, 6);
        } else {
            if (pc.spc == 8) PM_CP(2, 8, 4);
            else if (pc.spc == 12) PM_CP(2, 12, 6);
            else PM_CP(2, 18, 6);
        }
#undef PM_CP
        return pm_check_launch("pm_gather_gemm_bf16(patch_cp)");
    }
    PatchD2Plan pd;
    static const bool d2_off = getenv("PM_NO_PATCH_D2") != nullptr;      // A/B switch for measurements
    const int rn_d2 = a.g.N > 32 ? 2 : 1;
    if (!d2_off && plan_patch_d2(a.g, G, rn_d2, pd)) {   // zero-dilated problems: four residue classes off one staged patch
        a.ksplit = 1;
        if (query) return PM_OK;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_d2_bf16_kernel<1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_d2_bf16_kernel<2>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            attr_set = true;
        }
        PM_KTAG("patch_d2_bf16_kernel<%d>", rn_d2);
        if (rn_d2 == 1) hipLaunchKernelGGL(patch_d2_bf16_kernel<1>, pd.grid, dim3(256), pd.lds, s, a, ws, npad, plane, pd.tw_log2, pd.ni);
        else hipLaunchKernelGGL(patch_d2_bf16_kernel<2>, pd.grid, dim3(256), pd.lds, s, a, ws, npad, plane, pd.tw_log2, pd.ni);
        return pm_check_launch("pm_gather_gemm_bf16(patch_d2)");
    }
    const GemmPlan p = plan_gemm(a.g, G, true);
    a.ksplit = p.ksplit;
    a.sk_stride = (long long)a.g.M * a.g.N;
    if (query) {
        *query = p.ksplit > 1 ? a.sk_stride * p.ksplit : 0;
        return PM_OK;
    }
    if (p.ksplit > 1) {
        if (sk) {
            if (sk_floats < a.sk_stride * p.ksplit || !aligned16(sk)) return PM_EINVAL;
            a.sk = sk;
        } else if (pm_zero_async(s, out, (size_t)a.g.M * a.g.N * sizeof(float))) return PM_ELAUNCH;
    }
    dim3 grid((a.g.M + 127) / 128, (a.g.N + 32 * rn - 1) / (32 * rn), G * a.ksplit);
    if (rn == 1 && d->d == 1) launch_direct_bf16<1, 1>(s, a, grid, ws, npad, plane);
    else if (rn == 1) launch_direct_bf16<1, 2>(s, a, grid, ws, npad, plane);
    else if (d->d == 1) launch_direct_bf16<2, 1>(s, a, grid, ws, npad, plane);
    else launch_direct_bf16<2, 2>(s, a, grid, ws, npad, plane);
    if (p.ksplit > 1) {
        long long total = (long long)a.g.M * a.g.N;
        long long blocks = (total + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)blocks, G), dim3(256), 0, s, a, total);
    }
    return pm_check_launch("pm_gather_gemm_bf16");
}

extern "C" int pm_gather_gemm_bf16(pm_stream_t stream, const pm_gather_desc* d, const float* in, const void* wsplit,
                                   const float* bias, const float* aux, const float* res, float* out) {
    return gather_gemm_bf16_impl(stream, d, in, wsplit, bias, aux, res, out, nullptr, PM_ACT_NONE);
}

extern "C" int pm_gather_gemm_bf16_insum(pm_stream_t stream, const pm_gather_desc* d, const float* in, const void* wsplit,
                                         const float* bias, const float* aux, const float* res, float* out, float* in_colsum) {
    if (!in_colsum) return PM_EINVAL;
    return gather_gemm_bf16_impl(stream, d, in, wsplit, bias, aux, res, out, nullptr, PM_ACT_NONE, in_colsum);
}

extern "C" int pm_image_conv_insum_applies(const pm_gather_desc* d) {
#ifndef PM_IMAGE_INSUM
    // The column-sum code inside image_conv_bf16_kernel is compiled out of the default build.  Present but UNUSED it split the
    // PM-VAE step into two regimes picked at random per process (8 runs of 300 steps on one box: 1.307 - 1.337 ms without the code,
    // 3 x ~1.30 and 5 x ~1.36 ms with it) - a few more registers / instructions in the staging phase of the step's most frequent
    // kernel are enough to move how the two streams' launches interleave.  -DPM_IMAGE_INSUM (tools/build_variant.sh) restores it.
    (void)d;
    return 0;
#else
    GemmArgs a;
    ImagePlan ip;
    if (!d || !fill_geom(d, a.g, true) || d->C % 8 != 0 || d->d != 1) return 0;
    if (plan_skinny(a.g, d->groups) || getenv("PM_NO_IMAGE_CONV") || !plan_image(a.g, d->groups, ip)) return 0;
    return image_insum_ok(a.g, ip) ? 1 : 0;
#endif
}

extern "C" int pm_gather_gemm_bf16_dual(pm_stream_t stream, const pm_gather_desc* d, const float* in, const void* wsplit,
                                        const float* bias, const float* aux, const float* res, float* out, float* out2,
                                        int act2) {
    if (!out2) return PM_EINVAL;
    return gather_gemm_bf16_impl(stream, d, in, wsplit, bias, aux, res, out, out2, act2);
}

// Split-K without atomics (round 4): the K slices of a short-grid GEMM store their partial tiles into the caller's scratch slabs
// and the epilogue kernel adds them in slice order.  pm_gemm_splitk_floats: how many floats of scratch (0: this problem does not
// split K - any scratch is ignored).  scratch == NULL is PM_EINVAL in the _sk entry points; the plain entry points keep the
// f32-atomic form.
extern "C" int pm_gemm_splitk_floats(const pm_gather_desc* d, int bf16, int in_aligned16, long long* floats) {
    if (!d || !floats) return PM_EINVAL;
    *floats = 0;
    if (bf16) return gather_gemm_bf16_impl(nullptr, d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, PM_ACT_NONE,
                                           nullptr, nullptr, 0, floats);
    const float* fake = in_aligned16 ? reinterpret_cast<const float*>(16) : reinterpret_cast<const float*>(4);
    return gather_gemm_impl(nullptr, d, fake, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, floats);
}

extern "C" int pm_gather_gemm_bf16_sk(pm_stream_t stream, const pm_gather_desc* d, const float* in, const void* wsplit,
                                      const float* bias, const float* aux, const float* res, float* out, float* out2,
                                      int act2, float* scratch, long long scratch_floats) {
    if (!scratch) return PM_EINVAL;
    return gather_gemm_bf16_impl(stream, d, in, wsplit, bias, aux, res, out, out2, act2, nullptr, scratch, scratch_floats);
}

extern "C" int pm_split_weights(pm_stream_t stream, const float* params, void* out_bf16, const pm_split_job* jobs_dev,
                                int njobs, int total_blocks) {
    if (!params || !out_bf16 || !jobs_dev || njobs <= 0 || total_blocks <= 0) return PM_EINVAL;
    static_assert(sizeof(pm_split_job) == sizeof(SplitJob), "pm_split_job layout");
    PM_KTAG("split_weights_kernel<%d>", total_blocks >= 32768 ? 8 : 2);
    if (total_blocks >= 32768)
        hipLaunchKernelGGL(split_weights_kernel<8>, dim3((total_blocks + 7) / 8), dim3(256), 0, (hipStream_t)stream, params,
                           reinterpret_cast<__bf16*>(out_bf16), reinterpret_cast<const SplitJob*>(jobs_dev), njobs, total_blocks);
    else
        hipLaunchKernelGGL(split_weights_kernel<2>, dim3((total_blocks + 1) / 2), dim3(256), 0, (hipStream_t)stream, params,
                           reinterpret_cast<__bf16*>(out_bf16), reinterpret_cast<const SplitJob*>(jobs_dev), njobs, total_blocks);
    return pm_check_launch("pm_split_weights");
}

static int launch_big_wgrad(pm_stream_t stream, WgradArgs& a, const pm_gather_desc* d, bool cpad, const pm_wgrad_part* part,
                            int* slots_out);

static int gather_wgrad_bf16_impl(pm_stream_t stream, const pm_gather_desc* d, const float* gathered, const float* dense,
                                  float* dw, float* db, const long long* gtab, bool tab_aligned,
                                  const pm_wgrad_part* part = nullptr, int* slots_out = nullptr) {
    WgradArgs a;
    if (part) { dw = part->w; db = part->b; }
    if (!fill_geom(d, a.g, false) || !gathered || !dense || (!dw && !slots_out)) return PM_EINVAL;
    const bool cpad = d->C % 32 != 0 && d->C > 32 && d->C < 64 && d->C % 4 == 0;      // one zero-padded tap per k-block
    if ((d->C % 32 != 0 && !cpad) || d->N % 4 != 0 || (d->d != 1 && d->d != 2)) return PM_EINVAL;
    if (!aligned16(gathered) || !aligned16(dense)) return PM_EINVAL;
    if (gtab ? !tab_aligned : (d->in_gs % 4 != 0 || d->out_gs % 4 != 0)) return PM_EINVAL;
    a.gathered = gathered; a.dense = dense; a.dw = dw; a.db = db; a.gtab = gtab;
    a.in_gs = d->in_gs; a.w_gs = d->w_gs; a.out_gs = d->out_gs; a.bias_gs = d->bias_gs;
    if (!set_part(a, part, db != nullptr)) return PM_EINVAL;
    {   // 32 gathered channels, both images of a sample fit LDS: image-resident persistent form (any stride)
        static const bool image_off = getenv("PM_NO_IMAGE_WGRAD") != nullptr;      // A/B switch for measurements
        const Geom& g = a.g;
        const int rn = g.N > 32 ? 2 : 1;
        const size_t lds = ((size_t)(g.IH * g.IW + 1) * 40 + (size_t)(g.OH * g.OW + 1) * (32 * rn + 8)) * 2 * 2;
        if (!image_off && !gtab && d->groups == 1 && g.d == 1 && g.C == 32 && g.in_act == PM_ACT_NONE && g.N <= 64 &&
            g.N % 4 == 0 && g.KH * g.KW <= 28 && g.KH * g.KW >= 9 && g.B >= 64 && g.OW >= 9 && g.OW <= 16 && g.OH >= 4 &&
            lds <= 158 * 1024 && g.IH * g.IW <= (rn == 2 ? 224 : 800) && g.OH * g.OW <= (rn == 2 ? 208 : 224)) {
            // Register-staged pieces per thread: 25 (64 columns: 7) / 13 (7).  The atomic flush moves G x (taps x 32 x N) / nsub
            // floats at about 1.2 TB/s chip-wide, so the taps are dealt to nsub = 2 classes of workgroups (each sample is then
            // staged by both).  Measured at B = 256 (14x14 32->64 / 28->14 stride 2 32->32, us): nsub 1 on 128 workgroups
            // 64 / 44, nsub 2 on 192: 54 / 44, on 256: 56 / 41, nsub 4 on 256: 53 / 43; the PM-VAE step is fastest with
            // 2 on 192 (PM_IW_GRID / PM_IW_SPLIT for experiments).
            static const int grid_env = getenv("PM_IW_GRID") ? atoi(getenv("PM_IW_GRID")) : 0;
            static const int sub_env = getenv("PM_IW_SPLIT") ? atoi(getenv("PM_IW_SPLIT")) : 0;
            const int nsub = sub_env == 1 || sub_env == 2 ? sub_env : 2;
            int grid = grid_env > 0 ? grid_env : (nsub == 1 ? 128 : 192);
            if (grid > g.B * nsub) grid = g.B * nsub;
            grid -= grid % nsub;
            PM_PART_SLOTS(grid / nsub);
            const int npi = (g.IH * g.IW * 8 + 255) / 256;
            hipStream_t s = (hipStream_t)stream;
#define PM_IW(RNv, NPIv, NPDv, TPWv)                                                                                  \
    do {                                                                                                             \
        static bool attr = false;                                                                                    \
        if (!attr) {                                                                                                 \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&image_wgrad_bf16_kernel<RNv, NPIv, NPDv, TPWv>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);                       \
            attr = true;                                                                                             \
        }                                                                                                            \
        PM_KTAG("image_wgrad_bf16_kernel<%d, %d, %d, %d>", RNv, NPIv, NPDv, TPWv);                                   \
        hipLaunchKernelGGL((image_wgrad_bf16_kernel<RNv, NPIv, NPDv, TPWv>), dim3(grid), dim3(256), lds, s, a, nsub); \
    } while (0)
#define PM_IW3(TPWv)                                                                                                  \
    do {                                                                                                             \
        if (rn == 1 && npi <= 7) PM_IW(1, 7, 7, TPWv);                                                               \
        else if (rn == 1) PM_IW(1, 25, 7, TPWv);                                                                     \
        else PM_IW(2, 7, 13, TPWv);                                                                                  \
    } while (0)
            if (nsub == 1) PM_IW3(7);
            else PM_IW3(4);
#undef PM_IW3
#undef PM_IW
            return pm_check_launch("pm_gather_wgrad_bf16(image)");
        }
    }
    {   // stride-1 problems with 32 gathered channels on grids >= 12 wide: patch-staged persistent form
        static const bool patch_off = getenv("PM_NO_PATCH_WGRAD") != nullptr;
        const Geom& g = a.g;
        const int taps = g.KH * g.KW;
        if (!patch_off && d->groups == 1 && g.a == 1 && g.d == 1 && g.C == 32 && g.in_act == PM_ACT_NONE && g.N <= 64 &&
            g.N % 4 == 0 && (taps == 25 || taps == 9) && g.OW >= 12 && g.OH >= 4) {
            const int tw_log2 = g.OW > 16 ? 5 : 4;
            const int TW = 1 << tw_log2, TH = 128 >> tw_log2;
            const int tiles_x = (g.OW + TW - 1) / TW, tiles_y = (g.OH + TH - 1) / TH;
            const int rn = g.N > 32 ? 2 : 1;
            const size_t patch = (size_t)(TH + g.KH - 1) * (TW + g.KW - 1) * (g.C + 8) * 2 * 2;
            const size_t lds = patch + (size_t)128 * (32 * rn + 8) * 2 * 2;
            const int ptotal = (TH + g.KH - 1) * (TW + g.KW - 1) * (g.C / 4);
            if ((long long)tiles_x * TW * tiles_y * TH * 2 <= 3LL * g.OH * g.OW && lds <= 150 * 1024 && ptotal <= 2560) {
                const int ntiles = g.B * tiles_y * tiles_x;
                // persistent workgroups: each flushes taps x 32 x N accumulators with atomics at the end, so give
                // every workgroup several tiles to amortise that (PM_PW_TILES: tiles per workgroup, for experiments)
                // measured (B = 256): 28x28 layers 7-8 tiles per workgroup, 14x14 layers 4; more tiles per workgroup
                // lengthen the serial tile loop faster than they shorten the flush
                static const int tpw_env = getenv("PM_PW_TILES") ? atoi(getenv("PM_PW_TILES")) : 0;
                const int tpw = tpw_env > 0 ? tpw_env : (ntiles >= 1024 ? 8 : 4);
                int grid = (ntiles + tpw - 1) / tpw;
                if (grid > 256) grid = 256;
                if (grid < 1) grid = 1;
                PM_PART_SLOTS(grid);
                hipStream_t s = (hipStream_t)stream;
                static bool attr_set = false;
                if (!attr_set) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_wgrad_bf16_kernel<1, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_wgrad_bf16_kernel<2, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_wgrad_bf16_kernel<1, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_wgrad_bf16_kernel<2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
                    attr_set = true;
                }
                PM_KTAG("patch_wgrad_bf16_kernel<%d, %d>", rn, taps == 25 ? 7 : 3);
                if (taps == 25 && rn == 1) hipLaunchKernelGGL((patch_wgrad_bf16_kernel<1, 7>), dim3(grid), dim3(256), lds, s, a, tw_log2, ntiles);
                else if (taps == 25) hipLaunchKernelGGL((patch_wgrad_bf16_kernel<2, 7>), dim3(grid), dim3(256), lds, s, a, tw_log2, ntiles);
                else if (rn == 1) hipLaunchKernelGGL((patch_wgrad_bf16_kernel<1, 3>), dim3(grid), dim3(256), lds, s, a, tw_log2, ntiles);
                else hipLaunchKernelGGL((patch_wgrad_bf16_kernel<2, 3>), dim3(grid), dim3(256), lds, s, a, tw_log2, ntiles);
                return pm_check_launch("pm_gather_wgrad_bf16(patch)");
            }
        }
    }
    {   // 64 x 64-channel layers on grids <= 8 wide (the PM-VAE's 14x14 <-> 7x7 stride-2 layers): one tap row per class
        static const bool rt_off = getenv("PM_NO_ROWTAP_WGRAD") != nullptr;          // A/B switch for measurements
        const Geom& g = a.g;
        if (!rt_off && !gtab && d->groups == 1 && g.d == 1 && g.C == 64 && g.N == 64 && g.in_act == PM_ACT_NONE &&
            g.KW == 5 && g.KH >= 1 && g.KH <= 8 && g.OW >= 4 && g.OW <= 8 && g.OH >= 2 && g.OH <= 8 && g.IW <= 16 &&
            g.IH <= 16 && g.B >= 32) {
            // sample groups = slots of the partial-sum arena (each 410 KB for a 5x5 kernel): 48 groups x 5 tap rows = 240
            // workgroups of ~5 samples each (PM_RT_GROUPS for experiments)
            static const int grp_env = getenv("PM_RT_GROUPS") ? atoi(getenv("PM_RT_GROUPS")) : 48;
            int ngroups = grp_env > 0 ? grp_env : 48;
            if (ngroups > g.B) ngroups = g.B;
            PM_PART_SLOTS(ngroups);
            const size_t lds = ((size_t)(g.OH * g.IW + 1) * 72 + (size_t)64 * 72) * 2 * sizeof(short);
            static bool attr_rt = false;
            if (!attr_rt) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rowtap_wgrad_bf16_kernel<5>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
                attr_rt = true;
            }
            PM_KTAG("rowtap_wgrad_bf16_kernel<5>");
            hipLaunchKernelGGL((rowtap_wgrad_bf16_kernel<5>), dim3(ngroups * g.KH), dim3(256), lds, (hipStream_t)stream, a, ngroups);
            return pm_check_launch("pm_gather_wgrad_bf16(rowtap)");
        }
    }
    a.cpad = cpad ? 1 : 0;
    {
        const int rc_big = launch_big_wgrad(stream, a, d, cpad, part, slots_out);
        if (rc_big != 1) return rc_big;
    }
    Geom gplan = a.g;
    if (cpad) {                 // planned as the 64-channel problem it is executed as
        gplan.C = 64;
        gplan.K = a.g.KH * a.g.KW * 64;
    }
    WgradPlan p = plan_wgrad(gplan, d->groups, true, true);
    if (cpad) p.rc = p.rn = 2;  // the sub kernel (64 x 64 tiles) is the only form with the padded-tap mode
    PM_PART_SLOTS(p.splits);
    a.chunks_per_split = p.chunks_per_split;
    const int hw = a.g.OH * a.g.OW;
    a.step_b = 128 / hw;
    a.step_p = (128 - a.step_b * hw) / a.g.OW;
    a.step_q = 128 - a.step_b * hw - a.step_p * a.g.OW;
    static const bool xcd_off = getenv("PM_WG_NOXCD") != nullptr;       // A/B switches for measurements
    static const bool sub_off = getenv("PM_WG_NOSUB") != nullptr;
    a.ntiles = p.nkb * p.nnb;
    a.nsplits = p.splits;
    a.xcd_map = (!xcd_off && p.splits % 8 == 0) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    const int dd = d->d;
    if ((!sub_off || cpad) && p.rc == 2 && p.rn == 2) {     // 64 x 64 tiles: the occupancy-oriented form, 64-row chunks
        a.chunks_per_split = 2 * p.chunks_per_split;
        a.step_b = 64 / hw;
        a.step_p = (64 - a.step_b * hw) / a.g.OW;
        a.step_q = 64 - a.step_b * hw - a.step_p * a.g.OW;
        dim3 grid2(a.ntiles * p.splits, 1, d->groups);
        PM_KTAG("gather_wgrad_bf16_sub_kernel<%d>", dd);
        if (dd == 1) hipLaunchKernelGGL((gather_wgrad_bf16_sub_kernel<1>), grid2, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((gather_wgrad_bf16_sub_kernel<2>), grid2, dim3(256), 0, s, a);
        return pm_check_launch("pm_gather_wgrad_bf16(sub)");
    }
    dim3 grid(a.ntiles * p.splits, 1, d->groups);
#define PM_WB(RCv, RNv)                                                                                            \
    do {                                                                                                            \
        if (dd == 1) hipLaunchKernelGGL((gather_wgrad_bf16_kernel<RCv, RNv, 1>), grid, dim3(256), 0, s, a);         \
        else hipLaunchKernelGGL((gather_wgrad_bf16_kernel<RCv, RNv, 2>), grid, dim3(256), 0, s, a);                 \
    } while (0)
    PM_KTAG("gather_wgrad_bf16_kernel<%d, %d, %d>", p.rc, p.rn, dd);
    if (p.rc == 2 && p.rn == 2) PM_WB(2, 2);
    else if (p.rc == 2) PM_WB(2, 1);
    else if (p.rn == 2) PM_WB(1, 2);
    else PM_WB(1, 1);
#undef PM_WB
    return pm_check_launch("pm_gather_wgrad_bf16");
}

extern "C" int pm_gather_wgrad_bf16(pm_stream_t stream, const pm_gather_desc* d, const float* gathered,
                                    const float* dense, float* dw, float* db) {
    return gather_wgrad_bf16_impl(stream, d, gathered, dense, dw, db, nullptr, false);
}

// d->groups weight gradients of ONE geometry whose operands are separately allocated: group g reads gathered + table[4g],
// dense + table[4g+1] and accumulates into dw + table[4g+2] (and db + table[4g+3]); table lives in device memory.
extern "C" int pm_gather_wgrad_table(pm_stream_t stream, const pm_gather_desc* d, const float* gathered, const float* dense,
                                     float* dw, float* db, const long long* table, int all_aligned16, int use_bf16) {
    if (!table || d->groups < 1) return PM_EINVAL;
    if (use_bf16) return gather_wgrad_bf16_impl(stream, d, gathered, dense, dw, db, table, all_aligned16 != 0);
    return gather_wgrad_impl(stream, d, gathered, dense, dw, db, table, all_aligned16 != 0);
}

// Partial-sum forms (see pm_wgrad_part in pmhip.h).  table may be NULL (plain / uniformly strided groups).
extern "C" int pm_gather_wgrad_part(pm_stream_t stream, const pm_gather_desc* d, const float* gathered, const float* dense,
                                    const long long* table, int all_aligned16, int use_bf16, const pm_wgrad_part* part) {
    if (!d || !part || d->groups < 1) return PM_EINVAL;
    if (use_bf16) return gather_wgrad_bf16_impl(stream, d, gathered, dense, nullptr, nullptr, table, all_aligned16 != 0, part);
    return gather_wgrad_impl(stream, d, gathered, dense, nullptr, nullptr, table, all_aligned16 != 0, part);
}

extern "C" int pm_wgrad_part_slots(const pm_gather_desc* d, const float* gathered, const float* dense, int has_table,
                                   int all_aligned16, int use_bf16, int* nslots) {
    if (!d || !nslots || d->groups < 1) return PM_EINVAL;
    static const long long dummy_tab[4] = {0, 0, 0, 0};
    const long long* tab = has_table ? dummy_tab : nullptr;       // only its presence enters the plan
    if (use_bf16) return gather_wgrad_bf16_impl(nullptr, d, gathered, dense, nullptr, nullptr, tab, all_aligned16 != 0, nullptr, nslots);
    return gather_wgrad_impl(nullptr, d, gathered, dense, nullptr, nullptr, tab, all_aligned16 != 0, nullptr, nslots);
}

// Large weight gradients with 128-channel-aligned taps: 128 x 128 tiles (PM_WG_NOBIG=1: the 64 x 64 form, for A/B runs).  Returns
// 1 when it did not apply (the caller goes on), else the launch status.  Defined at the END of this file so that the kernel's
// instantiations are emitted behind every other kernel of the code object (see gather_wgrad_bf16_big_kernel).
static int launch_big_wgrad(pm_stream_t stream, WgradArgs& a, const pm_gather_desc* d, bool cpad, const pm_wgrad_part* part,
                            int* slots_out) {
        const Geom& g = a.g;
        const long long work = (long long)g.M * g.K * g.N * d->groups;
        static const int big_mink = getenv("PM_WG_BIG_MINK") ? atoi(getenv("PM_WG_BIG_MINK")) : 512;     // A/B knob
        if (!getenv("PM_WG_NOBIG") && !cpad && g.C % 128 == 0 && g.K >= big_mink && g.N >= 128 && g.M >= 2048 && work >= (1LL << 30)) {
            const int nkb = g.K / 128, nnb = (g.N + 127) / 128;
            static const int bmc = getenv("PM_WG_BIG_BMC") ? atoi(getenv("PM_WG_BIG_BMC")) : 32;   // rows per chunk (A/B knob)
            const int total_chunks = (g.M + bmc - 1) / bmc;
            long long tiles = (long long)nkb * nnb * d->groups;
            static const int target = getenv("PM_WG_BIG_TARGET") ? atoi(getenv("PM_WG_BIG_TARGET")) : 1536;   // workgroups wanted (A/B knob)
            int splits = (int)((target + tiles - 1) / tiles);
            if (splits > total_chunks / 16) splits = total_chunks / 16;
            if (splits < 1) splits = 1;
            a.chunks_per_split = (total_chunks + splits - 1) / splits;
            splits = (total_chunks + a.chunks_per_split - 1) / a.chunks_per_split;
            PM_PART_SLOTS(splits);
            const int hw = g.OH * g.OW;
            a.step_b = bmc / hw;
            a.step_p = (bmc - a.step_b * hw) / g.OW;
            a.step_q = bmc - a.step_b * hw - a.step_p * g.OW;
            a.ntiles = nkb * nnb;
            a.nsplits = splits;
            a.xcd_map = 0;
            const size_t lds = (size_t)2 * bmc * (136 + 136) * sizeof(short);
            static bool attr_big = false;
            if (!attr_big) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_wgrad_bf16_big_kernel<1, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_wgrad_bf16_big_kernel<2, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
                attr_big = true;
            }
            dim3 gridb(a.ntiles * splits, 1, d->groups);
            PM_KTAG("gather_wgrad_bf16_big_kernel<%d, %d>", d->d, bmc);
            hipStream_t sb = (hipStream_t)stream;
            if (d->d == 1 && bmc == 64) hipLaunchKernelGGL((gather_wgrad_bf16_big_kernel<1, 64>), gridb, dim3(256), lds, sb, a);
            else if (d->d == 1) hipLaunchKernelGGL((gather_wgrad_bf16_big_kernel<1, 32>), gridb, dim3(256), lds, sb, a);
            else if (bmc == 64) hipLaunchKernelGGL((gather_wgrad_bf16_big_kernel<2, 64>), gridb, dim3(256), lds, sb, a);
            else hipLaunchKernelGGL((gather_wgrad_bf16_big_kernel<2, 32>), gridb, dim3(256), lds, sb, a);
            return pm_check_launch("pm_gather_wgrad_bf16(big)");
        }
    return 1;
}
