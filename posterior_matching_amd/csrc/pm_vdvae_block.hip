// A VDVAE bottleneck Block (reference vdvae.py:263-299) as ONE launch forward and ONE launch for its data gradients.
//
//   forward :  h1 = c1(xg)   h2 = c2(gelu h1)   h3 = c3(gelu h2)   out = c4(gelu h3) + res       (xg = gelu(x), built by the caller)
//   backward:  dh3 = (dout W4^T) gelu'(h3)   dh2 = (c3^T dh3) gelu'(h2)   dh1 = (c2^T dh2) gelu'(h1)   dxg = dh1 W1^T [gelu'(x) + res]
//
// c1 / c4 are 1x1, c2 / c3 are 3x3 SAME (1x1 at resolutions <= 2), the bottleneck width is `mid` (48 for every reference
// config) and the step of 1 860 launches was bound by the dependent chain of these tiny GEMMs (15 - 27 us each, one kernel in
// flight 75 % of the time: DESIGN.md section 6).  Here a workgroup owns a band of R rows of NI images: the `mid`-channel
// intermediates never leave the CU - they sit in LDS as hi / lo bf16 planes (zero halo columns left and right, zero rows
// outside the image), the 3x3 taps are shifted 16-byte LDS reads, and the halo rows a band needs from its neighbours are
// recomputed (1x1 stage on R + 4 rows, first 3x3 on R + 2).  Arithmetic is the engine's bf16x3 (three bf16 MFMA products
// per operand pair, f32 accumulate) on v_mfma_f32_16x16x32_bf16: 16-column tiles cover mid = 48 exactly.  Weights are
// the K-contiguous hi / lo copies pm_split_weights already keeps per layer and direction ([plane][tap][32-chunk][npad][32]),
// read as MFMA B fragments straight from L2, two k-steps ahead.
// The pre-activations h1..h3 and their gelus g1..g3 (inputs of the weight gradients) are stored exactly as the unfused
// path stores them, so the weight-gradient launches and every parity test are unchanged.
#include <cstdlib>
#include "pm_common.h"

namespace {

typedef __bf16 vb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 vb_bf16x2 __attribute__((ext_vector_type(2)));
typedef float vb_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned vb_u32x4 __attribute__((ext_vector_type(4)));

constexpr int MIDP = 64;         // LDS channels per position: mid padded to whole 32-chunks (zeros past mid)
constexpr int PS = MIDP + 8;     // bf16 per position incl. 16 B pad: 144-byte pitch, conflict-free 16-byte reads
constexpr int MAXM = 256;        // positions a workgroup may own per stage (16 m-tiles of 16)

struct BlockArgs {
    // tensors
    const float* xin;            // forward: xg [B,H,W,Cin] (or the raw first source when gelu_in);  backward: dout [B,H,W,Cout]
    const float* xin2;           // forward, gelu_in: second raw source (channels Ca..Cin), may be NULL
    float* xg_out;               // forward, gelu_in: gelu([xin | xin2]) is stored here for the weight gradient
    int Ca, gelu_in;
    const float* res;            // forward: residual added to out (may be NULL);  backward: added to dxg (may be NULL)
    const float* xpre;           // backward only: dxg *= gelu'(xpre) (may be NULL)
    float* hh[3];                // h1, h2, h3 [B,H,W,mid]       (backward: read)
    float* gg[3];                // forward: g1, g2, g3 written;  backward: dh1, dh2, dh3 written (index = layer - 1)
    float* out;                  // forward: out [B,H,W,Cout];    backward: dxg [B,H,W,Cin]
    const __bf16* w[4];          // split weights of c1..c4 for this direction
    long long plane[4];          // elements between the hi and lo planes of each
    const float* bias[4];        // forward only
    int B, H, W, Cin, Cout, mid, k3;   // k3: 3 (3x3 middle convs) or 1
    int R, NI, bands;            // rows per band, images per workgroup, bands per image
    int dense;                   // the k3 x k3 layers' weights are dense_k split copies: K = tap * mid + c, no per-tap padding
};

__device__ __forceinline__ void vb_split8(const f32x4& x0, const f32x4& x1, vb_bf16x8& hi, vb_bf16x8& lo) {
    vb_u32x4 hp, lp;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a0 = j < 2 ? x0[2 * j] : x1[2 * j - 4];
        const float a1 = j < 2 ? x0[2 * j + 1] : x1[2 * j - 3];
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(vb_f32x2{a0, a1}, vb_bf16x2));
        const float f0 = __builtin_bit_cast(float, h << 16);
        const float f1 = __builtin_bit_cast(float, h & 0xffff0000u);
        hp[j] = h;
        lp[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(vb_f32x2{a0 - f0, a1 - f1}, vb_bf16x2));
    }
    hi = __builtin_bit_cast(vb_bf16x8, hp);
    lo = __builtin_bit_cast(vb_bf16x8, lp);
}

// jax.nn.gelu(approximate=True) and its derivative through ONE exponential each: tanh(u) = 1 - 2 / (1 + e^(2u)).
// (tanhf expands to a long polynomial / branch sequence; inlined 72 times it alone overflowed the instruction cache)
__device__ __forceinline__ float vb_tanh(float u) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * u)); }
__device__ __forceinline__ float vb_gelu(float x) {
    return 0.5f * x * (1.f + vb_tanh(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float vb_gelu_d(float x) {
    const float t = vb_tanh(0.7978845608028654f * (x + 0.044715f * x * x * x));
    return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * 0.7978845608028654f * (1.f + 0.134145f * x * x);
}

// keeps a value materialised HERE: without it the compiler sinks "acc + bias" (and with it the wait for the bias load)
// into every conditional store block, behind the stores of the previous element
__device__ __forceinline__ float vb_pin(float x) {
    asm volatile("" : "+v"(x));
    return x;
}

__device__ __forceinline__ void vb_store_split(__bf16* hi, __bf16* lo, int off, float v) {
    const __bf16 h = (__bf16)v;
    hi[off] = h;
    lo[off] = (__bf16)(v - (float)h);
}

constexpr int MAXBLK = 4;        // Blocks of one geometry (same B, H, W, mid, k3) per launch: blockIdx.y picks one
struct MultiArgs {
    BlockArgs b[MAXBLK];
};

// the few integers the helpers need, copied out of the kernel-argument block into registers (handing the by-value argument
// struct itself to the helpers made the compiler spill it to scratch memory: every field read became a private-memory load)
struct Dims {
    int B, H, W, mid, k3;
};

// One stage of the chain.  `rows` band rows starting at image row ys (may be negative / past H: those rows are zero),
// NI images; position p = (img_local * rows + y) * W + x.
struct Stage {
    int rows, ys;                // rows of this stage's band, image row of its first row
};

// ---- shared helpers: a GEMM of M positions x N columns over K, A from global rows (1x1 conv) or from an LDS band ----------
//   acc[mt][nt] 16x16 tiles: wave w owns m-tiles w, w + NW, ...; every wave computes all n-tiles.
constexpr int NW = 8;                   // waves per workgroup (512 threads: two per SIMD hide each other's load latency)
constexpr int NTHR = 64 * NW;
constexpr int MAXMT = MAXM / 16 / NW;   // m-tiles per wave
constexpr int MAXNT = 3;                // n-tiles of the mid-wide stages (mid <= 48)

// B fragment of (k-chunk kc, n-tile nt): lane (col = lane & 15, kg = lane >> 4) reads 8 consecutive k
__device__ __forceinline__ void load_b(const __bf16* w, long long plane, int npad, int kc, int nt, int lane, vb_bf16x8& bh,
                                        vb_bf16x8& bl) {
    const size_t o = ((size_t)kc * npad + 16 * nt + (lane & 15)) * 32 + 8 * (lane >> 4);
    bh = *reinterpret_cast<const vb_bf16x8*>(w + o);
    bl = *reinterpret_cast<const vb_bf16x8*>(w + plane + o);
}

__device__ __forceinline__ f32x4 mma3(const vb_bf16x8& ah, const vb_bf16x8& al, const vb_bf16x8& bh, const vb_bf16x8& bl, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
    return c;
}

// position p of a stage band -> (image, y in image, x) and validity (inside the batch and the image)
struct Pos {
    int img, y, x;
    bool ok;
};
__device__ __forceinline__ Pos decode(const Dims a, int img0, const Stage& s, int p, int M) {
    Pos r;
    const int per = s.rows * a.W;
    const int il = p / per;
    const int rem = p - il * per;
    const int yl = rem / a.W;
    r.x = rem - yl * a.W;
    r.y = s.ys + yl;
    r.img = img0 + il;
    r.ok = p < M && r.img < a.B && r.y >= 0 && r.y < a.H;
    return r;
}

// LDS offset (in bf16 elements) of position (il, yl, x) of a band buffer with `rows` rows: [NI][rows][W + 2][PS], x = -1 and x = W are
// the zero halo columns
__device__ __forceinline__ int band_off(int il, int yl, int x, int rows, int W) { return ((il * rows + yl) * (W + 2) + x + 1) * PS; }

// ---- the weight feeder --------------------------------------------------------------------------------------------------
// Every wave of the workgroup multiplies by the SAME B block in a k-step, so the block (<= 8 KB: [2 planes][<= 64 columns][32 k])
// is brought in ONCE by all 512 threads (16 bytes each) instead of by every wave for itself, through a two-slot LDS ring;
// each thread keeps the pieces of the next PD k-steps in registers (4 VGPRs per k-step), so the L2 round trip (~1 us) of a
// k-step is covered by the PD - 1 steps in front of it.  One workgroup barrier per k-step: slot (ks & 1) is rewritten
// only after every wave has passed the barrier of step ks - 1, i.e. has finished multiplying step ks - 2.
#ifndef PM_VB_PD
#define PM_VB_PD 6
#endif
constexpr int PD = PM_VB_PD;    // k-steps of weights in flight per thread (even: ring slot = step & 1)
#ifndef PM_VB_AD
#define PM_VB_AD 3
#endif
constexpr int AD = PM_VB_AD;    // k-steps of global A rows in flight per wave in the 1x1-from-global stage (divides PD): width
                                // 192 = 6 k-steps, so the whole operand of the stage is requested before the first MFMA
constexpr int BROWP = 40;        // ring row: 32 bf16 + 16 B pad (80-byte pitch: conflict-free 16-byte fragment reads)
constexpr int RING_SLOT = 2 * 64 * BROWP;   // bf16 elements of one slot (2 planes x 64 columns)

struct Feeder {
    const __bf16* w;             // split weights of the stage
    long long plane;
    int npad;                    // columns per k-chunk in the global layout
    int ncols;                   // columns brought in per step (<= 64)
    int pc_plane, pc_n, pc_q;    // this thread's piece: plane, column, 16-byte quarter of the 32-k row (pc_plane < 0: idle)
};

__device__ __forceinline__ Feeder make_feeder(const __bf16* w, long long plane, int npad, int ncols, int tid) {
    Feeder f{w, plane, npad, ncols, -1, 0, 0};
    if (tid < 2 * ncols * 4) {
        f.pc_plane = tid / (ncols * 4);
        f.pc_n = (tid >> 2) % ncols;
        f.pc_q = tid & 3;
    }
    return f;
}
// piece of (k-chunk kc, first column n0) -> registers
// (branch-free: a load under an `if` makes the compiler wait for EVERY outstanding load - s_waitcnt vmcnt(0) - at the join, which
// serialises the whole prefetch ring; threads without a piece re-read piece 0 and never store it)
__device__ __forceinline__ vb_u32x4 feed_load(const Feeder& f, int kc, int n0) {
    const int pl = f.pc_plane >= 0 ? f.pc_plane : 0;
    return *reinterpret_cast<const vb_u32x4*>(f.w + (size_t)pl * f.plane + ((size_t)kc * f.npad + n0 + f.pc_n) * 32 + 8 * f.pc_q);
}
__device__ __forceinline__ void feed_store(const Feeder& f, __bf16* slot, const vb_u32x4& v) {
    if (f.pc_plane >= 0) *reinterpret_cast<vb_u32x4*>(slot + (f.pc_plane * 64 + f.pc_n) * BROWP + 8 * f.pc_q) = v;
}
// B fragment of n-tile nt from a ring slot
__device__ __forceinline__ void ring_b(const __bf16* slot, int nt, int lane, vb_bf16x8& bh, vb_bf16x8& bl) {
    const int o = (16 * nt + (lane & 15)) * BROWP + 8 * (lane >> 4);
    bh = *reinterpret_cast<const vb_bf16x8*>(slot + o);
    bl = *reinterpret_cast<const vb_bf16x8*>(slot + 64 * BROWP + o);
}

// ---- 1x1 stage with A from GLOBAL memory: C[p][n] = sum_k A[p][k] W[k][n],  N <= 48 (mid-wide) ------------------------------
// A rows are [K] floats at ain + ((img*H + y)*W + x) * K; rows outside the image contribute zeros.  A rows are prefetched AD
// k-steps ahead in registers (each wave reads its own rows).
// GELU_IN (forward stage 1): the rows are the Block's raw input x = [ain | ain2] (Ca channels from ain, K - Ca from ain2, Ca a
// multiple of 32 when there are two sources); gelu is applied as the rows arrive and gelu(x) of the rows the band owns
// ([own_lo, own_hi)) is stored to xg_out for the weight gradient - the separate gelu launch in front of every Block is gone.
template <int NT, bool GELU_IN>
__device__ __forceinline__ void gemm_global_mid(const Dims a, const float* __restrict__ ain, const float* __restrict__ ain2,
                                                int Ca, float* __restrict__ xg_out, int own_lo, int own_hi, int K,
                                                const __bf16* w, long long plane, __bf16* ring, int img0, const Stage& s, int M,
                                                int wave, int lane, int tid, f32x4 (&acc)[MAXMT][NT]) {
    const int kch = (K + 31) / 32;
    const int npad = (a.mid + 31) / 32 * 32;               // the split layout pads N to whole 32s
    const int nmt = (M + 15) / 16;
    const float* rowp[MAXMT];
    const float* rowp2[MAXMT];
    float* xgp[MAXMT];
    bool rok[MAXMT];
    const int Cb = K - Ca;
#pragma unroll
    for (int t = 0; t < MAXMT; ++t) {
        const int mt = wave + NW * t;
        const Pos q = decode(a, img0, s, mt * 16 + (lane & 15), M);
        rok[t] = mt < nmt && q.ok;
        const size_t row = rok[t] ? (size_t)(q.img * a.H + q.y) * a.W + q.x : 0;
        rowp[t] = ain + row * (GELU_IN ? Ca : K) + 8 * (lane >> 4);
        rowp2[t] = (GELU_IN && ain2) ? ain2 + row * Cb + 8 * (lane >> 4) : nullptr;
        xgp[t] = (GELU_IN && rok[t] && q.y >= own_lo && q.y < own_hi) ? xg_out + row * K + 8 * (lane >> 4) : nullptr;
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int kg = 8 * (lane >> 4);
    const Feeder fd = make_feeder(w, plane, npad, npad, tid);
    // register rings indexed by compile-time constants (the k-loop is unrolled PD times, every load is unconditional and its
    // index clamped): the body is straight-line code, so the compiler counts the loads in flight exactly (s_waitcnt vmcnt(N))
    // instead of draining them all at a control-flow join
    vb_u32x4 pre[PD];
    f32x4 xa[AD][MAXMT][2];
    auto load_a = [&](int kc, f32x4 (&x)[MAXMT][2]) {
        kc = kc < kch ? kc : kch - 1;
        const bool kin = 32 * kc + kg + 8 <= K;              // K % 8 == 0: a lane's 8 channels are all inside or all outside
#pragma unroll
        for (int t = 0; t < MAXMT; ++t) {
            const float* src = (GELU_IN && 32 * kc >= Ca) ? rowp2[t] + (32 * kc - Ca) : rowp[t] + (kin ? 32 * kc : 0);
            x[t][0] = *reinterpret_cast<const f32x4*>(src);
            x[t][1] = *reinterpret_cast<const f32x4*>(src + 4);
        }
    };
    auto kclamp = [&](int k) { return k < kch ? k : kch - 1; };
#pragma unroll
    for (int j = 0; j < PD; ++j) pre[j] = feed_load(fd, kclamp(j), 0);
#pragma unroll
    for (int j = 0; j < AD; ++j) load_a(j, xa[j]);
    for (int ks0 = 0; ks0 < kch; ks0 += PD) {
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            const int ks = ks0 + j;                          // steps past the end only move (clamped) data around
            __bf16* slot = ring + (j & 1) * RING_SLOT;       // PD is even: ks & 1 == j & 1
            feed_store(fd, slot, pre[j]);
            pre[j] = feed_load(fd, kclamp(ks + PD), 0);
            __syncthreads();
            if (ks < kch) {                                  // uniform
                const bool kin = 32 * ks + kg + 8 <= K;
#pragma unroll
                for (int t = 0; t < MAXMT; ++t) {
                    if (wave + NW * t >= nmt) continue;      // wave-uniform
                    f32x4 v0 = xa[j % AD][t][0], v1 = xa[j % AD][t][1];
                    if (!(rok[t] && kin)) {
                        v0 = f32x4{0.f, 0.f, 0.f, 0.f};
                        v1 = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    if (GELU_IN) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v0[e] = vb_gelu(v0[e]);
                            v1[e] = vb_gelu(v1[e]);
                        }
                        if (xgp[t] && kin) {
                            *reinterpret_cast<f32x4*>(xgp[t] + 32 * ks) = v0;
                            *reinterpret_cast<f32x4*>(xgp[t] + 32 * ks + 4) = v1;
                        }
                    }
                    vb_bf16x8 ah, al;
                    vb_split8(v0, v1, ah, al);
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        vb_bf16x8 bh, bl;
                        ring_b(slot, n, lane, bh, bl);
                        acc[t][n] = mma3(ah, al, bh, bl, acc[t][n]);
                    }
                }
            }
            load_a(ks + AD, xa[j % AD]);                     // refill the set just consumed
        }
    }
    __syncthreads();                                        // the ring is free for the next stage
}

// ---- k3 x k3 stage with A from an LDS band: C[p][n] = sum_{tap, c} band[p + tap][c] W[tap][c][n] --------------------------------
// `sign` = +1 forward (source = p + (tap - centre)), -1 data gradient (source = p - (tap - centre)).
// DENSE (a run-time flag, uniform over the launch: one copy of the loop in the kernel): the weights are a dense_k split copy - k-step ks holds k = 32 ks .. 32 ks + 31 of the (tap, c) range, so the four
// 8-channel groups of a k-step (lane >> 4) may sit in two different taps: the tap shift is per lane group, not per k-step
// (mid = 48: 14 k-steps per 3x3 layer instead of 18, none of them half empty except the last).
template <int NT>
__device__ __forceinline__ void gemm_band_mid(const Dims a, const __bf16* bh_, const __bf16* bl_, int in_rows, int in_ys,
                                              const __bf16* w, long long plane, __bf16* ring, int img0, const Stage& s, int M,
                                              int sign, const bool DENSE, int wave, int lane, int tid, f32x4 (&acc)[MAXMT][NT]) {
    const int npad = (a.mid + 31) / 32 * 32, cch = (a.mid + 31) / 32;   // 32-channel chunks per tap (zero-padded past mid)
    const int nmt = (M + 15) / 16;
    const int taps = a.k3 * a.k3, ctr = a.k3 / 2;
    int base[MAXMT];
#pragma unroll
    for (int t = 0; t < MAXMT; ++t) {
        const int mt = wave + NW * t;
        int p = mt * 16 + (lane & 15);
        p = p < M ? p : M - 1;                               // padded slots recompute the last position (never stored)
        const int per = s.rows * a.W;
        const int il = p / per, rem = p - il * per;
        const int yl = rem / a.W, x = rem - yl * a.W;
        // row of the input band that holds image row (s.ys + yl): yl + (s.ys - in_ys)
        base[t] = band_off(il, yl + (s.ys - in_ys), x, in_rows, a.W) + (DENSE ? 0 : 8 * (lane >> 4));
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int nk = DENSE ? (taps * a.mid + 31) / 32 : taps * cch;
    const Feeder fd = make_feeder(w, plane, npad, npad, tid);
    vb_u32x4 pre[PD];
    auto kclamp = [&](int k) { return k < nk ? k : nk - 1; };
    // DENSE: this lane group's walk over the (tap, c) range, 32 channels per k-step (mid % 8 == 0: a group of 8 never straddles taps)
    int dk = 8 * (lane >> 4), dtap = 0;
    if (DENSE)
        while (dk >= a.mid) { dk -= a.mid; ++dtap; }
#pragma unroll
    for (int j = 0; j < PD; ++j) pre[j] = feed_load(fd, kclamp(j), 0);
    for (int ks0 = 0; ks0 < nk; ks0 += PD) {
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            const int ks = ks0 + j;
            __bf16* slot = ring + (j & 1) * RING_SLOT;
            feed_store(fd, slot, pre[j]);
            pre[j] = feed_load(fd, kclamp(ks + PD), 0);
            __syncthreads();
            if (ks < nk) {                                   // uniform
                int shift;
                if (DENSE) {
                    const int tp = dtap < taps ? dtap : ctr * a.k3 + ctr;        // past the end of K: zero weights, any finite A
                    const int ky = tp / a.k3, kx = tp - ky * a.k3;
                    shift = sign * ((ky - ctr) * (a.W + 2) + (kx - ctr)) * PS + (dtap < taps ? dk : 0);
                    dk += 32;                                                      // the next k-step of this lane group
                    while (dk >= a.mid) { dk -= a.mid; ++dtap; }
                } else {
                    const int tap = ks / cch, cc = ks - tap * cch;
                    const int ky = tap / a.k3, kx = tap - ky * a.k3;
                    shift = sign * ((ky - ctr) * (a.W + 2) + (kx - ctr)) * PS + 32 * cc;
                }
#pragma unroll
                for (int t = 0; t < MAXMT; ++t) {
                    if (wave + NW * t >= nmt) continue;
                    const vb_bf16x8 ah = *reinterpret_cast<const vb_bf16x8*>(bh_ + base[t] + shift);
                    const vb_bf16x8 al = *reinterpret_cast<const vb_bf16x8*>(bl_ + base[t] + shift);
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        vb_bf16x8 bh, bl;
                        ring_b(slot, n, lane, bh, bl);
                        acc[t][n] = mma3(ah, al, bh, bl, acc[t][n]);
                    }
                }
            }
        }
    }
    __syncthreads();
}

// ---- wide 1x1 stage with A from an LDS band (mid channels) and N = Nout columns, 32 columns (two n-tiles) per step: the last layer --------
// epi(t, e, n, value) is called for every element: row (wave + NW t)*16 + 4*(lane >> 4) + e, column n
// pre_fn(t, e, n, a, b) loads what the element's epilogue needs (bias / residual / gelu' argument) BEFORE the step's MFMAs are
// issued; comb(value, a, b) folds them in once the MFMAs are done; fin(t, e, n, result) only stores.  Loads and stores interleaved element by element make every load wait
// for the stores in front of it (vmcnt counts both): 16 round trips per step instead of one.
template <typename Pre, typename Comb, typename Fin>
__device__ __forceinline__ void gemm_band_wide(const Dims a, const __bf16* bh_, const __bf16* bl_, int in_rows, int in_ys,
                                               const __bf16* w, long long plane, __bf16* ring, int Nout, int img0, const Stage& s,
                                               int M, int wave, int lane, int tid, Pre pre_fn, Comb comb, Fin fin) {
    const int npad = (Nout + 31) / 32 * 32, cch = (a.mid + 31) / 32;
    const int nmt = (M + 15) / 16, nsteps = npad / 32;       // 32 columns per step
    int base[MAXMT];
#pragma unroll
    for (int t = 0; t < MAXMT; ++t) {
        const int mt = wave + NW * t;
        int p = mt * 16 + (lane & 15);
        p = p < M ? p : M - 1;
        const int per = s.rows * a.W;
        const int il = p / per, rem = p - il * per;
        const int yl = rem / a.W, x = rem - yl * a.W;
        base[t] = band_off(il, yl + (s.ys - in_ys), x, in_rows, a.W) + 8 * (lane >> 4);
    }
    vb_bf16x8 ah[MAXMT][2], al[MAXMT][2];                    // the A fragments (K <= 64) are loaded once and reused for every column
#pragma unroll
    for (int t = 0; t < MAXMT; ++t)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            ah[t][cc] = *reinterpret_cast<const vb_bf16x8*>(bh_ + base[t] + 32 * cc);
            al[t][cc] = *reinterpret_cast<const vb_bf16x8*>(bl_ + base[t] + 32 * cc);
        }
    // a step brings in [2 planes][cch chunks][32 columns][32 k]: the ring slot is used as [plane][chunk * 32 + column]
    // (the slot has 64 rows per plane: cch <= 2 chunks of 32 columns)
    Feeder fd{w, plane, npad, 32, -1, 0, 0};
    int pc_chunk = 0;
    if (tid < 2 * cch * 32 * 4) {
        fd.pc_plane = tid / (cch * 32 * 4);
        pc_chunk = (tid / (32 * 4)) % cch;
        fd.pc_n = (tid >> 2) & 31;
        fd.pc_q = tid & 3;
    }
    const int wpl = fd.pc_plane >= 0 ? fd.pc_plane : 0;
    auto wload = [&](int st) {
        st = st < nsteps ? st : nsteps - 1;
        return *reinterpret_cast<const vb_u32x4*>(w + (size_t)wpl * plane + ((size_t)pc_chunk * npad + 32 * st + fd.pc_n) * 32 + 8 * fd.pc_q);
    };
    vb_u32x4 pre[PD];
#pragma unroll
    for (int j = 0; j < PD; ++j) pre[j] = wload(j);
    for (int st0 = 0; st0 < nsteps; st0 += PD) {
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            const int st = st0 + j;
            __bf16* slot = ring + (j & 1) * RING_SLOT;
            if (fd.pc_plane >= 0)
                *reinterpret_cast<vb_u32x4*>(slot + (fd.pc_plane * 64 + pc_chunk * 32 + fd.pc_n) * BROWP + 8 * fd.pc_q) = pre[j];
            pre[j] = wload(st + PD);
            __syncthreads();
            if (st < nsteps) {                               // uniform
                float ea[2][MAXMT][4], eb[2][MAXMT][4];
#pragma unroll
                for (int half = 0; half < 2; ++half)
#pragma unroll
                    for (int t = 0; t < MAXMT; ++t)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            ea[half][t][e] = 0.f;
                            eb[half][t][e] = 0.f;
                            if (wave + NW * t < nmt) pre_fn(t, e, 16 * (2 * st + half) + (lane & 15), ea[half][t][e], eb[half][t][e]);
                        }
                f32x4 cacc[2][MAXMT];
#pragma unroll
                for (int half = 0; half < 2; ++half)         // the step's two n-tiles
#pragma unroll
                    for (int t = 0; t < MAXMT; ++t) {
                        f32x4 c = {0.f, 0.f, 0.f, 0.f};
                        if (16 * (2 * st + half) < Nout && wave + NW * t < nmt) {
#pragma unroll
                            for (int cc = 0; cc < 2; ++cc) {
                                if (cc >= cch) continue;
                                const int o = (cc * 32 + 16 * half + (lane & 15)) * BROWP + 8 * (lane >> 4);
                                const vb_bf16x8 bh = *reinterpret_cast<const vb_bf16x8*>(slot + o);
                                const vb_bf16x8 bl = *reinterpret_cast<const vb_bf16x8*>(slot + 64 * BROWP + o);
                                c = mma3(ah[t][cc], al[t][cc], bh, bl, c);
                            }
                        }
                        cacc[half][t] = c;
                    }
#pragma unroll
                for (int half = 0; half < 2; ++half)
#pragma unroll
                    for (int t = 0; t < MAXMT; ++t)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            cacc[half][t][e] = vb_pin(comb(cacc[half][t][e], ea[half][t][e], eb[half][t][e]));
                        }
#pragma unroll
                for (int half = 0; half < 2; ++half)
#pragma unroll
                    for (int t = 0; t < MAXMT; ++t)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (wave + NW * t < nmt) fin(t, e, 16 * (2 * st + half) + (lane & 15), cacc[half][t][e]);
            }
        }
    }
    __syncthreads();
}

// zero the band buffers (halo columns, rows outside the image and the padding channels stay zero afterwards)
__device__ __forceinline__ void zero_lds(__bf16* base, int elems, int tid) {
    vb_u32x4 z = {0u, 0u, 0u, 0u};
    for (int e = tid * 8; e < elems; e += NTHR * 8) *reinterpret_cast<vb_u32x4*>(base + e) = z;
}

// output row offsets ((img*H + y)*W + x, or -1) of the positions a lane's accumulator rows hold
__device__ __forceinline__ void row_offsets(const Dims a, int img0, const Stage& s, int M, int wave, int lane,
                                            int (&ro)[MAXMT][4], int (&bo)[MAXMT][4], int (&yy)[MAXMT][4]) {
    const int per = s.rows * a.W;
#pragma unroll
    for (int t = 0; t < MAXMT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int p = (wave + NW * t) * 16 + 4 * (lane >> 4) + e;
            const Pos q = decode(a, img0, s, p, M);
            ro[t][e] = q.ok ? (q.img * a.H + q.y) * a.W + q.x : -1;
            yy[t][e] = q.y;
            const int il = p / per, yl = (p - il * per) / a.W;
            bo[t][e] = band_off(il, yl, q.x, s.rows, a.W);
        }
}

// =================================================== forward ===================================================================
__global__ __launch_bounds__(NTHR) void vdvae_block_fwd_kernel(MultiArgs multi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const BlockArgs& ka = multi.b[blockIdx.y];               // independent Blocks of one geometry share a launch
    const Dims a{ka.B, ka.H, ka.W, ka.mid, ka.k3};
    const int aR = ka.R, aNI = ka.NI, abands = ka.bands, aCin = ka.Cin, aCout = ka.Cout;
    const bool dense = ka.dense != 0 && ka.k3 == 3;
    const float* __restrict__ xin = ka.xin;
    const float* __restrict__ resp = ka.res;
    [[maybe_unused]] const float* __restrict__ xpre = ka.xpre;
    float* __restrict__ outp = ka.out;
    float* const hh0 = ka.hh[0]; float* const hh1 = ka.hh[1]; float* const hh2 = ka.hh[2];
    float* const gg0 = ka.gg[0]; float* const gg1 = ka.gg[1]; float* const gg2 = ka.gg[2];
    const __bf16* const w0 = ka.w[0]; const __bf16* const w1 = ka.w[1]; const __bf16* const w2 = ka.w[2]; const __bf16* const w3 = ka.w[3];
    const long long pl0 = ka.plane[0], pl1 = ka.plane[1], pl2 = ka.plane[2], pl3 = ka.plane[3];
    [[maybe_unused]] const float* const bs0 = ka.bias[0]; [[maybe_unused]] const float* const bs1 = ka.bias[1];
    [[maybe_unused]] const float* const bs2 = ka.bias[2]; [[maybe_unused]] const float* const bs3 = ka.bias[3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int halo = a.k3 / 2;                                // 1 for 3x3 middle convs, 0 for 1x1
    const int band = blockIdx.x % abands, img0 = (blockIdx.x / abands) * aNI;
    const int y0 = band * aR;
    const Stage s1{aR + 4 * halo, y0 - 2 * halo}, s2{aR + 2 * halo, y0 - halo}, s3{aR, y0};
    const int M1 = aNI * s1.rows * a.W, M2 = aNI * s2.rows * a.W, M3 = aNI * s3.rows * a.W;
    const int e1 = aNI * s1.rows * (a.W + 2) * PS, e2 = aNI * s2.rows * (a.W + 2) * PS;
    // LDS: band 1 (g1, later g3) | band 2 (g2); each hi then lo; one spare position in front for the padded-slot address
    __bf16* b1h = reinterpret_cast<__bf16*>(smem_raw) + PS;
    __bf16* b1l = b1h + e1 + PS;
    __bf16* b2h = b1l + e1 + PS;
    __bf16* b2l = b2h + e2 + PS;
    __bf16* ring = b2l + e2;                                  // two slots of the weight feeder
    zero_lds(reinterpret_cast<__bf16*>(smem_raw), 2 * (e1 + PS) + 2 * (e2 + PS), tid);
    __syncthreads();

    f32x4 acc[MAXMT][MAXNT];
    // ---- stage 1: h1 = c1(xg) on the rows of s1 -> g1 band ----
    const int own_hi = y0 + aR < a.H ? y0 + aR : a.H;
    if (ka.gelu_in)
        gemm_global_mid<MAXNT, true>(a, xin, ka.xin2, ka.Ca, ka.xg_out, y0, own_hi, aCin, w0, pl0, ring, img0, s1, M1, wave, lane,
                                     tid, acc);
    else
        gemm_global_mid<MAXNT, false>(a, xin, nullptr, aCin, nullptr, 0, 0, aCin, w0, pl0, ring, img0, s1, M1, wave, lane, tid, acc);
    int ro[MAXMT][4], bo[MAXMT][4], yy[MAXMT][4];
    // (pointers are passed explicitly: indexing the kernel-argument arrays with a runtime layer number would force the whole
    // argument block into scratch memory)
    auto epilogue_mid = [&](const Stage& s, int M, const float* __restrict__ biasp, float* __restrict__ hdst,
                            float* __restrict__ gdst, __bf16* oh, __bf16* ol, int own_lo, int own_hi) {
        const int nmt = (M + 15) / 16;
        row_offsets(a, img0, s, M, wave, lane, ro, bo, yy);
        // pass 1: the only loads (three bias values) are consumed before any store is issued - a load behind a store waits
        // for that store too (vmcnt counts both), once per element in the interleaved form
#pragma unroll
        for (int n = 0; n < MAXNT; ++n) {
            const int col = 16 * n + (lane & 15);
            const float bv = col < a.mid ? biasp[col] : 0.f;
#pragma unroll
            for (int t = 0; t < MAXMT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[t][n][e] = vb_pin(acc[t][n][e] + bv);
                }
        }
        // pass 2: stores only
#pragma unroll
        for (int n = 0; n < MAXNT; ++n) {
            const int col = 16 * n + (lane & 15);
            if (col >= a.mid) continue;
#pragma unroll
            for (int t = 0; t < MAXMT; ++t) {
                if (wave + NW * t >= nmt) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (ro[t][e] < 0) continue;                 // outside the image: the band keeps its zeros (SAME padding)
                    const float v = acc[t][n][e];
                    const float gv = vb_gelu(v);
                    vb_store_split(oh, ol, bo[t][e] + col, gv);
                    if (yy[t][e] >= own_lo && yy[t][e] < own_hi) {   // rows this band owns: stored for the backward pass
                        const size_t o = (size_t)ro[t][e] * a.mid + col;
                        hdst[o] = v;
                        gdst[o] = gv;
                    }
                }
            }
        }
    };
    epilogue_mid(s1, M1, bs0, hh0, gg0, b1h, b1l, y0, own_hi);
    __syncthreads();
    // ---- stage 2: h2 = c2(g1) on the rows of s2 -> g2 band ----
    gemm_band_mid<MAXNT>(a, b1h, b1l, s1.rows, s1.ys, w1, pl1, ring, img0, s2, M2, +1, dense, wave, lane, tid, acc);
    epilogue_mid(s2, M2, bs1, hh1, gg1, b2h, b2l, y0, own_hi);
    __syncthreads();
    // ---- stage 3: h3 = c3(g2) on the owned rows -> g3 into band 1 (g1 is dead) ----
    gemm_band_mid<MAXNT>(a, b2h, b2l, s2.rows, s2.ys, w2, pl2, ring, img0, s3, M3, +1, dense, wave, lane, tid, acc);
    __syncthreads();                                          // every wave is done reading band 1's successor inputs (band 2 only)
    {   // g3 goes into band 1 laid out with s3's rows; stale g1 there is overwritten or unread (c4 is 1x1: no halo, and
        // the padding channels mid..63 of every slot were zeroed once and are never written)
        epilogue_mid(s3, M3, bs2, hh2, gg2, b1h, b1l, y0, own_hi);
    }
    __syncthreads();
    // ---- stage 4: out = c4(g3) + bias + res ----
    row_offsets(a, img0, s3, M3, wave, lane, ro, bo, yy);
    gemm_band_wide(a, b1h, b1l, s3.rows, s3.ys, w3, pl3, ring, aCout, img0, s3, M3, wave, lane, tid,
                   [&](int t, int e, int n, float& ea, float& eb) {
                       if (ro[t][e] < 0 || n >= aCout) return;
                       ea = bs3[n];
                       if (resp) eb = resp[(size_t)ro[t][e] * aCout + n];
                   },
                   [&](float v, float ea, float eb) { return v + ea + eb; },
                   [&](int t, int e, int n, float v) {
                       if (ro[t][e] < 0 || n >= aCout) return;
                       outp[(size_t)ro[t][e] * aCout + n] = v;
                   });
}

// =================================================== backward ==================================================================
// stage T1: dg3 = dout W4^T on s1 rows (K = Cout from global), dh3 = dg3 * gelu'(h3) -> band 1
// stage T2: dg2 = c3^T(dh3) on s2 rows, dh2 = dg2 * gelu'(h2) -> band 2
// stage T3: dg1 = c2^T(dh2) on s3 rows, dh1 = dg1 * gelu'(h1) -> band 1
// stage T4: dxg = dh1 W1^T (N = Cin) [* gelu'(xpre)] [+ res]
__global__ __launch_bounds__(NTHR) void vdvae_block_bwd_kernel(MultiArgs multi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const BlockArgs& ka = multi.b[blockIdx.y];               // independent Blocks of one geometry share a launch
    const Dims a{ka.B, ka.H, ka.W, ka.mid, ka.k3};
    const int aR = ka.R, aNI = ka.NI, abands = ka.bands, aCin = ka.Cin, aCout = ka.Cout;
    const bool dense = ka.dense != 0 && ka.k3 == 3;
    const float* __restrict__ xin = ka.xin;
    const float* __restrict__ resp = ka.res;
    [[maybe_unused]] const float* __restrict__ xpre = ka.xpre;
    float* __restrict__ outp = ka.out;
    float* const hh0 = ka.hh[0]; float* const hh1 = ka.hh[1]; float* const hh2 = ka.hh[2];
    float* const gg0 = ka.gg[0]; float* const gg1 = ka.gg[1]; float* const gg2 = ka.gg[2];
    const __bf16* const w0 = ka.w[0]; const __bf16* const w1 = ka.w[1]; const __bf16* const w2 = ka.w[2]; const __bf16* const w3 = ka.w[3];
    const long long pl0 = ka.plane[0], pl1 = ka.plane[1], pl2 = ka.plane[2], pl3 = ka.plane[3];
    [[maybe_unused]] const float* const bs0 = ka.bias[0]; [[maybe_unused]] const float* const bs1 = ka.bias[1];
    [[maybe_unused]] const float* const bs2 = ka.bias[2]; [[maybe_unused]] const float* const bs3 = ka.bias[3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int halo = a.k3 / 2;
    const int band = blockIdx.x % abands, img0 = (blockIdx.x / abands) * aNI;
    const int y0 = band * aR;
    const Stage s1{aR + 4 * halo, y0 - 2 * halo}, s2{aR + 2 * halo, y0 - halo}, s3{aR, y0};
    const int M1 = aNI * s1.rows * a.W, M2 = aNI * s2.rows * a.W, M3 = aNI * s3.rows * a.W;
    const int e1 = aNI * s1.rows * (a.W + 2) * PS, e2 = aNI * s2.rows * (a.W + 2) * PS;
    __bf16* b1h = reinterpret_cast<__bf16*>(smem_raw) + PS;
    __bf16* b1l = b1h + e1 + PS;
    __bf16* b2h = b1l + e1 + PS;
    __bf16* b2l = b2h + e2 + PS;
    __bf16* ring = b2l + e2;                                  // two slots of the weight feeder
    zero_lds(reinterpret_cast<__bf16*>(smem_raw), 2 * (e1 + PS) + 2 * (e2 + PS), tid);
    __syncthreads();

    f32x4 acc[MAXMT][MAXNT];
    const int own_hi = y0 + aR < a.H ? y0 + aR : a.H;
    int ro[MAXMT][4], bo[MAXMT][4], yy[MAXMT][4];
    auto epilogue_mid = [&](const Stage& s, int M, const float* __restrict__ hsrc, float* __restrict__ dhdst, __bf16* oh,
                            __bf16* ol) {
        const int nmt = (M + 15) / 16;
        row_offsets(a, img0, s, M, wave, lane, ro, bo, yy);
        // pass 1: every h the gelu' factors need, loaded back to back and multiplied in before the first store
        float hv[MAXMT][MAXNT][4];
#pragma unroll
        for (int n = 0; n < MAXNT; ++n) {
            const int col = 16 * n + (lane & 15);
#pragma unroll
            for (int t = 0; t < MAXMT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {        // unconditional (clamped) loads: 24 in flight, ONE wait
                    const bool ok = col < a.mid && ro[t][e] >= 0;
                    hv[t][n][e] = hsrc[ok ? (size_t)ro[t][e] * a.mid + col : 0];
                }
        }
#pragma unroll
        for (int n = 0; n < MAXNT; ++n)
#pragma unroll
            for (int t = 0; t < MAXMT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][n][e] = vb_pin(acc[t][n][e] * vb_gelu_d(hv[t][n][e]));
#pragma unroll
        for (int n = 0; n < MAXNT; ++n) {
            const int col = 16 * n + (lane & 15);
            if (col >= a.mid) continue;
#pragma unroll
            for (int t = 0; t < MAXMT; ++t) {
                if (wave + NW * t >= nmt) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (ro[t][e] < 0) continue;
                    const size_t o = (size_t)ro[t][e] * a.mid + col;
                    const float v = acc[t][n][e];
                    vb_store_split(oh, ol, bo[t][e] + col, v);
                    if (yy[t][e] >= y0 && yy[t][e] < own_hi) dhdst[o] = v;
                }
            }
        }
    };
    gemm_global_mid<MAXNT, false>(a, xin, nullptr, aCout, nullptr, 0, 0, aCout, w3, pl3, ring, img0, s1, M1, wave, lane, tid, acc);
    epilogue_mid(s1, M1, hh2, gg2, b1h, b1l);
    __syncthreads();
    gemm_band_mid<MAXNT>(a, b1h, b1l, s1.rows, s1.ys, w2, pl2, ring, img0, s2, M2, -1, dense, wave, lane, tid, acc);
    epilogue_mid(s2, M2, hh1, gg1, b2h, b2l);
    __syncthreads();
    gemm_band_mid<MAXNT>(a, b2h, b2l, s2.rows, s2.ys, w1, pl1, ring, img0, s3, M3, -1, dense, wave, lane, tid, acc);
    __syncthreads();
    epilogue_mid(s3, M3, hh0, gg0, b1h, b1l);
    __syncthreads();
    row_offsets(a, img0, s3, M3, wave, lane, ro, bo, yy);
    gemm_band_wide(a, b1h, b1l, s3.rows, s3.ys, w0, pl0, ring, aCin, img0, s3, M3, wave, lane, tid,
                   [&](int t, int e, int n, float& ea, float& eb) {
                       if (ro[t][e] < 0 || n >= aCin) return;
                       const size_t o = (size_t)ro[t][e] * aCin + n;
                       if (xpre) ea = xpre[o];
                       if (resp) eb = resp[o];
                   },
                   [&](float v, float ea, float eb) { return (xpre ? v * vb_gelu_d(ea) : v) + eb; },
                   [&](int t, int e, int n, float v) {
                       if (ro[t][e] < 0 || n >= aCin) return;
                       outp[(size_t)ro[t][e] * aCin + n] = v;
                   });
}

// band plan: rows per band and images per workgroup so that no stage owns more than MAXM positions and the LDS fits
struct BandPlan { int R, NI, bands; size_t lds; };
bool plan_block(int B, int H, int W, int k3, int nblk, BandPlan& bp) {
    const int halo = k3 / 2;
    if (W > 62 || H < 1) return false;
    int R = H, NI = 1;
    while (R > 1 && (R + 4 * halo) * W > MAXM) R = (R + 1) / 2;
    if ((R + 4 * halo) * W > MAXM) return false;
    // The chain inside a workgroup is serial and a launch is one link of a dependent chain: while the whole launch (all its
    // Blocks) still fits one round of one workgroup per CU, halve the bands - a workgroup then walks fewer rows (28x28 at
    // per-GPU 8: 4-row bands = 56 workgroups per Block, 2-row bands = 112; the halo rows it recomputes run on CUs that were
    // idle).  Whole small images (resolution <= 3) are packed several per workgroup only when the grid stays >= 128 workgroups.
    static const int round_wgs = getenv("PM_VB_ROUND") ? atoi(getenv("PM_VB_ROUND")) : 256;      // A/B knob
    while (halo > 0 && R > 2) {
        const int R2 = (R + 1) / 2;
        if ((long long)B * ((H + R2 - 1) / R2) * nblk > round_wgs) break;
        R = R2;
    }
    if (R == H)
        while (NI < B && (NI + 1) * (H + 4 * halo) * W <= MAXM && (B + NI) / (NI + 1) >= 128) ++NI;
    bp.R = R; bp.NI = NI; bp.bands = (H + R - 1) / R;
    const size_t e1 = (size_t)NI * (R + 4 * halo) * (W + 2) * PS, e2 = (size_t)NI * (R + 2 * halo) * (W + 2) * PS;
    bp.lds = (2 * (e1 + PS) + 2 * (e2 + PS) + 2 * RING_SLOT) * sizeof(__bf16);
    return bp.lds <= 158 * 1024;
}

int launch_blocks(hipStream_t stream, bool backward, MultiArgs& m, int n) {
    BandPlan bp;
    const BlockArgs& a0 = m.b[0];
    if (n < 1 || n > MAXBLK || !plan_block(a0.B, a0.H, a0.W, a0.k3, n, bp)) return PM_EINVAL;
    for (int i = 0; i < n; ++i) {
        if (m.b[i].B != a0.B || m.b[i].H != a0.H || m.b[i].W != a0.W || m.b[i].mid != a0.mid || m.b[i].k3 != a0.k3) return PM_EINVAL;
        m.b[i].R = bp.R; m.b[i].NI = bp.NI; m.b[i].bands = bp.bands;
    }
    const unsigned grid = (unsigned)(((a0.B + bp.NI - 1) / bp.NI) * bp.bands);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vdvae_block_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vdvae_block_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
        attr_set = true;
    }
    if (backward) hipLaunchKernelGGL(vdvae_block_bwd_kernel, dim3(grid, n), dim3(NTHR), bp.lds, stream, m);
    else hipLaunchKernelGGL(vdvae_block_fwd_kernel, dim3(grid, n), dim3(NTHR), bp.lds, stream, m);
    return pm_check_launch(backward ? "pm_vdvae_blocks_bwd" : "pm_vdvae_blocks_fwd");
}

bool block_shape_ok(int B, int H, int W, int Cin, int Cout, int mid, int k3) {
    return B > 0 && H > 0 && W > 0 && Cin % 8 == 0 && Cout % 8 == 0 && mid % 8 == 0 && mid <= 48 && (k3 == 1 || k3 == 3) &&
           (long long)B * H * W * (Cin > Cout ? Cin : Cout) < (1LL << 31);
}

}  // namespace

static bool fill_fwd(BlockArgs& a, const pm_vdvae_block_io& io, int B, int H, int W, int mid, int k3) {
    if (!io.x || !io.out || !block_shape_ok(B, H, W, io.Cin, io.Cout, mid, k3)) return false;
    // xg_out != NULL: x (and x2) are the RAW inputs, Ca channels from x and Cin - Ca from x2; gelu is applied on load
    if (io.xg_out && (io.Ca <= 0 || io.Ca > io.Cin || io.Ca % 8 != 0 || (io.Ca < io.Cin && (!io.x2 || io.Ca % 32 != 0)))) return false;
    a.xin = io.x; a.res = io.res; a.xpre = nullptr; a.out = io.out;
    a.xin2 = io.xg_out && io.Ca < io.Cin ? io.x2 : nullptr; a.xg_out = io.xg_out; a.Ca = io.xg_out ? io.Ca : io.Cin;
    a.gelu_in = io.xg_out ? 1 : 0;
    for (int i = 0; i < 3; ++i) { a.hh[i] = io.h[i]; a.gg[i] = io.g[i]; if (!io.h[i] || !io.g[i]) return false; }
    for (int i = 0; i < 4; ++i) {
        a.w[i] = reinterpret_cast<const __bf16*>(io.w[i]); a.plane[i] = io.plane[i]; a.bias[i] = io.bias[i];
        if (!io.w[i] || !io.bias[i]) return false;
    }
    a.B = B; a.H = H; a.W = W; a.Cin = io.Cin; a.Cout = io.Cout; a.mid = mid; a.k3 = k3;
    a.dense = io.dense_k3;
    return true;
}

static bool fill_bwd(BlockArgs& a, const pm_vdvae_block_io& io, int B, int H, int W, int mid, int k3) {
    if (!io.x || !io.out || !block_shape_ok(B, H, W, io.Cin, io.Cout, mid, k3)) return false;
    a.xin = io.x; a.res = io.res; a.xpre = io.xpre; a.out = io.out;       // x = dout, out = dxg
    a.xin2 = nullptr; a.xg_out = nullptr; a.Ca = io.Cout; a.gelu_in = 0;
    for (int i = 0; i < 3; ++i) { a.hh[i] = io.h[i]; a.gg[i] = io.g[i]; if (!io.h[i] || !io.g[i]) return false; }
    for (int i = 0; i < 4; ++i) {
        a.w[i] = reinterpret_cast<const __bf16*>(io.w[i]); a.plane[i] = io.plane[i]; a.bias[i] = nullptr;
        if (!io.w[i]) return false;
    }
    a.B = B; a.H = H; a.W = W; a.Cin = io.Cin; a.Cout = io.Cout; a.mid = mid; a.k3 = k3;
    a.dense = io.dense_k3;
    return true;
}

extern "C" int pm_vdvae_blocks_fwd(pm_stream_t stream, const pm_vdvae_block_io* blocks, int nblocks, int B, int H, int W, int mid,
                                   int k3) {
    if (!blocks || nblocks < 1 || nblocks > MAXBLK) return PM_EINVAL;
    MultiArgs m;
    for (int i = 0; i < nblocks; ++i)
        if (!fill_fwd(m.b[i], blocks[i], B, H, W, mid, k3)) return PM_EINVAL;
    for (int i = nblocks; i < MAXBLK; ++i) m.b[i] = m.b[0];
    PM_KTAG("vdvae_block_fwd_kernel");
    return launch_blocks((hipStream_t)stream, false, m, nblocks);
}

extern "C" int pm_vdvae_blocks_bwd(pm_stream_t stream, const pm_vdvae_block_io* blocks, int nblocks, int B, int H, int W, int mid,
                                   int k3) {
    if (!blocks || nblocks < 1 || nblocks > MAXBLK) return PM_EINVAL;
    MultiArgs m;
    for (int i = 0; i < nblocks; ++i)
        if (!fill_bwd(m.b[i], blocks[i], B, H, W, mid, k3)) return PM_EINVAL;
    for (int i = nblocks; i < MAXBLK; ++i) m.b[i] = m.b[0];
    PM_KTAG("vdvae_block_bwd_kernel");
    return launch_blocks((hipStream_t)stream, true, m, nblocks);
}
