// DCNv2 backward, data gradients (grad_x, grad_offset, grad_mask) for gfx950 -- register-resident
// grad-column kernel.
//
// Replaces upstream's dcn_v2_backward data path (columns in HBM + cuBLAS + atomic col2im) behind
// `from .DCNv2.dcn_v2 import DCN` (reference: src/lib/models/networks/pose_dla_dcn.py:16,354).
//
// One workgroup = a TILE of TH rows x 16 pixels of one image (one wave per row) x a slice of the
// input channels, walked in chunks of 16 channels.  Per chunk and wave:
//
//   gcol^T[c][px] (tap t) = sum_co W[co][c][t] * go[co][px]         v_mfma_f32_16x16x4_f32
//
// is computed TRANSPOSED: the weights are the A operand (rows = the chunk's 16 channels), the
// wave's 16 grad_out pixels the B operand (held in registers for the whole kernel), one
// accumulator per tap.  The result lands with the PIXEL on the lane (lane & 15) and four
// CHANNELS in the accumulator registers (4 * (lane >> 4) + reg): exactly the layout the
// consumption needs, so the grad columns never touch LDS or HBM, all 9 x 16 columns of the
// m-tiles are useful (the previous kernel padded 36 columns to 48) and the matrix operands cost
// 36 ds_read_b128 per 144 MFMAs (weights pre-permuted into fragment order by a prologue kernel).
//
// Consumption per (pixel, tap, channel): 4 corner reads from an LDS-staged input region
// ((TH+6) x 22 cells per channel, coalesced row loads), the bilinear algebra for grad_mask /
// grad_offset (register sums over the lane's channels, shuffled together once per tile), and 4
// grad_x contributions accumulated in a matching LDS region in 64-bit fixed point.
//   * fixed point (round 4: 32-bit): contribution -> (int)rint(gcol * m * w * scale), added with ds_add_u32 (LDS float
//     atomics run 25x slower on gfx950, tools/micro/lds_atomic_rate.hip; round 2-3 used 64-bit cells with a scale from
//     an L1 bound of grad_out -- the 8-byte atomics took twice the LDS passes of a dword atomic and 52 % of their
//     cycles were bank conflicts).  `scale` is a power of two chosen per tile AND chunk from the chunk's ACTUAL largest
//     |gcol * m| (a wave / workgroup max over the accumulators right after the matrix phase): the largest contribution
//     maps to < 2^20, a cell receives at most TH*16*9 <= 1728 < 2^11 adds, so the two's-complement sum cannot wrap and
//     is exact whatever the number and order of adds; one add is off by at most 2^-21 of the chunk's largest column
//     value.  Order-independent, i.e. run-to-run deterministic.  With the exact-f32 arithmetic (flag
//     CP_DCN_BWD_EXACT_F32, template BF = false) the cells stay 64 bits wide as in rounds 2-3: contribution ->
//     fma((double)gcol*m, (double)w*scale, 2^52+2^51), raw bits added with ds_add_u64 -- every add carries 0x4338<<48 in
//     its top 16 bits and the rounded integer in two's complement below, the largest column value maps below 2^35: one
//     add is off by at most 2^-36 of it, finer than the fp32 rounding of the product itself.
//   * flush: after each chunk the region sums are written with PLAIN coalesced stores to the
//     tile's slab in the caller's workspace; a second kernel adds, for every grad_x element, the
//     (at most four) slabs whose region covers it, in a fixed order.  No global float atomics on
//     the hot path: the round-1 kernel issued 2.0e7 64-B atomic requests (1.3 GB) per launch at
//     64->64 @256x512 x4 and ran at the chip's atomic rate (profiles/r02_dcn_bwd_pmc_baseline_*).
// Taps whose corners leave the region (|offset| > 2 px) take a cold path with global gathers and
// float atomics straight into grad_x (which is why grad_x stays "accumulated into").
//
// Small-spatial layers split the input channels over workgroups (grid z); their partial
// grad_offset / grad_mask sums meet in a reduce kernel.
#include "cp_common.h"

#include <stdlib.h>
#include <type_traits>

namespace {

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

constexpr int TAPS = 9;
constexpr int TW = 16;                     // tile width: one MFMA n-tile of pixels per wave
constexpr int KC = 16;                     // channels per chunk: one MFMA m-tile
constexpr int HALO = 3;                    // region halo in pixels (offsets within +-2 px stay in LDS)
constexpr int A_F4 = TAPS * 4 * 64;        // float4 elements of one weight stage (64 co x 16 c x 9 taps)
constexpr unsigned OOB = 0x80000000u;      // buffer offset past every tensor (reads 0)
constexpr int FX_BITS = 20;                // 32-bit cells: the chunk's largest |gcol * m| maps below 2^20 (<= 2^11 adds per cell)
constexpr int FX_BITS_WIDE = 35;           // 64-bit cells (exact-f32 arithmetic): below 2^35, 48-bit sums
constexpr double FX_MAGIC = 6755399441055744.0;   // 2^52 + 2^51: fma(x, s, FX_MAGIC) holds rint(x * s) in its low 48 bits

constexpr int HALO_L = 4;                  // columns left of the tile: the tile's own 16 columns start 16-B aligned
constexpr int RWD = 24;                    // region row: 4 + 16 + 4 columns (x offsets within about +-3 px)

template <int TH>
struct Geo {
  static constexpr int RH = TH + 2 * HALO, RSZ = RH * RWD;
  // channel stride of the LDS regions: = 4 (mod 8), so the four channel groups of a lane quad
  // (4 * RSZP apart) start 16 banks apart
  static constexpr int RSZP = RSZ + ((4 - RSZ % 8) + 8) % 8;
  static constexpr int NPX = TH * TW, NTHR = TH * 64;
};

struct D2Args {
  const float* x;
  const float* offset;
  const float* mask;
  const float* go;
  const f32x4* wp;           // permuted weights [chunk][slab][tap][ks4][lane] x 4 floats
  const float* wmax;         // max |W|
  unsigned* cold_flag;       // set when any wave took the cold path (float atomics straight into grad_x)
  float* gx;                 // cold path only (may be null)
  float* slab;               // [B][tiles][Cin][RSZ]
  float* goff;               // slices == 1: final tensors; else partial sums [slice][B][27][HW]
  float* gmask;
  long long offset_bstride, mask_bstride, goff_bstride, gmask_bstride, part_sstride;
  int B, Cin, H, W, Cout;
  int pad, dil, mask_is_logit;
  int tpr, ntiles;           // tiles per row, tiles per image
  int chunks, chunks_per_slice;
#ifdef CP_STAMP
  unsigned long long* dbg;   // diagnostic build only: per-wave cycle sums [block][wave][8]
#endif
};

#ifdef CP_STAMP
#define CP_T() __builtin_amdgcn_s_memtime()
#define CP_ACC(var, t0) var += (CP_T() - (t0))
#else
#define CP_T() 0ull
#define CP_ACC(var, t0) (void)0
#endif

// ---- prologue: weights into MFMA-fragment order + max |W| -------------------------------------
// wp[((chunk*NS + slab)*9 + t)*4 + ks4][lane][j] = W[co = slab*64 + 4*(4*ks4 + j) + (lane >> 4)]
//                                                   [ci = chunk*16 + (lane & 15)][t]
__global__ __launch_bounds__(256) void dcn_bwd_wperm_kernel(const float* __restrict__ w, float4* __restrict__ wp,
                                                            unsigned* __restrict__ wmax_bits, int Cin, int Cout,
                                                            int NS, int total_f4) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  float mx = 0.f;
  if (e < total_f4) {
    const int lane = e & 63;
    int r = e >> 6;
    const int ks4 = r & 3;
    r >>= 2;
    const int t = r % TAPS;
    r /= TAPS;
    const int slab = r % NS, chunk = r / NS;
    const int ci = chunk * KC + (lane & 15);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = slab * 64 + 4 * (4 * ks4 + j) + (lane >> 4);
      v[j] = (co < Cout && ci < Cin) ? w[((long long)co * Cin + ci) * TAPS + t] : 0.f;
      mx = fmaxf(mx, fabsf(v[j]));
    }
    wp[e] = make_float4(v[0], v[1], v[2], v[3]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(wmax_bits, __float_as_uint(mx));   // positive floats order as uints
}

// Split-bf16 form of the same stage (contraction on the bf16 matrix cores as three products of bf16 halves,
// w = wh + wl, wh = bf16(w), wl = bf16(w - wh)): per (chunk, slab, tap) two k-steps of 32 output channels, each
// a hi and a lo fragment of 8 bf16 per lane -- the A-operand layout of v_mfma_f32_16x16x32_bf16 (row = lane & 15,
// k = 8 (lane >> 4) + j).  Same 36 864 bytes per stage as the f32 form, so the LDS-DMA path is shared.
// wpb[(((chunk*NS + slab)*9 + t)*2 + k32)*2 + hl][lane][j] = half(hl) of W[co = slab*64 + 32 k32 + 8 (lane >> 4) + j]
//                                                                     [ci = chunk*16 + (lane & 15)][t]
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void dcn_bwd_wperm_bf16_kernel(const float* __restrict__ w, bf16x8* __restrict__ wpb,
                                                                 unsigned* __restrict__ wmax_bits, int Cin, int Cout,
                                                                 int NS, int total_frag) {
  const int e = blockIdx.x * 256 + threadIdx.x;                      // one (hi or lo) fragment of 8 values
  float mx = 0.f;
  if (e < total_frag) {
    const int lane = e & 63;
    int r = e >> 6;
    const int hl = r & 1;
    r >>= 1;
    const int k32 = r & 1;
    r >>= 1;
    const int t = r % TAPS;
    r /= TAPS;
    const int slab = r % NS, chunk = r / NS;
    const int ci = chunk * KC + (lane & 15);
    bf16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int co = slab * 64 + 32 * k32 + 8 * (lane >> 4) + j;
      const float v = (co < Cout && ci < Cin) ? w[((long long)co * Cin + ci) * TAPS + t] : 0.f;
      mx = fmaxf(mx, fabsf(v));
      const __bf16 h = (__bf16)v;
      out[j] = hl ? (__bf16)(v - (float)h) : h;
    }
    wpb[e] = out;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(wmax_bits, __float_as_uint(mx));
}

// ---- main kernel ---------------------------------------------------------------------------------
// Schedule: every wave runs the matrix phase of a chunk (NS stages, one 64-channel weight slab each,
// published by LDS-DMA into a double buffer behind ONE barrier per stage), then its consumption.  A
// staggered variant (half the waves in the matrix phase while the other half consumes) was built and
// measured equal within 2 %: on gfx950 the f32 MFMA executes on the SIMD's vector ALUs
// (SQ_VALU_MFMA_COEXEC_CYCLES = 0 in every profile of this kernel), so the two phases of one SIMD never
// overlap whichever waves run them.
template <int CP, int TH, bool WANT_GX, bool BF>
__global__ __launch_bounds__(TH * 64, TH / 4) void dcn_bwd_data2_kernel(D2Args a) {
  using G = Geo<TH>;
  constexpr int RSZ = G::RSZ, RSZP = G::RSZP, RH = G::RH, NPX = G::NPX, NTHR = G::NTHR;
  constexpr int NS = CP / 64;                              // 64-wide output-channel slabs
  constexpr int NB = CP / 4;                               // B-operand registers per lane
  // TH = 8: two waves per SIMD, weight stages double-buffered.  TH = 12 (64 output channels only): three waves
  // per SIMD -- more latency hiding for the LDS-bound consumption -- which leaves LDS for ONE weight buffer.
  constexpr bool DBUF = TH <= 8;
  static_assert(DBUF || NS == 1, "the single-buffer schedule is written for one slab per chunk");
  __shared__ f32x4 Abuf[DBUF ? 2 : 1][A_F4];               // weights of one (chunk, slab), fragment order
  // input region of the chunk's channels, channel-interleaved in groups of four ([group][cell] float4, round 4): a lane
  // (pixel, channels 4g .. 4g+3) reads a corner of its four channels with ONE ds_read_b128 -- four reads per tap where
  // the planar layout of rounds 2-3 took sixteen ds_read_b32; 16 consecutive lanes = 16 neighbouring cells = 256 bytes
  __shared__ __attribute__((aligned(16))) f32x4 xreg[(KC / 4) * RSZ];
  constexpr bool WIDE = !BF;                               // 64-bit cells under the exact-f32 arithmetic
  using cell_t = typename std::conditional<WIDE, unsigned long long, unsigned>::type;
  __shared__ __attribute__((aligned(16))) cell_t gacc[WANT_GX ? KC * RSZP : 4];   // fixed-point grad_x region sums
  __shared__ float4 rec[TAPS * NPX];                       // per (tap, pixel): ly, lx, mask, region index
  __shared__ float wred[TH];

  const unsigned long long t_entry = CP_T();
  (void)t_entry;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lpx = lane & 15, g = lane >> 4;
  const int b = blockIdx.y;
  const int tyi = blockIdx.x / a.tpr, txi = blockIdx.x - tyi * a.tpr;
  const int ty0 = tyi * TH, tx0 = txi * TW;
  const int ry0 = ty0 - HALO, rx0 = tx0 - HALO_L;
  const int HW = a.H * a.W;
  const int py = ty0 + wid, pxx = tx0 + lpx;
  const bool p_ok = py < a.H && pxx < a.W;
  const int p = p_ok ? py * a.W + pxx : 0;
  const int wpx = wid * TW + lpx;                          // pixel slot in the tile
  const int cs0 = blockIdx.z * a.chunks_per_slice;
  const int cs1 = min(a.chunks, cs0 + a.chunks_per_slice);
  const int n = cs1 - cs0;

  // ---- per (pixel, tap) sampling recipe into LDS; lane (px, g) builds taps g, g+4, g+8 ----
  bool lane_fb = false;
  float lane_mmax = 1.f;                                   // the bound of |gcol * m| also holds for masks > 1
  {
    const float* off = a.offset + (long long)b * a.offset_bstride;
    const float* msk = a.mask + (long long)b * a.mask_bstride;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int t = g + 4 * k;
      if (t < TAPS) {
        const int ky = t / 3, kx = t - ky * 3;
        float oy = 0.f, ox = 0.f, m = 0.f;
        if (p_ok) {
          oy = off[(long long)(2 * t) * HW + p];
          ox = off[(long long)(2 * t + 1) * HW + p];
          m = msk[(long long)t * HW + p];
          if (a.mask_is_logit) m = 1.f / (1.f + __expf(-m));
        }
        const float sy = (float)(py - a.pad + ky * a.dil) + oy;
        const float sx = (float)(pxx - a.pad + kx * a.dil) + ox;
        const bool inside = p_ok && sy > -1.f && sx > -1.f && sy < (float)a.H && sx < (float)a.W;
        const float fy = floorf(sy), fx = floorf(sx);
        const int y0 = (int)fy, x0 = (int)fx;
        const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= a.H - 1;
        const bool x0ok = x0 >= 0, x1ok = x0 + 1 <= a.W - 1;
        const int y0c = min(max(y0, 0), a.H - 1), x0c = min(max(x0, 0), a.W - 1);
        const int vb = (y0ok && x0ok ? 1 : 0) | (y0ok && x1ok ? 2 : 0) | (y1ok && x0ok ? 4 : 0) |
                       (y1ok && x1ok ? 8 : 0);
        const int ry = y0 - ry0, rx = x0 - rx0;
        const bool in_region = inside && ry >= 0 && ry + 1 < RH && rx >= 0 && rx + 1 < RWD;
        // >= 0: region offset of the top-left corner; -2: contributes nothing; <= -3: cold path,
        // packs the clamped top-left index and the corner validity bits
        const int rb = in_region ? ry * RWD + rx : (inside ? -(3 + (((y0c * a.W + x0c) << 4) | vb)) : -2);
        lane_fb |= rb <= -3;
        lane_mmax = fmaxf(lane_mmax, fabsf(m));
        rec[t * NPX + wpx] = make_float4(sy - fy, sx - fx, p_ok ? m : 0.f, __int_as_float(rb));
      }
    }
  }
  const bool any_fallback = __builtin_amdgcn_ballot_w64(lane_fb) != 0ull;
  if (WANT_GX && any_fallback && lane == 0) atomicOr(a.cold_flag, 1u);   // the reduce then adds onto grad_x
  const unsigned long long t_rec = CP_T();
  (void)t_rec;

  // ---- B operand: this wave's 16 grad_out pixels, all output channels, in registers ----
  const float* gob = a.go + (long long)b * a.Cout * HW;
  const __amdgpu_buffer_rsrc_t rs_go = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(gob), 0, (int)((unsigned)a.Cout * (unsigned)HW * 4u), 0x00020000);
  // f32 form: breg[q] = go[co = 4 q + g]; split-bf16 form: per slab and k-step of 32 channels the hi / lo halves of
  // go[co = 64 slab + 32 k32 + 8 g + j], j = 0..7 (B-operand layout of the 16x16x32 instruction) -- converted ONCE
  // per tile; the weights are split by the prologue kernel, so the matrix phase carries no conversion at all.
  float breg[BF ? 1 : NB];
  bf16x8 bh[BF ? NS * 2 : 1], bl[BF ? NS * 2 : 1];
  if constexpr (!BF) {
    const unsigned gbase = p_ok ? ((unsigned)g * (unsigned)HW + (unsigned)p) * 4u : OOB;
    const unsigned gstep = p_ok ? (unsigned)HW * 16u : 0u;          // 4 output channels
#pragma unroll
    for (int q = 0; q < NB; ++q)                                    // rows past Cout read 0
      breg[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_go, gbase + (unsigned)q * gstep, 0, 0));
  } else {
    const unsigned gstep = p_ok ? (unsigned)HW * 4u : 0u;           // 1 output channel
#pragma unroll
    for (int kk = 0; kk < NS * 2; ++kk) {
      const unsigned gbase = p_ok ? ((unsigned)(32 * kk + 8 * g) * (unsigned)HW + (unsigned)p) * 4u : OOB;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_go, gbase + (unsigned)j * gstep, 0, 0));
        const __bf16 h = (__bf16)v;
        bh[kk][j] = h;
        bl[kk][j] = (__bf16)(v - (float)h);
      }
    }
  }
  // largest |mask| of the wave's pixels (>= 1): the chunk's column maximum times this bounds |gcol * m|
  float fx_scale = 0.f, fx_inv = 0.f;
  if (WANT_GX) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lane_mmax = fmaxf(lane_mmax, __shfl_xor(lane_mmax, o, 64));
    for (int e = tid; e < KC * RSZP; e += NTHR) gacc[e] = 0;
  }
  const unsigned long long t_b0 = CP_T();
  (void)t_b0;
  __syncthreads();                                                  // rec, gacc initialised
  // Fixed-point scale of a chunk: after its matrix phase every wave leaves max |acc| * mask-max in wred[], the barrier
  // that publishes the input region publishes these too, and every lane derives the same power of two.
  auto publish_colmax = [&](const f32x4 (&acc_)[TAPS]) __attribute__((always_inline)) {
    auto nan_max = [](float p, float q) { return p != p ? p : (q != q ? q : fmaxf(p, q)); };   // (fmaxf drops a NaN)
    float m_ = 0.f;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) m_ = nan_max(m_, fabsf(acc_[t][r]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m_ = nan_max(m_, __shfl_xor(m_, o, 64));
    if (lane == 0) wred[wid] = m_ * lane_mmax;
  };
  auto derive_scale = [&]() __attribute__((always_inline)) {
    float cm = wred[0];
#pragma unroll
    for (int i = 1; i < TH; ++i) cm = (cm != cm) ? cm : (wred[i] != wred[i] ? wred[i] : fmaxf(cm, wred[i]));
    const float gb = cm * 1.0001f;
    if (gb > 0.f && gb < 3.0e38f) {
      int ge = 0;
      (void)frexpf(gb, &ge);                                        // gb < 2^ge
      ge = max(ge, -80);                                            // (the scale stays a finite float for vanishing gradients)
      constexpr int FXB = WIDE ? FX_BITS_WIDE : FX_BITS;
      fx_scale = ldexpf(1.f, FXB - ge);
      fx_inv = ldexpf(1.f, ge - FXB);
    } else if (gb == 0.f) {
      fx_scale = 0.f;
      fx_inv = 0.f;
    } else {
      // the chunk's columns hold Inf / NaN (a loss overflow): no fixed-point scale exists.  The sums stay 0 and the
      // flush multiplies them by NaN, so the tile's grad_x comes out NaN, as the reference's float chain would,
      // instead of silently zero (grad_offset / grad_mask carry the NaN through their float sums anyway).
      fx_scale = 0.f;
      fx_inv = __builtin_nanf("");
    }
  };

  // ---- staging ----
  // Weights: LDS-DMA (global_load_lds_dwordx4), no registers: a stage is 36 pieces of 1 KB (64 lanes x
  // 16 B, fragment order = linear order), wave w moves pieces w, w+TH, ...; the DMA of stage j+1 is
  // issued right behind the barrier that opens stage j and lands in the other buffer.
  // Input region: the X waves fetch chunk i's cells into registers while they run MFMA(i) (their VALU
  // is idle there and their register pressure low) and store them at the next odd boundary.
  const unsigned plane_bytes = (unsigned)HW * 4u;
  const float* xb = a.x + (long long)b * a.Cin * HW;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xb), 0, (int)((unsigned)a.Cin * plane_bytes), 0x00020000);
  constexpr int NPIECE = A_F4 / 64;                                 // 36
  // (inline asm: the compiler does not know LDS-DMA targets only Abuf and would drain vmcnt before every
  // LDS atomic of the consumption; the waits are placed by hand at the stage boundaries)
  auto dma_a = [&](int stage_global, int buf) {
    const f32x4* src = a.wp + (long long)stage_global * A_F4;
#pragma unroll
    for (int i = 0; i < (NPIECE + TH - 1) / TH; ++i) {
      const int piece = wid + TH * i;
      if (piece < NPIECE) {
        const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(&Abuf[buf][piece * 64]);
        const f32x4* gp = src + piece * 64 + lane;
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(gp) : "memory", "m0");
      }
    }
  };
  constexpr int NXT = NTHR;                                         // threads that stage the input region
  constexpr int XRX = ((KC / 4) * RSZ + NXT - 1) / NXT;             // (group, cell) float4 items per X thread
  f32x4 xr[XRX];
  auto load_x = [&](int ch) {                                       // X waves only
    const unsigned cbase = (unsigned)(ch * KC) * plane_bytes;       // rides in voffset (range-checked);
#pragma unroll                                                      // OOB + cbase stays past the tensor
    for (int i = 0; i < XRX; ++i) {
      int e = tid + NXT * i;
      asm volatile("" : "+v"(e));                                   // keep the index math out of the loop-invariant set
      const int gq = e / RSZ, cell = e - gq * RSZ;
      const int ry = cell / RWD, rx = cell - ry * RWD;
      const int gy_ = ry0 + ry, gx_ = rx0 + rx;
      const bool ok = e < (KC / 4) * RSZ && gy_ >= 0 && gy_ < a.H && gx_ >= 0 && gx_ < a.W;
      const unsigned off = ok ? (unsigned)(4 * gq) * plane_bytes + 4u * (unsigned)(gy_ * a.W + gx_) : OOB;
#pragma unroll
      for (int q = 0; q < 4; ++q)                                   // (everything in voffset: the range check covers channels past Cin)
        xr[i][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, off + cbase + (unsigned)q * plane_bytes, 0, 0));
    }
  };
  auto store_x = [&]() {                                            // X waves only
#pragma unroll
    for (int i = 0; i < XRX; ++i) {
      int e = tid + NXT * i;
      asm volatile("" : "+v"(e));
      if (e < (KC / 4) * RSZ) xreg[e] = xr[i];
    }
  };

  float gm[TAPS], gy[TAPS], gxo[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) gm[t] = gy[t] = gxo[t] = 0.f;
  const int cbase_lds = 4 * g * RSZP;                               // this lane's first channel in the regions
  float* slab_tile = WANT_GX ? a.slab + ((long long)(b * a.ntiles + blockIdx.x) * a.Cin) * RSZ : nullptr;
  constexpr int NQ = KC * RSZ / 4;                                  // 4-cell groups of the chunk's regions
  constexpr int XQ = (NQ + NTHR - 1) / NTHR;
  static_assert(RSZ % 4 == 0 && RSZP % 4 == 0, "flush moves aligned groups of four cells");
  auto flush = [&](int c0, float inv) {                             // region sums -> slab, plain coalesced stores
    float* dst = slab_tile + (long long)c0 * RSZ;
#pragma unroll
    for (int i = 0; i < XQ; ++i) {
      int q = tid + NTHR * i;
      asm volatile("" : "+v"(q));                                   // keep the index math out of the loop-invariant set
      if (q < NQ) {
        const int c = q / (RSZ / 4);
        const int l = c * RSZP + (q - c * (RSZ / 4)) * 4;
        f32x4 o;
        if constexpr (WIDE) {
          typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
          u64x2* gp = reinterpret_cast<u64x2*>(&gacc[l]);
          const u64x2 a0 = gp[0], a1 = gp[1];
          gp[0] = u64x2{0ull, 0ull};
          gp[1] = u64x2{0ull, 0ull};
          const double dinv = (double)inv;
          o.x = (float)((double)(((long long)(a0.x << 16)) >> 16) * dinv);   // low 48 bits, sign-extended
          o.y = (float)((double)(((long long)(a0.y << 16)) >> 16) * dinv);
          o.z = (float)((double)(((long long)(a1.x << 16)) >> 16) * dinv);
          o.w = (float)((double)(((long long)(a1.y << 16)) >> 16) * dinv);
        } else {
          typedef int i32x4 __attribute__((ext_vector_type(4)));
          i32x4* gp = reinterpret_cast<i32x4*>(&gacc[l]);
          const i32x4 a0 = gp[0];
          gp[0] = i32x4{0, 0, 0, 0};
          o.x = (float)a0.x * inv;                                  // |sum| < 2^31: one rounding to fp32, then an exact scaling
          o.y = (float)a0.y * inv;
          o.z = (float)a0.z * inv;
          o.w = (float)a0.w * inv;
        }
        if (inv != inv) o = f32x4{inv, inv, inv, inv};              // (non-finite chunk: see derive_scale)
        if (c0 + c < a.Cin) *reinterpret_cast<f32x4*>(dst + 4 * q) = o;
      }
    }
  };

  f32x4 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (n > 0) dma_a(cs0 * NS, 0);
  int stage = 0;                                                    // counts published weight stages
  unsigned long long t_mfma = 0, t_cons = 0, t_bodd = 0, t_bar = 0, t_flush = 0, t_kernel0 = CP_T();
  (void)t_mfma; (void)t_cons; (void)t_bodd; (void)t_bar; (void)t_flush; (void)t_kernel0;   // (diagnostic build only)

  auto mfma_slab = [&](auto S_, int buf) __attribute__((always_inline)) {
    constexpr int s = decltype(S_)::value;
    if constexpr (BF) {
      const bf16x8* Ab = reinterpret_cast<const bf16x8*>(&Abuf[buf][0]);      // [tap][k32][hl][lane]
#pragma unroll
      for (int k32 = 0; k32 < 2; ++k32) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          const bf16x8 ah = Ab[((t * 2 + k32) * 2 + 0) * 64 + lane];
          const bf16x8 al = Ab[((t * 2 + k32) * 2 + 1) * 64 + lane];
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[s * 2 + k32], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[s * 2 + k32], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[s * 2 + k32], acc[t], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int ks4 = 0; ks4 < 4; ++ks4) {
#pragma unroll
        for (int t0 = 0; t0 < TAPS; t0 += 3) {
          f32x4 af[3];
#pragma unroll
          for (int u = 0; u < 3; ++u) af[u] = Abuf[buf][((t0 + u) * 4 + ks4) * 64 + lane];
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            acc[t0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u].x, breg[s * 16 + ks4 * 4 + 0], acc[t0 + u], 0, 0, 0);
            acc[t0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u].y, breg[s * 16 + ks4 * 4 + 1], acc[t0 + u], 0, 0, 0);
            acc[t0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u].z, breg[s * 16 + ks4 * 4 + 2], acc[t0 + u], 0, 0, 0);
            acc[t0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u].w, breg[s * 16 + ks4 * 4 + 3], acc[t0 + u], 0, 0, 0);
          }
        }
      }
    }
  };
  // Consumption segment s of a chunk: lane = (pixel lpx, channels c0 + 4g .. 4g+3), one tap at a time.
  // The region indices of the segment's taps are read first, so the corner reads of tap t+1 depend on no
  // LDS read and are issued AHEAD of tap t's adds (LDS operations of a wave complete in order: a read
  // queued behind sixteen adds would make the wave wait for all of them).
  auto consume_seg = [&](auto S_, int c0_cons) __attribute__((always_inline)) {
    constexpr int s = decltype(S_)::value;
    constexpr int T0 = (TAPS * s + NS - 1) / NS, T1 = (TAPS * (s + 1) + NS - 1) / NS;
    int rbs[T1 - T0];
#pragma unroll
    for (int t = T0; t < T1; ++t) rbs[t - T0] = __float_as_int(rec[t * NPX + wpx].w);
    float4 rcb[2];                                                  // recipe and corners of taps t (slot t & 1) and t+1
    f32x4 vc_[2][4];                                                // corners 00, 01, 10, 11 x the lane's four channels
    auto fetch = [&](int t, int slot) __attribute__((always_inline)) {
      rcb[slot] = rec[t * NPX + wpx];
      const int rbc = max(rbs[t - T0], 0) + g * RSZ;
      vc_[slot][0] = xreg[rbc];
      vc_[slot][1] = xreg[rbc + 1];
      vc_[slot][2] = xreg[rbc + RWD];
      vc_[slot][3] = xreg[rbc + RWD + 1];
    };
    fetch(T0, T0 & 1);
#pragma unroll
    for (int t = T0; t < T1; ++t) {
      const float ly = rcb[t & 1].x, lx = rcb[t & 1].y, m = rcb[t & 1].z;
      const int rb = rbs[t - T0];
      const f32x4 (&vc)[4] = vc_[t & 1];
      if (t + 1 < T1) fetch(t + 1, (t + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);                            // keep those reads ahead of the adds below
      if (rb >= 0) {                                                // out-of-image cells of the region hold 0
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float w00 = hy * hx, w01 = hy * lx, w10 = ly * hx, w11 = ly * lx;
        const int cell = cbase_lds + rb;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gc = acc[t][r];
          const float v00 = vc[0][r], v01 = vc[1][r], v10 = vc[2][r], v11 = vc[3][r];
          // d/dy and d/dx of the bilinear interpolant, and the interpolant itself from them (14 operations where the
          // three separate corner forms took 16): val = v00 + lx (v01 - v00) + ly (hx (v10 - v00) + lx (v11 - v01))
          const float dyv = hx * (v10 - v00) + lx * (v11 - v01);
          const float dxv = hy * (v01 - v00) + ly * (v11 - v10);
          gm[t] += gc * fmaf(ly, dyv, fmaf(lx, v01 - v00, v00));
          const float gcm = gc * m;
          gy[t] += gcm * dyv;
          gxo[t] += gcm * dxv;
          if (WANT_GX) {                                            // cells outside the image are never read back
            cell_t* q = &gacc[cell + r * RSZP];
            if constexpr (WIDE) {
              const double dg = (double)gcm, ds = (double)fx_scale;
              atomicAdd(q, (unsigned long long)__double_as_longlong(fma(dg, (double)w00 * ds, FX_MAGIC)));
              atomicAdd(q + 1, (unsigned long long)__double_as_longlong(fma(dg, (double)w01 * ds, FX_MAGIC)));
              atomicAdd(q + RWD, (unsigned long long)__double_as_longlong(fma(dg, (double)w10 * ds, FX_MAGIC)));
              atomicAdd(q + RWD + 1, (unsigned long long)__double_as_longlong(fma(dg, (double)w11 * ds, FX_MAGIC)));
            } else {
              const float gs = gcm * fx_scale;                       // (power of two: exact)
              atomicAdd(q, (unsigned)__float2int_rn(gs * w00));
              atomicAdd(q + 1, (unsigned)__float2int_rn(gs * w01));
              atomicAdd(q + RWD, (unsigned)__float2int_rn(gs * w10));
              atomicAdd(q + RWD + 1, (unsigned)__float2int_rn(gs * w11));
            }
          }
        }
      }
    }
    // cold path (skipped wave-uniformly when no pixel of the wave has such a tap): corners that leave the
    // region are gathered from memory, their grad_x contributions go there as float atomics
    if (s == NS - 1 && any_fallback) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {                              // static t: acc / gm stay in registers
        const float4 rc = rec[t * NPX + wpx];
        const int rb = __float_as_int(rc.w);
        if (rb <= -3) {
          const int code = -rb - 3;
          const unsigned vb = (unsigned)(code & 15);
          const int fbase = code >> 4;
          const int dx = ((vb & 3u) == 3u || (vb & 12u) == 12u) ? 1 : 0;
          const int dy = ((vb & 5u) == 5u || (vb & 10u) == 10u) ? a.W : 0;
          const float ly = rc.x, lx = rc.y, m = rc.z, hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = c0_cons + 4 * g + r;
            if (c < a.Cin) {
              const float* q = xb + (long long)c * HW + fbase;
              const float v00 = (vb & 1u) ? q[0] : 0.f;
              const float v01 = (vb & 2u) ? q[dx] : 0.f;
              const float v10 = (vb & 4u) ? q[dy] : 0.f;
              const float v11 = (vb & 8u) ? q[dy + dx] : 0.f;
              const float gc = acc[t][r];
              gm[t] += gc * (hy * hx * v00 + hy * lx * v01 + ly * hx * v10 + ly * lx * v11);
              const float gcm = gc * m;
              gy[t] += gcm * (hx * (v10 - v00) + lx * (v11 - v01));
              gxo[t] += gcm * (hy * (v01 - v00) + ly * (v11 - v10));
              if (WANT_GX && a.gx) {
                float* o = a.gx + ((long long)b * a.Cin + c) * HW + fbase;
                if (vb & 1u) atomicAdd(o, gcm * hy * hx);
                if (vb & 2u) atomicAdd(o + dx, gcm * hy * lx);
                if (vb & 4u) atomicAdd(o + dy, gcm * ly * hx);
                if (vb & 8u) atomicAdd(o + dy + dx, gcm * ly * lx);
              }
            }
          }
        }
      }
    }
  };
  {
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned long long tw0 = CP_T();
      (void)tw0;
      if constexpr (DBUF) {
        load_x(cs0 + i);                                            // lands during the matrix phase
        static_for<0, NS>([&](auto S_) __attribute__((always_inline)) {
          constexpr int s = decltype(S_)::value;
          const int buf = stage & 1;
          __builtin_amdgcn_s_waitcnt(0x0f70);                       // vmcnt(0): this stage's DMA (and the region loads)
          __syncthreads();                                          // ... of every wave; previous consumption finished
          const int ni = s + 1 < NS ? i : i + 1, ns = s + 1 < NS ? s + 1 : 0;
          if (ni < n) dma_a((cs0 + ni) * NS + ns, buf ^ 1);
          ++stage;
          mfma_slab(S_, buf);
        });
        CP_ACC(t_mfma, tw0);
        const unsigned long long tf0 = CP_T();
        (void)tf0;
        store_x();                                                  // (loads waited for at the stage barrier above)
        if (WANT_GX) publish_colmax(acc);
        if (WANT_GX && i >= 1) flush((cs0 + i - 1) * KC, fx_inv);   // (still the previous chunk's scale)
        __syncthreads();                                            // region + emptied sums + column maxima visible
        if (WANT_GX) derive_scale();
        CP_ACC(t_flush, tf0);
      } else {
        __builtin_amdgcn_s_waitcnt(0x0f70);                         // vmcnt(0): this chunk's weight DMA has landed
        __syncthreads();                                            // ... for every wave; previous consumption finished
        mfma_slab(std::integral_constant<int, 0>{}, 0);
        CP_ACC(t_mfma, tw0);
        const unsigned long long tf0 = CP_T();
        (void)tf0;
        if (WANT_GX) publish_colmax(acc);
        __syncthreads();                                            // every wave has read the weight buffer
        load_x(cs0 + i);                                            // region registers live only across the flush
        if (WANT_GX && i >= 1) flush((cs0 + i - 1) * KC, fx_inv);   // (still the previous chunk's scale)
        store_x();
        if (i + 1 < n) dma_a((cs0 + i + 1) * NS, 0);                // lands during the consumption
        __syncthreads();                                            // region + emptied sums visible
        if (WANT_GX) derive_scale();
        CP_ACC(t_flush, tf0);
      }
      const unsigned long long tw1 = CP_T();
      (void)tw1;
      static_for<0, NS>([&](auto S_) __attribute__((always_inline)) { consume_seg(S_, (cs0 + i) * KC); });
      CP_ACC(t_cons, tw1);
    }
  }
  if (WANT_GX && n > 0) {
    __syncthreads();                                                // the last chunk's adds are in LDS
    flush((cs1 - 1) * KC, fx_inv);
  }

#ifdef CP_STAMP
  if (a.dbg && lane == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 1024) {
    unsigned long long* d = a.dbg + ((long long)blockIdx.x * TH + wid) * 8;
    d[0] = t_mfma; d[1] = t_cons; d[2] = t_bodd; d[3] = t_bar; d[4] = t_flush; d[5] = CP_T() - t_kernel0;
    d[6] = t_rec - t_entry; d[7] = t_kernel0 - t_b0;
    if (wid == 0) { d[8 + 6] = t_b0 - t_rec; d[8 + 7] = t_entry; }   // (wave 1's slots 6/7 reused: breg phase, entry time)
  }
  if (a.dbg && lane == 0 && wid == 0 && blockIdx.z == 0) {           // every workgroup: entry, exit, hardware id
    unsigned long long* d = a.dbg + 1024 * 8 * 8 + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 4;
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    d[0] = t_entry; d[1] = CP_T(); d[2] = hwid; d[3] = xcc;
  }
#endif
  // ---- grad_offset / grad_mask: add the four channel groups of each pixel, lane g stores taps g, g+4, g+8 ----
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    gm[t] += __shfl_xor(gm[t], 16, 64);
    gm[t] += __shfl_xor(gm[t], 32, 64);
    gy[t] += __shfl_xor(gy[t], 16, 64);
    gy[t] += __shfl_xor(gy[t], 32, 64);
    gxo[t] += __shfl_xor(gxo[t], 16, 64);
    gxo[t] += __shfl_xor(gxo[t], 32, 64);
  }
  if (p_ok) {
    float* goffb = a.goff ? a.goff + (long long)blockIdx.z * a.part_sstride + (long long)b * a.goff_bstride : nullptr;
    float* gmaskb = a.gmask ? a.gmask + (long long)blockIdx.z * a.part_sstride + (long long)b * a.gmask_bstride : nullptr;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      if ((t & 3) != g) continue;
      if (goffb) {
        goffb[(long long)(2 * t) * HW + p] = gy[t];
        goffb[(long long)(2 * t + 1) * HW + p] = gxo[t];
      }
      if (gmaskb) {
        float v = gm[t];
        if (a.mask_is_logit) {
          const float m = rec[t * NPX + wpx].z;
          v *= m * (1.f - m);
        }
        gmaskb[(long long)t * HW + p] = v;
      }
    }
  }
}

// ---- grad_x[b][c][y][x] += sum of the slabs whose region covers (y, x), fixed order -------------
// A tile's slab holds, per channel, RH rows of 24 cells: columns 0..3 = left halo, 4..19 = the tile's
// own pixels (16-byte aligned), 20..23 = right halo; rows 0..2 / RH-3..RH-1 = the vertical halos.  One
// thread adds up one float4 of grad_x: its own tile's cells, the horizontal neighbour's halo when it is
// the tile's first / last float4, the vertical neighbours' for the tile's first / last three rows, and
// the diagonal ones.
template <bool VEC>
__global__ __launch_bounds__(256) void dcn_bwd_gx_reduce_kernel(const float* __restrict__ slab, float* __restrict__ gx,
                                                                const unsigned* __restrict__ cold_flag, int BC, int Cin,
                                                                int H, int W, int tpr, int tpc, int TH) {
  constexpr int V = VEC ? 4 : 1;
  const int xq = blockIdx.x * 64 + (threadIdx.x & 63);              // float4 (or pixel) index in the row
  const int x = xq * V;
  const int y = blockIdx.y;
  const int bc = blockIdx.z * 4 + (threadIdx.x >> 6);               // b * Cin + c
  if (x >= W || bc >= BC) return;
  const int b = bc / Cin, c = bc - b * Cin;
  const int RH = TH + 2 * HALO, RSZ = RH * RWD;
  const int txi = x / TW, tyi = y / TH;
  const int ntiles = tpr * tpc;
  float s[V];
#pragma unroll
  for (int i = 0; i < V; ++i) s[i] = 0.f;
#pragma unroll
  for (int dy = 0; dy <= 2; ++dy) {                                 // own tile first, then above, then below
    const int tyy = tyi + (dy == 0 ? 0 : (dy == 1 ? -1 : 1));
    const int ry = y - (tyy * TH - HALO);
    if (tyy < 0 || tyy >= tpc || ry < 0 || ry >= RH) continue;
#pragma unroll
    for (int dx = 0; dx <= 2; ++dx) {
      const int txx = txi + (dx == 0 ? 0 : (dx == 1 ? -1 : 1));
      const int rx = x - (txx * TW - HALO_L);
      if (txx < 0 || txx >= tpr || rx < 0 || rx + V > RWD) continue;
      const float* q = slab + (((long long)b * ntiles + tyy * tpr + txx) * Cin + c) * RSZ + ry * RWD + rx;
      if (VEC) {
        const float4 v = *reinterpret_cast<const float4*>(q);
        s[0] += v.x; s[1 % V] += v.y; s[2 % V] += v.z; s[3 % V] += v.w;
      } else {
        s[0] += q[0];
      }
    }
  }
  // grad_x already holds something only if a wave gathered outside its region (float atomics of the cold path):
  // otherwise it is overwritten and its 4 bytes per element are not read
  const bool add = *cold_flag != 0u;
  float* o = gx + ((long long)bc * H + y) * W + x;
  if (VEC) {
    float4 v = add ? *reinterpret_cast<float4*>(o) : make_float4(0.f, 0.f, 0.f, 0.f);
    v.x += s[0]; v.y += s[1 % V]; v.z += s[2 % V]; v.w += s[3 % V];
    *reinterpret_cast<float4*>(o) = v;
  } else {
    o[0] = (add ? o[0] : 0.f) + s[0];
  }
}

// ---- out[b][ch][p] = sum_s part[s][b][ch][p]  (channel-sliced launches) -------------------------
__global__ __launch_bounds__(256) void dcn_bwd_part_reduce_kernel(const float* __restrict__ part, long long sstride,
                                                                  int slices, float* __restrict__ goff,
                                                                  long long goff_bstride, float* __restrict__ gmask,
                                                                  long long gmask_bstride, int B, int HW) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;     // over B * 27 * HW
  const long long per_b = 27ll * HW;
  if (i >= (long long)B * per_b) return;
  const int b = (int)(i / per_b);
  const long long r = i - (long long)b * per_b;
  const int ch = (int)(r / HW);
  const int p = (int)(r - (long long)ch * HW);
  float s = 0.f;
  for (int k = 0; k < slices; ++k) s += part[(long long)k * sstride + i];
  if (ch < 18) {
    if (goff) goff[(long long)b * goff_bstride + (long long)ch * HW + p] = s;
  } else if (gmask) {
    gmask[(long long)b * gmask_bstride + (long long)(ch - 18) * HW + p] = s;
  }
}

constexpr int TH_DEFAULT = 8;
constexpr int TH_WIDE = 12;                // three waves per SIMD, layers with <= 64 output channels

struct D2Plan {
  int TH;
  int tpr, tpc, ntiles, chunks, NS, slices, chunks_per_slice;
  size_t off_wmax, off_wp, off_slab, off_part, total;
};

}  // namespace

// Shapes the kernel takes: stride 1, output size = input size (pad == dil), Cout <= 256,
// 32-bit byte offsets inside one image.
bool cp_dcn_bwd_data2_supported(const cp_dcn_shape* s) {
  if (s->stride != 1 || s->pad != s->dil || s->kh != 3 || s->kw != 3 || s->Cout > 256) return false;
  const unsigned long long hw = (unsigned long long)s->H * s->W;
  const unsigned long long cin_pad = (unsigned long long)((s->Cin + KC - 1) / KC * KC);
  if (hw >= (1ull << 27)) return false;                              // cold-path index packed with 4 bits
  if (cin_pad * hw * 4ull >= 0x70000000ull) return false;            // OOB + chunk base must not wrap
  if (256ull * hw * 4ull >= 0x70000000ull) return false;
  if (s->B > 65535) return false;
  return true;
}

// narrow = CP_DCN_BWD_NARROW_TILES: the 8-row tiles everywhere (A/B timing)
static D2Plan d2_plan(const cp_dcn_shape* s, bool want_gx, bool want_om, bool narrow) {
  D2Plan p;
  // 12-row tiles (three waves per SIMD) for the 64-channel-output layers when they still fill the chip twice:
  // measured 2..7 % faster than 8-row tiles at B >= 4, slower at B = 1 (fewer workgroups)
  const long long wide_wgs = (long long)((s->W + TW - 1) / TW) * ((s->H + TH_WIDE - 1) / TH_WIDE) * s->B;
  p.TH = (s->Cout <= 64 && want_gx && wide_wgs >= 1024 && !narrow) ? TH_WIDE : TH_DEFAULT;
  p.tpr = (s->W + TW - 1) / TW;
  p.tpc = (s->H + p.TH - 1) / p.TH;
  p.ntiles = p.tpr * p.tpc;
  p.chunks = (s->Cin + KC - 1) / KC;
  p.NS = s->Cout <= 64 ? 1 : (s->Cout <= 128 ? 2 : 4);
  // channel slices: at least ~3 rounds of workgroups on 256 CUs (one workgroup per CU)
  const long long wgs = (long long)p.ntiles * s->B;
  int slices = (int)((768 + wgs - 1) / wgs);
  if (slices < 1) slices = 1;
  if (slices > p.chunks) slices = p.chunks;
  p.chunks_per_slice = (p.chunks + slices - 1) / slices;
  p.slices = (p.chunks + p.chunks_per_slice - 1) / p.chunks_per_slice;
  size_t o = 0;
  p.off_wmax = o; o += 256;
  p.off_wp = o; o += cp_align_up((size_t)p.chunks * p.NS * A_F4 * sizeof(f32x4), 256);
  p.off_slab = o;
  if (want_gx)
    o += cp_align_up((size_t)s->B * p.ntiles * s->Cin * (p.TH + 2 * HALO) * RWD * sizeof(float), 256);
  p.off_part = o;
  if (want_om && p.slices > 1) o += cp_align_up((size_t)p.slices * s->B * 27 * s->H * s->W * sizeof(float), 256);
#ifdef CP_STAMP
  o += 1024 * 16 * 8 * 8;
#endif
  p.total = o;
  return p;
}

size_t cp_dcn_bwd_data2_workspace_bytes(const cp_dcn_shape* s) {      // (enough for either tile form)
  if (!cp_dcn_bwd_data2_supported(s)) return 0;
  const size_t a = d2_plan(s, true, true, false).total, b = d2_plan(s, true, true, true).total;
  return a > b ? a : b;
}

// Contraction of the grad columns: split-bf16 x3 on the bf16 matrix cores (default) or the exact f32 MFMA chain
// (flag CP_DCN_BWD_EXACT_F32).  The f32-input MFMA executes on the SIMD's vector ALUs and serialises with the
// consumption's VALU work; the bf16 cores run beside it.  |error| of a grad column <= 3 * 2^-17 * sum|w go|.

template <int CP, bool WANT_GX, bool BF>
static void d2_launch_bf(const D2Args& a, const D2Plan& p, int B, hipStream_t st) {
  if constexpr (CP == 64 && WANT_GX) {
    if (p.TH == TH_WIDE) {
      hipLaunchKernelGGL((dcn_bwd_data2_kernel<64, TH_WIDE, true, BF>), dim3(p.ntiles, B, p.slices),
                         dim3(TH_WIDE * 64), 0, st, a);
      return;
    }
  }
  hipLaunchKernelGGL((dcn_bwd_data2_kernel<CP, TH_DEFAULT, WANT_GX, BF>), dim3(p.ntiles, B, p.slices),
                     dim3(TH_DEFAULT * 64), 0, st, a);
}

template <int CP, bool WANT_GX>
static void d2_launch(const D2Args& a, const D2Plan& p, int B, bool exact, hipStream_t st) {
  if (exact) d2_launch_bf<CP, WANT_GX, false>(a, p, B, st);
  else d2_launch_bf<CP, WANT_GX, true>(a, p, B, st);
}

// Data gradients of cp_dcn_v2_backward on the caller's workspace.  Returns CP_OK, or CP_EINVAL when
// the workspace is too small.
int cp_dcn_bwd_data2(const cp_dcn_shape* s, const float* x, const float* offset, int64_t offset_bstride,
                     const float* mask, int64_t mask_bstride, int32_t mask_is_logit, const float* weight,
                     const float* grad_out, float* grad_x, float* grad_offset, int64_t grad_offset_bstride,
                     float* grad_mask, int64_t grad_mask_bstride, int32_t flags, void* workspace, size_t workspace_bytes,
                     hipStream_t st) {
  const bool want_gx = grad_x != nullptr, want_om = grad_offset != nullptr || grad_mask != nullptr;
  const bool exact = (flags & CP_DCN_BWD_EXACT_F32) != 0;
  const D2Plan p = d2_plan(s, want_gx, want_om, (flags & CP_DCN_BWD_NARROW_TILES) != 0);
  if (!workspace || workspace_bytes < p.total) return CP_EINVAL;
  char* ws = (char*)workspace;
  unsigned* wmax_bits = (unsigned*)(ws + p.off_wmax);
  f32x4* wp = (f32x4*)(ws + p.off_wp);
  (void)hipMemsetAsync(wmax_bits, 0, 8, st);                 // max |W| and the cold-path flag
  const int total_f4 = p.chunks * p.NS * A_F4;
  if (exact)
    hipLaunchKernelGGL(dcn_bwd_wperm_kernel, dim3((total_f4 + 255) / 256), dim3(256), 0, st, weight, (float4*)wp,
                       wmax_bits, s->Cin, s->Cout, p.NS, total_f4);
  else                                                     // (a bf16x8 fragment is 16 bytes: same count, same bytes)
    hipLaunchKernelGGL(dcn_bwd_wperm_bf16_kernel, dim3((total_f4 + 255) / 256), dim3(256), 0, st, weight, (bf16x8*)wp,
                       wmax_bits, s->Cin, s->Cout, p.NS, total_f4);
  const int HW = s->H * s->W;
  D2Args a;
  a.x = x; a.offset = offset; a.mask = mask; a.go = grad_out; a.wp = wp; a.wmax = (const float*)wmax_bits; a.cold_flag = wmax_bits + 1;
  a.gx = grad_x; a.slab = want_gx ? (float*)(ws + p.off_slab) : nullptr;
  a.offset_bstride = offset_bstride; a.mask_bstride = mask_bstride;
  a.B = s->B; a.Cin = s->Cin; a.H = s->H; a.W = s->W; a.Cout = s->Cout;
  a.pad = s->pad; a.dil = s->dil; a.mask_is_logit = mask_is_logit;
  a.tpr = p.tpr; a.ntiles = p.ntiles; a.chunks = p.chunks; a.chunks_per_slice = p.chunks_per_slice;
#ifdef CP_STAMP
  a.dbg = (unsigned long long*)(ws + p.total - 1024 * 16 * 8 * 8);
#endif
  float* part = (float*)(ws + p.off_part);
  if (p.slices > 1 && want_om) {           // partial sums [slice][B][27][HW]: offsets at ch 0, mask at ch 18
    a.goff = part; a.gmask = part + 18ll * HW;
    a.goff_bstride = a.gmask_bstride = 27ll * HW;
    a.part_sstride = (long long)s->B * 27 * HW;
  } else {
    a.goff = grad_offset; a.gmask = grad_mask;
    a.goff_bstride = grad_offset_bstride; a.gmask_bstride = grad_mask_bstride;
    a.part_sstride = 0;
  }
  if (want_gx) {
    if (p.NS == 1) d2_launch<64, true>(a, p, s->B, exact, st);
    else if (p.NS == 2) d2_launch<128, true>(a, p, s->B, exact, st);
    else d2_launch<256, true>(a, p, s->B, exact, st);
    const bool vec = s->W % 4 == 0 && ((uintptr_t)grad_x & 15) == 0;
    const int per_row = vec ? s->W / 4 : s->W;
    const dim3 grid((per_row + 63) / 64, s->H, (s->B * s->Cin + 3) / 4);
    if (vec)
      hipLaunchKernelGGL(dcn_bwd_gx_reduce_kernel<true>, grid, dim3(256), 0, st, a.slab, grad_x, a.cold_flag,
                         s->B * s->Cin, s->Cin, s->H, s->W, p.tpr, p.tpc, p.TH);
    else
      hipLaunchKernelGGL(dcn_bwd_gx_reduce_kernel<false>, grid, dim3(256), 0, st, a.slab, grad_x, a.cold_flag,
                         s->B * s->Cin, s->Cin, s->H, s->W, p.tpr, p.tpc, p.TH);
  } else {
    if (p.NS == 1) d2_launch<64, false>(a, p, s->B, exact, st);
    else if (p.NS == 2) d2_launch<128, false>(a, p, s->B, exact, st);
    else d2_launch<256, false>(a, p, s->B, exact, st);
  }
  if (p.slices > 1 && want_om) {
    const long long n = (long long)s->B * 27 * HW;
    hipLaunchKernelGGL(dcn_bwd_part_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part,
                       a.part_sstride, p.slices, grad_offset, (long long)grad_offset_bstride, grad_mask,
                       (long long)grad_mask_bstride, s->B, HW);
  }
  return cp_launch_status();
}
