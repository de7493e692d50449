// DCNv2 backward for gfx950.
//
// Replaces upstream's dcn_v2_backward (columns materialised in HBM + cuBLAS + an
// atomic col2im) behind `from .DCNv2.dcn_v2 import DCN`
// (reference: src/lib/models/networks/pose_dla_dcn.py:16,354).
//
// Two kernels over the same 64-pixel tiles as the forward pass; the im2col /
// grad-column tiles exist only in LDS:
//   data kernel    gcol[px][k] = sum_co go[px][co] * W[co][k]   (fp32 MFMA, K = Cout)
//                  then per lane (= pixel), per (channel, tap): re-gather the 4
//                  corners and turn gcol into
//                    grad_mask   += gcol * sampled value            (summed over channels)
//                    grad_offset += gcol * mask * d(bilinear)/d(y,x)
//                    grad_x      += gcol * mask * corner weight     (float atomics)
//   weight kernel  gW[co][k] += sum_px go[px][co] * col[px][k]   (fp32 MFMA, K = 64 pixels
//                  per workgroup, col re-sampled in LDS), one float-atomic tile per chunk.
// grad_bias is a plain reduction of grad_out.
//
// Fast data kernel (`tiled`, Cout <= 256 and W a multiple of 64, i.e. every DLA-34 layer at
// the Cityscapes shape): the grad_out tile is staged once; per 4-channel chunk an
// (2R+2) x (64+2R+1) input REGION around the tile's row segment is staged in LDS with
// coalesced row loads, corners are read from it, and grad_x contributions are accumulated
// into a matching LDS region, then flushed with one coalesced global atomic per region cell
// (4x fewer global atomics than one per corner).  The LDS accumulation is 64-bit FIXED POINT
// (ds_add_u64, measured 7.6 cycles per wave-instruction vs 193 for ds_add_f32 on gfx950,
// tools/micro/lds_atomic_rate.hip) with a per-chunk power-of-two scale taken from max|gcol|:
// every fp32 contribution converts exactly, so the region sum is exact and order-independent
// and is rounded to fp32 once, at the flush.  Taps whose corners leave
// the region fall back to global gathers / atomics.  The generic kernel below it covers
// every other shape.
#include "cp_common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 64;
constexpr int TAPS = 9;
constexpr int KC = 4;             // channels per chunk
constexpr int KK = KC * TAPS;     // 36
constexpr int LDK = KK + 1;       // 37
constexpr int COC = 128;          // output-channel slab staged per pass
constexpr int LDC = COC + 1;

struct DcnBwdArgs {
  const float* x;
  const float* offset;
  const float* mask;
  const float* weight;
  const float* go;
  float* gx;
  float* goff;
  float* gmask;
  float* gw;
  long long offset_bstride, mask_bstride, goff_bstride, gmask_bstride;
  int B, Cin, H, W, Cout, Ho, Wo;
  int stride, pad, dil, mask_is_logit;
  int tpr;      // tiled kernels: 64-pixel tiles per image row (the last one may be partial)
#ifdef CP_ABLATE
  int ablate;   // timing-only build: bit0 no consumption, bit1 no flush, bit2 no MFMA, bit3 no loads
#endif
};

#ifdef CP_ABLATE
#include <stdlib.h>
#define CP_ABL(bit) (a.ablate & (bit))
#else
#define CP_ABL(bit) 0
#endif

// Per-pixel, per-tap sampling recipe shared by both kernels.
struct Recipe {
  float ly[TAPS], lx[TAPS], m[TAPS];
  int base[TAPS];
  unsigned step;        // bit 2t: +1 column step valid, bit 2t+1: +W row step valid
  unsigned valid_lo;    // 4 bits per tap (taps 0..7): corner validity 00,01,10,11
  unsigned valid_hi;    // tap 8
};

__device__ __forceinline__ unsigned corner_bits(const Recipe& r, int t) {
  return t < 8 ? (r.valid_lo >> (4 * t)) & 15u : r.valid_hi & 15u;
}

__device__ __forceinline__ void build_recipe(const DcnBwdArgs& a, int b, int p, bool p_ok,
                                             Recipe& r) {
  const int HWo = a.Ho * a.Wo;
  const int ho = p_ok ? p / a.Wo : 0;
  const int wo = p_ok ? p - ho * a.Wo : 0;
  const float* off = a.offset + (long long)b * a.offset_bstride;
  const float* msk = a.mask + (long long)b * a.mask_bstride;
  r.step = 0;
  r.valid_lo = 0;
  r.valid_hi = 0;
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    const int ky = t / 3, kx = t - ky * 3;
    float oy = 0.f, ox = 0.f, m = 0.f;
    if (p_ok) {
      oy = off[(long long)(2 * t) * HWo + p];
      ox = off[(long long)(2 * t + 1) * HWo + p];
      m = msk[(long long)t * HWo + p];
      if (a.mask_is_logit) m = 1.f / (1.f + __expf(-m));
    }
    const float py = (float)(ho * a.stride - a.pad + ky * a.dil) + oy;
    const float px = (float)(wo * a.stride - a.pad + kx * a.dil) + ox;
    const bool inside = p_ok && py > -1.f && px > -1.f && py < (float)a.H && px < (float)a.W;
    const float fy = floorf(py), fx = floorf(px);
    const int y0 = (int)fy, x0 = (int)fx;
    r.ly[t] = py - fy;
    r.lx[t] = px - fx;
    r.m[t] = m;
    const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= a.H - 1;
    const bool x0ok = x0 >= 0, x1ok = x0 + 1 <= a.W - 1;
    const int y0c = min(max(y0, 0), a.H - 1), x0c = min(max(x0, 0), a.W - 1);
    r.base[t] = inside ? y0c * a.W + x0c : 0;
    if (inside && x0ok && x1ok) r.step |= 1u << (2 * t);
    if (inside && y0ok && y1ok) r.step |= 2u << (2 * t);
    unsigned v = 0;
    if (inside && y0ok && x0ok) v |= 1u;
    if (inside && y0ok && x1ok) v |= 2u;
    if (inside && y1ok && x0ok) v |= 4u;
    if (inside && y1ok && x1ok) v |= 8u;
    if (t < 8) r.valid_lo |= v << (4 * t);
    else r.valid_hi = v;
  }
}

// ------------------------------------------------------------- data kernel ---
__global__ __launch_bounds__(256) void dcn_bwd_data_kernel(DcnBwdArgs a) {
  extern __shared__ float lds[];
  float* goT = lds;                       // [BM][LDC]    grad_out slab, row = pixel
  float* wT = goT + BM * LDC;             // [COC][LDK]   weight slab for this chunk
  float* gcT = wT + COC * LDK;            // [BM][48+1]   grad columns of this chunk
  constexpr int LDG = 49;
  float* red = lds;                       // [4][27][64]  cross-wave reduction (aliases goT/wT
                                          //              after the channel loop)

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.y;
  const int HWo = a.Ho * a.Wo, HW = a.H * a.W;
  const int p = blockIdx.x * BM + lane;
  const bool p_ok = p < HWo;
  const int Ktot = a.Cin * TAPS;

  Recipe r;
  build_recipe(a, b, p, p_ok, r);
  float gm[TAPS], gy[TAPS], gxo[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) gm[t] = gy[t] = gxo[t] = 0.f;

  const float* gob = a.go + (long long)b * a.Cout * HWo;
  const float* xb = a.x + (long long)b * a.Cin * HW;
  float* gxb = a.gx ? a.gx + (long long)b * a.Cin * HW : nullptr;

  for (int c0 = 0; c0 < a.Cin; c0 += KC) {
    // gcol tile [64 px][36] = go[64][Cout] * W[Cout][36], Cout walked in slabs of COC.
    // wave w owns m-tile w (16 pixels) and all three 16-wide n-tiles.
    f32x4 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int co0 = 0; co0 < a.Cout; co0 += COC) {
      __syncthreads();
      for (int idx = tid; idx < COC * BM; idx += 256) {      // go slab, lanes along pixels
        const int co = idx / BM, pp = idx - co * BM;
        const int pg = blockIdx.x * BM + pp;
        float v = 0.f;
        if (co0 + co < a.Cout && pg < HWo) v = gob[(long long)(co0 + co) * HWo + pg];
        goT[pp * LDC + co] = v;
      }
      for (int idx = tid; idx < COC * KK; idx += 256) {      // weight slab
        const int co = idx / KK, kk = idx - co * KK;
        const int kg = c0 * TAPS + kk;
        float v = 0.f;
        if (co0 + co < a.Cout && kg < Ktot) v = a.weight[(long long)(co0 + co) * Ktot + kg];
        wT[co * LDK + kk] = v;
      }
      __syncthreads();
      // A[row = pixel][k = co], B[k = co][col = kk]
      const int arow = (wid * 16 + (lane & 15)) * LDC + (lane >> 4);
#pragma unroll 4
      for (int ks = 0; ks < COC / 4; ++ks) {
        const float af = goT[arow + ks * 4];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int col = j * 16 + (lane & 15);
          const float bf = col < KK ? wT[(ks * 4 + (lane >> 4)) * LDK + col] : 0.f;
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[j], 0, 0, 0);
        }
      }
    }
    // C layout: col = lane&15 (kk), row = (lane>>4)*4 + reg (pixel within the m-tile)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        gcT[(wid * 16 + (lane >> 4) * 4 + q) * LDG + j * 16 + (lane & 15)] = acc[j][q];
    __syncthreads();

    // ---- per-lane consumption: wave w handles channel c0 + w ----
    const int c = c0 + wid;
    if (c < a.Cin && p_ok) {
      const float* xc = xb + (long long)c * HW;
      float g[TAPS][4];
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int dx = (r.step >> (2 * t)) & 1;
        const int dy = ((r.step >> (2 * t + 1)) & 1) ? a.W : 0;
        const float* q = xc + r.base[t];
        g[t][0] = q[0];
        g[t][1] = q[dx];
        g[t][2] = q[dy];
        g[t][3] = q[dy + dx];
      }
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const unsigned vb = corner_bits(r, t);
        const float v00 = (vb & 1u) ? g[t][0] : 0.f, v01 = (vb & 2u) ? g[t][1] : 0.f;
        const float v10 = (vb & 4u) ? g[t][2] : 0.f, v11 = (vb & 8u) ? g[t][3] : 0.f;
        const float ly = r.ly[t], lx = r.lx[t], hy = 1.f - ly, hx = 1.f - lx;
        const float gc = gcT[lane * LDG + wid * TAPS + t];
        const float val = hy * hx * v00 + hy * lx * v01 + ly * hx * v10 + ly * lx * v11;
        gm[t] += gc * val;
        const float gcm = gc * r.m[t];
        gy[t] += gcm * (hx * (v10 - v00) + lx * (v11 - v01));
        gxo[t] += gcm * (hy * (v01 - v00) + ly * (v11 - v10));
        if (gxb && vb) {
          const int dx = (r.step >> (2 * t)) & 1;
          const int dy = ((r.step >> (2 * t + 1)) & 1) ? a.W : 0;
          float* q = gxb + (long long)c * HW + r.base[t];
          if (vb & 1u) atomicAdd(q, gcm * hy * hx);
          if (vb & 2u) atomicAdd(q + dx, gcm * hy * lx);
          if (vb & 4u) atomicAdd(q + dy, gcm * ly * hx);
          if (vb & 8u) atomicAdd(q + dy + dx, gcm * ly * lx);
        }
      }
    }
  }

  // ---- reduce the 4 waves' per-pixel sums and store offset / mask gradients ----
  __syncthreads();
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    red[(wid * 27 + t) * 64 + lane] = gy[t];
    red[(wid * 27 + 9 + t) * 64 + lane] = gxo[t];
    // (mask-logit chain rule applied here with a STATIC tap index: a runtime-indexed read of
    // the recipe would push the whole struct to scratch memory)
    red[(wid * 27 + 18 + t) * 64 + lane] = a.mask_is_logit ? gm[t] * r.m[t] * (1.f - r.m[t]) : gm[t];
  }
  __syncthreads();
  if (p_ok) {
    for (int q = wid; q < 27; q += 4) {
      const float v = red[(0 * 27 + q) * 64 + lane] + red[(1 * 27 + q) * 64 + lane] +
                      red[(2 * 27 + q) * 64 + lane] + red[(3 * 27 + q) * 64 + lane];
      if (q < 18) {
        if (a.goff) {
          const int t = q < 9 ? q : q - 9;
          const int ch = q < 9 ? 2 * t : 2 * t + 1;       // (dy, dx) interleaved
          a.goff[(long long)b * a.goff_bstride + (long long)ch * HWo + p] = v;
        }
      } else if (a.gmask) {
        a.gmask[(long long)b * a.gmask_bstride + (long long)(q - 18) * HWo + p] = v;
      }
    }
  }
}

// Exact round-to-nearest float -> int64 for |x| < 2^51 in 3 VALU instructions (cvt to double,
// add 2^52 + 2^51, integer-subtract the constant's bit pattern); __float2ll_rn is 12.
__device__ __forceinline__ unsigned long long fx_from_float(float x) {
  const double d = (double)x + 6755399441055744.0;
  return (unsigned long long)(__double_as_longlong(d) - 0x4338000000000000ll);
}

// ------------------------------------------------------- tiled data kernel ---
constexpr int RR = 3;                    // region halo (pixels)
constexpr int RH = 2 * RR + 2;           // 8 rows
constexpr int RW = BM + 2 * RR + 1;      // 71 columns
constexpr int RWP = 72;                  // padded row
constexpr int RSZ = RH * RWP;            // 576 floats per channel

// Slim per-tap recipe of the LDS-region kernels: bilinear fractions, mask, and ONE integer that is
// either the region offset of the tap's top-left corner (>= 0), -2 (contributes nothing), or, for
// a tap whose corners leave the region, -(3 + (clamped top-left index << 4 | corner validity
// bits)) -- what the cold fallback path needs, without nine more registers in the hot loop.
__device__ __forceinline__ void slim_recipe_load(const DcnBwdArgs& a, int b, int p, bool p_ok,
                                                 float (&raw)[3 * TAPS]) {
  const int HWo = a.Ho * a.Wo;
  const float* off = a.offset + (long long)b * a.offset_bstride;
  const float* msk = a.mask + (long long)b * a.mask_bstride;
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    raw[3 * t] = p_ok ? off[(long long)(2 * t) * HWo + p] : 0.f;
    raw[3 * t + 1] = p_ok ? off[(long long)(2 * t + 1) * HWo + p] : 0.f;
    raw[3 * t + 2] = p_ok ? msk[(long long)t * HWo + p] : 0.f;
  }
}

__device__ __forceinline__ void slim_recipe_build(const DcnBwdArgs& a, bool p_ok, int ty, int tx0, int lane,
                                                  const float (&raw)[3 * TAPS], float (&rly)[TAPS],
                                                  float (&rlx)[TAPS], float (&rm)[TAPS], int (&rbase)[TAPS]) {
  const int ry0 = ty - RR, rx0 = tx0 - RR;
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    const int ky = t / 3, kx = t - ky * 3;
    float m = raw[3 * t + 2];
    if (a.mask_is_logit) m = 1.f / (1.f + __expf(-m));
    if (!p_ok) m = 0.f;
    const float py = (float)(ty * a.stride - a.pad + ky * a.dil) + raw[3 * t];
    const float px = (float)((tx0 + lane) * a.stride - a.pad + kx * a.dil) + raw[3 * t + 1];
    const bool inside = p_ok && py > -1.f && px > -1.f && py < (float)a.H && px < (float)a.W;
    const float fy = floorf(py), fx = floorf(px);
    const int y0 = (int)fy, x0 = (int)fx;
    rly[t] = py - fy;
    rlx[t] = px - fx;
    rm[t] = m;
    const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= a.H - 1;
    const bool x0ok = x0 >= 0, x1ok = x0 + 1 <= a.W - 1;
    const int y0c = min(max(y0, 0), a.H - 1), x0c = min(max(x0, 0), a.W - 1);
    const int vb = (y0ok && x0ok ? 1 : 0) | (y0ok && x1ok ? 2 : 0) | (y1ok && x0ok ? 4 : 0) | (y1ok && x1ok ? 8 : 0);
    const int ry = y0 - ry0, rx = x0 - rx0;
    const bool in_region = inside && ry >= 0 && ry + 1 < RH && rx >= 0 && rx + 1 < RW;
    rbase[t] = in_region ? ry * RWP + rx : (inside ? -(3 + (((y0c * a.W + x0c) << 4) | vb)) : -2);
  }
}

__device__ __forceinline__ void slim_recipe(const DcnBwdArgs& a, int b, int p, bool p_ok, int ty, int tx0,
                                            int lane, float (&rly)[TAPS], float (&rlx)[TAPS],
                                            float (&rm)[TAPS], int (&rbase)[TAPS]) {
  float raw[3 * TAPS];
  slim_recipe_load(a, b, p, p_ok, raw);
  slim_recipe_build(a, p_ok, ty, tx0, lane, raw, rly, rlx, rm, rbase);
}

// Decode of a fallback tap (rb <= -3): clamped top-left index, validity bits, and the column / row
// steps (a step exists when some row / column has both of its corners valid).
__device__ __forceinline__ void fallback_decode(int rb, int W, int& fbase, unsigned& vb, int& dx, int& dy) {
  const int code = -rb - 3;
  vb = (unsigned)(code & 15);
  fbase = code >> 4;
  dx = ((vb & 3u) == 3u || (vb & 12u) == 12u) ? 1 : 0;
  dy = ((vb & 5u) == 5u || (vb & 10u) == 10u) ? W : 0;
}

template <int CP, int WPS>                // Cout rounded up to 64/128/256; waves per SIMD
__global__ __launch_bounds__(256, WPS) void dcn_bwd_data_tiled_kernel(DcnBwdArgs a) {
  constexpr int LDO = CP + 2;             // stride 2 (mod 32): the lane = (row, k) fragment reads are conflict-free
  constexpr int WPT = (CP * KK + 255) / 256;
  constexpr int LDG = 49;
  constexpr int LDW = 49;                 // weight rows padded to 48 columns (+1): the three
                                          // 16-wide n-tiles read unconditionally, cols 36..47 = 0
  extern __shared__ float lds[];
  float* goT = lds;                       // [BM][LDO]
  float* wT = goT + BM * LDO;             // [CP][LDK]
  float* gcT = wT + CP * LDW;             // [BM][LDG]
  float* xreg = gcT + BM * LDG;           // [KC][RSZ]
  float* smax = xreg + KC * RSZ;          // [4] per-wave max |gcol| (+ pad to 8-byte alignment)
  unsigned long long* greg = (unsigned long long*)(smax + 4 + ((BM * LDO + CP * LDW + BM * LDG) & 1));
                                          // [KC][RSZ] fixed-point accumulators
  float* red = lds;                       // aliases goT after the loop

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.y;
  const int HWo = a.Ho * a.Wo, HW = a.H * a.W;
  // the tile is a segment of up to 64 pixels of ONE row (the last tile of a row is partial
  // when W is not a multiple of 64: its surplus lanes carry an empty recipe)
  const int ty = blockIdx.x / a.tpr, tx0 = (blockIdx.x - ty * a.tpr) * BM;
  const int p0 = ty * a.W + tx0;
  const bool p_ok = tx0 + lane < a.W;
  const int p = p_ok ? p0 + lane : p0;
  const int Ktot = a.Cin * TAPS;
  const int ry0 = ty - RR, rx0 = tx0 - RR;

  float rly[TAPS], rlx[TAPS], rm[TAPS];
  int rbase[TAPS];
  slim_recipe(a, b, p, p_ok, ty, tx0, lane, rly, rlx, rm, rbase);
  bool any_fallback = false;
#pragma unroll
  for (int t = 0; t < TAPS; ++t) any_fallback |= rbase[t] <= -3;
  float gm[TAPS], gy[TAPS], gxo[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) gm[t] = gy[t] = gxo[t] = 0.f;

  const float* gob = a.go + (long long)b * a.Cout * HWo;
  const float* xb = a.x + (long long)b * a.Cin * HW;
  float* gxb = a.gx ? a.gx + (long long)b * a.Cin * HW : nullptr;

  // grad_out tile, staged once: goT[px][co]
  for (int idx = tid; idx < CP * BM; idx += 256) {
    const int co = idx / BM, pp = idx - co * BM;
    goT[pp * LDO + co] = (co < a.Cout && tx0 + pp < a.W) ? gob[(long long)co * HWo + p0 + pp] : 0.f;
  }
  for (int e = tid; e < KC * RSZ; e += 256) greg[e] = 0ull;
  for (int e = tid; e < CP * (LDW - KK); e += 256) {       // zero the pad columns once
    const int co = e / (LDW - KK);
    wT[co * LDW + KK + (e - co * (LDW - KK))] = 0.f;
  }

  // Region cells and weights are read with raw buffer loads: wave-uniform descriptor, per-lane
  // 32-bit byte offset fixed for the whole channel loop (cells outside the image carry an
  // out-of-range offset and read 0), channel plane in the scalar offset (clamped: soffset is not
  // range-checked).  Wave w stages, consumes and flushes the region of ITS channel c0 + w.
  constexpr int RPW = RSZ / 64;            // region cells per lane (9)
  static_assert(RSZ % 64 == 0, "one wave sweeps a region in whole passes");
  const unsigned plane_bytes = (unsigned)HW * 4u;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xb), 0, (int)((unsigned)a.Cin * plane_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.weight), 0, (int)((unsigned)a.Cout * (unsigned)Ktot * 4u), 0x00020000);
  unsigned roff[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int e = lane + 64 * i;
    const int ry = e / RWP, rx = e - ry * RWP;
    const int gy_ = ry0 + ry, gx_ = rx0 + rx;
    const bool ok = rx < RW && gy_ >= 0 && gy_ < a.H && gx_ >= 0 && gx_ < a.W;
    roff[i] = ok ? 4u * (unsigned)(gy_ * a.W + gx_) : 0xf0000000u;
  }
  unsigned woff[WPT];
#pragma unroll
  for (int i = 0; i < WPT; ++i) {
    const int idx = tid + i * 256;
    const int co = idx / KK, kk = idx - co * KK;
    woff[i] = (idx < CP * KK && co < a.Cout) ? ((unsigned)co * (unsigned)Ktot + (unsigned)kk) * 4u : 0xf0000000u;
  }
  const int swid = __builtin_amdgcn_readfirstlane(wid);
  float wreg[WPT], xr[RPW];
  auto issue = [&](int c0) {
    const unsigned wk = (unsigned)(c0 * TAPS) * 4u;       // rides in voffset (range-checked)
#pragma unroll
    for (int i = 0; i < WPT; ++i)
      wreg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_w, woff[i] + wk, 0, 0));
    const unsigned xsoff = (unsigned)min(c0 + swid, a.Cin - 1) * plane_bytes;
#pragma unroll
    for (int i = 0; i < RPW; ++i)
      xr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, roff[i], xsoff, 0));
  };
  issue(0);
  __syncthreads();                         // goT / greg / pad columns initialised
  for (int c0 = 0; c0 < a.Cin; c0 += KC) {
    // Two barriers per chunk.  wT is only read in the MFMA phase (closed by the barrier after
    // it), xreg / greg belong to ONE wave (LDS operations of a wave complete in order), and
    // gcT / smax are rewritten only after the staging barrier, which every wave reaches after
    // it has consumed the previous chunk.
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int idx = tid + i * 256;
      if (idx < CP * KK) {
        const int co = idx / KK;
        wT[co * LDW + (idx - co * KK)] = wreg[i];
      }
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) xreg[swid * RSZ + lane + 64 * i] = xr[i];
    __syncthreads();
    if (c0 + KC < a.Cin && !CP_ABL(8)) issue(c0 + KC);   // next chunk's loads fly during the MFMA phase

    // gcol tile [64 px][36] = goT[64][Cout] * wT[Cout][36]; wave w owns m-tile w
    f32x4 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int arow = (wid * 16 + (lane & 15)) * LDO + (lane >> 4);
    const int brow = (lane >> 4) * LDW + (lane & 15);
#pragma unroll 8
    for (int ks = 0; ks < (CP_ABL(4) ? 0 : CP / 4); ++ks) {
      const float af = goT[arow + ks * 4];
      const float b0 = wT[brow + ks * 4 * LDW];
      const float b1 = wT[brow + ks * 4 * LDW + 16];
      const float b2 = wT[brow + ks * 4 * LDW + 32];
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, b2, acc[2], 0, 0, 0);
    }
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        gcT[(wid * 16 + (lane >> 4) * 4 + q) * LDG + j * 16 + (lane & 15)] = acc[j][q];
        amax = fmaxf(amax, fabsf(acc[j][q]));
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (lane == 0) smax[wid] = amax;
    __syncthreads();
    // fixed-point scale of this chunk: 2^(50 - exponent(max|gcol|)); |contribution| <= max|gcol|
    const float gmax = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    int gexp = 0;
    (void)frexpf(gmax, &gexp);
    const float fx_scale = ldexpf(1.f, 50 - gexp);
    const double fx_inv = ldexp(1.0, gexp - 50);

    // ---- consumption: wave w handles channel c0 + w ----
    const int c = c0 + wid;
    if (c < a.Cin && !CP_ABL(1) && gmax > 0.f) {
      const float* xw = xreg + wid * RSZ;
      unsigned long long* gw_ = greg + wid * RSZ;
      // hot loop: taps served by the LDS region (one divergent `if`, no fallback state live)
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int rb = rbase[t];
        if (rb >= 0) {                     // out-of-image cells of the region hold 0
          const float v00 = xw[rb], v01 = xw[rb + 1], v10 = xw[rb + RWP], v11 = xw[rb + RWP + 1];
          float ly = rly[t], lx = rlx[t];
          // opaque to the optimiser: otherwise 1-ly, 1-lx and the four corner products of all
          // nine taps are hoisted out of the channel loop and held in ~50 registers (spills)
          asm volatile("" : "+v"(ly), "+v"(lx));
          const float hy = 1.f - ly, hx = 1.f - lx;
          const float gc = gcT[lane * LDG + wid * TAPS + t];
          gm[t] += gc * (hy * hx * v00 + hy * lx * v01 + ly * hx * v10 + ly * lx * v11);
          const float gcm = gc * rm[t];
          gy[t] += gcm * (hx * (v10 - v00) + lx * (v11 - v01));
          gxo[t] += gcm * (hy * (v01 - v00) + ly * (v11 - v10));
          if (gxb) {                       // cells outside the image are dropped at the flush
            const float gs = gcm * fx_scale;
            atomicAdd(&gw_[rb], fx_from_float(gs * (hy * hx)));
            atomicAdd(&gw_[rb + 1], fx_from_float(gs * (hy * lx)));
            atomicAdd(&gw_[rb + RWP], fx_from_float(gs * (ly * hx)));
            atomicAdd(&gw_[rb + RWP + 1], fx_from_float(gs * (ly * lx)));
          }
        }
      }
      // cold loop (skipped wave-uniformly when no lane of the tile has such a tap): corners that
      // leave the region go to memory.  rb <= -3 packs the clamped top-left index and the corner
      // validity bits; a column / row step exists when some row / column has both corners valid.
      if (__builtin_amdgcn_ballot_w64(any_fallback) != 0ull && !CP_ABL(16)) {
        const float* xc = xb + (long long)c * HW;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          const int rb = rbase[t];
          if (rb > -3) continue;
          int fbase, dx, dy;
          unsigned vb;
          fallback_decode(rb, a.W, fbase, vb, dx, dy);
          const float* q = xc + fbase;
          const float v00 = (vb & 1u) ? q[0] : 0.f;
          const float v01 = (vb & 2u) ? q[dx] : 0.f;
          const float v10 = (vb & 4u) ? q[dy] : 0.f;
          const float v11 = (vb & 8u) ? q[dy + dx] : 0.f;
          const float ly = rly[t], lx = rlx[t], hy = 1.f - ly, hx = 1.f - lx;
          const float gc = gcT[lane * LDG + wid * TAPS + t];
          gm[t] += gc * (hy * hx * v00 + hy * lx * v01 + ly * hx * v10 + ly * lx * v11);
          const float gcm = gc * rm[t];
          gy[t] += gcm * (hx * (v10 - v00) + lx * (v11 - v01));
          gxo[t] += gcm * (hy * (v01 - v00) + ly * (v11 - v10));
          if (gxb && vb) {
            float* g = gxb + (long long)c * HW + fbase;
            if (vb & 1u) atomicAdd(g, gcm * hy * hx);
            if (vb & 2u) atomicAdd(g + dx, gcm * hy * lx);
            if (vb & 4u) atomicAdd(g + dy, gcm * ly * hx);
            if (vb & 8u) atomicAdd(g + dy + dx, gcm * ly * lx);
          }
        }
      }
    }
    // ---- flush: each wave empties its own channel's region (no barrier: wave-private), one
    // coalesced global atomic per touched in-image cell (the cell's byte offset is the staging
    // offset) ----
    if (gxb && !CP_ABL(2) && c < a.Cin) {
      unsigned long long* gw_ = greg + swid * RSZ;
      float* gplane = gxb + (long long)c * HW;
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const long long qv = (long long)gw_[lane + 64 * i];
        if (qv != 0) {
          gw_[lane + 64 * i] = 0ull;
          if (roff[i] != 0xf0000000u)
            atomicAdd(gplane + (roff[i] >> 2), (float)((double)qv * fx_inv));
        }
      }
    }
  }

  __syncthreads();
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    red[(wid * 27 + t) * 64 + lane] = gy[t];
    red[(wid * 27 + 9 + t) * 64 + lane] = gxo[t];
    // (mask-logit chain rule applied here with a STATIC tap index: a runtime-indexed read of
    // the recipe would push the whole struct to scratch memory)
    red[(wid * 27 + 18 + t) * 64 + lane] = a.mask_is_logit ? gm[t] * rm[t] * (1.f - rm[t]) : gm[t];
  }
  __syncthreads();
  for (int q = wid; q < 27; q += 4) {
    const float v = red[(0 * 27 + q) * 64 + lane] + red[(1 * 27 + q) * 64 + lane] +
                    red[(2 * 27 + q) * 64 + lane] + red[(3 * 27 + q) * 64 + lane];
    if (!p_ok) continue;
    if (q < 18) {
      if (a.goff) {
        const int t = q < 9 ? q : q - 9;
        const int ch = q < 9 ? 2 * t : 2 * t + 1;
        a.goff[(long long)b * a.goff_bstride + (long long)ch * HWo + p] = v;
      }
    } else if (a.gmask) {
      a.gmask[(long long)b * a.gmask_bstride + (long long)(q - 18) * HWo + p] = v;
    }
  }
}

template <int CP, int WPS>
void launch_tiled(const DcnBwdArgs& a, int tiles, hipStream_t st) {
  const size_t lds = (size_t)(BM * (CP + 2) + CP * 49 + BM * 49 + KC * RSZ + 6 + 2 * KC * RSZ) * sizeof(float);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)dcn_bwd_data_tiled_kernel<CP, WPS>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((dcn_bwd_data_tiled_kernel<CP, WPS>), dim3(tiles, a.B), dim3(256), lds, st, a);
}

// ----------------------------------------------------------- weight kernel ---
// gW[co][k] += sum over this workgroup's 64 pixels of go[px][co] * col[px][k].
__global__ __launch_bounds__(256) void dcn_bwd_weight_kernel(DcnBwdArgs a) {
  extern __shared__ float lds[];
  float* goT = lds;                       // [BM][LDC]   row = pixel, col = co (slab of COC)
  float* colT = goT + BM * LDC;           // [BM][LDK]   row = pixel, col = kk

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.y;
  const int co0 = blockIdx.z * COC;
  const int HWo = a.Ho * a.Wo, HW = a.H * a.W;
  const int p = blockIdx.x * BM + lane;
  const bool p_ok = p < HWo;
  const int Ktot = a.Cin * TAPS;

  Recipe r;
  build_recipe(a, b, p, p_ok, r);
  const float* gob = a.go + (long long)b * a.Cout * HWo;
  const float* xb = a.x + (long long)b * a.Cin * HW;

  for (int idx = tid; idx < COC * BM; idx += 256) {
    const int co = idx / BM, pp = idx - co * BM;
    const int pg = blockIdx.x * BM + pp;
    float v = 0.f;
    if (co0 + co < a.Cout && pg < HWo) v = gob[(long long)(co0 + co) * HWo + pg];
    goT[pp * LDC + co] = v;
  }

  for (int c0 = 0; c0 < a.Cin; c0 += KC) {
    __syncthreads();
    const int c = c0 + wid;
    {
      const bool c_ok = c < a.Cin && p_ok;
      const float* xc = xb + (long long)(c < a.Cin ? c : 0) * HW;
      float g[TAPS][4];
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int dx = (r.step >> (2 * t)) & 1;
        const int dy = ((r.step >> (2 * t + 1)) & 1) ? a.W : 0;
        const float* q = xc + r.base[t];
        g[t][0] = q[0];
        g[t][1] = q[dx];
        g[t][2] = q[dy];
        g[t][3] = q[dy + dx];
      }
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const unsigned vb = corner_bits(r, t);
        const float ly = r.ly[t], lx = r.lx[t], hy = 1.f - ly, hx = 1.f - lx;
        const float val = ((vb & 1u) ? hy * hx * g[t][0] : 0.f) + ((vb & 2u) ? hy * lx * g[t][1] : 0.f) +
                          ((vb & 4u) ? ly * hx * g[t][2] : 0.f) + ((vb & 8u) ? ly * lx * g[t][3] : 0.f);
        colT[lane * LDK + wid * TAPS + t] = c_ok ? val * r.m[t] : 0.f;
      }
    }
    __syncthreads();
    // D[co][kk] = sum_px goT[px][co] * colT[px][kk]:  A[row = co][k = px], B[k = px][col = kk]
    // wave w owns co rows [w*32, w*32+32) of the slab: 2 m-tiles x 3 n-tiles.
    f32x4 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int ks = 0; ks < BM / 4; ++ks) {
      const int px = ks * 4 + (lane >> 4);
      float af[2], bf[3];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = goT[px * LDC + wid * 32 + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int col = j * 16 + (lane & 15);
        bf[j] = col < KK ? colT[px * LDK + col] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    // C layout: col = lane&15 (kk), row = (lane>>4)*4 + reg (co within the m-tile)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int kk = j * 16 + (lane & 15);
        const int kg = c0 * TAPS + kk;
        if (kk >= KK || kg >= Ktot) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = co0 + wid * 32 + i * 16 + (lane >> 4) * 4 + q;
          if (co < a.Cout) atomicAdd(&a.gw[(long long)co * Ktot + kg], acc[i][j][q]);
        }
      }
  }
}

// ------------------------------------------------------ tiled weight kernel ---
// One workgroup = (group of T consecutive 64-pixel tiles of one image) x (slice of WSC input
// channels) x (slab of 128 output channels).  The gW tile [128 co][WSC*9] stays in registers
// across all T tiles and is flushed with float atomics ONCE (T x fewer atomics than a flush per
// tile); corners come from the LDS-staged input region (coalesced row loads, prefetched one
// chunk ahead), with a global-gather fallback for taps that leave the region.
constexpr int WSC = 16;                   // input channels per workgroup
constexpr int WCH = WSC / KC;             // chunks per workgroup (4)

struct WTiledExtra {
  int T;                                  // tiles per group
  int groups_per_image;
};

template <int SLAB, int WPS>                // output channels per workgroup: 64 or 128
__global__ __launch_bounds__(256, WPS) void dcn_bwd_weight_tiled_kernel(DcnBwdArgs a, WTiledExtra ex) {
  constexpr int MT = SLAB / 64;            // 16-row m-tiles per wave
  constexpr int LDS_ = SLAB + 1;
  extern __shared__ float lds[];
  float* goT = lds;                       // [BM][LDS_]
  float* colT = goT + BM * LDS_;           // [BM][LDW2]  columns of one chunk, padded to 48
  constexpr int LDW2 = 49;
  float* xreg = colT + BM * LDW2;         // [KC][RSZ]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.x / ex.groups_per_image;
  const int grp = blockIdx.x - b * ex.groups_per_image;
  const int cs0 = blockIdx.y * WSC;       // first input channel of this slice
  const int co0 = blockIdx.z * SLAB;
  const int HWo = a.Ho * a.Wo, HW = a.H * a.W;
  const int Ktot = a.Cin * TAPS;
  const int tiles = a.H * a.tpr;
  const float* gob = a.go + (long long)b * a.Cout * HWo;
  const float* xb = a.x + (long long)b * a.Cin * HW;
  // raw buffer loads (see the data kernel): per-lane 32-bit offsets, range-checked by hardware
  constexpr int RPW = RSZ / 64;
  const unsigned plane_bytes = (unsigned)HW * 4u;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xb), 0, (int)((unsigned)a.Cin * plane_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_go = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(gob), 0, (int)((unsigned)a.Cout * (unsigned)HWo * 4u), 0x00020000);
  const int swid = __builtin_amdgcn_readfirstlane(wid);

  f32x4 acc[WCH][MT][3];
#pragma unroll
  for (int h = 0; h < WCH; ++h)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int e = tid; e < BM * (LDW2 - KK); e += 256) {          // zero the pad columns once
    const int pp = e / (LDW2 - KK);
    colT[pp * LDW2 + KK + (e - pp * (LDW2 - KK))] = 0.f;
  }

  const int t_begin = grp * ex.T, t_end = min(tiles, t_begin + ex.T);
  // the 64 slab has the registers to fetch the NEXT tile's offsets / mask while this tile is
  // processed (27 loads whose latency otherwise opens every tile)
  constexpr bool PREFETCH = SLAB == 64;
  float raw[3 * TAPS];
  auto tile_geom = [&](int tile, int& ty, int& tx0, int& p, bool& p_ok) {
    ty = tile / a.tpr;
    tx0 = (tile - ty * a.tpr) * BM;
    p_ok = tx0 + lane < a.W;                // partial last tile of a row when W % 64 != 0
    p = ty * a.W + tx0 + (p_ok ? lane : 0);
  };
  if (PREFETCH && t_begin < t_end) {
    int ty, tx0, p; bool p_ok;
    tile_geom(t_begin, ty, tx0, p, p_ok);
    slim_recipe_load(a, b, p, p_ok, raw);
  }
  for (int tile = t_begin; tile < t_end; ++tile) {
    int ty, tx0, p; bool p_ok;
    tile_geom(tile, ty, tx0, p, p_ok);
    const int p0 = ty * a.W + tx0;
    const int ry0 = ty - RR, rx0 = tx0 - RR;
    float rly[TAPS], rlx[TAPS], rm[TAPS];
    int rbase[TAPS];
    if (!PREFETCH) slim_recipe_load(a, b, p, p_ok, raw);
    slim_recipe_build(a, p_ok, ty, tx0, lane, raw, rly, rlx, rm, rbase);
    if (PREFETCH && tile + 1 < t_end) {
      int ty2, tx2, p2; bool ok2;
      tile_geom(tile + 1, ty2, tx2, p2, ok2);
      slim_recipe_load(a, b, p2, ok2, raw);
    }
    bool any_fallback = false;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) any_fallback |= rbase[t] <= -3;
    unsigned roff[RPW];                    // this wave's region cells (channel c0 + w), per tile
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int e = lane + 64 * i;
      const int ry = e / RWP, rx = e - ry * RWP;
      const int gy_ = ry0 + ry, gx_ = rx0 + rx;
      const bool ok = rx < RW && gy_ >= 0 && gy_ < a.H && gx_ >= 0 && gx_ < a.W;
      roff[i] = ok ? 4u * (unsigned)(gy_ * a.W + gx_) : 0xf0000000u;
    }
    float xr[RPW];
    auto issue = [&](int c0) {
      const unsigned xsoff = (unsigned)min(c0 + swid, a.Cin - 1) * plane_bytes;
#pragma unroll
      for (int i = 0; i < RPW; ++i)
        xr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, roff[i], xsoff, 0));
    };
    issue(cs0);
    __syncthreads();                       // previous tile's MFMA reads of goT/colT are done
    {                                      // grad_out tile: thread -> pixel lane, rows wid, wid+4, ...
      // lanes of a partial tile keep one out-of-range offset for every row (reads 0)
      const unsigned gbase = p_ok ? ((unsigned)(co0 + swid) * (unsigned)HWo + (unsigned)(p0 + lane)) * 4u
                                  : 0xf0000000u;
      const unsigned gstep = p_ok ? (unsigned)HWo * 16u : 0u;       // 4 rows of grad_out
#pragma unroll
      for (int i = 0; i < SLAB / 4; ++i)   // rows past Cout are past num_records and read 0
        goT[lane * LDS_ + swid + 4 * i] = __builtin_bit_cast(
            float, __builtin_amdgcn_raw_buffer_load_b32(rs_go, gbase + (unsigned)i * gstep, 0, 0));
    }
#pragma unroll
    for (int h = 0; h < WCH; ++h) {
      const int c0 = cs0 + h * KC;
      __syncthreads();                     // colT free again (xreg is wave-private: no barrier
#pragma unroll                             // between its store and the reads below)
      for (int i = 0; i < RPW; ++i) xreg[swid * RSZ + lane + 64 * i] = xr[i];
      if (h + 1 < WCH) issue(c0 + KC);     // next chunk's region loads fly during sampling + MFMA
      // ---- sample: wave w handles channel c0 + w ----
      // (the 64-channel slab has the registers for the batched, branch-free form: -11 % on
      // 64->64 @256x512; on the 128 slab the same form cost +13..26 %, it keeps the per-tap form)
      if constexpr (SLAB == 64)
      {
        const int c = c0 + wid;
        const bool c_ok = c < a.Cin;
        const float* xw = xreg + wid * RSZ;
        // hot pass, branch-free: all 36 region reads are issued back to back (clamped address,
        // the result is dropped by a select when the tap is not served by the region)
        float v[TAPS][4];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          const int rbc = max(rbase[t], 0);
          v[t][0] = xw[rbc];
          v[t][1] = xw[rbc + 1];
          v[t][2] = xw[rbc + RWP];
          v[t][3] = xw[rbc + RWP + 1];
        }
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          const float ly = rly[t], lx = rlx[t], hy = 1.f - ly, hx = 1.f - lx;
          const float val = (hy * hx * v[t][0] + hy * lx * v[t][1] + ly * hx * v[t][2] + ly * lx * v[t][3]) * rm[t];
          colT[lane * LDW2 + wid * TAPS + t] = (rbase[t] >= 0 && c_ok) ? val : 0.f;
        }
        // cold pass (wave-uniform skip): taps whose corners leave the region gather from memory
        if (c_ok && __builtin_amdgcn_ballot_w64(any_fallback) != 0ull) {
          const float* xc = xb + (long long)c * HW;
#pragma unroll
          for (int t = 0; t < TAPS; ++t) {
            if (rbase[t] > -3) continue;
            int fbase, dx, dy;
            unsigned vb;
            fallback_decode(rbase[t], a.W, fbase, vb, dx, dy);
            const float* q = xc + fbase;
            const float v00 = (vb & 1u) ? q[0] : 0.f;
            const float v01 = (vb & 2u) ? q[dx] : 0.f;
            const float v10 = (vb & 4u) ? q[dy] : 0.f;
            const float v11 = (vb & 8u) ? q[dy + dx] : 0.f;
            const float ly = rly[t], lx = rlx[t], hy = 1.f - ly, hx = 1.f - lx;
            colT[lane * LDW2 + wid * TAPS + t] =
                (hy * hx * v00 + hy * lx * v01 + ly * hx * v10 + ly * lx * v11) * rm[t];
          }
        }
      }
      else
      {                                   // per-tap form, fallback taps in a wave-uniformly skipped loop
        const int c = c0 + wid;
        const bool c_ok = c < a.Cin;
        const float* xw = xreg + wid * RSZ;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          const int rb = rbase[t];
          float val = 0.f;
          if (rb >= 0 && c_ok) {
            const float v00 = xw[rb], v01 = xw[rb + 1], v10 = xw[rb + RWP], v11 = xw[rb + RWP + 1];
            const float ly = rly[t], lx = rlx[t], hy = 1.f - ly, hx = 1.f - lx;
            val = (hy * hx * v00 + hy * lx * v01 + ly * hx * v10 + ly * lx * v11) * rm[t];
          }
          colT[lane * LDW2 + wid * TAPS + t] = val;
        }
        if (c_ok && __builtin_amdgcn_ballot_w64(any_fallback) != 0ull) {
          const float* xc = xb + (long long)c * HW;
#pragma unroll
          for (int t = 0; t < TAPS; ++t) {
            if (rbase[t] > -3) continue;
            int fbase, dx, dy;
            unsigned vb;
            fallback_decode(rbase[t], a.W, fbase, vb, dx, dy);
            const float* q = xc + fbase;
            const float v00 = (vb & 1u) ? q[0] : 0.f;
            const float v01 = (vb & 2u) ? q[dx] : 0.f;
            const float v10 = (vb & 4u) ? q[dy] : 0.f;
            const float v11 = (vb & 8u) ? q[dy + dx] : 0.f;
            const float ly = rly[t], lx = rlx[t], hy = 1.f - ly, hx = 1.f - lx;
            colT[lane * LDW2 + wid * TAPS + t] =
                (hy * hx * v00 + hy * lx * v01 + ly * hx * v10 + ly * lx * v11) * rm[t];
          }
        }
      }
      __syncthreads();
      // ---- D[co][kk] += sum_px goT[px][co] * colT[px][kk]; wave w owns co rows [16*MT*w, 16*MT*(w+1)) ----
#pragma unroll 8
      for (int ks = 0; ks < BM / 4; ++ks) {
        const int px = ks * 4 + (lane >> 4);
        const float b0 = colT[px * LDW2 + (lane & 15)];
        const float b1 = colT[px * LDW2 + 16 + (lane & 15)];
        const float b2 = colT[px * LDW2 + 32 + (lane & 15)];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const float ai = goT[px * LDS_ + wid * (16 * MT) + i * 16 + (lane & 15)];
          acc[h][i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, b0, acc[h][i][0], 0, 0, 0);
          acc[h][i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, b1, acc[h][i][1], 0, 0, 0);
          acc[h][i][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, b2, acc[h][i][2], 0, 0, 0);
        }
      }
    }
  }
  // ---- one atomic flush of the [128 co][WSC*9] tile ----
#pragma unroll
  for (int h = 0; h < WCH; ++h)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int kk = j * 16 + (lane & 15);
        const int kg = (cs0 + h * KC) * TAPS + kk;
        if (kk >= KK || kg >= Ktot) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = co0 + wid * (16 * MT) + i * 16 + (lane >> 4) * 4 + q;
          if (co < a.Cout) atomicAdd(&a.gw[(long long)co * Ktot + kg], acc[h][i][j][q]);
        }
      }
}

// grad_bias[co] += sum_{b,p} go[b][co][p]; grid = (Cout, segments of the B*HW range).
constexpr int BIAS_SEG = 16384;
__global__ __launch_bounds__(256) void dcn_bwd_bias_kernel(const float* __restrict__ go,
                                                           float* __restrict__ gb, int B, int Cout,
                                                           int HWo) {
  const int co = blockIdx.x;
  const long long total = (long long)B * HWo;
  const long long s0 = (long long)blockIdx.y * BIAS_SEG;
  const long long s1 = s0 + BIAS_SEG < total ? s0 + BIAS_SEG : total;
  float s = 0.f;
  if ((HWo & 3) == 0) {                       // 16-byte loads: a segment never straddles an image (BIAS_SEG % 4 == 0)
    for (long long i = s0 + 4 * threadIdx.x; i < s1; i += 1024) {
      const long long b = i / HWo, p = i - b * HWo;
      const float4 v = *reinterpret_cast<const float4*>(go + (b * Cout + co) * HWo + p);
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (long long i = s0 + threadIdx.x; i < s1; i += 256) {
      const long long b = i / HWo, p = i - b * HWo;
      s += go[(b * Cout + co) * HWo + p];
    }
  }
  s = cp_wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&gb[co], red[0] + red[1] + red[2] + red[3]);
}

inline int out_extent(int in, int pad, int dil, int stride) {
  return (in + 2 * pad - (dil * 2 + 1)) / stride + 1;
}

}  // namespace

// dcn_bwd_data.hip: the register-resident grad-column kernel for the data gradients
bool cp_dcn_bwd_data2_supported(const cp_dcn_shape* s);
size_t cp_dcn_bwd_data2_workspace_bytes(const cp_dcn_shape* s);
int cp_dcn_bwd_data2(const cp_dcn_shape* s, const float* x, const float* offset, int64_t offset_bstride,
                     const float* mask, int64_t mask_bstride, int32_t mask_is_logit, const float* weight,
                     const float* grad_out, float* grad_x, float* grad_offset, int64_t grad_offset_bstride,
                     float* grad_mask, int64_t grad_mask_bstride, int32_t flags, void* workspace, size_t workspace_bytes,
                     hipStream_t st);

// dcn_bwd_weight.hip: the weight gradient with columns sampled straight into the MFMA operand
bool cp_dcn_bwd_weight2_supported(const cp_dcn_shape* s);
int cp_dcn_bwd_weight2(const cp_dcn_shape* s, const float* x, const float* offset, int64_t offset_bstride,
                       const float* mask, int64_t mask_bstride, int32_t mask_is_logit, const float* grad_out,
                       float* grad_weight, float* grad_bias, int32_t flags, hipStream_t st);


extern "C" size_t cp_dcn_v2_backward_workspace_bytes(const cp_dcn_shape* s) {
  if (!s || s->kh != 3 || s->kw != 3 || s->deformable_groups != 1) return 0;
  return cp_dcn_bwd_data2_workspace_bytes(s);
}

extern "C" int cp_dcn_v2_backward(const cp_dcn_shape* s, const float* x, const float* offset,
                                  int64_t offset_bstride, const float* mask,
                                  int64_t mask_bstride, int32_t mask_is_logit,
                                  const float* weight, const float* grad_out, float* grad_x,
                                  float* grad_offset, int64_t grad_offset_bstride,
                                  float* grad_mask, int64_t grad_mask_bstride, float* grad_weight,
                                  float* grad_bias, int32_t flags, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  CP_CHECK_ARG(s && x && offset && mask && weight && grad_out);
  CP_CHECK_ARG(s->B > 0 && s->Cin > 0 && s->H > 0 && s->W > 0 && s->Cout > 0);
  CP_CHECK_ARG(s->stride > 0 && s->dil > 0 && s->pad >= 0);
  if (s->kh != 3 || s->kw != 3 || s->deformable_groups != 1) return CP_EUNSUPPORTED;
  const int Ho = out_extent(s->H, s->pad, s->dil, s->stride);
  const int Wo = out_extent(s->W, s->pad, s->dil, s->stride);
  CP_CHECK_ARG(Ho > 0 && Wo > 0);
  if ((long long)s->H * s->W >= (1ll << 31) || (long long)Ho * Wo >= (1ll << 31)) return CP_EUNSUPPORTED;
  if (s->B > 65535) return CP_EUNSUPPORTED;
  DcnBwdArgs a;
  a.x = x; a.offset = offset; a.mask = mask; a.weight = weight; a.go = grad_out;
  a.gx = grad_x; a.goff = grad_offset; a.gmask = grad_mask; a.gw = grad_weight;
  a.offset_bstride = offset_bstride; a.mask_bstride = mask_bstride;
  a.goff_bstride = grad_offset_bstride; a.gmask_bstride = grad_mask_bstride;
  a.B = s->B; a.Cin = s->Cin; a.H = s->H; a.W = s->W; a.Cout = s->Cout; a.Ho = Ho; a.Wo = Wo;
  a.stride = s->stride; a.pad = s->pad; a.dil = s->dil; a.mask_is_logit = mask_is_logit;
#ifdef CP_ABLATE
  {
    const char* e = getenv("CP_DCN_ABLATE");
    a.ablate = e ? atoi(e) : 0;
  }
#endif
  hipStream_t st = (hipStream_t)stream;
  CP_CHECK_ARG((flags & ~(CP_DCN_BWD_EXACT_F32 | CP_DCN_BWD_NARROW_TILES | CP_DCN_BWD_ROUND1_KERNELS)) == 0);
  const bool round1 = (flags & CP_DCN_BWD_ROUND1_KERNELS) != 0;
  // grad_x is OVERWRITTEN: every kernel below accumulates into it (region sums + cold-path / fallback float atomics),
  // so the library zero-fills it here and the caller may hand over uninitialised memory
  if (grad_x && hipMemsetAsync(grad_x, 0, (size_t)s->B * s->Cin * s->H * s->W * sizeof(float), st) != hipSuccess) return CP_EHIP;
  const int tiles = (Ho * Wo + BM - 1) / BM;
  a.tpr = (s->W + BM - 1) / BM;
  const int row_tiles = s->H * a.tpr;       // tiled kernels: tiles never straddle rows
  const bool same_size = s->stride == 1 && Wo == s->W && Ho == s->H &&
                         (unsigned long long)s->Cout * Ho * Wo * 4ull < 0xE0000000ull &&
                         (long long)s->H * s->W < (1ll << 27);     // fallback index packed with 4 bits
  bool data_done = false;
  if ((grad_x || grad_offset || grad_mask) && cp_dcn_bwd_data2_supported(s) && !round1 &&
      workspace && workspace_bytes >= cp_dcn_bwd_data2_workspace_bytes(s)) {
    const int rc = cp_dcn_bwd_data2(s, x, offset, offset_bstride, mask, mask_bstride, mask_is_logit, weight, grad_out,
                                    grad_x, grad_offset, grad_offset_bstride, grad_mask, grad_mask_bstride, flags, workspace,
                                    workspace_bytes, st);
    if (rc != CP_OK) return rc;
    data_done = true;
  }
  if (!data_done && (grad_x || grad_offset || grad_mask)) {
    const bool tiled = s->Cout <= 256 && same_size;
    if (tiled) {
      if (s->Cout <= 64) launch_tiled<64, 2>(a, row_tiles, st);
      else if (s->Cout <= 128) launch_tiled<128, 1>(a, row_tiles, st);
      else launch_tiled<256, 1>(a, row_tiles, st);
    } else {
      const size_t lds = (size_t)(BM * LDC + COC * LDK + BM * 49) * sizeof(float);
      hipLaunchKernelGGL(dcn_bwd_data_kernel, dim3(tiles, s->B), dim3(256), lds, st, a);
    }
  }
  bool weight_done = false;
  if (grad_weight && cp_dcn_bwd_weight2_supported(s) && !round1) {
    // (grad_bias rides along: the weight kernel has every grad_out tile in LDS anyway)
    const int rc = cp_dcn_bwd_weight2(s, x, offset, offset_bstride, mask, mask_bstride, mask_is_logit, grad_out,
                                      grad_weight, grad_bias, flags, st);
    if (rc != CP_OK) return rc;
    weight_done = true;
    grad_bias = nullptr;
  }
  if (grad_weight && !weight_done) {
    const bool tiled_w = same_size;
    const int slab = s->Cout <= 64 ? 64 : COC;
    const int slabs = (s->Cout + slab - 1) / slab;
    if (tiled_w) {
      const int slices = (s->Cin + WSC - 1) / WSC;
      const long long blocks1 = (long long)row_tiles * s->B * slices * slabs;
      // tiles per workgroup: as many as leave ONE resident round of workgroups (256 CUs x 2 per
      // CU): every extra tile amortises the accumulator flush and the per-workgroup set-up, a second
      // round only adds a tail (swept on the GPU: 512 workgroups beat 1024 / 2048 by 5..30 %)
      int T = (int)((blocks1 + 511) / 512);
      if (T < 1) T = 1;
      if (T > 64) T = 64;
      if (T > row_tiles) T = row_tiles;
      WTiledExtra ex;
      ex.T = T;
      ex.groups_per_image = (row_tiles + T - 1) / T;
      const size_t lds = (size_t)(BM * (slab + 1) + BM * 49 + KC * RSZ) * sizeof(float);
      const dim3 grid(ex.groups_per_image * s->B, slices, slabs);
      if (slab == 64)
        hipLaunchKernelGGL((dcn_bwd_weight_tiled_kernel<64, 2>), grid, dim3(256), lds, st, a, ex);
      else
        hipLaunchKernelGGL((dcn_bwd_weight_tiled_kernel<128, 1>), grid, dim3(256), lds, st, a, ex);
    } else {
      const size_t lds = (size_t)(BM * LDC + BM * LDK) * sizeof(float);
      hipLaunchKernelGGL(dcn_bwd_weight_kernel, dim3(tiles, s->B, (s->Cout + COC - 1) / COC),
                         dim3(256), lds, st, a);
    }
  }
  if (grad_bias) {
    const long long total = (long long)s->B * Ho * Wo;
    hipLaunchKernelGGL(dcn_bwd_bias_kernel, dim3(s->Cout, (unsigned)((total + BIAS_SEG - 1) / BIAS_SEG)),
                       dim3(256), 0, st, grad_out, grad_bias, s->B, s->Cout, Ho * Wo);
  }
  return cp_launch_status();
}

// out[c] += sum_{b,p} x[b][c][p]: the bias gradient of a library convolution (the 27-channel
// conv_offset_mask of every DCN; torch's generic reduction takes 148 us on [4,27,256,512]).
extern "C" int cp_channel_sum_accumulate(const float* x, float* out, int32_t B, int32_t C, int64_t HW,
                                         void* stream) {
  CP_CHECK_ARG(x && out && B > 0 && C > 0 && HW > 0);
  if (HW >= (1ll << 31) || C > 65535) return CP_EUNSUPPORTED;
  const long long total = (long long)B * HW;
  hipLaunchKernelGGL(dcn_bwd_bias_kernel, dim3(C, (unsigned)((total + BIAS_SEG - 1) / BIAS_SEG)), dim3(256), 0,
                     (hipStream_t)stream, x, out, B, C, (int)HW);
  return cp_launch_status();
}
