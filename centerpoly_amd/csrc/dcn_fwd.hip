// DCNv2 forward for gfx950, gather kernels + dispatch: bilinear-sampled, mask-modulated im2col tile built in LDS
// (never in HBM) and contracted against the weight tile on the matrix cores.  The large maps go to the LDS-region
// kernel of dcn_fwd_region.hip (split-bf16); this file holds the per-lane-gather kernels -- exact fp32 MFMA
// (dcn_fwd_pipe_kernel) and split-bf16 (dcn_fwd_pipe_bf16_kernel) -- that serve the small, deep maps (K split over
// workgroups), strides / dilations / channel counts the region kernel does not take, and CP_DCN_F32.
//
// Replaces the native extension behind `from .DCNv2.dcn_v2 import DCN`
// (reference: src/lib/models/networks/pose_dla_dcn.py:16,354).
//
// Tiling: one workgroup (4 waves) = 64 consecutive output pixels of one image x BN output channels; every lane owns
// one pixel and keeps that pixel's 9 sampling recipes (4 corner weights, pre-multiplied by mask and validity, + the
// byte offsets of the two row pairs) in registers for the whole K loop.  Layers whose grid would not fill the 256
// CUs split K over workgroups; the partial sums meet in a small reduce + epilogue kernel (workspace supplied by the
// caller).  Details at each kernel.
#include "cp_common.h"
#include "dcn_internal.h"
#include <stdlib.h>

namespace {

constexpr int BM = 64;       // pixels per workgroup
constexpr int TAPS = 9;      // 3x3 only in this kernel
// LDS row stride of the column / weight tiles = k extent + LDPAD dwords.  The MFMA fragment reads
// (lane = (row r, k = q)) hit bank (LD*r + q) mod 32 inside each 32-lane group: LD = 2*odd makes
// LD*r cover the 16 even banks, so the reads are conflict-free (an odd LD leaves three 2-way
// conflicts per read = 2x the LDS cycles; PMC: 44 % of LDS cycles were conflicts).  The lane =
// pixel stores become 2-way conflicted, which a ds_write_b32 absorbs at no cost.
#ifndef CP_DCN_LDPAD
#define CP_DCN_LDPAD 2
#endif
constexpr int LDPAD = CP_DCN_LDPAD;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct DcnFwdArgs {
  const float* x;
  const float* offset;
  const float* mask;
  const float* weight;
  const float* bias;
  const float* ep_scale;
  const float* ep_shift;
  float* out;
  float* partial;            // [S][B][Cout][Ho*Wo] when splitk > 1
  long long offset_bstride, mask_bstride;
  int B, Cin, H, W, Cout, Ho, Wo;
  int stride, pad, dil;
  int mask_is_logit, relu;
  int splitk, c_per_split;   // K split: this many input channels per z-slice
#ifdef CP_ABLATE
  int ablate;                // timing-only build: bit0 no gathers, bit1 no colT writes,
                             // bit2 no weight staging, bit3 no MFMA (results are wrong)
#endif
};

#ifdef CP_ABLATE
#define CP_ABL(bit) (a.ablate & (bit))
#else
#define CP_ABL(bit) 0
#endif
// compile-time mask for the interleaved kernel (a runtime flag would change its schedule)
#ifdef CP_ABLATE_MASK
#define CP_ABLP(bit) ((CP_ABLATE_MASK) & (bit))
#else
#define CP_ABLP(bit) 0
#endif

// Per-pixel sampling recipe shared by the fp32 kernels (W >= 2).
// Per tap: the two x-neighbours of a row are ONE 8-byte gather.  The pair starts at
// xl = clamp(x0, 0, W-2) so both dwords are always inside the plane; the corner weights move
// to the slot their column landed in (x0 = -1 -> the valid right corner sits in the low slot,
// x0 = W-1 -> the valid left corner sits in the high slot).  Half the vector-memory
// instructions of the 4-dword form (the gather stream is what bounds this kernel) and 18
// fewer address registers.
__device__ __forceinline__ void pair_recipe(const DcnFwdArgs& a, int b, int p, bool p_ok,
                                            float (&cw)[TAPS][4], unsigned (&coff)[TAPS][2]) {
  const int HWo = a.Ho * a.Wo;
  const int ho = p_ok ? p / a.Wo : 0;
  const int wo = p_ok ? p - ho * a.Wo : 0;
  const float* off = a.offset + (long long)b * a.offset_bstride;
  const float* msk = a.mask + (long long)b * a.mask_bstride;
  float oy[TAPS], ox[TAPS], mk[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    oy[t] = p_ok ? off[(long long)(2 * t) * HWo + p] : 0.f;
    ox[t] = p_ok ? off[(long long)(2 * t + 1) * HWo + p] : 0.f;
    mk[t] = p_ok ? msk[(long long)t * HWo + p] : 0.f;
  }
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    const int ky = t / 3, kx = t - ky * 3;
    float m = mk[t];
    if (a.mask_is_logit) m = 1.f / (1.f + __expf(-m));
    const float py = (float)(ho * a.stride - a.pad + ky * a.dil) + oy[t];
    const float px = (float)(wo * a.stride - a.pad + kx * a.dil) + ox[t];
    const bool inside = p_ok && py > -1.f && px > -1.f && py < (float)a.H && px < (float)a.W;
    const float fy = floorf(py), fx = floorf(px);
    const int y0 = (int)fy, x0 = (int)fx;
    const float ly = py - fy, lx = px - fx;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= a.H - 1;
    const int y0c = min(max(y0, 0), a.H - 1), y1c = min(max(y0 + 1, 0), a.H - 1);
    const int xl = min(max(x0, 0), a.W - 2);
    // column weights of the pair's low / high dword (columns xl and xl + 1)
    const float wlo = (x0 == xl) ? hx : ((x0 + 1 == xl) ? lx : 0.f);
    const float whi = (x0 == xl) ? lx : ((x0 == xl + 1) ? hx : 0.f);
    const float wy0 = (inside && y0ok) ? hy : 0.f, wy1 = (inside && y1ok) ? ly : 0.f;
    cw[t][0] = wy0 * wlo * m;
    cw[t][1] = wy0 * whi * m;
    cw[t][2] = wy1 * wlo * m;
    cw[t][3] = wy1 * whi * m;
    coff[t][0] = inside ? 4u * (unsigned)(y0c * a.W + xl) : 0u;
    coff[t][1] = inside ? 4u * (unsigned)(y1c * a.W + xl) : 0u;
  }
}

// ------------------------------------------------------- interleaved kernel ---
// One workgroup (4 waves) = 64 consecutive output pixels x BN output channels; every lane owns one pixel and keeps
// its 9 sampling recipes in registers.  The matrix pipe never waits for a sampling phase (a phase-separated
// first version: MFMA busy 28 %, all waves reaching their MFMA phase together).  The K loop body is
// one k-step = one tap:
//     MFMAs of chunk i, k-step t, from LDS buffer A            (matrix pipe)
//     sample tap t of chunk i+1 from registers -> LDS buffer B  (VALU + ds_write, in the
//     store weight element t of chunk i+1 -> LDS buffer B        shadow of the MFMAs)
//     re-issue tap t's 4 gathers + weight element t for chunk i+2 into the SAME registers
// so each gather has a full chunk time to land, nothing is double-buffered in registers, and
// there is ONE barrier per chunk.  KC = 4 channels per chunk, wave w samples channel c0 + w.
template <int BN, int WPS>
__global__ __launch_bounds__(256, WPS) void dcn_fwd_pipe_kernel(DcnFwdArgs a) {
  constexpr int KC = 4;
  constexpr int KK = KC * TAPS;          // 36
  constexpr int LD = KK + LDPAD;         // 38
  constexpr int NT = BN / 32;
  constexpr int WPT = BN * KK / 256;     // weight elements per thread per chunk (9 / 18)
  constexpr int WPK = WPT / TAPS;        // ... per k-step (1 / 2)
  constexpr int BUF = (BM + BN) * LD;
  static_assert(WPT % TAPS == 0, "weights spread evenly over the k-steps");
  extern __shared__ float lds[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.z % a.B;
  const int n0 = blockIdx.y * BN;
  const int HWo = a.Ho * a.Wo, HW = a.H * a.W;
  const int p = blockIdx.x * BM + lane;
  const bool p_ok = p < HWo;

  float cw[TAPS][4];                          // {row0 lo, row0 hi, row1 lo, row1 hi}
  unsigned coff[TAPS][2];                     // byte offsets of the two row pairs
  pair_recipe(a, b, p, p_ok, cw, coff);

  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wm = wid >> 1, wn = wid & 1;
  const int Ktot = a.Cin * TAPS;
  const float* xb = a.x + (long long)b * a.Cin * HW;
  const int ksl = (a.splitk > 1) ? (int)(blockIdx.z / a.B) : 0;
  const int c_begin = ksl * a.c_per_split;
  const int c_end = min(a.Cin, c_begin + a.c_per_split);

  const unsigned plane_bytes = (unsigned)HW * 4u;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xb), 0, (int)((unsigned)a.Cin * plane_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.weight), 0, (int)((unsigned)a.Cout * (unsigned)Ktot * 4u), 0x00020000);
  unsigned woff[WPT];
  int wlds[WPT];
#pragma unroll
  for (int i = 0; i < WPT; ++i) {
    const int idx = tid + i * 256;
    const int co = idx / KK, kk = idx - co * KK;
    woff[i] = n0 + co < a.Cout ? ((unsigned)(n0 + co) * (unsigned)Ktot + (unsigned)kk) * 4u : 0xf0000000u;
    wlds[i] = BM * LD + co * LD + kk;          // offset inside one buffer
  }
  const int swid = __builtin_amdgcn_readfirstlane(wid);
  const int colw = lane * LD + swid * TAPS;    // this lane's column slot (+ tap) inside a buffer

  float g[TAPS][4];
  float wreg[WPT];
  auto load_tap = [&](int c0, int t) {         // gathers of tap t (+ its share of the weights)
    // soffset is not range-checked by the hardware (only voffset is): clamp the channel plane
    // (foreign samples are zeroed in build_tap) and carry the chunk's k offset in voffset
    const unsigned xsoff = (unsigned)min(c0 + swid, a.Cin - 1) * plane_bytes;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (CP_ABLP(1)) {
        g[t][2 * k] = g[t][2 * k + 1] = 1.f;
      } else {
        // (bit_cast of the whole vector: a bit_cast of the element expression `v.y` reads v.x)
        const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_x, coff[t][k], xsoff, 0));
        g[t][2 * k] = v.x;
        g[t][2 * k + 1] = v.y;
      }
    }
    const unsigned wk = (unsigned)(c0 * TAPS) * 4u;
#pragma unroll
    for (int q = 0; q < WPK; ++q) {
      wreg[t * WPK + q] = CP_ABLP(4) ? 1.f : __builtin_bit_cast(
          float, __builtin_amdgcn_raw_buffer_load_b32(rs_w, woff[t * WPK + q] + wk, 0, 0));
    }
  };
  auto build_tap = [&](float* buf, int c0, int t) {   // tap t of chunk c0 -> LDS buffer `buf`
    const float v = cw[t][0] * g[t][0] + cw[t][1] * g[t][1] + cw[t][2] * g[t][2] + cw[t][3] * g[t][3];
    if (CP_ABLP(2)) {
      if (v == 12345.f) buf[0] = v + wreg[t * WPK];
      return;
    }
    buf[colw + t] = (c0 + swid < c_end) ? v : 0.f;     // split-K: foreign channels contribute 0
#pragma unroll
    for (int q = 0; q < WPK; ++q) buf[wlds[t * WPK + q]] = wreg[t * WPK + q];
  };

  // prologue: chunk 0 -> buffer 0, chunk 1's loads in flight
#pragma unroll
  for (int t = 0; t < TAPS; ++t) load_tap(c_begin, t);
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    build_tap(lds, c_begin, t);
    load_tap(c_begin + KC, t);
  }
  __syncthreads();

  const int arow = (wm * 32 + (lane & 15)) * LD + (lane >> 4);
  const int brow = BM * LD + (wn * (BN / 2) + (lane & 15)) * LD + (lane >> 4);
  int par = 0;
  for (int c0 = c_begin; c0 < c_end; c0 += KC, par ^= 1) {
    const float* cur = lds + par * BUF;
    float* nxt = lds + (par ^ 1) * BUF;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {             // k-step t of chunk c0  ||  tap t of chunk c0+KC
      float af[2], bf[NT];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = CP_ABLP(32) ? cw[t][i] : cur[arow + i * 16 * LD + t * 4];
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j] = CP_ABLP(32) ? cw[t][2 + (j & 1)] : cur[brow + j * 16 * LD + t * 4];
      if (!CP_ABLP(8)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      } else {
        acc[0][0][0] += af[0] + af[1] + bf[0] + bf[NT - 1];
      }
      build_tap(nxt, c0 + KC, t);
      load_tap(c0 + 2 * KC, t);
      // keep hipcc from sinking the re-issued gathers to the loop bottom (it did: the next
      // iteration then waited vmcnt(0), i.e. the full latency, every chunk)
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                             // nxt complete, cur free
  }

  const bool raw = a.splitk > 1;
  float* ob = raw ? a.partial + ((long long)ksl * a.B + b) * a.Cout * HWo
                  : a.out + (long long)b * a.Cout * HWo;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = n0 + wn * (BN / 2) + j * 16 + (lane & 15);
    if (co >= a.Cout) continue;
    float sc = 1.f, sh = 0.f;
    if (!raw) {
      if (a.ep_scale) sc = a.ep_scale[co];
      if (a.ep_shift) sh = a.ep_shift[co];
      else if (a.bias) sh = a.bias[co];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pp = blockIdx.x * BM + wm * 32 + i * 16 + (lane >> 4) * 4;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[i][j][r] * sc + sh;
        if (a.relu && !raw) v[r] = fmaxf(v[r], 0.f);
      }
      float* dst = ob + (long long)co * HWo + pp;
      if (pp + 3 < HWo && (HWo & 3) == 0) {
        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (pp + r < HWo) dst[r] = v[r];
      }
    }
  }
}

// ---------------------------------------------------------- split-bf16 kernel ---
// The interleaved kernel's tiling, pair gathers and pipelining with the contraction on the bf16 matrix cores as
// THREE products of bf16 halves with fp32 accumulation:
//     a*b ~= ah*bh + ah*bl + al*bh,   x = xh + xl,  xh = bf16(x), xl = bf16(x - xh)
// (the dropped al*bl term is 2^-16 relative), i.e. an fp32 emulation, not a bf16 result.  Why: the f32-input MFMA
// executes on the SIMD's vector ALUs (SQ_VALU_MFMA_COEXEC_CYCLES = 0 in every profile of this path), so the layers
// with >= 128 output channels -- 2304 MFMA cycles per 4-channel chunk against 1152 cycles of gathers -- are bound by
// it; the bf16 cores run beside the VALU and leave the texture addresser as the only bound.
// Chunks are 8 input channels = 72 k values (k = 9 channel + tap), zero-padded to 96 = three 32-deep MFMA steps.
//   * weights: split and laid out in B-fragment order by dcn_fwd_wperm_kernel (once per weight tensor; cached by the
//     caller at inference) and read straight from global memory -- no weight tile in LDS, no conversion here;
//   * columns: wave w samples channels 2w, 2w + 1 of the chunk (lane = pixel), splits each value and stores the
//     bf16 halves of its 18 k-values as 9 + 9 dwords into the pixel's row of a double-buffered [64][104] tile
//     (208-byte rows: conflict-free 16-byte fragment reads);
//   * per k-step: 2 x NT x 3 MFMAs of chunk i  ||  six values of chunk i + 1 built and stored  ||  their gathers
//     re-issued for chunk i + 2 into the same registers; ONE barrier per chunk.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// wp[((ct * nchunk + chunk) * 3 + ks) * 2 + hl][lane][j] = half(hl) of W[co = 16 ct + (lane & 15)][ci = 8 chunk + k / 9][k % 9],
// k = 32 ks + 8 (lane >> 4) + j  (zero for k >= 72, co >= Cout, ci >= Cin)
__global__ __launch_bounds__(256) void dcn_fwd_wperm_kernel(const float* __restrict__ w, bf16x8* __restrict__ wp, int Cout,
                                                            int Cin, int nchunk, int total) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int lane = e & 63;
  int r = e >> 6;
  const int hl = r & 1;
  r >>= 1;
  const int ks = r % 3;
  r /= 3;
  const int chunk = r % nchunk, ct = r / nchunk;
  const int co = ct * 16 + (lane & 15);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 32 * ks + 8 * (lane >> 4) + j;
    const int ci = chunk * 8 + k / TAPS, t = k % TAPS;
    const float v = (k < 72 && co < Cout && ci < Cin) ? w[((long long)co * Cin + ci) * TAPS + t] : 0.f;
    const __bf16 h = (__bf16)v;
    o[j] = hl ? (__bf16)(v - (float)h) : h;
  }
  wp[e] = o;
}

template <int BN, int WPS>
__global__ __launch_bounds__(256, WPS) void dcn_fwd_pipe_bf16_kernel(DcnFwdArgs a, const bf16x8* __restrict__ wp) {
  constexpr int KC = 8;
  constexpr int KK = KC * TAPS;          // 72
  constexpr int KP = 96;                 // padded k extent (3 MFMA steps of 32)
  constexpr int LDA = KP + 8;            // 104 bf16 = 208 B rows
  constexpr int NT = BN / 32;
  constexpr int HALF = BM * LDA, BUF = 2 * HALF;      // bf16 elements: one half, one buffer (hi | lo)
  extern __shared__ float lds[];
  __bf16* col = (__bf16*)lds;            // [2 buffers][hi | lo][BM][LDA]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.z % a.B;
  const int n0 = blockIdx.y * BN;
  const int HWo = a.Ho * a.Wo, HW = a.H * a.W;
  const int p = blockIdx.x * BM + lane;
  const bool p_ok = p < HWo;

  float cw[TAPS][4];
  unsigned coff[TAPS][2];
  pair_recipe(a, b, p, p_ok, cw, coff);

  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wm = wid >> 1, wn = wid & 1;
  const float* xb = a.x + (long long)b * a.Cin * HW;
  const int ksl = (a.splitk > 1) ? (int)(blockIdx.z / a.B) : 0;
  const int c_begin = ksl * a.c_per_split;
  const int c_end = min(a.Cin, c_begin + a.c_per_split);
  const int nchunk = (a.Cin + KC - 1) / KC;

  const unsigned plane_bytes = (unsigned)HW * 4u;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xb), 0, (int)((unsigned)a.Cin * plane_bytes), 0x00020000);
  const int swid = __builtin_amdgcn_readfirstlane(wid);

  // zero the k padding [72, 96) of both buffers' halves once
  for (int e = tid; e < 4 * BM * (KP - KK); e += 256) {
    const int row = e / (KP - KK), kk = KK + (e - row * (KP - KK));
    col[row * LDA + kk] = (__bf16)0.f;                    // rows run over [buffer][half][pixel]
  }

  float g[2][TAPS][4];                                    // gathers of this wave's two channels
  auto load_unit = [&](int c0, int u) __attribute__((always_inline)) {   // unit u = cc * 9 + t
    const int cc = u / TAPS, t = u % TAPS;
    // soffset is not range-checked: clamp the channel plane (foreign samples are zeroed when built)
    const unsigned xsoff = (unsigned)min(c0 + 2 * swid + cc, a.Cin - 1) * plane_bytes;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_x, coff[t][k], xsoff, 0));
      g[cc][t][2 * k] = v.x;
      g[cc][t][2 * k + 1] = v.y;
    }
  };
  auto value = [&](int c0, int u) __attribute__((always_inline)) -> float {
    const int cc = u / TAPS, t = u % TAPS;
    const float v = cw[t][0] * g[cc][t][0] + cw[t][1] * g[cc][t][1] + cw[t][2] * g[cc][t][2] + cw[t][3] * g[cc][t][3];
    return (c0 + 2 * swid + cc < c_end) ? v : 0.f;        // split-K / ragged Cin: foreign channels contribute 0
  };
  // dwords 3 s .. 3 s + 2 of this wave's 18 k-values (k = 18 wid + 2 d, + 1) of chunk c0 -> buffer `buf`,
  // then the same six units' gathers for chunk c0 + KC
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  auto build_third = [&](__bf16* buf, int c0, int s_, bool reissue) __attribute__((always_inline)) {
    unsigned* rowH = reinterpret_cast<unsigned*>(buf + lane * LDA + 18 * swid);
    unsigned* rowL = reinterpret_cast<unsigned*>(buf + HALF + lane * LDA + 18 * swid);
#pragma unroll
    for (int d = 3 * s_; d < 3 * s_ + 3; ++d) {
      const float v0 = value(c0, 2 * d), v1 = value(c0, 2 * d + 1);
      bf16x2 h, l;
      h[0] = (__bf16)v0;
      h[1] = (__bf16)v1;
      l[0] = (__bf16)(v0 - (float)h[0]);
      l[1] = (__bf16)(v1 - (float)h[1]);
      rowH[d] = __builtin_bit_cast(unsigned, h);
      rowL[d] = __builtin_bit_cast(unsigned, l);
    }
    if (reissue) {
#pragma unroll
      for (int u = 6 * s_; u < 6 * s_ + 6; ++u) load_unit(c0 + KC, u);
    }
  };

  // prologue: chunk 0 -> buffer 0, chunk 1's gathers in flight
#pragma unroll
  for (int u = 0; u < 2 * TAPS; ++u) load_unit(c_begin, u);
#pragma unroll
  for (int s_ = 0; s_ < 3; ++s_) build_third(col, c_begin, s_, true);
  __syncthreads();

  const int arow = (wm * 32 + (lane & 15)) * LDA + 8 * (lane >> 4);
  const bf16x8* wq = wp + lane;
  const int ct0 = (n0 + wn * (BN / 2)) / 16;
  int par = 0;
  for (int c0 = c_begin; c0 < c_end; c0 += KC, par ^= 1) {
    const __bf16* cur = col + par * BUF;
    __bf16* nxt = col + (par ^ 1) * BUF;
    const int chunk = c0 / KC;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {                       // k-step ks of chunk c0  ||  a third of chunk c0 + KC
      bf16x8 bh[NT], bl[NT], ah[2], al[2];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const bf16x8* q = wq + ((((long long)(ct0 + j) * nchunk + chunk) * 3 + ks) * 2) * 64;
        bh[j] = q[0];
        bl[j] = q[64];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8*>(cur + arow + i * 16 * LDA + ks * 32);
        al[i] = *reinterpret_cast<const bf16x8*>(cur + HALF + arow + i * 16 * LDA + ks * 32);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
      build_third(nxt, c0 + KC, ks, true);
      __builtin_amdgcn_sched_barrier(0);                   // keep the re-issued gathers where they are
    }
    __syncthreads();                                       // nxt complete, cur free
  }

  const bool raw = a.splitk > 1;
  float* ob = raw ? a.partial + ((long long)ksl * a.B + b) * a.Cout * HWo
                  : a.out + (long long)b * a.Cout * HWo;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = n0 + wn * (BN / 2) + j * 16 + (lane & 15);
    if (co >= a.Cout) continue;
    float sc = 1.f, sh = 0.f;
    if (!raw) {
      if (a.ep_scale) sc = a.ep_scale[co];
      if (a.ep_shift) sh = a.ep_shift[co];
      else if (a.bias) sh = a.bias[co];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pp = blockIdx.x * BM + wm * 32 + i * 16 + (lane >> 4) * 4;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[i][j][r] * sc + sh;
        if (a.relu && !raw) v[r] = fmaxf(v[r], 0.f);
      }
      float* dst = ob + (long long)co * HWo + pp;
      if (pp + 3 < HWo && (HWo & 3) == 0) {
        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (pp + r < HWo) dst[r] = v[r];
      }
    }
  }
}

// Sum the K-split partials and apply bias / folded-BN / ReLU.
__global__ __launch_bounds__(256) void dcn_splitk_reduce_kernel(DcnFwdArgs a, long long n_per_b) {
  const long long total = (long long)a.B * n_per_b;
  const int HWo = a.Ho * a.Wo;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (long long)gridDim.x * 256) {
    float v = 0.f;
    for (int sidx = 0; sidx < a.splitk; ++sidx) v += a.partial[(long long)sidx * total + i];
    const int co = (int)((i % n_per_b) / HWo);
    float sc = 1.f, sh = 0.f;
    if (a.ep_scale) sc = a.ep_scale[co];
    if (a.ep_shift) sh = a.ep_shift[co];
    else if (a.bias) sh = a.bias[co];
    v = v * sc + sh;
    if (a.relu) v = fmaxf(v, 0.f);
    a.out[i] = v;
  }
}

struct Plan {
  int bn, splitk, c_per_split;
};

Plan make_plan(int B, int Cin, int Cout, int HWo) {
  // Widest N tile (up to 128) that is not mostly padding: every extra N tile re-samples the
  // columns.  Under-filled grids are topped up by splitting K, not by narrowing N.  (A 256-wide
  // tile samples once but only fits the phase-separated kernel at one wave per SIMD: measured
  // 0.170 vs 0.109 ms on 256->256 @64x128, so Cout > 128 takes two 128-wide tiles.)
  const long long tiles_m = (long long)((HWo + BM - 1) / BM) * B;
  Plan p;
  p.bn = Cout > 64 ? 128 : 64;
  const long long blocks = tiles_m * ((Cout + p.bn - 1) / p.bn);
  p.splitk = 1;
  const int kc = 4;
  if (blocks < 384) {
    int s = (int)((512 + blocks - 1) / blocks);
    const int max_s = (Cin + 4 * kc - 1) / (4 * kc);    // keep >= 4 chunks per slice
    if (s > max_s) s = max_s;
    if (s > 16) s = 16;
    if (s < 1) s = 1;
    p.splitk = s;
  }
  int cps = (Cin + p.splitk - 1) / p.splitk;
  cps = (cps + 7) / 8 * 8;               // whole chunks for both the fp32 (4) and bf16x3 (8) kernels
  p.c_per_split = cps;
  p.splitk = (Cin + cps - 1) / cps;
  return p;
}

template <int BN, int WPS>
int launch_pipe(const DcnFwdArgs& a, hipStream_t st) {
  const size_t lds = (size_t)2 * (BM + BN) * (4 * TAPS + LDPAD) * sizeof(float);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)dcn_fwd_pipe_kernel<BN, WPS>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid((a.Ho * a.Wo + BM - 1) / BM, (a.Cout + BN - 1) / BN, a.B * a.splitk);
  hipLaunchKernelGGL((dcn_fwd_pipe_kernel<BN, WPS>), grid, dim3(256), lds, st, a);
  if (a.splitk > 1) {
    const long long n_per_b = (long long)a.Cout * a.Ho * a.Wo;
    long long nb = ((long long)a.B * n_per_b + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(dcn_splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, a, n_per_b);
  }
  return cp_launch_status();
}

size_t wperm_bytes_gather(int Cin, int Cout) {             // 16-row tiles padded to whole groups of 8 (BN = 128)
  return (size_t)((Cout + 127) / 128 * 8) * ((Cin + 7) / 8) * 3 * 2 * 64 * 16;
}

// head of the workspace: room for the permuted weights of whichever split-bf16 kernel the plan picks
size_t wperm_bytes(const cp_dcn_shape* s) {
  size_t n = wperm_bytes_gather(s->Cin, s->Cout);
  if (cp_dcn_region_supported(s)) {
    const size_t r = cp_dcn_region_wperm_bytes(s);
    if (r > n) n = r;
  }
  return cp_align_up(n, 256);
}

// The region kernel (dcn_fwd_region.hip) pays where its 8 x 32 pixel tiles x 64-channel blocks fill the chip; small
// maps keep the gather kernels with their K split.
bool region_pays(const cp_dcn_shape* s) {
  const long long wgs = (long long)((s->H + 7) / 8) * ((s->W + 31) / 32) * ((s->Cout + 63) / 64) * s->B;
  if (wgs >= 256 || (wgs >= 128 && s->Cin <= 128)) return true;
  // round 4: deep, small maps take the region kernel with its input channels split over grid z (>= 256 workgroups of >= 2
  // chunks); before, they kept the gather kernels (256->256 @64x128: 89 us there)
  return wgs * cp_dcn_region_ksplit(s) >= 256;
}

// slices of the region kernel's K split for this shape (1 = none)
int region_ksplit(const cp_dcn_shape* s) {
  const long long wgs = (long long)((s->H + 7) / 8) * ((s->W + 31) / 32) * ((s->Cout + 63) / 64) * s->B;
  if (wgs >= 256) return 1;
  if (wgs >= 128 && s->Cin <= 64) return 1;                 // (4 chunks: the fused module launch is worth more than a split)
  return cp_dcn_region_ksplit(s);                           // 128 .. 255 workgroups leave half the CUs with one, or none
}

template <int BN, int WPS>
int launch_bf16x3(const DcnFwdArgs& a, const void* wp, hipStream_t st) {
  const size_t lds = (size_t)2 * 2 * BM * 104 * 2;         // two buffers x (hi | lo) x 64 rows of 104 bf16
  dim3 grid((a.Ho * a.Wo + BM - 1) / BM, (a.Cout + BN - 1) / BN, a.B * a.splitk);
  hipLaunchKernelGGL((dcn_fwd_pipe_bf16_kernel<BN, WPS>), grid, dim3(256), lds, st, a, (const bf16x8*)wp);
  if (a.splitk > 1) {
    const long long n_per_b = (long long)a.Cout * a.Ho * a.Wo;
    long long nb = ((long long)a.B * n_per_b + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(dcn_splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, a, n_per_b);
  }
  return cp_launch_status();
}

int out_extent(int in, int pad, int dil, int stride) {
  return (in + 2 * pad - (dil * 2 + 1)) / stride + 1;
}

}  // namespace

extern "C" size_t cp_dcn_v2_forward_workspace_bytes(const cp_dcn_shape* s) {
  if (!s || s->B <= 0 || s->Cin <= 0 || s->Cout <= 0) return 0;
  const int Ho = out_extent(s->H, s->pad, s->dil, s->stride);
  const int Wo = out_extent(s->W, s->pad, s->dil, s->stride);
  if (Ho <= 0 || Wo <= 0) return 0;
  const Plan p = make_plan(s->B, s->Cin, s->Cout, Ho * Wo);
  // [permuted weights of the split-bf16 contraction | K-split partial sums]
  size_t part = p.splitk <= 1 ? 0 : (size_t)p.splitk * s->B * s->Cout * Ho * Wo * sizeof(float);
  if (cp_dcn_region_supported(s)) {
    const size_t rp = region_ksplit(s) <= 1 ? 0 : (size_t)region_ksplit(s) * s->B * s->Cout * Ho * Wo * sizeof(float);
    if (rp > part) part = rp;
  }
  return wperm_bytes(s) + part;
}

extern "C" int cp_dcn_v2_forward_kernel(const cp_dcn_shape* s, int32_t contraction) {
  if (!s) return CP_EINVAL;
  if (contraction == CP_DCN_F32) return 0;
  const bool force = contraction == CP_DCN_BF16X3_REGION || contraction == CP_DCN_BF16X3_REGION_PREPARED;
  if (contraction != CP_DCN_BF16X3 && contraction != CP_DCN_BF16X3_PREPARED && !force) return CP_EINVAL;
  if (cp_dcn_region_supported(s) && (force || region_pays(s))) return 2;
  return force ? CP_EUNSUPPORTED : 1;
}

extern "C" int cp_dcn_v2_forward(const cp_dcn_shape* s, const float* x, const float* offset,
                                 int64_t offset_bstride, const float* mask,
                                 int64_t mask_bstride, int32_t mask_is_logit,
                                 const float* weight, const float* bias, const float* ep_scale,
                                 const float* ep_shift, int32_t relu, int32_t contraction,
                                 float* out, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  CP_CHECK_ARG(s && x && offset && mask && weight && out);
  CP_CHECK_ARG(s->B > 0 && s->Cin > 0 && s->H > 0 && s->W > 0 && s->Cout > 0);
  CP_CHECK_ARG(s->stride > 0 && s->dil > 0 && s->pad >= 0);
  if (s->kh != 3 || s->kw != 3 || s->deformable_groups != 1) return CP_EUNSUPPORTED;
  const int Ho = out_extent(s->H, s->pad, s->dil, s->stride);
  const int Wo = out_extent(s->W, s->pad, s->dil, s->stride);
  CP_CHECK_ARG(Ho > 0 && Wo > 0);
  if ((long long)s->H * s->W >= (1ll << 31) || (long long)Ho * Wo >= (1ll << 31)) return CP_EUNSUPPORTED;
  if (s->W < 2) return CP_EUNSUPPORTED;          // the x-pair gathers need two columns
  const Plan p = make_plan(s->B, s->Cin, s->Cout, Ho * Wo);
  if ((long long)s->B * p.splitk > 65535) return CP_EUNSUPPORTED;
  const size_t wpb = wperm_bytes(s);
  const bool force_region = contraction == CP_DCN_BF16X3_REGION || contraction == CP_DCN_BF16X3_REGION_PREPARED;
  const bool prepared = contraction == CP_DCN_BF16X3_PREPARED || contraction == CP_DCN_BF16X3_REGION_PREPARED;
  const bool bf = contraction == CP_DCN_BF16X3 || contraction == CP_DCN_BF16X3_PREPARED || force_region;
  if (force_region && !cp_dcn_region_supported(s)) return CP_EUNSUPPORTED;
  if (bf && cp_dcn_region_supported(s) && (force_region || region_pays(s))) {
    if (!workspace || workspace_bytes < cp_dcn_v2_forward_workspace_bytes(s)) return CP_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (!prepared) {
      const int rc = cp_dcn_region_prepare(s, weight, workspace, st);
      if (rc != CP_OK) return rc;
    }
    const int ks = region_ksplit(s);
    float* partial = ks > 1 ? (float*)((char*)workspace + wpb) : nullptr;
    const int rc = cp_dcn_region_forward(s, x, offset, offset_bstride, mask, mask_bstride, mask_is_logit, workspace, bias,
                                         ep_scale, ep_shift, relu, out, nullptr, nullptr, nullptr, partial, ks, st);
    if (rc != CP_OK || ks <= 1) return rc;
    DcnFwdArgs ra;                                          // the slices' raw sums -> out, + bias / folded BN / ReLU
    ra.partial = partial; ra.out = out; ra.bias = bias; ra.ep_scale = ep_scale; ra.ep_shift = ep_shift; ra.relu = relu;
    ra.B = s->B; ra.Ho = Ho; ra.Wo = Wo; ra.Cout = s->Cout; ra.splitk = ks;
    const long long n_per_b = (long long)s->Cout * Ho * Wo;
    long long nb = ((long long)s->B * n_per_b + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(dcn_splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, ra, n_per_b);
    return cp_launch_status();
  }
  if (p.splitk > 1 || bf) {
    if (!workspace || workspace_bytes < cp_dcn_v2_forward_workspace_bytes(s)) return CP_EWORKSPACE;
  }
  DcnFwdArgs a;
  a.x = x; a.offset = offset; a.mask = mask; a.weight = weight; a.bias = bias;
  a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.out = out; a.partial = (float*)((char*)workspace + (workspace ? wpb : 0));
  a.offset_bstride = offset_bstride; a.mask_bstride = mask_bstride;
  a.B = s->B; a.Cin = s->Cin; a.H = s->H; a.W = s->W; a.Cout = s->Cout; a.Ho = Ho; a.Wo = Wo;
  a.stride = s->stride; a.pad = s->pad; a.dil = s->dil;
  a.mask_is_logit = mask_is_logit; a.relu = relu;
  a.splitk = p.splitk; a.c_per_split = p.c_per_split;
#ifdef CP_ABLATE
  {
    const char* e = getenv("CP_DCN_ABLATE");
    a.ablate = e ? atoi(e) : 0;
  }
#endif
  hipStream_t st = (hipStream_t)stream;
  CP_CHECK_ARG(contraction == CP_DCN_F32 || bf);
  if ((unsigned long long)s->Cout * s->Cin * 9ull * 4ull >= 0xE0000000ull) return CP_EUNSUPPORTED;
  if (bf && p.bn <= 128 && (p.c_per_split % 8) == 0) {
    if (!prepared) {                                        // (PREPARED: the workspace already holds them)
      const int nchunk = (s->Cin + 7) / 8;
      const int total = (int)(wperm_bytes_gather(s->Cin, s->Cout) / 16);
      hipLaunchKernelGGL(dcn_fwd_wperm_kernel, dim3((total + 255) / 256), dim3(256), 0, st, weight, (bf16x8*)workspace,
                         s->Cout, s->Cin, nchunk, total);
    }
    if (p.bn == 64) return launch_bf16x3<64, 2>(a, workspace, st);
    return launch_bf16x3<128, 2>(a, workspace, st);
  }
  if (p.bn == 64) return launch_pipe<64, 3>(a, st);        // interleaved gather kernel, exact f32 MFMA
  return launch_pipe<128, 2>(a, st);
}

// --------------------------------------------------------------------------------------------------------------------
// DCN with its conv_offset_mask inside the kernel (dcn_fwd_region.hip, template FUSE).
static size_t fused_region_bytes(const cp_dcn_shape* s) { return cp_align_up(cp_dcn_region_wperm_bytes(s), 256); }

// One 64-channel block only (with more, every block's workgroups would repeat the offset convolution of their tile) and
// at most 64 input channels: measured at 128 -> 64 @128x256 the in-kernel pass (8 chunks on 128 workgroups, 81 us in
// all) does not beat the stand-alone convolution with its K split (58 + 20 us); at 64 -> 64 @256x512 it does (64 vs 89).
extern "C" int cp_dcn_v2_forward_fused_supported(const cp_dcn_shape* s) {
  return s && s->B > 0 && s->Cout > 0 && s->Cout <= 64 && s->Cin <= 64 && cp_dcn_region_supported(s) && region_pays(s) &&
                 region_ksplit(s) == 1
             ? 1
             : 0;                                 // (the fused form never splits its input channels)
}

extern "C" size_t cp_dcn_v2_forward_fused_workspace_bytes(const cp_dcn_shape* s) {
  if (!cp_dcn_v2_forward_fused_supported(s)) return 0;
  return fused_region_bytes(s) + cp_align_up(cp_dcn_region_om_wperm_bytes(s), 256);
}

extern "C" int cp_dcn_v2_forward_fused(const cp_dcn_shape* s, const float* x, const float* om_weight, const float* om_bias,
                                       const float* weight, const float* bias, const float* ep_scale,
                                       const float* ep_shift, int32_t relu, int32_t prepared, float* om_out, float* out,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  CP_CHECK_ARG(s && x && om_weight && om_bias && weight && out);
  if (!cp_dcn_v2_forward_fused_supported(s)) return CP_EUNSUPPORTED;
  if (!workspace || workspace_bytes < cp_dcn_v2_forward_fused_workspace_bytes(s)) return CP_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  void* om_wp = (char*)workspace + fused_region_bytes(s);
  if (!prepared) {
    int rc = cp_dcn_region_prepare(s, weight, workspace, st);
    if (rc != CP_OK) return rc;
    rc = cp_dcn_region_prepare_om(s, om_weight, om_wp, st);
    if (rc != CP_OK) return rc;
  }
  return cp_dcn_region_forward(s, x, nullptr, 0, nullptr, 0, 1, workspace, bias, ep_scale, ep_shift, relu, out, om_wp, om_bias,
                               om_out, nullptr, 1, st);
}
