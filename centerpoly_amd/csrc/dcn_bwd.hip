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
#include "cp_common.h"

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
};

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
    red[(wid * 27 + 18 + t) * 64 + lane] = gm[t];
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
        const int t = q - 18;
        float gv = v;
        if (a.mask_is_logit) {
          const float m = r.m[t];
          gv *= m * (1.f - m);
        }
        a.gmask[(long long)b * a.gmask_bstride + (long long)t * HWo + p] = gv;
      }
    }
  }
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

// grad_bias[co] += sum_{b,p} go[b][co][p]; one workgroup per output channel.
__global__ __launch_bounds__(256) void dcn_bwd_bias_kernel(const float* __restrict__ go,
                                                           float* __restrict__ gb, int B, int Cout,
                                                           int HWo) {
  const int co = blockIdx.x;
  double s = 0;
  for (int b = 0; b < B; ++b) {
    const float* q = go + ((long long)b * Cout + co) * HWo;
    for (int i = threadIdx.x; i < HWo; i += 256) s += (double)q[i];
  }
  s = cp_wave_sum_d(s);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) gb[co] += (float)(red[0] + red[1] + red[2] + red[3]);
}

inline int out_extent(int in, int pad, int dil, int stride) {
  return (in + 2 * pad - (dil * 2 + 1)) / stride + 1;
}

}  // namespace

extern "C" size_t cp_dcn_v2_backward_workspace_bytes(const cp_dcn_shape*) { return 0; }

extern "C" int cp_dcn_v2_backward(const cp_dcn_shape* s, const float* x, const float* offset,
                                  int64_t offset_bstride, const float* mask,
                                  int64_t mask_bstride, int32_t mask_is_logit,
                                  const float* weight, const float* grad_out, float* grad_x,
                                  float* grad_offset, int64_t grad_offset_bstride,
                                  float* grad_mask, int64_t grad_mask_bstride, float* grad_weight,
                                  float* grad_bias, void* /*workspace*/,
                                  size_t /*workspace_bytes*/, void* stream) {
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
  hipStream_t st = (hipStream_t)stream;
  const int tiles = (Ho * Wo + BM - 1) / BM;
  if (grad_x || grad_offset || grad_mask) {
    const size_t lds = (size_t)(BM * LDC + COC * LDK + BM * 49) * sizeof(float);
    hipLaunchKernelGGL(dcn_bwd_data_kernel, dim3(tiles, s->B), dim3(256), lds, st, a);
  }
  if (grad_weight) {
    const size_t lds = (size_t)(BM * LDC + BM * LDK) * sizeof(float);
    hipLaunchKernelGGL(dcn_bwd_weight_kernel, dim3(tiles, s->B, (s->Cout + COC - 1) / COC),
                       dim3(256), lds, st, a);
  }
  if (grad_bias)
    hipLaunchKernelGGL(dcn_bwd_bias_kernel, dim3(s->Cout), dim3(256), 0, st, grad_out, grad_bias,
                       s->B, s->Cout, Ho * Wo);
  return cp_launch_status();
}
