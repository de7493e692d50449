// DCNv2 forward for gfx950: bilinear-sampled, mask-modulated im2col tile built in
// LDS (never in HBM) and contracted against the weight tile with fp32 MFMA.
//
// Replaces the native extension behind `from .DCNv2.dcn_v2 import DCN`
// (reference: src/lib/models/networks/pose_dla_dcn.py:16,354).
//
// Tiling: one workgroup (4 waves) = 64 consecutive output pixels of one image x
// BN output channels.  K = Cin*kh*kw is walked in chunks of KC input channels:
//   1. every lane owns one pixel and keeps that pixel's kh*kw sampling recipes
//      (4 corner weights, pre-multiplied by mask and validity, + 4 clamped
//      indices) in registers for the whole K loop;
//   2. wave w samples channels {c0 + w*KC/4 ...} of the chunk: lanes = adjacent
//      pixels, so the 4 corner reads of a (channel, tap) are near-contiguous;
//   3. the weight chunk [BN][KC*9] is staged to LDS with coalesced row reads;
//   4. v_mfma_f32_16x16x4_f32 over the chunk (exact fp32 fma chain).
// LDS rows are padded to an odd dword count so both the lane=pixel writes and the
// lane=(row, k) fragment reads are bank-conflict free.
#include "cp_common.h"

namespace {

constexpr int BM = 64;       // pixels per workgroup
constexpr int TAPS = 9;      // 3x3 only in this kernel

struct DcnFwdArgs {
  const float* x;
  const float* offset;
  const float* mask;
  const float* weight;
  const float* bias;
  const float* ep_scale;
  const float* ep_shift;
  float* out;
  long long offset_bstride, mask_bstride;
  int B, Cin, H, W, Cout, Ho, Wo;
  int stride, pad, dil;
  int mask_is_logit, relu;
};

template <int BN, int KC>
__global__ __launch_bounds__(256) void dcn_fwd_kernel(DcnFwdArgs a) {
  constexpr int KK = KC * TAPS;          // k extent of one chunk
  constexpr int LD = KK + 1;             // odd row stride (dwords)
  constexpr int NT = BN / 32;            // 16-wide n tiles per wave
  constexpr int CPW = KC / 4;            // channels sampled per wave per chunk
  static_assert(KK % 4 == 0, "chunk must be a multiple of the MFMA k");
  extern __shared__ float lds[];
  float* colT = lds;                     // [BM][LD]
  float* wT = lds + BM * LD;             // [BN][LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int b = blockIdx.z;
  const int n0 = blockIdx.y * BN;
  const int HWo = a.Ho * a.Wo;
  const int HW = a.H * a.W;
  const int p = blockIdx.x * BM + lane;
  const bool p_ok = p < HWo;

  // ---- per-pixel sampling recipes, kept in registers ----
  float cw[TAPS][4];
  int ci[TAPS][4];
  {
    const int ho = p_ok ? p / a.Wo : 0;
    const int wo = p_ok ? p - ho * a.Wo : 0;
    const float* off = a.offset + (long long)b * a.offset_bstride;
    const float* msk = a.mask + (long long)b * a.mask_bstride;
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
      const float ly = py - fy, lx = px - fx;
      const float hy = 1.f - ly, hx = 1.f - lx;
      const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= a.H - 1;
      const bool x0ok = x0 >= 0, x1ok = x0 + 1 <= a.W - 1;
      const int y0c = min(max(y0, 0), a.H - 1), y1c = min(max(y0 + 1, 0), a.H - 1);
      const int x0c = min(max(x0, 0), a.W - 1), x1c = min(max(x0 + 1, 0), a.W - 1);
      cw[t][0] = (inside && y0ok && x0ok) ? hy * hx * m : 0.f;
      cw[t][1] = (inside && y0ok && x1ok) ? hy * lx * m : 0.f;
      cw[t][2] = (inside && y1ok && x0ok) ? ly * hx * m : 0.f;
      cw[t][3] = (inside && y1ok && x1ok) ? ly * lx * m : 0.f;
      ci[t][0] = inside ? y0c * a.W + x0c : 0;
      ci[t][1] = inside ? y0c * a.W + x1c : 0;
      ci[t][2] = inside ? y1c * a.W + x0c : 0;
      ci[t][3] = inside ? y1c * a.W + x1c : 0;
    }
  }

  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wm = wid >> 1, wn = wid & 1;
  const int Ktot = a.Cin * TAPS;
  const float* xb = a.x + (long long)b * a.Cin * HW;

  for (int c0 = 0; c0 < a.Cin; c0 += KC) {
    __syncthreads();   // previous chunk's fragment reads are done
    // ---- sample this wave's channels of the chunk ----
#pragma unroll
    for (int cc = 0; cc < CPW; ++cc) {
      const int cl = wid * CPW + cc;
      const int c = c0 + cl;
      const bool c_ok = c < a.Cin;
      const float* xc = xb + (long long)(c_ok ? c : 0) * HW;
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        float v = cw[t][0] * xc[ci[t][0]] + cw[t][1] * xc[ci[t][1]] +
                  cw[t][2] * xc[ci[t][2]] + cw[t][3] * xc[ci[t][3]];
        colT[lane * LD + cl * TAPS + t] = c_ok ? v : 0.f;
      }
    }
    // ---- stage the weight chunk [BN][KK] ----
    for (int idx = tid; idx < BN * KK; idx += 256) {
      const int co = idx / KK;
      const int kk = idx - co * KK;
      const int kg = c0 * TAPS + kk;
      float w = 0.f;
      if (n0 + co < a.Cout && kg < Ktot) w = a.weight[(long long)(n0 + co) * Ktot + kg];
      wT[co * LD + kk] = w;
    }
    __syncthreads();
    // ---- MFMA over the chunk ----
    const int arow = (wm * 32 + (lane & 15)) * LD + (lane >> 4);
    const int brow = (wn * (BN / 2) + (lane & 15)) * LD + (lane >> 4);
#pragma unroll
    for (int ks = 0; ks < KK / 4; ++ks) {
      float af[2], bf[NT];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = colT[arow + i * 16 * LD + ks * 4];
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j] = wT[brow + j * 16 * LD + ks * 4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: C layout col = lane&15 (cout), row = (lane>>4)*4 + reg (pixel) ----
  float* ob = a.out + (long long)b * a.Cout * HWo;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = n0 + wn * (BN / 2) + j * 16 + (lane & 15);
    if (co >= a.Cout) continue;
    float sc = 1.f, sh = 0.f;
    if (a.ep_scale) sc = a.ep_scale[co];
    if (a.ep_shift) sh = a.ep_shift[co];
    else if (a.bias) sh = a.bias[co];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pp = blockIdx.x * BM + wm * 32 + i * 16 + (lane >> 4) * 4;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[i][j][r] * sc + sh;
        if (a.relu) v[r] = fmaxf(v[r], 0.f);
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

template <int BN, int KC>
int launch(const DcnFwdArgs& a, hipStream_t st) {
  constexpr int LD = KC * TAPS + 1;
  const size_t lds = (size_t)(BM + BN) * LD * sizeof(float);
  dim3 grid((a.Ho * a.Wo + BM - 1) / BM, (a.Cout + BN - 1) / BN, a.B);
  hipLaunchKernelGGL((dcn_fwd_kernel<BN, KC>), grid, dim3(256), lds, st, a);
  return cp_launch_status();
}

}  // namespace

extern "C" int cp_dcn_v2_forward(const cp_dcn_shape* s, const float* x, const float* offset,
                                 int64_t offset_bstride, const float* mask,
                                 int64_t mask_bstride, int32_t mask_is_logit,
                                 const float* weight, const float* bias, const float* ep_scale,
                                 const float* ep_shift, int32_t relu, float* out, void* stream) {
  CP_CHECK_ARG(s && x && offset && mask && weight && out);
  CP_CHECK_ARG(s->B > 0 && s->Cin > 0 && s->H > 0 && s->W > 0 && s->Cout > 0);
  CP_CHECK_ARG(s->stride > 0 && s->dil > 0 && s->pad >= 0);
  if (s->kh != 3 || s->kw != 3 || s->deformable_groups != 1) return CP_EUNSUPPORTED;
  const int Ho = (s->H + 2 * s->pad - (s->dil * 2 + 1)) / s->stride + 1;
  const int Wo = (s->W + 2 * s->pad - (s->dil * 2 + 1)) / s->stride + 1;
  CP_CHECK_ARG(Ho > 0 && Wo > 0);
  if ((long long)s->H * s->W >= (1ll << 31) || (long long)Ho * Wo >= (1ll << 31)) return CP_EUNSUPPORTED;
  if (s->B > 65535) return CP_EUNSUPPORTED;
  DcnFwdArgs a;
  a.x = x; a.offset = offset; a.mask = mask; a.weight = weight; a.bias = bias;
  a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.out = out;
  a.offset_bstride = offset_bstride; a.mask_bstride = mask_bstride;
  a.B = s->B; a.Cin = s->Cin; a.H = s->H; a.W = s->W; a.Cout = s->Cout; a.Ho = Ho; a.Wo = Wo;
  a.stride = s->stride; a.pad = s->pad; a.dil = s->dil;
  a.mask_is_logit = mask_is_logit; a.relu = relu;
  hipStream_t st = (hipStream_t)stream;
  if (s->Cout <= 64) return launch<64, 8>(a, st);
  if (s->Cout <= 128) return launch<128, 8>(a, st);
  return launch<256, 4>(a, st);
}
