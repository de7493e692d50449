// The Hourglass stem: 7x7 / stride 2 / pad 3 convolution of a 3-channel image to Cout channels (+ folded-BN shift, ReLU)
// (src/lib/models/networks/large_hourglass.py:287-290 `pre = convolution(7, 3, 128, stride=2)`; the reference runs it as
// a cuDNN convolution, the library here as im2col + GEMM: 0.31 ms of BASELINE config 4's 18.6 ms).
//
//   out[b][co][y][x] = act(bias[co] + sum_{ci, ky, kx} w[co][ci][ky][kx] * in[b][ci][2 y - 3 + ky][2 x - 3 + kx])
//
// Split-bf16 x3 on v_mfma_f32_16x16x32_bf16 (the arithmetic of conv_mfma.hip).  K = 3 x 7 x 7 = 147 is laid out as
// 21 (ky, ci) pairs x 8 columns: the 8 k-values of a lane are the input columns 2 x - 4 .. 2 x + 3 of one (ky, ci) row
// (column 2 x - 4 carries a zero weight), 4 pairs per 32-wide k-step, 6 k-steps (pairs 21 .. 23 are zero): 192 / 147 of
// the useful matrix work.
//   * B (the image): the tile's input rows are staged ONCE per workgroup as 16-byte records rec[ci][row][x] = the 8
//     columns 2 x - 4 .. 2 x + 3 of that row, split to bf16 hi | lo -- every input element sits in four records, which
//     buys ONE conflict-free ds_read_b128 per fragment half (16 lanes x 16 B contiguous) instead of four 4-byte-aligned
//     dword reads.  (3 x 21 rows x 32 records x 16 B x 2 halves = 63 KB for the 8 x 32 output tile.)
//   * A (the weights): split and laid out in fragment order by a prologue kernel (48 KB per 64 output channels,
//     L2-resident), one 16-byte load per lane and fragment, a k-step ahead.
//   * workgroup = 64 output channels x 8 rows x 32 pixels, 4 waves (2 rows each: 4 x 4 accumulator tiles).
#include "cp_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TH = 8, TW = 32;                     // output tile
constexpr int IR = 2 * TH + 5;                     // staged input rows: 2 y0 - 3 .. 2 y0 + 2 TH + 1
constexpr int KSTEPS = 6, PAIRS = 21;
constexpr int PLANE = 3 * IR * TW;                 // records per half
constexpr unsigned OOB = 0x80000000u;

// wp[((ct * 6 + s) * 2 + hl) * 64 + lane][j] = half(hl) of w[co = ct * 16 + (lane & 15)][ci][ky][j - 1]
// with (ky, ci) = pair 4 s + (lane >> 4) (ky = pair / 3, ci = pair % 3); j = 0, pairs >= 21 and co >= Cout: zero.
__global__ __launch_bounds__(256) void stem_wperm_kernel(const float* __restrict__ w, bf16x8* __restrict__ wp, int Cout,
                                                         int total) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int lane = e & 63;
  int r = e >> 6;
  const int hl = r & 1;
  r >>= 1;
  const int s = r % KSTEPS, ct = r / KSTEPS;
  const int co = ct * 16 + (lane & 15), pair = 4 * s + (lane >> 4);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = 0.f;
    if (co < Cout && pair < PAIRS && j >= 1) v = w[((long long)(co * 3 + pair % 3) * 7 + pair / 3) * 7 + (j - 1)];
    const __bf16 h = (__bf16)v;
    o[j] = hl ? (__bf16)(v - (float)h) : h;
  }
  wp[e] = o;
}

struct StemArgs {
  const float* x;
  const bf16x8* wp;
  const float* bias;
  float* out;
  int H, W, Ho, Wo, Cout, ncot, tiles_x, relu;
};

__global__ __launch_bounds__(256, 2) void conv_stem_kernel(StemArgs a) {
  __shared__ bf16x8 Xs[2 * PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, c = lane & 15;
  const int cot = blockIdx.x % a.ncot, tile = blockIdx.x / a.ncot;
  const int x0 = (tile % a.tiles_x) * TW, y0 = (tile / a.tiles_x) * TH, b = blockIdx.y;
  const int HW = a.H * a.W;

  // first weight fragments (k-step 0) while the image tile is staged
  const long long tstride = (long long)KSTEPS * 2 * 64;        // fragments per 16-row weight tile
  const bf16x8* wq = a.wp + (long long)cot * 4 * tstride + lane;
  bf16x8 af[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    af[m][0] = wq[m * tstride];
    af[m][1] = wq[m * tstride + 64];
  }

  // staging: record (ci, row, xr) = input columns 2 (x0 + xr) - 4 .. + 7 of input row 2 y0 - 3 + row
  {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x + (long long)b * 3 * HW), 0, (int)(3u * (unsigned)HW * 4u), 0x00020000);
    for (int u = tid; u < PLANE; u += 256) {
      const int xr = u % TW, row = (u / TW) % IR, ci = u / (TW * IR);
      const int gy = 2 * y0 - 3 + row, gx0 = 2 * (x0 + xr) - 4;
      const bool rok = gy >= 0 && gy < a.H;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int gx = gx0 + j;
        const unsigned off = (rok && gx >= 0 && gx < a.W) ? ((unsigned)ci * (unsigned)HW + (unsigned)(gy * a.W + gx)) * 4u : OOB;
        v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
      }
      bf16x8 h, l;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const __bf16 hh = (__bf16)v[j];
        h[j] = hh;
        l[j] = (__bf16)(v[j] - (float)hh);
      }
      Xs[u] = h;
      Xs[PLANE + u] = l;
    }
  }
  __syncthreads();

  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    bf16x8 an[4][2];
    {
      const bf16x8* nq = wq + (long long)((s < KSTEPS - 1 ? s + 1 : s) * 2) * 64;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        an[m][0] = nq[m * tstride];
        an[m][1] = nq[m * tstride + 64];
      }
    }
    const int pair = min(4 * s + g, PAIRS - 1);               // (pairs past 20 carry zero weights: any record will do)
    const int ky = pair / 3, ci = pair - 3 * ky;
    bf16x8 bh[4], bl[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int yy = wid * 2 + (n >> 1);                       // output row within the tile
      const int idx = (ci * IR + 2 * yy + ky) * TW + (n & 1) * 16 + c;
      bh[n] = Xs[idx];
      bl[n] = Xs[PLANE + idx];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        f32x4& d = acc[m][n];
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][0], bh[n], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][0], bl[n], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][1], bh[n], d, 0, 0, 0);
      }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      af[m][0] = an[m][0];
      af[m][1] = an[m][1];
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // epilogue: D[row = 4 g + r (co)][col = c (pixel)]; invalid elements get an offset past the descriptor (dropped)
  const int HWo = a.Ho * a.Wo;
  const int cot0 = cot * 64;
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(
      a.out + ((long long)b * a.Cout + cot0) * HWo, 0, (int)((unsigned)min(a.Cout - cot0, 64) * (unsigned)HWo * 4u), 0x00020000);
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = m * 16 + 4 * g + r;
      const float bv = (a.bias && cot0 + col < a.Cout) ? a.bias[cot0 + col] : 0.f;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int y = y0 + wid * 2 + (n >> 1), x = x0 + (n & 1) * 16 + c;
        const bool ok = cot0 + col < a.Cout && y < a.Ho && x < a.Wo;
        float v = acc[m][n][r] + bv;
        if (a.relu) v = fmaxf(v, 0.f);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_o,
                                              ok ? ((unsigned)col * (unsigned)HWo + (unsigned)(y * a.Wo + x)) * 4u : OOB, 0, 0);
      }
    }
}

}  // namespace

extern "C" {

int cp_conv7x7s2_c3_supported(int32_t Cout, int32_t H, int32_t W) {
  if (Cout < 1 || H < 1 || W < 1) return 0;
  const long long Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  return 3ll * H * W * 4 < 0x7FFFFFF0ll && 64ll * Ho * Wo * 4 < 0x7FFFFFF0ll;
}

size_t cp_conv7x7s2_c3_weight_bytes(int32_t Cout) { return (size_t)((Cout + 63) / 64 * 4) * KSTEPS * 2 * 64 * 16; }

int cp_conv7x7s2_c3_prepare(const float* weight, int32_t Cout, void* wperm, void* stream) {
  CP_CHECK_ARG(weight && wperm && Cout >= 1);
  const int total = ((Cout + 63) / 64 * 4) * KSTEPS * 2 * 64;
  hipLaunchKernelGGL(stem_wperm_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, weight,
                     (bf16x8*)wperm, Cout, total);
  return cp_launch_status();
}

int cp_conv7x7s2_c3_forward(const float* x, const void* wperm, const float* bias, float* out, int32_t B, int32_t H,
                            int32_t W, int32_t Cout, int32_t relu, void* stream) {
  CP_CHECK_ARG(x && wperm && out && B >= 1 && B <= 65535);
  if (!cp_conv7x7s2_c3_supported(Cout, H, W)) return CP_EUNSUPPORTED;
  StemArgs a;
  a.x = x; a.wp = (const bf16x8*)wperm; a.bias = bias; a.out = out;
  a.H = H; a.W = W; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
  a.Cout = Cout; a.ncot = (Cout + 63) / 64; a.tiles_x = (a.Wo + TW - 1) / TW; a.relu = relu;
  const long long wgs = (long long)a.tiles_x * ((a.Ho + TH - 1) / TH) * a.ncot;
  if (wgs > 0x7FFFFFFFll) return CP_EUNSUPPORTED;
  hipLaunchKernelGGL(conv_stem_kernel, dim3((unsigned)wgs, B), dim3(256), 0, (hipStream_t)stream, a);
  return cp_launch_status();
}

}  // extern "C"
