// The 7x7 / pad 3 convolutions of a 3-channel image (+ folded-BN shift, ReLU): the Hourglass stem, stride 2 to 128 channels
// (src/lib/models/networks/large_hourglass.py:287-290 `pre = convolution(7, 3, 128, stride=2)`; cuDNN in the reference, im2col
// + GEMM in the library here: 0.31 ms of BASELINE config 4's 18.6 ms), and DLA's `base_layer`, stride 1 to 16 channels
// (pose_dla_dcn.py:236-241; 108 us on the exact-f32 direct kernel of conv_direct.hip).
//
//   out[b][co][y][x] = act(bias[co] + sum_{ci, ky, kx} w[co][ci][ky][kx] * in[b][ci][S y - 3 + ky][S x - 3 + kx])
//
// Split-bf16 x3 on v_mfma_f32_16x16x32_bf16 (the arithmetic of conv_mfma.hip).  K = 3 x 7 x 7 = 147 is laid out as
// 21 (ky, ci) pairs x 8 columns: the 8 k-values of a lane are the input columns S x - 4 .. S x + 3 of one (ky, ci) row
// (column S x - 4 carries a zero weight), 4 pairs per 32-wide k-step, 6 k-steps (pairs 21 .. 23 are zero): 192 / 147 of
// the useful matrix work.
//   * B (the image): the tile's input rows are staged ONCE per workgroup as 16-byte records rec[ci][row][x] = the 8
//     columns S x - 4 .. S x + 3 of that row, split to bf16 hi | lo -- every input element sits in 8 / S records, which
//     buys ONE conflict-free ds_read_b128 per fragment half (16 lanes x 16 B contiguous) instead of four 4-byte-aligned
//     dword reads.  (Stride 2: 3 x 21 rows x 32 records x 16 B x 2 halves = 63 KB for the 8 x 32 output tile; stride 1: 43 KB.)
//   * A (the weights): split and laid out in fragment order by a prologue kernel (48 KB per 64 output channels,
//     L2-resident), one 16-byte load per lane and fragment, a k-step ahead.
//   * workgroup = 16 MT output channels (MT = 4, or 1 for the 16-channel base layer) x 8 rows x 32 pixels, 4 waves
//     (2 rows each: MT x 4 accumulator tiles).  Stride 1 puts every input element in eight records (43 KB per tile).
#include "cp_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TH = 8, TW = 32;                     // output tile
constexpr int KSTEPS = 6, PAIRS = 21;
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

template <int S, int MT>
__global__ __launch_bounds__(256, 2) void conv_stem_kernel(StemArgs a) {
  constexpr int IR = S * (TH - 1) + 7;             // staged input rows: S y0 - 3 .. S (y0 + TH - 1) + 3
  // LDS addressing of record xr of a (channel, row): RP = 34 slots per row, one pad slot after every 16 records.  The
  // staging writes record r of segment seg of four rows per 16-lane group: with 32 slots per row and the segments 8
  // apart every lane of a group fell on banks {0, 32} + 4 r -- 8-way conflicts, 67 % of the LDS-active cycles (round-4
  // PMC passes, 74.9 us); with the pad the segments start at dwords {0, 32, 68, 100} and the rows 136 apart: the 16
  // lanes take the 16 different 4-bank groups, and a fragment read (the 16 records of one 16-aligned tile) stays
  // contiguous.
  constexpr int RP = TW + TW / 16;                 // 34 slots per row
  constexpr int PLANE = 3 * IR * RP;               // slots per half
  __shared__ bf16x8 Xs[2 * PLANE];
  auto slot = [](int xr) { return xr + (xr >> 4); };
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, c = lane & 15;
  const int cot = blockIdx.x % a.ncot, tile = blockIdx.x / a.ncot;
  const int x0 = (tile % a.tiles_x) * TW, y0 = (tile / a.tiles_x) * TH, b = blockIdx.y;
  const int HW = a.H * a.W;

  // first weight fragments (k-step 0) while the image tile is staged
  const long long tstride = (long long)KSTEPS * 2 * 64;        // fragments per 16-row weight tile
  const bf16x8* wq = a.wp + (long long)cot * MT * tstride + lane;
  bf16x8 af[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    af[m][0] = wq[m * tstride];
    af[m][1] = wq[m * tstride + 64];
  }

  // staging: record (ci, row, xr) = input columns S (x0 + xr) - 4 .. + 7 of input row S y0 - 3 + row
  {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x + (long long)b * 3 * HW), 0, (int)(3u * (unsigned)HW * 4u), 0x00020000);
    if ((a.W & 3) == 0) {
      // one thread = 8 consecutive records of one (channel, row): their 8 S + 8 input columns start 16-byte aligned, are
      // loaded as float4 and split ONCE into packed bf16 pairs; a record is four consecutive pairs (an odd first column,
      // stride 1 only, takes them through v_alignbit).  (The per-record form below loads and splits every element 8 / S
      // times: the stride-1 base layer was bound by it.)
      constexpr int NV = 2 * S + 2;                            // float4 per thread: 16 (stride 1) or 24 (stride 2) columns
      if (tid < 3 * IR * 4) {
        const int seg = tid & 3, row = (tid >> 2) % IR, ci = (tid >> 2) / IR;
        const int gy = S * y0 - 3 + row, gx0 = S * (x0 + 8 * seg) - 4;
        const bool rok = gy >= 0 && gy < a.H;
        unsigned ph[2 * NV], pl[2 * NV];                       // packed (column 2 i, 2 i + 1) pairs, hi and lo halves
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const int gx = gx0 + 4 * q;
          const unsigned off = (rok && gx >= 0 && gx < a.W) ? ((unsigned)ci * (unsigned)HW + (unsigned)(gy * a.W + gx)) * 4u : OOB;
          const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
          const float e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
          typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
          bf16x2 h0, h1, l0, l1;
          h0[0] = (__bf16)e0; h0[1] = (__bf16)e1; h1[0] = (__bf16)e2; h1[1] = (__bf16)e3;
          l0[0] = (__bf16)(e0 - (float)h0[0]); l0[1] = (__bf16)(e1 - (float)h0[1]);
          l1[0] = (__bf16)(e2 - (float)h1[0]); l1[1] = (__bf16)(e3 - (float)h1[1]);
          ph[2 * q] = __builtin_bit_cast(unsigned, h0); ph[2 * q + 1] = __builtin_bit_cast(unsigned, h1);
          pl[2 * q] = __builtin_bit_cast(unsigned, l0); pl[2 * q + 1] = __builtin_bit_cast(unsigned, l1);
        }
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4* Xu = reinterpret_cast<u32x4*>(Xs);
        const int u0 = (ci * IR + row) * RP + 8 * seg + (seg >> 1);
#pragma unroll
        for (int r = 0; r < 8; ++r) {                          // record r starts at column offset S r (pair S r / 2)
          u32x4 rh, rl;
          if ((S * r) % 2 == 0) {
            const int k = S * r / 2;
            rh = u32x4{ph[k], ph[k + 1], ph[k + 2], ph[k + 3]};
            rl = u32x4{pl[k], pl[k + 1], pl[k + 2], pl[k + 3]};
          } else {
            const int k = (S * r - 1) / 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              rh[j] = __builtin_amdgcn_alignbit(ph[k + j + 1], ph[k + j], 16);
              rl[j] = __builtin_amdgcn_alignbit(pl[k + j + 1], pl[k + j], 16);
            }
          }
          Xu[u0 + r] = rh;
          Xu[PLANE + u0 + r] = rl;
        }
      }
    } else {
      for (int u = tid; u < 3 * IR * TW; u += 256) {
        const int xr = u % TW, row = (u / TW) % IR, ci = u / (TW * IR);
        const int su = (ci * IR + row) * RP + slot(xr);
        const int gy = S * y0 - 3 + row, gx0 = S * (x0 + xr) - 4;
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
        Xs[su] = h;
        Xs[PLANE + su] = l;
      }
    }
  }
  __syncthreads();

  f32x4 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    bf16x8 an[MT][2];
    {
      const bf16x8* nq = wq + (long long)((s < KSTEPS - 1 ? s + 1 : s) * 2) * 64;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
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
      const int idx = (ci * IR + S * yy + ky) * RP + slot((n & 1) * 16 + c);
      bh[n] = Xs[idx];
      bl[n] = Xs[PLANE + idx];
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        f32x4& d = acc[m][n];
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][0], bh[n], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][0], bl[n], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][1], bh[n], d, 0, 0, 0);
      }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      af[m][0] = an[m][0];
      af[m][1] = an[m][1];
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // epilogue: D[row = 4 g + r (co)][col = c (pixel)]; invalid elements get an offset past the descriptor (dropped)
  const int HWo = a.Ho * a.Wo;
  const int cot0 = cot * MT * 16;
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(
      a.out + ((long long)b * a.Cout + cot0) * HWo, 0, (int)((unsigned)min(a.Cout - cot0, MT * 16) * (unsigned)HWo * 4u), 0x00020000);
#pragma unroll
  for (int m = 0; m < MT; ++m)
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

int cp_conv7x7_c3_supported(int32_t Cout, int32_t H, int32_t W, int32_t stride) {
  if (Cout < 1 || H < 1 || W < 1 || (stride != 1 && stride != 2)) return 0;
  const long long Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  return 3ll * H * W * 4 < 0x7FFFFFF0ll && 64ll * Ho * Wo * 4 < 0x7FFFFFF0ll;
}

size_t cp_conv7x7_c3_weight_bytes(int32_t Cout) { return (size_t)((Cout + 63) / 64 * 4) * KSTEPS * 2 * 64 * 16; }

int cp_conv7x7_c3_prepare(const float* weight, int32_t Cout, void* wperm, void* stream) {
  CP_CHECK_ARG(weight && wperm && Cout >= 1);
  const int total = ((Cout + 63) / 64 * 4) * KSTEPS * 2 * 64;
  hipLaunchKernelGGL(stem_wperm_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, weight,
                     (bf16x8*)wperm, Cout, total);
  return cp_launch_status();
}

int cp_conv7x7_c3_forward(const float* x, const void* wperm, const float* bias, float* out, int32_t B, int32_t H,
                          int32_t W, int32_t Cout, int32_t stride, int32_t relu, void* stream) {
  CP_CHECK_ARG(x && wperm && out && B >= 1 && B <= 65535);
  if (!cp_conv7x7_c3_supported(Cout, H, W, stride)) return CP_EUNSUPPORTED;
  StemArgs a;
  a.x = x; a.wp = (const bf16x8*)wperm; a.bias = bias; a.out = out;
  a.H = H; a.W = W; a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;
  const int mt = Cout <= 16 ? 1 : 4;                 // 16 output channels per workgroup for the 16-channel base layer
  a.Cout = Cout; a.ncot = (Cout + 16 * mt - 1) / (16 * mt); a.tiles_x = (a.Wo + TW - 1) / TW; a.relu = relu;
  const long long wgs = (long long)a.tiles_x * ((a.Ho + TH - 1) / TH) * a.ncot;
  if (wgs > 0x7FFFFFFFll) return CP_EUNSUPPORTED;
  const dim3 grid((unsigned)wgs, B);
  hipStream_t st = (hipStream_t)stream;
  if (stride == 2 && mt == 4) hipLaunchKernelGGL((conv_stem_kernel<2, 4>), grid, dim3(256), 0, st, a);
  else if (stride == 2) hipLaunchKernelGGL((conv_stem_kernel<2, 1>), grid, dim3(256), 0, st, a);
  else if (mt == 4) hipLaunchKernelGGL((conv_stem_kernel<1, 4>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv_stem_kernel<1, 1>), grid, dim3(256), 0, st, a);
  return cp_launch_status();
}

}  // extern "C"
