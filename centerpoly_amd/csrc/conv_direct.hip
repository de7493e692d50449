// Direct convolution of the full-resolution, low-channel layers of the DLA base for gfx950, with the
// folded-BatchNorm shift and the ReLU in the epilogue.
//
// Replaces, at inference, `base.base_layer` (7x7, 3 -> 16), `base.level0` (3x3, 16 -> 16) and
// `base.level1` (3x3 stride 2, 16 -> 32) of the reference's DLA-34
// (src/lib/models/networks/pose_dla_dcn.py:236-246,266-276): Conv2d(bias=False) -> BatchNorm2d ->
// ReLU.  The library ran them as Winograd / implicit-GEMM kernels plus a separate bias+ReLU pass:
// 371 + 240 + 112 us and 123 us of epilogues per 2048x1024 image (profiles/r01_infer_*), for layers
// whose 16 channels make the work a thin contraction (K = 147 / 144) over 2 M pixels.
//
// Implicit GEMM on the f32 matrix instruction, out[co][px] = sum_k W[co][k] * x_patch[k][px]:
//   A (weights, BN scale folded in): the wave's registers for the whole kernel -- lane (co = lane & 15,
//     k = 4 ks + (lane >> 4)) holds one value per k-step (36..39 k-steps, x2 for 32 output channels);
//   B (input patch): read per k-step from an LDS-staged halo tile, lane = (pixel lane & 15, k lane >> 4);
//     k-order = tap-major for the 16-channel layers (a k-step = 4 input channels of one tap, so the
//     four lane groups read the same tile position in four planes whose stride is 16 (stride 1) or 17
//     (stride 2) banks: conflict-free), channel-major for the stem (a k-step = 4 taps of one channel,
//     per-lane tap offsets precomputed, 49 taps padded to 52);
//   D: pixel on the lane (16 consecutive pixels of a row), four output channels in the accumulator
//     registers: bias + ReLU and 64-byte row-segment stores, no transposition.
// One workgroup = 4 waves = an 8 x 64 (stride 1) or 4 x 64 (stride 2) output tile; the halo tile is
// staged once with coalesced row loads.  Exact fp32 (fma chain of the MFMA), k-order differs from the
// library's, so results agree with F.conv2d to rounding (tests: 1e-5 relative).
#include "cp_common.h"

namespace {

constexpr unsigned OOBC = 0x80000000u;

struct ConvArgs {
  const float* x;
  const float* w;       // [Cout][Cin][k][k], scale folded in by the caller
  const float* bias;    // [Cout] or null
  float* out;
  int B, H, W, Ho, Wo;
  int relu;
};

// ---------------------------------------------------------------- 3x3, 16 input channels ---
template <int COUT, int STRIDE>
__global__ __launch_bounds__(256) void conv3x3_c16_kernel(ConvArgs a) {
  constexpr int CIN = 16, MT = COUT / 16, KS = 9 * CIN / 4;          // 36 k-steps
  constexpr int TH = STRIDE == 1 ? 8 : 4, TW = 64;                   // output tile
  constexpr int IH = (TH - 1) * STRIDE + 3;
  constexpr int NC4 = ((TW - 1) * STRIDE + 3 + 3 + 3) / 4;           // float4 chunks per tile row (from column -4)
  constexpr int PITCH = STRIDE == 1 ? 72 : 137;                      // plane = IH * PITCH = 16 / 17 (mod 32)
  constexpr int PLANE = IH * PITCH;
  static_assert(PITCH >= 4 * NC4, "pitch");
  static_assert((PLANE % 32) == (STRIDE == 1 ? 16 : 17), "bank phase of the channel planes");
  __shared__ __attribute__((aligned(16))) float xs[CIN * PLANE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lpx = lane & 15, g = lane >> 4;
  const int b = blockIdx.z;
  const int oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
  const int iy0 = oy0 * STRIDE - 1;
  const int HW = a.H * a.W;

  // weights -> registers: A[co = 16 m + lpx][k = 4 ks + g], k = tap * 16 + ci (gathers straight from global memory, in
  // flight together with the tile's loads; staging them through LDS as the bf16 kernel below does was measured here:
  // level0 125 -> 150 us, level1 71 -> 84 -- two more barriers in front of a kernel that runs two workgroups per CU)
  float wa[MT][KS];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = 4 * ks + g, tap = k >> 4, ci = k & 15;
      wa[m][ks] = a.w[((16 * m + lpx) * CIN + ci) * 9 + tap];
    }

  // stage the halo tile, zero outside the image.  The tile's first column is the 16-byte aligned
  // ox0 * STRIDE - 4 (tap column kx sits at tile column 3 + kx): whole float4 chunks are inside or
  // outside the image when W % 4 == 0, and all of a thread's loads are in flight together.
  const float* xb = a.x + (long long)b * CIN * HW;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xb), 0, (int)((unsigned)CIN * (unsigned)HW * 4u), 0x00020000);
  const int gx0 = ox0 * STRIDE - 4;
  if ((a.W & 3) == 0) {
    constexpr int NQ = CIN * IH * NC4, NI = (NQ + 255) / 256;
    f32x4 v[NI];
    int dst[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = tid + 256 * i;
      const int ci = q / (IH * NC4), r = q - ci * (IH * NC4);
      const int ry = r / NC4, c4 = r - ry * NC4;
      const int gy = iy0 + ry, gx = gx0 + 4 * c4;
      const bool ok = q < NQ && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      const unsigned off = ok ? ((unsigned)ci * (unsigned)HW + (unsigned)(gy * a.W + gx)) * 4u : OOBC;
      v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
      dst[i] = q < NQ ? ci * PLANE + ry * PITCH + 4 * c4 : -1;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (dst[i] >= 0) {
        if (PLANE % 4 == 0 && PITCH % 4 == 0) {
          *reinterpret_cast<f32x4*>(&xs[dst[i]]) = v[i];
        } else {
          xs[dst[i]] = v[i].x; xs[dst[i] + 1] = v[i].y; xs[dst[i] + 2] = v[i].z; xs[dst[i] + 3] = v[i].w;
        }
      }
    }
  } else {
    for (int e = tid; e < CIN * IH * (4 * NC4); e += 256) {
      const int ci = e / (IH * 4 * NC4), r = e - ci * (IH * 4 * NC4);
      const int ry = r / (4 * NC4), rx = r - ry * (4 * NC4);
      const int gy = iy0 + ry, gx = gx0 + rx;
      const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      const unsigned off = ok ? ((unsigned)ci * (unsigned)HW + (unsigned)(gy * a.W + gx)) * 4u : OOBC;
      xs[ci * PLANE + ry * PITCH + rx] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, off, 0, 0));
    }
  }
  __syncthreads();

  // wave w: rows [w * TH / 4, ...) of the tile, all four 16-pixel column tiles
  constexpr int RPW = TH / 4 > 0 ? TH / 4 : 1;                       // rows per wave (2 or 1)
  float bias_r[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) bias_r[m][q] = a.bias ? a.bias[16 * m + 4 * g + q] : 0.f;
  float* ob = a.out + (long long)b * COUT * a.Ho * a.Wo;
#pragma unroll 1
  for (int rr = 0; rr < RPW; ++rr) {
    const int ty = wid * RPW + rr;
    // the row's four 16-pixel tiles together: independent accumulator chains keep the matrix pipe issuing (one tile at
    // a time was a chain of 36 dependent MFMAs per output-channel fragment)
    constexpr int NTX = TW / 16;
    f32x4 acc[MT][NTX];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int tx = 0; tx < NTX; ++tx) acc[m][tx] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* base = xs + g * PLANE + (ty * STRIDE) * PITCH + lpx * STRIDE + 3;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {                               // k-step = tap * 4 + c4: channels 4 c4 + g
        float bv[NTX];
#pragma unroll
        for (int tx = 0; tx < NTX; ++tx) bv[tx] = base[c4 * 4 * PLANE + ky * PITCH + kx + tx * 16 * STRIDE];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int tx = 0; tx < NTX; ++tx)
            acc[m][tx] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[m][tap * 4 + c4], bv[tx], acc[m][tx], 0, 0, 0);
      }
    }
    const int oy = oy0 + ty;
#pragma unroll
    for (int tx = 0; tx < NTX; ++tx) {
      const int ox = ox0 + tx * 16 + lpx;
      if (oy < a.Ho && ox < a.Wo) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float v = acc[m][tx][q] + bias_r[m][q];
            if (a.relu) v = fmaxf(v, 0.f);
            ob[((long long)(16 * m + 4 * g + q) * a.Ho + oy) * a.Wo + ox] = v;
          }
      }
    }
  }
}

// ------------------------------------------- 3x3, 16 input channels, split-bf16 x3 (round 4) ---
// The same layers on the bf16 matrix cores (the arithmetic of conv_mfma.hip: a * b ~ ah*bh + ah*bl + al*bh, fp32
// accumulate, ~2^-16 per product).  The exact kernel above sits on the f32 matrix pipe (level0: 104 us against 61 at
// that pipe's peak); three v_mfma_f32_16x16x32_bf16 per 32 k-values are a fifth of its matrix time, which leaves the
// layers at their memory traffic.  K = tap * 16 + ci, 144 padded to 160 = five k-steps of two taps x 16 channels: lane
// (pixel lane & 15, group g = lane >> 4) supplies the 8 channels 8 (g & 1) .. of tap 2 s + (g >> 1) -- ONE ds_read_b128
// per half (hi / lo) from a halo tile staged as [hi | lo][channel half][cell][8 x bf16] (16 lanes of a group read 16
// neighbouring cells: 256 contiguous bytes at stride 1).  Weights: split once into the wave's registers (A fragments).
// D as above: pixel on the lane, four output channels in the accumulator registers.
typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cbf16x2 __attribute__((ext_vector_type(2)));
typedef float cf32x2 __attribute__((ext_vector_type(2)));
typedef unsigned cu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void csplit2(float v0, float v1, unsigned& hi, unsigned& lo) {
  const cbf16x2 h = __builtin_convertvector(cf32x2{v0, v1}, cbf16x2);
  const unsigned hb = __builtin_bit_cast(unsigned, h);
  const float h0 = __builtin_bit_cast(float, hb << 16), h1 = __builtin_bit_cast(float, hb & 0xffff0000u);
  const cbf16x2 l = __builtin_convertvector(cf32x2{v0 - h0, v1 - h1}, cbf16x2);
  hi = hb;
  lo = __builtin_bit_cast(unsigned, l);
}

template <int COUT, int STRIDE>
__global__ __launch_bounds__(256) void conv3x3_c16_bf16_kernel(ConvArgs a) {
  constexpr int CIN = 16, MT = COUT / 16, KS = 5;                    // five k-steps of 32 (taps 2 s, 2 s + 1; tap 9 = padding)
  constexpr int TH = STRIDE == 1 ? 8 : 4, TW = 64;                   // output tile
  constexpr int IH = (TH - 1) * STRIDE + 3;
  constexpr int NC4 = ((TW - 1) * STRIDE + 3 + 3 + 3) / 4;           // float4 chunks per tile row (from column -4)
  constexpr int IW = 4 * NC4;                                        // cells per tile row
  constexpr int CELLS = IH * IW;
  constexpr int HALF = CELLS * 16;                                   // bytes of one (hi | lo, channel half) plane
  __shared__ __attribute__((aligned(16))) unsigned char xs[4 * HALF];   // [hi | lo][half][cell][8 bf16]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lpx = lane & 15, g = lane >> 4;
  const int b = blockIdx.z;
  const int oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
  const int iy0 = oy0 * STRIDE - 1;
  const int HW = a.H * a.W;

  // weights -> registers: A[co = 16 m + lpx][k = 32 s + 8 g + j] = W[co][ci = 8 (g & 1) + j][tap = 2 s + (g >> 1)].
  // Through LDS (round 4): the tensor is read ONCE per workgroup with coalesced loads and every lane picks its values
  // with ds_reads, under the latency of the tile's loads -- straight from global memory each wave issued 40 MT dword
  // gathers over 16 rows 576 bytes apart (32+ cache lines per instruction through the texture addresser the CU's 12
  // waves share): level0 105 -> 91 us.
  cbf16x8 wh[MT][KS], wl[MT][KS];
  auto weights_via_lds = [&]() {
    constexpr int NWT = COUT * CIN * 9;
    static_assert(NWT * 4 <= 4 * HALF, "the weights fit the tile buffer");
    float* wsh = reinterpret_cast<float*>(xs);
    for (int e = tid; e < NWT; e += 256) wsh[e] = a.w[e];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int s_ = 0; s_ < KS; ++s_) {
        const int tap = 2 * s_ + (g >> 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = tap < 9 ? wsh[((16 * m + lpx) * CIN + 8 * (g & 1) + j) * 9 + tap] : 0.f;
          const __bf16 h = (__bf16)v;
          wh[m][s_][j] = h;
          wl[m][s_][j] = (__bf16)(v - (float)h);
        }
      }
    __syncthreads();                                                 // (the tile overwrites the buffer)
  };
  // per-lane byte offset of the lane's tap in k-step s (tile cell of tap (ky, kx) relative to the output pixel's cell)
  int toff[KS];
#pragma unroll
  for (int s_ = 0; s_ < KS; ++s_) {
    const int tap = min(2 * s_ + (g >> 1), 8);                       // (the padding tap multiplies zero weights)
    toff[s_] = ((tap / 3) * IW + (tap % 3)) * 16 + (g & 1) * HALF;
  }

  // stage the halo tile: one item = (channel half, row, float4 chunk): 8 channels x 4 pixels -> 4 cells x (hi, lo)
  const float* xb = a.x + (long long)b * CIN * HW;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xb), 0, (int)((unsigned)CIN * (unsigned)HW * 4u), 0x00020000);
  const int gx0 = ox0 * STRIDE - 4;                                  // tap column kx sits at tile column 3 + kx
  const bool vec = (a.W & 3) == 0;
  constexpr int NITEM = 2 * IH * NC4, NI = (NITEM + 255) / 256;
  if (vec) {
    // every load of the thread's items in flight before the first split (one memory latency per workgroup, not NI)
    f32x4 v[NI][8];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int it = tid + 256 * i;
      const int half = it / (IH * NC4), r = it - half * (IH * NC4);
      const int ry = r / NC4, c4 = r - ry * NC4;
      const int gy = iy0 + ry, gx = gx0 + 4 * c4;
      const bool ok = it < NITEM && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;   // (whole chunks are inside or outside)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned off = ok ? ((unsigned)(8 * half + j) * (unsigned)HW + (unsigned)(gy * a.W + gx)) * 4u : OOBC;
        v[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
      }
    }
    weights_via_lds();                                               // (under the tile loads' latency)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int it = tid + 256 * i;
      if (it < NITEM) {
        const int half = it / (IH * NC4), r = it - half * (IH * NC4);
        const int ry = r / NC4, c4 = r - ry * NC4;
        unsigned char* dst = xs + half * HALF + (ry * IW + 4 * c4) * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          unsigned hi[4], lo[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) csplit2(v[i][2 * jj][q], v[i][2 * jj + 1][q], hi[jj], lo[jj]);
          *reinterpret_cast<cu32x4*>(dst + q * 16) = cu32x4{hi[0], hi[1], hi[2], hi[3]};
          *reinterpret_cast<cu32x4*>(dst + 2 * HALF + q * 16) = cu32x4{lo[0], lo[1], lo[2], lo[3]};
        }
      }
    }
  } else {
    weights_via_lds();
#pragma unroll 1
    for (int it = tid; it < NITEM; it += 256) {
      const int half = it / (IH * NC4), r = it - half * (IH * NC4);
      const int ry = r / NC4, c4 = r - ry * NC4;
      const int gy = iy0 + ry, gx = gx0 + 4 * c4;
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool ok = gy >= 0 && gy < a.H && gx + q >= 0 && gx + q < a.W;
          const unsigned off = ok ? ((unsigned)(8 * half + j) * (unsigned)HW + (unsigned)(gy * a.W + gx + q)) * 4u : OOBC;
          v[j][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, off, 0, 0));
        }
      unsigned char* dst = xs + half * HALF + (ry * IW + 4 * c4) * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned hi[4], lo[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) csplit2(v[2 * jj][q], v[2 * jj + 1][q], hi[jj], lo[jj]);
        *reinterpret_cast<cu32x4*>(dst + q * 16) = cu32x4{hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<cu32x4*>(dst + 2 * HALF + q * 16) = cu32x4{lo[0], lo[1], lo[2], lo[3]};
      }
    }
  }
  __syncthreads();

  constexpr int RPW = TH / 4 > 0 ? TH / 4 : 1;                       // rows per wave (2 or 1)
  float bias_r[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) bias_r[m][q] = a.bias ? a.bias[16 * m + 4 * g + q] : 0.f;
  float* ob = a.out + (long long)b * COUT * a.Ho * a.Wo;
#pragma unroll 1
  for (int rr = 0; rr < RPW; ++rr) {
    const int ty = wid * RPW + rr;
    // the row's four 16-pixel tiles together: four independent accumulator chains per output-channel fragment (one
    // tile at a time was ONE chain of 15 dependent MFMAs per wave -- the matrix pipe idled on its own latency)
    constexpr int NTX = TW / 16;
    f32x4 acc[MT][NTX];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int tx = 0; tx < NTX; ++tx) acc[m][tx] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* base = xs + ((ty * STRIDE) * IW + lpx * STRIDE + 3) * 16;
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) {
      cbf16x8 bh[NTX], bl[NTX];
#pragma unroll
      for (int tx = 0; tx < NTX; ++tx) {
        bh[tx] = *reinterpret_cast<const cbf16x8*>(base + tx * 16 * STRIDE * 16 + toff[s_]);
        bl[tx] = *reinterpret_cast<const cbf16x8*>(base + tx * 16 * STRIDE * 16 + toff[s_] + 2 * HALF);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int tx = 0; tx < NTX; ++tx) {
          acc[m][tx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[m][s_], bh[tx], acc[m][tx], 0, 0, 0);
          acc[m][tx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[m][s_], bl[tx], acc[m][tx], 0, 0, 0);
          acc[m][tx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[m][s_], bh[tx], acc[m][tx], 0, 0, 0);
        }
    }
    const int oy = oy0 + ty;
#pragma unroll
    for (int tx = 0; tx < NTX; ++tx) {
      const int ox = ox0 + tx * 16 + lpx;
      if (oy < a.Ho && ox < a.Wo) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float v = acc[m][tx][q] + bias_r[m][q];
            if (a.relu) v = fmaxf(v, 0.f);
            ob[((long long)(16 * m + 4 * g + q) * a.Ho + oy) * a.Wo + ox] = v;
          }
      }
    }
  }
}

// ---------------------------------------------------------------- 7x7, 3 -> 16 (the stem) ---
__global__ __launch_bounds__(256) void conv7x7_c3_kernel(ConvArgs a) {
  constexpr int CIN = 3, TAPS7 = 49, TPAD = 52, KS1 = TPAD / 4, KS = CIN * KS1;   // 13 k-steps per channel
  constexpr int TH = 8, TW = 64, IH = TH + 6, NC4 = 18, PITCH = 72, PLANE = IH * PITCH;   // columns -4 .. 67
  static_assert(PLANE % 32 == 16, "bank phase");
  __shared__ __attribute__((aligned(16))) float xs[CIN * PLANE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lpx = lane & 15, g = lane >> 4;
  const int b = blockIdx.z;
  const int oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
  const int iy0 = oy0 - 3;
  const int HW = a.H * a.W;

  // A[co = lpx][k = ci * 52 + tap], taps 49..51 are zero; per-lane tile offset of tap 4 ks + g
  float wa[KS];
  int toff[KS1];
#pragma unroll
  for (int ks = 0; ks < KS1; ++ks) {
    const int tap = 4 * ks + g;
    const int tc = tap < TAPS7 ? tap : 0;
    toff[ks] = (tc / 7) * PITCH + (tc % 7);
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
      wa[ci * KS1 + ks] = tap < TAPS7 ? a.w[(lpx * CIN + ci) * TAPS7 + tap] : 0.f;
  }
  const float* xb = a.x + (long long)b * CIN * HW;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xb), 0, (int)((unsigned)CIN * (unsigned)HW * 4u), 0x00020000);
  const int gx0 = ox0 - 4;                                          // aligned; tap column kx sits at tile column 1 + kx
  if ((a.W & 3) == 0) {
    constexpr int NQ = CIN * IH * NC4, NI = (NQ + 255) / 256;
    f32x4 v[NI];
    int dst[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = tid + 256 * i;
      const int ci = q / (IH * NC4), r = q - ci * (IH * NC4);
      const int ry = r / NC4, c4 = r - ry * NC4;
      const int gy = iy0 + ry, gx = gx0 + 4 * c4;
      const bool ok = q < NQ && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      const unsigned off = ok ? ((unsigned)ci * (unsigned)HW + (unsigned)(gy * a.W + gx)) * 4u : OOBC;
      v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
      dst[i] = q < NQ ? ci * PLANE + ry * PITCH + 4 * c4 : -1;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (dst[i] >= 0) *reinterpret_cast<f32x4*>(&xs[dst[i]]) = v[i];
  } else {
    for (int e = tid; e < CIN * IH * (4 * NC4); e += 256) {
      const int ci = e / (IH * 4 * NC4), r = e - ci * (IH * 4 * NC4);
      const int ry = r / (4 * NC4), rx = r - ry * (4 * NC4);
      const int gy = iy0 + ry, gx = gx0 + rx;
      const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      const unsigned off = ok ? ((unsigned)ci * (unsigned)HW + (unsigned)(gy * a.W + gx)) * 4u : OOBC;
      xs[ci * PLANE + ry * PITCH + rx] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, off, 0, 0));
    }
  }
  __syncthreads();

  float bias_r[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bias_r[q] = a.bias ? a.bias[4 * g + q] : 0.f;
  float* ob = a.out + (long long)b * 16 * a.Ho * a.Wo;
#pragma unroll 1
  for (int rr = 0; rr < 2; ++rr) {
    const int ty = wid * 2 + rr;
#pragma unroll 1
    for (int tx = 0; tx < TW / 16; ++tx) {
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;          // two chains: 40-cycle dependent latency
      const float* base = xs + ty * PITCH + tx * 16 + lpx + 1;
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
          const float bv = base[ci * PLANE + toff[ks]];
          if ((ks & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ci * KS1 + ks], bv, acc0, 0, 0, 0);
          else acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ci * KS1 + ks], bv, acc1, 0, 0, 0);
        }
      const int oy = oy0 + ty, ox = ox0 + tx * 16 + lpx;
      if (oy < a.Ho && ox < a.Wo) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v = acc0[q] + acc1[q] + bias_r[q];
          if (a.relu) v = fmaxf(v, 0.f);
          ob[((long long)(4 * g + q) * a.Ho + oy) * a.Wo + ox] = v;
        }
      }
    }
  }
}

}  // namespace

extern "C" int cp_conv_direct_supported(int32_t Cin, int32_t Cout, int32_t k, int32_t stride, int32_t pad) {
  if (k == 7 && Cin == 3 && Cout == 16 && stride == 1 && pad == 3) return 1;
  if (k == 3 && Cin == 16 && pad == 1 && ((Cout == 16 && stride == 1) || (Cout == 32 && stride == 2) ||
                                          (Cout == 16 && stride == 2) || (Cout == 32 && stride == 1)))
    return 1;
  return 0;
}

extern "C" int cp_conv_direct_forward(const float* x, const float* w, const float* bias, float* out, int32_t B,
                                      int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t k, int32_t stride,
                                      int32_t pad, int32_t relu, void* stream) {
  return cp_conv_direct_forward_ex(x, w, bias, out, B, Cin, H, W, Cout, k, stride, pad, relu, 0, stream);
}

// split_bf16 != 0: the stride-1 3x3 / 16-input-channel layers (level0) contract in split-bf16 x3 on the bf16 matrix cores
// (the 7x7 stem has its own split-bf16 kernel, cp_conv7x7_c3_forward); 0: exact fp32 fma chains.
extern "C" int cp_conv_direct_forward_ex(const float* x, const float* w, const float* bias, float* out, int32_t B,
                                         int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t k, int32_t stride,
                                         int32_t pad, int32_t relu, int32_t split_bf16, void* stream) {
  CP_CHECK_ARG(x && w && out && B > 0 && H > 0 && W > 0);
  if (!cp_conv_direct_supported(Cin, Cout, k, stride, pad)) return CP_EUNSUPPORTED;
  if ((unsigned long long)Cin * H * W * 4ull >= 0x70000000ull || B > 65535) return CP_EUNSUPPORTED;
  ConvArgs a;
  a.x = x; a.w = w; a.bias = bias; a.out = out; a.B = B; a.H = H; a.W = W; a.relu = relu;
  a.Ho = (H + 2 * pad - k) / stride + 1;
  a.Wo = (W + 2 * pad - k) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  if (k == 7) {
    hipLaunchKernelGGL(conv7x7_c3_kernel, dim3((a.Wo + 63) / 64, (a.Ho + 7) / 8, B), dim3(256), 0, st, a);
  } else if (stride == 1) {
    const dim3 grid((a.Wo + 63) / 64, (a.Ho + 7) / 8, B);
    if (split_bf16) {
      if (Cout == 16) hipLaunchKernelGGL((conv3x3_c16_bf16_kernel<16, 1>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((conv3x3_c16_bf16_kernel<32, 1>), grid, dim3(256), 0, st, a);
    } else {
      if (Cout == 16) hipLaunchKernelGGL((conv3x3_c16_kernel<16, 1>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((conv3x3_c16_kernel<32, 1>), grid, dim3(256), 0, st, a);
    }
  } else {
    // (stride 2 stays on the exact kernel under either arithmetic: the split-bf16 form -- built, parity-green -- ran
    // level1 in 93 us against 70: its B fragments sit two cells apart, two-way bank conflicts on every read, and the
    // staging splits twice the cells per output; tools/probe_conv_direct_bf16.py history in DESIGN 4.10)
    const dim3 grid((a.Wo + 63) / 64, (a.Ho + 3) / 4, B);
    if (Cout == 16) hipLaunchKernelGGL((conv3x3_c16_kernel<16, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_c16_kernel<32, 2>), grid, dim3(256), 0, st, a);
  }
  return cp_launch_status();
}

// =====================================================================================================================
// Weight gradient of the same full-resolution, low-channel layers (what the reference gets from cuDNN's
// backward-filter for `base_layer` / `level0` / `level1`, pose_dla_dcn.py:236-246,266-276):
//   gw[co][ci][ky][kx] += sum_{b, y, x} go[b][co][y][x] * in[b][ci][S y + ky - pad][S x + kx - pad]
// The library ran these as NHWC implicit GEMMs behind two layout transposes of full-resolution maps (stem 1.8 ms,
// level1 0.9 ms per B = 4 step); the layers are a few GFLOP over 100-300 MB, i.e. bandwidth-sized.
// Contraction over PIXELS on the exact f32 MFMA (v_mfma_f32_16x16x4_f32): D[co][n] += A[co][px] * B[px][n] with
// n = (ci, ky, kx) flattened -- 147 / 144 columns = 10 / 9 fragments of 16.  One workgroup walks pixel tiles of
// 8 x 32 (stride 2: 4 x 32) outputs (grid-stride); per tile grad_out ([co][256 px], pitch 260) and the input's halo region
// ([ci][S*8 + K - S rows][S*32 + K - S cols]) are staged in LDS; wave w owns the k-steps w, w + 4, ... (4 pixels each)
// for ALL column fragments: per k-step one A read per 16 output channels and, per fragment, ONE B read at
// (the lane's column offset ci*plane + ky*pitch + kx) + S * (pixel offset) -- no column tile is ever built.  The
// partial gradient stays in accumulator registers over the whole run and is added to gw once per workgroup with
// coalesced float atomics (a row of D is contiguous in gw).
// =====================================================================================================================
namespace {

struct WgSmallArgs {
  const float* x;
  const float* go;
  float* gw;
  int B, H, W, Ho, Wo, pad, tiles_x, tiles_y, ntiles;
};

template <int COUT, int CIN, int K, int S>
__global__ __launch_bounds__(256, 2) void conv_wgrad_small_kernel(WgSmallArgs a) {
  constexpr int TH = S == 2 ? 4 : 8, TW = 32, NPX = TH * TW;   // (stride 2: 4 rows keep two workgroups per CU)
  constexpr int NN = CIN * K * K, NTL = (NN + 15) / 16, MTL = COUT / 16;
  constexpr int RH = S * TH + K - S, RWC = S * TW + K - S, RP = RWC + 1, PLANE = RH * RP;
  constexpr int GP = NPX + 4;                                  // grad_out pitch: 4 (mod 32) -> conflict-free A reads
  constexpr int XCELLS = CIN * RH * RWC, NXL = (XCELLS + 255) / 256, XB = 8;      // region loads per thread, batch
  constexpr int NG4 = COUT * NPX / 4 / 256;                                       // grad_out float4 per thread
  __shared__ float gs[COUT * GP];
  __shared__ float xs[CIN * PLANE + 1];                        // + one zero cell for the padding columns

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, n = lane & 15;
  const int HWo = a.Ho * a.Wo, HWi = a.H * a.W;
  if (tid == 0) xs[CIN * PLANE] = 0.f;

  // this lane's column of every fragment: LDS offset of (ci, ky, kx); padding columns read the zero cell
  int noff[NTL], nmul[NTL];
#pragma unroll
  for (int j = 0; j < NTL; ++j) {
    const int idx = 16 * j + n;
    const int ci = idx / (K * K), r = idx - ci * (K * K), ky = r / K, kx = r - ky * K;
    noff[j] = idx < NN ? ci * PLANE + ky * RP + kx : CIN * PLANE;
    nmul[j] = idx < NN ? S : 0;
  }
  f32x4 acc[NTL][MTL];
#pragma unroll
  for (int j = 0; j < NTL; ++j)
#pragma unroll
    for (int m = 0; m < MTL; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int per_img = a.tiles_x * a.tiles_y;
  for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
    const int b = t / per_img, tt = t - b * per_img;
    const int oy0 = (tt / a.tiles_x) * TH, ox0 = (tt % a.tiles_x) * TW;
    const int iy0 = S * oy0 - a.pad, ix0 = S * ox0 - a.pad;
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.go + (long long)b * COUT * HWo), 0, (int)((unsigned)COUT * (unsigned)HWo * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x + (long long)b * CIN * HWi), 0, (int)((unsigned)CIN * (unsigned)HWi * 4u), 0x00020000);
    __syncthreads();                                          // the previous tile's reads are done
    {                                                         // grad_out tile: all loads first, then the stores
      float gv[NG4][4];
      const bool vec = (a.Wo & 3) == 0;                       // rows of 32: float4 when the row pitch allows
#pragma unroll
      for (int i = 0; i < NG4; ++i) {
        const int u = tid + 256 * i, co = u / (NPX / 4), q = u - co * (NPX / 4), oy = oy0 + q / 8, ox = ox0 + 4 * (q % 8);
        const unsigned base = ((unsigned)co * (unsigned)HWo + (unsigned)(oy * a.Wo + ox)) * 4u;
        if (vec) {
          const bool ok = oy < a.Ho && ox < a.Wo;
          const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, ok ? base : 0xFFFFFFF0u, 0, 0));
          gv[i][0] = v.x; gv[i][1] = v.y; gv[i][2] = v.z; gv[i][3] = v.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = oy < a.Ho && ox + e < a.Wo;
            gv[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_g, ok ? base + 4u * e : 0xFFFFFFF0u, 0, 0));
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NG4; ++i) {
        const int u = tid + 256 * i, co = u / (NPX / 4), q = u - co * (NPX / 4);
        *reinterpret_cast<f32x4*>(&gs[co * GP + 4 * q]) = f32x4{gv[i][0], gv[i][1], gv[i][2], gv[i][3]};
      }
    }
#pragma unroll 1
    for (int i0 = 0; i0 < NXL; i0 += XB) {                    // input halo region, zero outside the image
      float xv[XB];
#pragma unroll
      for (int i = 0; i < XB; ++i) {
        const int e = tid + 256 * (i0 + i);
        const int ci = e / (RH * RWC), r = e - ci * (RH * RWC), ry = r / RWC, rx = r - ry * RWC;
        const int iy = iy0 + ry, ix = ix0 + rx;
        const bool ok = i0 + i < NXL && e < XCELLS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        const unsigned o = ok ? ((unsigned)ci * (unsigned)HWi + (unsigned)(iy * a.W + ix)) * 4u : 0xFFFFFFF0u;
        xv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, o, 0, 0));
      }
#pragma unroll
      for (int i = 0; i < XB; ++i) {
        const int e = tid + 256 * (i0 + i);
        if (i0 + i < NXL && e < XCELLS) {
          const int ci = e / (RH * RWC), r = e - ci * (RH * RWC), ry = r / RWC, rx = r - ry * RWC;
          xs[ci * PLANE + ry * RP + rx] = xv[i];
        }
      }
    }
    __syncthreads();

    // wave w contracts the k-steps w, w + 4, ... (4 pixels each) against ALL column fragments
#pragma unroll 2
    for (int it = 0; it < NPX / 16; ++it) {
      const int ks = wid + 4 * it;
      const int p = 4 * ks + g, py = p / TW, px = p % TW;
      const int poff = py * RP + px;
      float av[MTL], bv[NTL];
#pragma unroll
      for (int m = 0; m < MTL; ++m) av[m] = gs[(16 * m + n) * GP + p];
#pragma unroll
      for (int j = 0; j < NTL; ++j) bv[j] = xs[noff[j] + nmul[j] * poff];
#pragma unroll
      for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int m = 0; m < MTL; ++m) acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[j], acc[j][m], 0, 0, 0);
    }
  }

  // flush: D[co = 16 m + 4 g + r][column = 16 j + n]; gw[co][column] is contiguous in the column
#pragma unroll
  for (int j = 0; j < NTL; ++j) {
    const int idx = 16 * j + n;
    if (idx < NN) {
#pragma unroll
      for (int m = 0; m < MTL; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(&a.gw[(16 * m + 4 * g + r) * NN + idx], acc[j][m][r]);
    }
  }
}

}  // namespace

extern "C" int cp_conv_direct_wgrad_supported(int32_t Cin, int32_t Cout, int32_t k, int32_t stride, int32_t pad) {
  return (k == 7 && Cin == 3 && Cout == 16 && stride == 1 && pad == 3) ||
         (k == 3 && Cin == 16 && Cout == 32 && stride == 2 && pad == 1) ||
         (k == 3 && Cin == 16 && Cout == 16 && stride == 1 && pad == 1);
}

// gw [Cout][Cin][k][k] is ACCUMULATED into (float atomics; zero it first).
extern "C" int cp_conv_direct_wgrad(const float* x, const float* go, float* gw, int32_t B, int32_t Cin, int32_t H,
                                    int32_t W, int32_t Cout, int32_t k, int32_t stride, int32_t pad, void* stream) {
  CP_CHECK_ARG(x && go && gw && B > 0 && H > 0 && W > 0);
  if (!cp_conv_direct_wgrad_supported(Cin, Cout, k, stride, pad)) return CP_EUNSUPPORTED;
  WgSmallArgs a;
  a.x = x; a.go = go; a.gw = gw; a.B = B; a.H = H; a.W = W; a.pad = pad;
  a.Ho = (H + 2 * pad - k) / stride + 1;
  a.Wo = (W + 2 * pad - k) / stride + 1;
  if ((unsigned long long)Cin * H * W * 4ull >= 0x70000000ull || (unsigned long long)Cout * a.Ho * a.Wo * 4ull >= 0x70000000ull)
    return CP_EUNSUPPORTED;
  a.tiles_x = (a.Wo + 31) / 32;
  const int th = stride == 2 ? 4 : 8;
  a.tiles_y = (a.Ho + th - 1) / th;
  a.ntiles = a.tiles_x * a.tiles_y * B;
  const int grid = a.ntiles < 512 ? a.ntiles : 512;            // two workgroups per CU, each a grid-stride run
  hipStream_t st = (hipStream_t)stream;
  if (k == 7) hipLaunchKernelGGL((conv_wgrad_small_kernel<16, 3, 7, 1>), dim3(grid), dim3(256), 0, st, a);
  else if (stride == 2) hipLaunchKernelGGL((conv_wgrad_small_kernel<32, 16, 3, 2>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv_wgrad_small_kernel<16, 16, 3, 1>), dim3(grid), dim3(256), 0, st, a);
  return cp_launch_status();
}
