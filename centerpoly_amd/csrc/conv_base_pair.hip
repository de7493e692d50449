// level0 + level1 of the DLA base as ONE kernel at inference (gfx950, split-bf16 x3 on the bf16 matrix cores):
//
//   y0 = relu(conv3x3(x, w0) + b0)            16 -> 16, stride 1, pad 1          (pose_dla_dcn.py:236-246,266-276:
//   y1 = relu(conv3x3_s2(y0, w1) + b1)        16 -> 32, stride 2, pad 1           `base.level0`, `base.level1`, BN folded)
//
// Why: y0 is the largest activation of the network (16 x H x W: 134 MB at 1024 x 2048) and has ONE consumer; as two
// kernels it is written and read back (268 MB of the 470 MB the two layers move) between two launches that each sit at
// twice their memory time.  Here the level0 tile a workgroup needs stays in LDS.
//
// One workgroup = 4 waves = a tile of 2 x 32 outputs of y1 = 5 x 65 positions of y0 = 7 x 67 positions of x.
//   * x region: 7 rows x 72 columns (from the 16-byte aligned column 2 x0 - 4), staged once with 16-byte row loads and
//     split to bf16 halves as [hi | lo][channel half][cell][8 x bf16] (the layout of conv3x3_c16_bf16_kernel: a B
//     fragment of any tap is one ds_read_b128 per half); cells outside the image hold zeros (level0's padding).
//   * stage B (level0): 5 rows x 4 tiles of 16 columns + one tile for column 64 of the five rows, 5 k-steps of two taps x 16
//     channels, three v_mfma_f32_16x16x32_bf16 per step (hi*hi + hi*lo + lo*hi); + b0, ReLU, ZERO outside the image
//     (level1's padding), split again and written to the y0 region [hi | lo][half][5 x 65 cells][8 x bf16]: the
//     accumulator tile (lane = pixel, 4 consecutive channels) is half a cell, one ds_write_b64 per plane.
//   * stage C (level1): one (row, 16-column tile) per wave, two 16-channel fragments, B fragments at twice the cell
//     stride; + b1, ReLU, 64-byte row-segment stores of the 32 channels.
//   * persistent: 512 workgroups (two per CU), each a run of tiles t, t + 512, ...; both weight tensors are read once
//     per workgroup (coalesced, through LDS) and stay in registers as A fragments (120 VGPRs) for the whole run; the
//     next tile's x loads are in flight over the current tile's two stages; two barriers per tile.
// LDS 32.3 + 20.8 KB.  y0 is bit-identical to conv3x3_c16_bf16_kernel's; y1 differs from the
// separate route only in its arithmetic (split-bf16 x3 instead of the exact f32 MFMA: ~2^-16 per product).
#include "cp_common.h"

namespace {

typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pbf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 pbf16x2 __attribute__((ext_vector_type(2)));
typedef float pf32x2 __attribute__((ext_vector_type(2)));
typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned pu32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned POOB = 0x80000000u;
constexpr int CIN = 16, C1 = 32, KS = 5;
constexpr int TH2 = 2, TW2 = 32;                       // y1 tile
constexpr int L0H = 2 * TH2 + 1, L0W = 2 * TW2 + 1;    // y0 region: 5 x 65
constexpr int XH = L0H + 2, NC4 = 18, XW = 4 * NC4;    // x region: 7 x 72 cells
constexpr int XCELLS = XH * XW, XHALF = XCELLS * 16;   // bytes of one (hi | lo, channel half) plane of x
constexpr int L0CELLS = L0H * L0W, L0HALF = L0CELLS * 16;
static_assert(2 * XH * NC4 <= 256, "one staging item per thread");
static_assert(C1 * CIN * 9 * 4 <= 4 * XHALF, "the raw weights fit the region they pass through");

struct PairArgs {
  const float* x;       // [B][16][H][W]
  const float* w0;      // [16][16][3][3], BN scale folded in
  const float* b0;      // [16] or null
  const float* w1;      // [32][16][3][3]
  const float* b1;      // [32] or null
  float* out;           // [B][32][Ho][Wo]
  int H, W, Ho, Wo;
  int tiles_x, tiles_per_image, ntiles;
};

__device__ __forceinline__ void psplit2(float v0, float v1, unsigned& hi, unsigned& lo) {
  const pbf16x2 h = __builtin_convertvector(pf32x2{v0, v1}, pbf16x2);
  const unsigned hb = __builtin_bit_cast(unsigned, h);
  const float h0 = __builtin_bit_cast(float, hb << 16), h1 = __builtin_bit_cast(float, hb & 0xffff0000u);
  const pbf16x2 l = __builtin_convertvector(pf32x2{v0 - h0, v1 - h1}, pbf16x2);
  hi = hb;
  lo = __builtin_bit_cast(unsigned, l);
}

__global__ __launch_bounds__(256, 2) void conv_base_pair_kernel(PairArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char xs[4 * XHALF];      // x region
  __shared__ __attribute__((aligned(16))) unsigned char ls[4 * L0HALF];     // y0 region

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lpx = lane & 15, g = lane >> 4;
  const int HW = a.H * a.W;

  // ---- once per workgroup: both weight tensors, coalesced, through LDS, into A fragments that stay in registers for
  // every tile of the workgroup's run (the first form of this kernel rebuilt them per tile: 120 conflicted ds_reads and
  // 240 conversions per lane and tile -- 183 us against 169 for the two separate launches)
  pbf16x8 wh0[KS], wl0[KS], wh1[2][KS], wl1[2][KS];
  {
    float* wsh = reinterpret_cast<float*>(xs);
#pragma unroll
    for (int i = 0; i < 9; ++i) wsh[tid + 256 * i] = a.w0[tid + 256 * i];
    __syncthreads();
    // A[co = lpx][k = 32 s + 8 g + j] = W[co][ci = 8 (g & 1) + j][tap = 2 s + (g >> 1)] (tap 9: zero)
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) {
      const int tap = 2 * s_ + (g >> 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float w = tap < 9 ? wsh[(lpx * CIN + 8 * (g & 1) + j) * 9 + tap] : 0.f;
        const __bf16 h = (__bf16)w;
        wh0[s_][j] = h;
        wl0[s_][j] = (__bf16)(w - (float)h);
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 18; ++i) wsh[tid + 256 * i] = a.w1[tid + 256 * i];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int s_ = 0; s_ < KS; ++s_) {
        const int tap = 2 * s_ + (g >> 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float w = tap < 9 ? wsh[((16 * m + lpx) * CIN + 8 * (g & 1) + j) * 9 + tap] : 0.f;
          const __bf16 h = (__bf16)w;
          wh1[m][s_][j] = h;
          wl1[m][s_][j] = (__bf16)(w - (float)h);
        }
      }
    __syncthreads();                                                 // (the first tile's cells overwrite the buffer)
  }
  // byte offset of the lane's tap in k-step s relative to the output position's cell, in the x region and in the y0 region
  int toffx[KS], toffl[KS];
#pragma unroll
  for (int s_ = 0; s_ < KS; ++s_) {
    const int tap = min(2 * s_ + (g >> 1), 8);                       // (the padding tap multiplies zero weights)
    toffx[s_] = ((tap / 3) * XW + (tap % 3)) * 16 + (g & 1) * XHALF;
    toffl[s_] = ((tap / 3) * L0W + (tap % 3)) * 16 + (g & 1) * L0HALF;
  }
  float b0r[4], b1r[2][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    b0r[q] = a.b0 ? a.b0[4 * g + q] : 0.f;
    b1r[0][q] = a.b1 ? a.b1[4 * g + q] : 0.f;
    b1r[1][q] = a.b1 ? a.b1[16 + 4 * g + q] : 0.f;
  }

  // the thread's staging item: (channel half, row, float4 chunk) of the x region
  const int ihalf = tid / (XH * NC4), ir = tid - ihalf * (XH * NC4);
  const int iry = ir / NC4, ic4 = ir - iry * NC4;
  const bool item = tid < 2 * XH * NC4;
  f32x4 v[8];
  auto load_item = [&](int t) {
    const int b = t / a.tiles_per_image, tt = t - b * a.tiles_per_image;
    const int x0 = (tt % a.tiles_x) * TW2, y0 = (tt / a.tiles_x) * TH2;
    const int gy = 2 * y0 - 2 + iry, gx = 2 * x0 - 4 + 4 * ic4;
    const bool ok = item && t < a.ntiles && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;   // (W % 4 == 0: whole chunks in or out)
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x + (long long)b * CIN * HW), 0, (int)((unsigned)CIN * (unsigned)HW * 4u), 0x00020000);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned off = ok ? ((unsigned)(8 * ihalf + j) * (unsigned)HW + (unsigned)(gy * a.W + gx)) * 4u : POOB;
      v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
    }
  };

  // ---- the workgroup's run of tiles: t, t + gridDim.x, ... (the workgroups in flight cover consecutive tiles of a band of
  // rows: neighbours share their halo columns in L2); the next tile's loads are in flight over this tile's two stages
  load_item(blockIdx.x);
#pragma unroll 1
  for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
    const int b = t / a.tiles_per_image, tt = t - b * a.tiles_per_image;
    const int x0 = (tt % a.tiles_x) * TW2, y0 = (tt / a.tiles_x) * TH2;      // y1 tile origin
    const int l0x0 = 2 * x0 - 1, l0y0 = 2 * y0 - 1;                          // y0 region origin (image coordinates)
    // x region origin (2 x0 - 4, l0y0 - 1): y0 column c, tap kx <-> x cell c + kx + 2
    if (item) {
      unsigned char* dst = xs + ihalf * XHALF + (iry * XW + 4 * ic4) * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned hi[4], lo[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) psplit2(v[2 * jj][q], v[2 * jj + 1][q], hi[jj], lo[jj]);
        *reinterpret_cast<pu32x4*>(dst + q * 16) = pu32x4{hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<pu32x4*>(dst + 2 * XHALF + q * 16) = pu32x4{lo[0], lo[1], lo[2], lo[3]};
      }
    }
    __syncthreads();                                                 // x region staged; the previous tile's stage C is done
    load_item(t + gridDim.x);

    // ---- stage B: y0 = relu(conv(x) + b0) on the 5 x 65 region, into LDS as split cells.  21 accumulator tiles: 5 rows x
    // 4 tiles of 16 columns, and ONE tile for column 64 of all five rows (lane = row); a wave takes its tiles two at a
    // time -- two independent MFMA chains (with one wave per SIMD and workgroup a single chain leaves the pipe idle)
    constexpr int NTILE = L0H * 4 + 1;
    auto geom = [&](int u, int& r, int& c, bool& valid) {
      if (u < L0H * 4) {
        r = u >> 2;
        c = (u & 3) * 16 + lpx;
        valid = true;
      } else {
        r = min(lpx, L0H - 1);
        c = L0W - 1;
        valid = lpx < L0H;
      }
    };
    auto emit = [&](const f32x4& acc, int r, int c) {
      const int iy = l0y0 + r, ix = l0x0 + c;
      const bool inside = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      float o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = inside ? fmaxf(acc[q] + b0r[q], 0.f) : 0.f;
      unsigned hi[2], lo[2];
      psplit2(o[0], o[1], hi[0], lo[0]);
      psplit2(o[2], o[3], hi[1], lo[1]);
      // channels 4 g .. 4 g + 3: half g >> 1, elements 4 (g & 1) .. of the cell
      unsigned char* dst = ls + (g >> 1) * L0HALF + (r * L0W + c) * 16 + (g & 1) * 8;
      *reinterpret_cast<pu32x2*>(dst) = pu32x2{hi[0], hi[1]};
      *reinterpret_cast<pu32x2*>(dst + 2 * L0HALF) = pu32x2{lo[0], lo[1]};
    };
#pragma unroll 1
    for (int u = wid; u < NTILE; u += 8) {
      int r0, c0, r1, c1;
      bool v0, v1;
      geom(u, r0, c0, v0);
      const bool two = u + 4 < NTILE;                                // (wave-uniform)
      geom(two ? u + 4 : u, r1, c1, v1);
      const unsigned char* base0 = xs + (r0 * XW + c0 + 2) * 16;
      const unsigned char* base1 = xs + (r1 * XW + c1 + 2) * 16;
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
      for (int s_ = 0; s_ < KS; ++s_) {
        const pbf16x8 bh0 = *reinterpret_cast<const pbf16x8*>(base0 + toffx[s_]);
        const pbf16x8 bl0 = *reinterpret_cast<const pbf16x8*>(base0 + toffx[s_] + 2 * XHALF);
        const pbf16x8 bh1 = *reinterpret_cast<const pbf16x8*>(base1 + toffx[s_]);
        const pbf16x8 bl1 = *reinterpret_cast<const pbf16x8*>(base1 + toffx[s_] + 2 * XHALF);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh0[s_], bh0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh0[s_], bh1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh0[s_], bl0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh0[s_], bl1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl0[s_], bh0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl0[s_], bh1, acc1, 0, 0, 0);
      }
      if (v0) emit(acc0, r0, c0);
      if (two && v1) emit(acc1, r1, c1);
    }
    __syncthreads();                                                 // y0 region complete, x region free for the next tile

    // ---- stage C: y1 = relu(conv_s2(y0) + b1): wave = (row wid >> 1, 16-column tile wid & 1)
    {
      const int row = wid >> 1, x2 = (wid & 1) * 16 + lpx;
      const unsigned char* base = ls + ((2 * row) * L0W + 2 * x2) * 16;
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int s_ = 0; s_ < KS; ++s_) {
        const pbf16x8 bh = *reinterpret_cast<const pbf16x8*>(base + toffl[s_]);
        const pbf16x8 bl = *reinterpret_cast<const pbf16x8*>(base + toffl[s_] + 2 * L0HALF);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh1[m][s_], bh, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh1[m][s_], bl, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl1[m][s_], bh, acc[m], 0, 0, 0);
        }
      }
      const int oy = y0 + row, ox = x0 + x2;
      if (oy < a.Ho && ox < a.Wo) {
        float* ob = a.out + (long long)b * C1 * a.Ho * a.Wo;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            ob[((long long)(16 * m + 4 * g + q) * a.Ho + oy) * a.Wo + ox] = fmaxf(acc[m][q] + b1r[m][q], 0.f);
      }
    }
  }
}

}  // namespace

extern "C" int cp_dla_base_pair_supported(int32_t H, int32_t W) {
  return H >= 1 && W >= 4 && (W & 3) == 0 && (unsigned long long)CIN * H * W * 4ull < 0x70000000ull;
}

// out = relu(conv3x3 stride 2 (relu(conv3x3(x, w0) + b0), w1) + b1): x [B][16][H][W] -> out [B][32][(H-1)/2+1][(W-1)/2+1]
extern "C" int cp_dla_base_pair_forward(const float* x, const float* w0, const float* b0, const float* w1, const float* b1,
                                        float* out, int32_t B, int32_t H, int32_t W, void* stream) {
  CP_CHECK_ARG(x && w0 && w1 && out && B >= 1 && B <= 65535);
  if (!cp_dla_base_pair_supported(H, W) || (reinterpret_cast<unsigned long long>(x) & 15)) return CP_EUNSUPPORTED;
  PairArgs a;
  a.x = x; a.w0 = w0; a.b0 = b0; a.w1 = w1; a.b1 = b1; a.out = out;
  a.H = H; a.W = W; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
  a.tiles_x = (a.Wo + TW2 - 1) / TW2;
  a.tiles_per_image = a.tiles_x * ((a.Ho + TH2 - 1) / TH2);
  const long long ntiles = (long long)a.tiles_per_image * B;
  if (ntiles > 0x3FFFFFFFll) return CP_EUNSUPPORTED;
  a.ntiles = (int)ntiles;
  const int wgs = (int)(ntiles < 512 ? ntiles : 512);              // two workgroups per CU, each a run of tiles
  hipLaunchKernelGGL(conv_base_pair_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, a);
  return cp_launch_status();
}
