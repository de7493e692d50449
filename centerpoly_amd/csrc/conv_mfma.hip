// 3x3 / stride 1 / pad 1 convolution of NCHW float32 maps as an implicit GEMM on the bf16 matrix cores (gfx950),
// float32 in and out: every product runs as three v_mfma_f32_16x16x32_bf16 of split halves
// (v = hi + lo, hi = bf16(v), lo = bf16(v - hi); hi*hi + hi*lo + lo*hi, float32 accumulation), so a product carries
// ~2^-16 relative error -- three orders below the path's 1e-3 parity bar -- at a third of the 2.5 PFLOP/s bf16 peak,
// where the f32-input MFMA (157 TFLOP/s, and on the vector ALUs) and the library's Winograd stop near 100 TFLOP/s.
//
// It serves the dense convolutions either side of the DCN layers in the reference's network files
// (src/lib/models/networks/pose_dla_dcn.py: BasicBlock :38-66 conv1 / conv2, the heads' 3x3 :480-488 `fc`,
// DCN.conv_offset_mask via DCNv2/dcn_v2.py:137-145; large_hourglass.py: convolution :24-37 / residual :55-81)
// and, with the weights transposed and flipped by the prologue, their input gradients.
//
//   out[b][co][y][x] = sum_{ci, dy, dx} W[co][ci][dy][dx] * in[b][ci][y + dy - 1][x + dx - 1]   (+ bias, + residual, ReLU)
//
// GEMM view: D[co][px] += A[co][ci] * B[ci][px] per tap, k-step = 32 input channels.
//   * A (weights): split and laid out in fragment order ONCE by conv_mfma_wperm_kernel; the main kernel reads its
//     fragments straight from global memory (L1/L2 resident: <= 72 KB per channel chunk and output-channel tile),
//     one 16-byte load per lane, prefetched a tap ahead.  No LDS, no barrier on this operand.
//   * B (activations): a (rows + 2) x 34 pixel tile of 32 channels is staged per chunk -- coalesced dword loads along x,
//     split once, stored as [hi|lo][channel group of 8][row][col][8 x bf16] -- so a B fragment of any tap is ONE
//     conflict-free ds_read_b128 per lane (16 lanes x 16 B contiguous) at a tap-shifted pixel address.
//   * a wave owns MT x (2 RW) accumulator tiles (64 VGPRs); per tap and chunk it issues 2 MT global + 4 RW LDS
//     fragment reads for 6 MT RW MFMAs: the two operand streams load the two pipes (L1 and LDS) about equally.
//   * 256 threads; LDS = (4 RW + 2) * 34 * 128 B (43.5 KB for RW = 2: three workgroups per CU; while one stages, the
//     others keep the matrix cores busy).
#include "cp_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int KC = 32;              // input channels per k-step
constexpr int TW = 32;              // tile width (pixels)
constexpr int LW = TW + 2;          // staged width
constexpr unsigned OOB = 0xFFFFFFF0u;

// wp[((((cot * nchunk + chunk) * 9 + tap) * MT + mt) * 2 + hl) * 64 + lane][j] =
//   half(hl) of Wsrc[m = (cot * MT + mt) * 16 + (lane & 15)][k = chunk * 32 + 8 (lane >> 4) + j][tap]
// Wsrc = W ([M][K][9]) or, transposed (input gradient): Wsrc[m][k][tap] = W[k][m][8 - tap] with W = [K][M][9].
__global__ __launch_bounds__(256) void conv_mfma_wperm_kernel(const float* __restrict__ w, bf16x8* __restrict__ wp, int M,
                                                              int K, int MT, int nchunk, int transposed, int total) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int lane = e & 63;
  int r = e >> 6;
  const int hl = r & 1;
  r >>= 1;
  const int mt = r % MT;
  r /= MT;
  const int tap = r % 9;
  r /= 9;
  const int chunk = r % nchunk, cot = r / nchunk;
  const int m = (cot * MT + mt) * 16 + (lane & 15);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = chunk * KC + 8 * (lane >> 4) + j;
    float v = 0.f;
    if (m < M && k < K) v = transposed ? w[((long long)k * M + m) * 9 + (8 - tap)] : w[((long long)m * K + k) * 9 + tap];
    const __bf16 h = (__bf16)v;
    o[j] = hl ? (__bf16)(v - (float)h) : h;
  }
  wp[e] = o;
}

struct CvArgs {
  const float* x;
  const bf16x8* wp;
  const float* bias;      // [Cout] or null
  const float* res;       // same shape as out, or null
  float* out;
  int Cin, H, W, Cout, nchunk, ncot, tiles_x, relu;
};

template <int MT, int RW>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(CvArgs a) {
  constexpr int TH = 4 * RW, LH = TH + 2, NT = 2 * RW, PLANE = 4 * LH * LW;      // PLANE: fragments per half
  constexpr int UNITS = PLANE, ITERS = (UNITS + 255) / 256, SB = ITERS <= 6 ? ITERS : 5;
  __shared__ bf16x8 Xs[2 * PLANE];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, c = lane & 15;
  const int cot = blockIdx.x % a.ncot, tile = blockIdx.x / a.ncot;
  const int x0 = (tile % a.tiles_x) * TW, y0 = (tile / a.tiles_x) * TH, b = blockIdx.y;
  const int HW = a.H * a.W;

  const float* xb = a.x + (long long)b * a.Cin * HW;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xb), 0, (int)((unsigned)a.Cin * (unsigned)HW * 4u), 0x00020000);

  // staging units of this thread: (channel group, row, col) -> byte offset of channel 0 of the group, or OOB
  unsigned soff[ITERS];
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int u = tid + i * 256;
    const int col = u % LW, r = (u / LW) % LH, cg = u / (LW * LH);
    const int gy = y0 - 1 + r, gx = x0 - 1 + col;
    const bool ok = u < UNITS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    soff[i] = ok ? ((unsigned)(cg * 8) * (unsigned)HW + (unsigned)(gy * a.W + gx)) * 4u : OOB;
  }
  const unsigned cstep = (unsigned)HW * 4u;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bf16x8* wq = a.wp + (long long)cot * a.nchunk * 9 * MT * 2 * 64 + lane;
  bf16x8 af[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    af[m][0] = wq[(m * 2 + 0) * 64];
    af[m][1] = wq[(m * 2 + 1) * 64];
  }

  // B fragment base of this lane: channel group g, wave's first row, col c (tile origin is (-1, -1))
  const int bbase = (g * LH + wid * RW) * LW + c;

  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    __syncthreads();                                          // the previous chunk's fragments have been read
    {
      const unsigned cb = (unsigned)chunk * KC * cstep;
#pragma unroll
      for (int i0 = 0; i0 < ITERS; i0 += SB) {                // batches of SB units: 8 SB loads in flight per thread
        float v[SB][8];
#pragma unroll
        for (int i = 0; i < SB; ++i) {
          const unsigned o = (i0 + i >= ITERS || soff[(i0 + i) % ITERS] == OOB) ? OOB : soff[(i0 + i) % ITERS] + cb;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            v[i][j] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, o == OOB ? OOB : o + j * cstep, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < SB; ++i) {
          const int u = tid + (i0 + i) * 256;
          if (i0 + i < ITERS && u < UNITS) {
            bf16x8 h, l;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const __bf16 hh = (__bf16)v[i][j];
              h[j] = hh;
              l[j] = (__bf16)(v[i][j] - (float)hh);
            }
            Xs[u] = h;
            Xs[PLANE + u] = l;
          }
        }
      }
    }
    __syncthreads();

#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
      // next tap's (or next chunk's first) weight fragments; past the end: re-read the last (harmless, in bounds)
      bf16x8 an[MT][2];
      {
        const bool last = tap == 8 && chunk == a.nchunk - 1;
        const bf16x8* nq = wq + (long long)((chunk * 9 + tap + (last ? 0 : 1)) * MT * 2) * 64;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          an[m][0] = nq[(m * 2 + 0) * 64];
          an[m][1] = nq[(m * 2 + 1) * 64];
        }
      }
      bf16x8 bh[NT], bl[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int idx = bbase + ((n >> 1) + dy) * LW + (n & 1) * 16 + dx;
        bh[n] = Xs[idx];
        bl[n] = Xs[PLANE + idx];
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][0], bh[n], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][0], bl[n], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][1], bh[n], acc[m][n], 0, 0, 0);
        }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        af[m][0] = an[m][0];
        af[m][1] = an[m][1];
      }
      __builtin_amdgcn_sched_barrier(0);                      // keep the taps' fragment reads from piling up
    }
  }

  // epilogue: D[row = 4 g + r (co)][col = c (pixel)]
  float* ob = a.out + (long long)b * a.Cout * HW;
  const float* rb = a.res ? a.res + (long long)b * a.Cout * HW : nullptr;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = (cot * MT + m) * 16 + 4 * g + r;
      if (co >= a.Cout) continue;
      const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int y = y0 + wid * RW + (n >> 1), x = x0 + (n & 1) * 16 + c;
        if (y < a.H && x < a.W) {
          const long long o = (long long)co * HW + (long long)y * a.W + x;
          float v = acc[m][n][r] + bv;
          if (rb) v += rb[o];
          if (a.relu) v = fmaxf(v, 0.f);
          ob[o] = v;
        }
      }
    }
  }
}

int mt_for(int Cout) { return Cout <= 32 ? 2 : 4; }

}  // namespace

extern "C" {

int cp_conv3x3_mfma_supported(int32_t Cin, int32_t Cout, int32_t H, int32_t W) {
  // channels past Cin (the last k-step of a ragged Cin) lie beyond the buffer bound and read as zero
  return Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1 && (long long)Cin * H * W * 4 < 0x7FFFFFF0ll &&
         (long long)(Cin + KC) * H * W * 4 < 0xFFFFFFF0ll;
}

size_t cp_conv3x3_mfma_weight_bytes(int32_t Cin, int32_t Cout) {
  const int MT = mt_for(Cout), ncot = (Cout + 16 * MT - 1) / (16 * MT), nchunk = (Cin + KC - 1) / KC;
  return (size_t)ncot * nchunk * 9 * MT * 2 * 64 * 16;
}

// weight: [Cout][Cin][3][3] (transposed = 0), or -- for the input gradient of a convolution whose weight is
// [K][M][3][3] -- the same tensor read as Wsrc[m][k][tap] = W[k][m][8 - tap] (transposed = 1; Cin := K, Cout := M).
int cp_conv3x3_mfma_prepare(const float* weight, int32_t Cin, int32_t Cout, int32_t transposed, void* wperm,
                            void* stream) {
  hipStream_t st = (hipStream_t)stream;
  CP_CHECK_ARG(weight && wperm && Cin >= 1 && Cout >= 1);
  const int MT = mt_for(Cout), ncot = (Cout + 16 * MT - 1) / (16 * MT), nchunk = (Cin + KC - 1) / KC;
  const int total = ncot * nchunk * 9 * MT * 2 * 64;
  hipLaunchKernelGGL(conv_mfma_wperm_kernel, dim3((total + 255) / 256), dim3(256), 0, st, weight, (bf16x8*)wperm, Cout,
                     Cin, MT, nchunk, transposed, total);
  return cp_launch_status();
}

int cp_conv3x3_mfma_forward(const float* x, const void* wperm, const float* bias, const float* residual, float* out,
                            int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t relu, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  CP_CHECK_ARG(x && wperm && out && B >= 1);
  if (!cp_conv3x3_mfma_supported(Cin, Cout, H, W)) return CP_EUNSUPPORTED;
  const int MT = mt_for(Cout);
  CvArgs a;
  a.x = x;
  a.wp = (const bf16x8*)wperm;
  a.bias = bias;
  a.res = residual;
  a.out = out;
  a.Cin = Cin;
  a.H = H;
  a.W = W;
  a.Cout = Cout;
  a.nchunk = (Cin + KC - 1) / KC;
  a.ncot = (Cout + 16 * MT - 1) / (16 * MT);
  a.tiles_x = (W + TW - 1) / TW;
  a.relu = relu;
  if (MT == 4) {
    const int tiles = a.tiles_x * ((H + 7) / 8);
    hipLaunchKernelGGL((conv3x3_mfma_kernel<4, 2>), dim3(tiles * a.ncot, B), dim3(256), 0, st, a);
  } else {
    const int tiles = a.tiles_x * ((H + 15) / 16);
    hipLaunchKernelGGL((conv3x3_mfma_kernel<2, 4>), dim3(tiles * a.ncot, B), dim3(256), 0, st, a);
  }
  return cp_launch_status();
}

}  // extern "C"
