// The detection heads at inference as ONE kernel (the `fc` Sequentials of DLASeg, src/lib/models/networks/
// pose_dla_dcn.py:445-462: Conv2d 3x3 (64 -> head_conv) + bias -> ReLU -> Conv2d 1x1 (head_conv -> classes) + bias,
// four of them over the same input): the 3x3 convolution of csrc/conv_mfma.hip (split-bf16 x3 implicit GEMM on the
// bf16 matrix cores) whose epilogue, instead of storing the head_conv-channel map, feeds it straight into the 1x1
// convolution -- a second MFMA stage -- so that the largest tensor of the network (4 x 256 channels at the output
// resolution, 537 MB at 256 x 512) never goes to memory.  (Separate kernels: the 3x3 launch wrote it, 0.47 ms of which
// ~30 % were the stores, and cp_conv1x1_act_forward read it back, 0.16 ms.)
//
// One workgroup = (8 x 32 pixel tile, one head).  For each of the head's HC / 64 output-channel tiles in turn it runs
// the 3x3 contraction exactly as conv_mfma_kernel<4, 2, 9> (64 channels x 256 pixels in accumulator registers), then
//   * adds the bias, applies ReLU, splits to bf16 halves IN PLACE: the accumulator layout of
//     v_mfma_f32_16x16x32_bf16 (lane = pixel column, 4 registers = 4 consecutive channel rows) is already a B operand of
//     the next MFMA -- rows 4g..4g+3 of two 16-row tiles give the 8 k-values of lane group g, with the 1x1 weights
//     permuted to the same k order by the prologue (no lane movement, no LDS);
//   * accumulates out2[class][pixel] += W2[class][those 32 channels] * that fragment (2 x 4 x 2 x 3 MFMAs per tile,
//     +11 % over the 3x3's 432).
// After the last tile: + bias2, 64-byte row-segment stores of the head's <= 32 (or <= 64) output channels.
#include "cp_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int KC = 32, TW = 32, TH = 8, LW = TW + 2, LH = TH + 2, MT = 4, NT = 4, RW = 2;
constexpr int PLANE = 4 * LH * LW, UNITS = PLANE, ITERS = (UNITS + 255) / 256;
constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int MAXH = 4;

// wp2[((mt2 * KS2 + ks) * 2 + hl) * 64 + lane][j] = half(hl) of W2[c = 16 mt2 + (lane & 15)][co(ks, lane >> 4, j)],
// ks = 2 cl + p (cl: 64-channel tile of the head, p: pair of 16-row fragments), g = lane >> 4,
// co = 64 cl + 16 (2 p + (j >> 2)) + 4 g + (j & 3): the k order in which the 3x3's accumulators present their rows
__global__ __launch_bounds__(256) void heads_w2perm_kernel(const float* __restrict__ w2, bf16x8* __restrict__ wp2, int cout,
                                                           int HC, int total) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int KS2 = HC / 32;
  const int lane = e & 63;
  int r = e >> 6;
  const int hl = r & 1;
  r >>= 1;
  const int ks = r % KS2, mt2 = r / KS2;
  const int c = 16 * mt2 + (lane & 15), g = lane >> 4, cl = ks >> 1, p = ks & 1;
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int co = 64 * cl + 16 * (2 * p + (j >> 2)) + 4 * g + (j & 3);
    const float v = c < cout ? w2[(long long)c * HC + co] : 0.f;
    const __bf16 h = (__bf16)v;
    o[j] = hl ? (__bf16)(v - (float)h) : h;
  }
  wp2[e] = o;
}

struct HeadsArgs {
  const float* x;
  const bf16x8* wp1;          // cp_conv_mfma_prepare layout of the concatenated 3x3 weights [nheads * HC][Cin][3][3]
  const float* b1;            // [nheads * HC]
  const bf16x8* wp2[MAXH];
  const float* b2[MAXH];      // [cout] or null
  float* out[MAXH];           // [B][cout][H][W]
  int cout[MAXH];
  int Cin, H, W, nchunk, nheads, HC, tiles_x;
};

// MT2: 16-class fragments per head (2: up to 32 output channels, 4: up to 64)
template <int MT2>
__global__ __launch_bounds__(256, 2) void conv_heads_fused_kernel(HeadsArgs a) {
  __shared__ bf16x8 Xs[2 * PLANE];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, c = lane & 15;
  // (XCD-aware order, as in conv_mfma.hip: logical neighbours -- the heads of one tile, then the next tile -- share an L2)
  int lw = blockIdx.x;
  if ((gridDim.x & 7) == 0) lw = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int head = lw % a.nheads, tile = lw / a.nheads;
  const int x0 = (tile % a.tiles_x) * TW, y0 = (tile / a.tiles_x) * TH, b = blockIdx.y;
  const int HW = a.H * a.W;
  const int ncl = a.HC / 64, KS2 = a.HC / 32;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x + (long long)b * a.Cin * HW), 0, (int)((unsigned)a.Cin * (unsigned)HW * 4u), 0x00020000);
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
  const long long tstride = (long long)a.nchunk * 9 * 2 * 64;          // fragments per 16-row tile of the 3x3 weights
  const int bbase = (g * LH + wid * RW) * LW + c;
  const bf16x8* wq2 = a.wp2[head] + lane;

  f32x4 acc2[MT2][NT];                                                 // the head's outputs: [16-class fragment][n]
#pragma unroll
  for (int m2 = 0; m2 < MT2; ++m2)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc2[m2][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int cl = 0; cl < ncl; ++cl) {
    const int cot = head * ncl + cl;                                   // 64-channel tile of the concatenated 3x3 output
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8* wq = a.wp1 + (long long)cot * MT * tstride + lane;
    bf16x8 af[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      af[m][0] = wq[m * tstride];
      af[m][1] = wq[m * tstride + 64];
    }

    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
      __syncthreads();                                                 // the previous fragments have been read
      {
        const unsigned cb = (unsigned)chunk * KC * cstep;
        float v[ITERS][8];
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
          const unsigned o = soff[i] == OOB ? OOB : soff[i] + cb;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            v[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, o == OOB ? OOB : o + j * cstep, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
          const int u = tid + i * 256;
          if (u < UNITS) {
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
      __syncthreads();
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap % 3;
        bf16x8 an[MT][2];
        {
          const bool last = tap == 8 && chunk == a.nchunk - 1;
          const bf16x8* nq = wq + (long long)((chunk * 9 + tap + (last ? 0 : 1)) * 2) * 64;
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            an[m][0] = nq[m * tstride];
            an[m][1] = nq[m * tstride + 64];
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
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- second stage: out2 += W2[:, these 64 channels] * relu(acc + b1) ----
    const float* b1p = a.b1 + cot * 64 + 4 * g;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      float bia[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) bia[j] = b1p[16 * (2 * p + (j >> 2)) + (j & 3)];
      bf16x8 a2h[MT2], a2l[MT2];
#pragma unroll
      for (int m2 = 0; m2 < MT2; ++m2) {
        const bf16x8* q = wq2 + (long long)(((m2 * KS2 + 2 * cl + p) * 2) * 64);
        a2h[m2] = q[0];
        a2l[m2] = q[64];
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        bf16x8 fh, fl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = fmaxf(acc[2 * p + (j >> 2)][n][j & 3] + bia[j], 0.f);
          const __bf16 h = (__bf16)v;
          fh[j] = h;
          fl[j] = (__bf16)(v - (float)h);
        }
#pragma unroll
        for (int m2 = 0; m2 < MT2; ++m2) {
          acc2[m2][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2h[m2], fh, acc2[m2][n], 0, 0, 0);
          acc2[m2][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2h[m2], fl, acc2[m2][n], 0, 0, 0);
          acc2[m2][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2l[m2], fh, acc2[m2][n], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: D[row = 4 g + r (class)][col = c (pixel)] ----
  const int co_n = a.cout[head];
  float* ob = a.out[head] + (long long)b * co_n * HW;
  const float* b2 = a.b2[head];
#pragma unroll
  for (int m2 = 0; m2 < MT2; ++m2) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cls = 16 * m2 + 4 * g + r;
      if (cls >= co_n) continue;
      const float bv = b2 ? b2[cls] : 0.f;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int y = y0 + wid * RW + (n >> 1), x = x0 + (n & 1) * 16 + c;
        if (y < a.H && x < a.W) ob[(long long)cls * HW + (long long)y * a.W + x] = acc2[m2][n][r] + bv;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Input-resident form (Cin <= 96: the whole (8 + 2) x (32 + 2) halo tile of every input channel fits in LDS as bf16
// hi | lo, 43.5 KB per 32 channels).  The form above re-stages -- loads, splits, stores -- the same input tile for each
// of a head's HC / 64 channel tiles, and the heads of a pixel tile sit in different workgroups on different XCDs: the
// PMC passes showed ~1 GB fetched for a 34 MB input.  Here one workgroup = 8 waves = (pixel tile, TWO heads): the tile
// is staged ONCE by all 512 threads, then waves 0-3 run head 2 i and waves 4-7 head 2 i + 1 over it, each through its
// channel tiles, with no further barrier (8 waves per CU as before, one staging instead of eight).
template <int MT2>
__global__ __launch_bounds__(512, 2) void conv_heads_fused_res_kernel(HeadsArgs a) {
  extern __shared__ __attribute__((aligned(16))) bf16x8 Xd[];          // [chunk][hi | lo][PLANE]

  const int tid = threadIdx.x, lane = tid & 63, wid = (tid >> 6) & 3, g = lane >> 4, c = lane & 15;
  const int hsel = __builtin_amdgcn_readfirstlane(tid >> 8);
  const int npair = (a.nheads + 1) / 2;
  // XCD-aware order: consecutive workgroups go round-robin over the 8 XCDs, so the head pairs of one tile (and the
  // tiles sharing its halo rows) fetched the input through different L2s -- 114 MB on the fabric side for a 34 MB input
  // (PMC FETCH_SIZE).  Workgroup w takes the logical index (w % 8) * (n / 8) + w / 8: logical neighbours share an XCD.
  int lw = blockIdx.x;
  if ((gridDim.x & 7) == 0) lw = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int head = 2 * (lw % npair) + hsel, tile = lw / npair;
  const int x0 = (tile % a.tiles_x) * TW, y0 = (tile / a.tiles_x) * TH, b = blockIdx.y;
  const int HW = a.H * a.W;
  const int ncl = a.HC / 64, KS2 = a.HC / 32;

  {
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x + (long long)b * a.Cin * HW), 0, (int)((unsigned)a.Cin * (unsigned)HW * 4u), 0x00020000);
    const unsigned cstep = (unsigned)HW * 4u;
    constexpr int IT = (UNITS + 511) / 512;
    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
      float v[IT][8];
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int u = tid + i * 512;
        const int col = u % LW, r = (u / LW) % LH, cg = u / (LW * LH);
        const int gy = y0 - 1 + r, gx = x0 - 1 + col;
        const bool ok = u < UNITS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const unsigned o = ok ? ((unsigned)(chunk * KC + cg * 8) * (unsigned)HW + (unsigned)(gy * a.W + gx)) * 4u : OOB;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          v[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, o == OOB ? OOB : o + j * cstep, 0, 0));
      }
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int u = tid + i * 512;
        if (u < UNITS) {
          bf16x8 h, l;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const __bf16 hh = (__bf16)v[i][j];
            h[j] = hh;
            l[j] = (__bf16)(v[i][j] - (float)hh);
          }
          Xd[chunk * 2 * PLANE + u] = h;
          Xd[chunk * 2 * PLANE + PLANE + u] = l;
        }
      }
    }
  }
  __syncthreads();
  if (head >= a.nheads) return;                                        // odd number of heads: the last pair's second half

  const long long tstride = (long long)a.nchunk * 9 * 2 * 64;
  const int bbase = (g * LH + wid * RW) * LW + c;
  const bf16x8* wq2 = a.wp2[head] + lane;

  f32x4 acc2[MT2][NT];
#pragma unroll
  for (int m2 = 0; m2 < MT2; ++m2)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc2[m2][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int cl = 0; cl < ncl; ++cl) {
    const int cot = head * ncl + cl;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8* wq = a.wp1 + (long long)cot * MT * tstride + lane;
    bf16x8 af[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      af[m][0] = wq[m * tstride];
      af[m][1] = wq[m * tstride + 64];
    }
    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
      const bf16x8* Xs = Xd + chunk * 2 * PLANE;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap % 3;
        bf16x8 an[MT][2];
        {
          const bool last = tap == 8 && chunk == a.nchunk - 1;
          const bf16x8* nq = wq + (long long)((chunk * 9 + tap + (last ? 0 : 1)) * 2) * 64;
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            an[m][0] = nq[m * tstride];
            an[m][1] = nq[m * tstride + 64];
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
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const float* b1p = a.b1 + cot * 64 + 4 * g;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      float bia[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) bia[j] = b1p[16 * (2 * p + (j >> 2)) + (j & 3)];
      bf16x8 a2h[MT2], a2l[MT2];
#pragma unroll
      for (int m2 = 0; m2 < MT2; ++m2) {
        const bf16x8* q = wq2 + (long long)(((m2 * KS2 + 2 * cl + p) * 2) * 64);
        a2h[m2] = q[0];
        a2l[m2] = q[64];
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        bf16x8 fh, fl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = fmaxf(acc[2 * p + (j >> 2)][n][j & 3] + bia[j], 0.f);
          const __bf16 h = (__bf16)v;
          fh[j] = h;
          fl[j] = (__bf16)(v - (float)h);
        }
#pragma unroll
        for (int m2 = 0; m2 < MT2; ++m2) {
          acc2[m2][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2h[m2], fh, acc2[m2][n], 0, 0, 0);
          acc2[m2][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2h[m2], fl, acc2[m2][n], 0, 0, 0);
          acc2[m2][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2l[m2], fh, acc2[m2][n], 0, 0, 0);
        }
      }
    }
  }

  const int co_n = a.cout[head];
  float* ob = a.out[head] + (long long)b * co_n * HW;
  const float* b2 = a.b2[head];
#pragma unroll
  for (int m2 = 0; m2 < MT2; ++m2) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cls = 16 * m2 + 4 * g + r;
      if (cls >= co_n) continue;
      const float bv = b2 ? b2[cls] : 0.f;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int y = y0 + wid * RW + (n >> 1), x = x0 + (n & 1) * 16 + c;
        if (y < a.H && x < a.W) ob[(long long)cls * HW + (long long)y * a.W + x] = acc2[m2][n][r] + bv;
      }
    }
  }
}

}  // namespace

extern "C" {

// (always four 16-class fragments: zero rows past cout, so that one layout serves both kernel forms)
size_t cp_heads_fused_w2_bytes(int32_t head_conv) { return (size_t)4 * (head_conv / 32) * 2 * 64 * 16; }

// w2: the head's 1x1 weight [cout][head_conv] (cout <= 64, head_conv a multiple of 64)
int cp_heads_fused_prepare_w2(const float* w2, int32_t cout, int32_t head_conv, void* w2perm, void* stream) {
  CP_CHECK_ARG(w2 && w2perm && cout >= 1 && cout <= 64 && head_conv >= 64 && head_conv % 64 == 0);
  const int total = 4 * (head_conv / 32) * 2 * 64;
  hipLaunchKernelGGL(heads_w2perm_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w2,
                     (bf16x8*)w2perm, cout, head_conv, total);
  return cp_launch_status();
}

// out[h][b][o][p] = b2[h][o] + sum_c w2[h][o][c] * relu(b1[h*HC + c] + conv3x3(x, w1)[h*HC + c][p])   for h < nheads
// wperm1: cp_conv_mfma_prepare(taps = 9) of the heads' 3x3 weights concatenated along the output channels.
int cp_heads_fused_forward(const float* x, const void* wperm1, const float* b1, const void* const* w2perm,
                           const float* const* b2, float* const* out, const int32_t* cout, int32_t nheads, int32_t B,
                           int32_t Cin, int32_t H, int32_t W, int32_t head_conv, void* stream) {
  CP_CHECK_ARG(x && wperm1 && b1 && w2perm && b2 && out && cout && B >= 1 && nheads >= 1 && nheads <= MAXH);
  if (Cin % KC != 0 || head_conv % 64 != 0 || head_conv < 64) return CP_EUNSUPPORTED;
  if ((long long)Cin * H * W * 4 >= 0x7FFFFFF0ll || B > 65535) return CP_EUNSUPPORTED;
  HeadsArgs a;
  a.x = x;
  a.wp1 = (const bf16x8*)wperm1;
  a.b1 = b1;
  for (int h = 0; h < MAXH; ++h) {
    const bool on = h < nheads;
    if (on) {
      CP_CHECK_ARG(w2perm[h] && out[h] && cout[h] >= 1);
      if (cout[h] > 64) return CP_EUNSUPPORTED;
    }
    a.wp2[h] = on ? (const bf16x8*)w2perm[h] : nullptr;
    a.b2[h] = on ? b2[h] : nullptr;
    a.out[h] = on ? out[h] : nullptr;
    a.cout[h] = on ? cout[h] : 0;
  }
  a.Cin = Cin;
  a.H = H;
  a.W = W;
  a.nchunk = Cin / KC;
  a.nheads = nheads;
  a.HC = head_conv;
  a.tiles_x = (W + TW - 1) / TW;
  const int tiles = a.tiles_x * ((H + TH - 1) / TH);
  int widest = 0;
  for (int h = 0; h < nheads; ++h) widest = cout[h] > widest ? cout[h] : widest;
  const size_t res_lds = (size_t)a.nchunk * 2 * PLANE * sizeof(bf16x8);          // input-resident form: Cin <= 96
  if (res_lds <= 160 * 1024 - 1024 && nheads >= 2) {
    const int npair = (nheads + 1) / 2;
    if (widest <= 32) {
      (void)hipFuncSetAttribute((const void*)conv_heads_fused_res_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)res_lds);
      hipLaunchKernelGGL(conv_heads_fused_res_kernel<2>, dim3(tiles * npair, B), dim3(512), res_lds, (hipStream_t)stream, a);
    } else {
      (void)hipFuncSetAttribute((const void*)conv_heads_fused_res_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)res_lds);
      hipLaunchKernelGGL(conv_heads_fused_res_kernel<4>, dim3(tiles * npair, B), dim3(512), res_lds, (hipStream_t)stream, a);
    }
  } else if (widest <= 32)
    hipLaunchKernelGGL(conv_heads_fused_kernel<2>, dim3(tiles * nheads, B), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(conv_heads_fused_kernel<4>, dim3(tiles * nheads, B), dim3(256), 0, (hipStream_t)stream, a);
  return cp_launch_status();
}

}  // extern "C"
