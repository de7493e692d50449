// Weight gradient of the 3x3 / stride 2 / pad 1 convolutions (the first convolution of DLA levels 2-5,
// src/lib/models/networks/pose_dla_dcn.py:32-40; cuDNN's backward-filter in the reference, MIOpen's NHWC implicit GEMM
// behind two layout transposes before this kernel), in the arithmetic of conv_mfma.hip (split-bf16 x3 on the bf16 matrix
// cores, fp32 accumulation):
//   gw[co][ci][ky][kx] += sum_{b, i, j} go[b][co][i][j] * x[b][ci][2 i + ky - 1][2 j + kx - 1]
// The structure is conv3x3_wgrad_kernel's (conv_mfma.hip): contraction over PIXELS, k-step = 32 pixels of one grad_out
// row, a workgroup = (a run of pixel tiles, 64 input channels, 64 output channels), 8 waves, wave w owns the input-channel
// fragment w & 3 and two output-channel fragments for all nine taps.  What the stride changes is the B operand: the 8
// consecutive grad_out pixels of a fragment meet input columns 2 j + kx - 1, two apart.  The input tile is therefore
// staged as two column planes per row -- E[jj] = x[.][2 (j0 + jj)] and O[jj] = x[.][2 (j0 + jj) + 1], jj = -1 .. 31 for O
// -- so that
//   kx = 1 is E[jj .. jj + 7]        one aligned ds_read_b128,
//   kx = 2 is O[jj .. jj + 7]        one aligned ds_read_b128,
//   kx = 0 is O[jj - 1 .. jj + 6]    the same register shifted by one bf16 with the preceding dword (v_alignbit),
// the de-interleave happening for free in the staging (a float4 of x is two E and two O values).  Tile = 2 grad_out rows
// x 32 pixels (5 input rows x 66 columns per channel): 132 KB of LDS, one workgroup per CU like the stride-1 kernel.
#include "cp_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int CO = 64, CI = 64, MTW = 2;
constexpr int GP = 2 * 32 + 8;             // go pitch per output channel (bf16 elements): 2 rows x 32 px
constexpr int XO = 8, XE = 48;             // element offsets of the O plane (O[-1] at XO - 1) and the E plane in a row
constexpr int XR = 88;                     // row pitch: 48 (O) + 40 (E)
constexpr int XP = 5 * XR + 16;            // channel pitch 456 el = 57 x 16 B: the 16 channels of a fragment hit 64 banks
constexpr int G_PLANE = CO * GP, X_PLANE = CI * XP;
constexpr int XQ = 17;                     // float4 units per (channel, row): columns 2 j0 - 4 .. 2 j0 + 63
constexpr int G_UNITS = CO * 2 * 8, G_ITERS = G_UNITS / 512;
constexpr int X_UNITS = CI * 5 * XQ, X_ITERS = (X_UNITS + 511) / 512;
constexpr int STAGE_BYTES = (2 * G_PLANE + 2 * X_PLANE) * 2;
constexpr int OP = CI * 9 + 1, OUT_BYTES = 32 * OP * 4;
constexpr int SMEM = STAGE_BYTES > OUT_BYTES ? STAGE_BYTES : OUT_BYTES;

struct Ws2Args {
  const float* x;
  const float* go;
  float* gw;
  int Cin, Cout, H, W, Ho, Wo, tiles_x, tiles_y, ntiles, tiles_per_wg, n_ci;
};

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 h, l;
  h[0] = (__bf16)a;
  h[1] = (__bf16)b;
  l[0] = (__bf16)(a - (float)h[0]);
  l[1] = (__bf16)(b - (float)h[1]);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

__global__ __launch_bounds__(512) void conv3x3s2_wgrad_kernel(Ws2Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  unsigned short* Gs = reinterpret_cast<unsigned short*>(smem);                // [2][G_PLANE]
  unsigned short* Xs = Gs + 2 * G_PLANE;                                       // [2][X_PLANE]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, c = lane & 15;
  const int nt = wid & 3, mtb = (wid >> 2) * MTW;
  const int ci0 = (blockIdx.y % a.n_ci) * CI, co0 = (blockIdx.y / a.n_ci) * CO;
  const int HW = a.H * a.W, HWo = a.Ho * a.Wo;
  const int k_begin = blockIdx.x * a.tiles_per_wg;
  const int k_end = min(a.ntiles, k_begin + a.tiles_per_wg);

  f32x4 acc[9][MTW];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int m = 0; m < MTW; ++m) acc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging units of this thread (tile-independent parts)
  int g_lds[G_ITERS], g_rc[G_ITERS];                     // LDS element; packed (co, row, col)
#pragma unroll
  for (int i = 0; i < G_ITERS; ++i) {
    const int u = tid + i * 512, q4 = u & 7, r = (u >> 3) & 1, co = u >> 4;
    g_lds[i] = co * GP + r * 32 + 4 * q4;
    g_rc[i] = (co << 16) | (r << 8) | (4 * q4);
  }
  // x unit u = tid + i * 512 -> (q = u % 17 - 1, row = u / 17 % 5, ci = u / 85), recomputed where used (registers);
  // q = -1: columns 2 j0 - 4 .. 2 j0 - 1, of which only the last (O[-1]) is kept

  f32x4 gv[G_ITERS], xv[X_ITERS];
  auto load_tile = [&](int k) __attribute__((always_inline)) {
    const int tx = k % a.tiles_x, ty = (k / a.tiles_x) % a.tiles_y, b = k / (a.tiles_x * a.tiles_y);
    const int j0 = tx * 32, i0 = ty * 2;
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.go + (long long)b * a.Cout * HWo), 0, (int)((unsigned)a.Cout * (unsigned)HWo * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x + (long long)b * a.Cin * HW), 0, (int)((unsigned)a.Cin * (unsigned)HW * 4u), 0x00020000);
#pragma unroll
    for (int i = 0; i < G_ITERS; ++i) {
      const int co = co0 + (g_rc[i] >> 16), y = i0 + ((g_rc[i] >> 8) & 255), x = j0 + (g_rc[i] & 255);
      const bool ok = co < a.Cout && y < a.Ho && x < a.Wo;
      const unsigned off = ok ? ((unsigned)co * (unsigned)HWo + (unsigned)(y * a.Wo + x)) * 4u : OOB;
      gv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, off, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < X_ITERS; ++i) {
      const int u = tid + i * 512;
      const int ci = ci0 + u / (5 * XQ), y = 2 * i0 - 1 + (u / XQ) % 5, x = 2 * j0 + 4 * (u % XQ - 1);
      const bool ok = u < X_UNITS && ci < a.Cin && y >= 0 && y < a.H && x >= 0 && x < a.W;
      const unsigned off = ok ? ((unsigned)ci * (unsigned)HW + (unsigned)(y * a.W + x)) * 4u : OOB;
      xv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
    }
  };

  if (k_begin < k_end) load_tile(k_begin);

  // fragment bases of this lane
  const int a_base = c * GP + 8 * g;                                 // + (mtb + m) * 16 * GP + r * 32
  const int b_base = (nt * 16 + c) * XP + 8 * g;                     // + (2 r + dy) * XR + XO | XE

  for (int k = k_begin; k < k_end; ++k) {
    __syncthreads();                                                 // the previous tile's fragments have been read
#pragma unroll
    for (int i = 0; i < G_ITERS; ++i) {
      unsigned h0, l0, h1, l1;
      split2(gv[i][0], gv[i][1], h0, l0);
      split2(gv[i][2], gv[i][3], h1, l1);
      *reinterpret_cast<u32x2*>(&Gs[g_lds[i]]) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(&Gs[G_PLANE + g_lds[i]]) = u32x2{l0, l1};
    }
#pragma unroll
    for (int i = 0; i < X_ITERS; ++i) {
      const int u = tid + i * 512;
      if (u < X_UNITS) {
        const int q = u % XQ - 1;                                    // float4 = columns 2 j0 + 4 q .. + 3 = E[2q], O[2q], E[2q+1], O[2q+1]
        const int xl = (u / (5 * XQ)) * XP + ((u / XQ) % 5) * XR;
        unsigned he, le, ho, lo;
        split2(xv[i][0], xv[i][2], he, le);
        split2(xv[i][1], xv[i][3], ho, lo);
        if (q >= 0) {
          *reinterpret_cast<unsigned*>(&Xs[xl + XE + 2 * q]) = he;
          *reinterpret_cast<unsigned*>(&Xs[X_PLANE + xl + XE + 2 * q]) = le;
          *reinterpret_cast<unsigned*>(&Xs[xl + XO + 2 * q]) = ho;
          *reinterpret_cast<unsigned*>(&Xs[X_PLANE + xl + XO + 2 * q]) = lo;
        } else {                                                     // columns 2 j0 - 4 .. 2 j0 - 1: only O[-1] = the last one
          Xs[xl + XO - 1] = (unsigned short)(ho >> 16);
          Xs[X_PLANE + xl + XO - 1] = (unsigned short)(lo >> 16);
        }
      }
    }
    __syncthreads();
    if (k + 1 < k_end) load_tile(k + 1);                             // in flight during the matrix phase

#pragma unroll
    for (int r = 0; r < 2; ++r) {
      bf16x8 ah[MTW], al[MTW];
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
        const int e = a_base + (mtb + m) * 16 * GP + r * 32;
        ah[m] = *reinterpret_cast<const bf16x8*>(&Gs[e]);
        al[m] = *reinterpret_cast<const bf16x8*>(&Gs[G_PLANE + e]);
      }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int e = b_base + (2 * r + dy) * XR;
        u32x4 bh[3], bl[3];                                          // kx = 0 (column 2 j - 1), 1 (2 j), 2 (2 j + 1)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) {
          const unsigned short* pl = Xs + hl * X_PLANE + e;
          const u32x4 qo = *reinterpret_cast<const u32x4*>(pl + XO);
          const unsigned pL = *reinterpret_cast<const unsigned*>(pl + XO - 2);
          const u32x4 qe = *reinterpret_cast<const u32x4*>(pl + XE);
          u32x4* dst = hl ? bl : bh;
          dst[0] = u32x4{__builtin_amdgcn_alignbit(qo[0], pL, 16), __builtin_amdgcn_alignbit(qo[1], qo[0], 16),
                         __builtin_amdgcn_alignbit(qo[2], qo[1], 16), __builtin_amdgcn_alignbit(qo[3], qo[2], 16)};
          dst[1] = qe;
          dst[2] = qo;
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const bf16x8 fh = __builtin_bit_cast(bf16x8, bh[dx]), fl = __builtin_bit_cast(bf16x8, bl[dx]);
#pragma unroll
          for (int m = 0; m < MTW; ++m) {
            f32x4 v = acc[dy * 3 + dx][m];
            v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], fh, v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], fl, v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m], fh, v, 0, 0, 0);
            acc[dy * 3 + dx][m] = v;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- flush: 32 output channels per pass through LDS, then coalesced atomics ----
  float* O = reinterpret_cast<float*>(smem);
  const int ciw = min(CI, a.Cin - ci0) * 9;
#pragma unroll
  for (int pass = 0; pass < CO / 32; ++pass) {
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      const int row0 = (mtb + m) * 16 + 4 * g;                       // rows row0 .. row0 + 3 lie in one pass
      if (row0 / 32 == pass) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) O[(row0 % 32 + r) * OP + (nt * 16 + c) * 9 + t] = acc[t][m][r];
      }
    }
    __syncthreads();
    for (int idx = tid; idx < 32 * ciw; idx += 512) {
      const int row = idx / ciw, col = idx - row * ciw;
      const int co = co0 + pass * 32 + row;
      if (co < a.Cout) atomicAdd(&a.gw[((long long)co * a.Cin + ci0) * 9 + col], O[row * OP + col]);
    }
  }
}

}  // namespace

extern "C" {

int cp_conv3x3_s2_wgrad_supported(int32_t Cin, int32_t Cout, int32_t H, int32_t W) {
  // (float4 loads of both maps: W % 8 == 0 makes the rows of x and of grad_out 16-byte aligned; H may be odd)
  return Cin >= 1 && Cout >= 1 && H >= 2 && W >= 8 && W % 8 == 0 && (long long)Cin * H * W * 4 < 0x7FFFFFF0ll &&
         (long long)Cout * ((H - 1) / 2 + 1) * (W / 2) * 4 < 0x7FFFFFF0ll;
}

// gw [Cout][Cin][3][3] += the weight gradient of a 3x3 / stride 2 / pad 1 convolution (float atomics: the caller zeroes
// gw or carries an accumulation).  x [B][Cin][H][W], grad_out [B][Cout][(H - 1) / 2 + 1][W / 2].
int cp_conv3x3_s2_wgrad(const float* x, const float* grad_out, float* gw, int32_t B, int32_t Cin, int32_t H, int32_t W,
                        int32_t Cout, void* stream) {
  CP_CHECK_ARG(x && grad_out && gw && B >= 1);
  if (!cp_conv3x3_s2_wgrad_supported(Cin, Cout, H, W)) return CP_EUNSUPPORTED;
  Ws2Args a;
  a.x = x; a.go = grad_out; a.gw = gw;
  a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.Ho = (H - 1) / 2 + 1; a.Wo = W / 2;
  a.tiles_x = (a.Wo + 31) / 32;
  a.tiles_y = (a.Ho + 1) / 2;
  a.ntiles = a.tiles_x * a.tiles_y * B;
  a.n_ci = (Cin + CI - 1) / CI;
  const int pairs = a.n_ci * ((Cout + CO - 1) / CO);
  int nsplit = (256 + pairs - 1) / pairs;                            // one workgroup per CU (132 KB of LDS)
  if (nsplit > a.ntiles) nsplit = a.ntiles;
  a.tiles_per_wg = (a.ntiles + nsplit - 1) / nsplit;
  nsplit = (a.ntiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
  hipLaunchKernelGGL(conv3x3s2_wgrad_kernel, dim3(nsplit, pairs), dim3(512), 0, (hipStream_t)stream, a);
  return cp_launch_status();
}

}  // extern "C"
