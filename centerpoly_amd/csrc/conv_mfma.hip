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
//   * float32 output: the instruction is issued as mfma(B, A) -- the operands share one register layout -- so the tile
//     comes out transposed (lane = 4 consecutive pixels of one channel): one 16-byte store / residual / mask access per
//     tile.  Split-plane output (template OSPLIT) and the stride-2 gradient (IG2) keep D[co][px] (lane = 4 channels).
//   * split planes (round 4): [hi | lo] x [C / 8][H][W][8 x bf16] written by a producer's epilogue, staged by the consumer
//     (template PRE) with 16-byte loads and no conversion arithmetic: cp_conv_mfma_forward_split.
#include "cp_common.h"

#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#ifndef CP_CVABL
#define CP_CVABL 0                  // timing-only ablation bits (tools/probe_conv_ablate.py; results wrong by construction):
#endif                              // 1 weight fragments never re-loaded, 2 no staging loads, 4 no B ds_reads, 8 no MFMA, 16 no staging at all
#ifdef CP_CVSTAMP                   // diagnostic build (tools/probe_conv_stamp.py): s_memtime stamps of wave 0 over out[]
#define CVSTAMP(i) do { if ((i) < 16) stamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CVSTAMP(i) do { } while (0)
#endif
constexpr int KC = 32;              // input channels per k-step
constexpr int TW = 32;              // tile width (pixels)
constexpr unsigned OOB = 0xFFFFFFF0u;

// wp[(((ct * nchunk + chunk) * taps + tap) * 2 + hl) * 64 + lane][j] =
//   half(hl) of Wsrc[m = ct * 16 + (lane & 15)][k = chunk * 32 + 8 (lane >> 4) + j][tap]        (ct: 16-row tile)
// taps = 9 (3x3) or 1 (1x1).  Wsrc = W ([M][K][taps]) or, transposed (input gradient):
// Wsrc[m][k][tap] = W[k][m][taps - 1 - tap] with W = [K][M][taps].
// transposed = 6 (the one-pass stride-2 input gradient, IG2): Wsrc[m][k][tap] = W[k][m][tap].
// Rows past M (the tile count is rounded up to a multiple of 4) and k past K are zero.
__device__ __forceinline__ void wperm_element(const float* __restrict__ w, bf16x8* __restrict__ wp, int M, int K, int nchunk,
                                              int taps, int transposed, int e) {
  const int lane = e & 63;
  int r = e >> 6;
  const int hl = r & 1;
  r >>= 1;
  const int tap = r % taps;
  r /= taps;
  const int chunk = r % nchunk, ct = r / nchunk;
  const int m = ct * 16 + (lane & 15);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = chunk * KC + 8 * (lane >> 4) + j;
    float v = 0.f;
    if (m < M && k < K) {
      if (transposed == 6) {                                   // IG2: transposed, taps in place
        v = w[((long long)k * M + m) * 9 + tap];
      } else {
        v = transposed ? w[((long long)k * M + m) * taps + (taps - 1 - tap)] : w[((long long)m * K + k) * taps + tap];
      }
    }
    const __bf16 h = (__bf16)v;
    o[j] = hl ? (__bf16)(v - (float)h) : h;
  }
  wp[e] = o;
}

__global__ __launch_bounds__(256) void conv_mfma_wperm_kernel(const float* __restrict__ w, bf16x8* __restrict__ wp, int M,
                                                              int K, int nchunk, int taps, int transposed, int total) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < total) wperm_element(w, wp, M, K, nchunk, taps, transposed, e);
}

// every job of a table in one launch (the training step's ~110 weight forms, permuted once after the optimizer's update
// instead of one 4-microsecond launch per use): the workgroup finds its job by bisection over the jobs' first blocks
__global__ __launch_bounds__(256) void conv_mfma_wperm_batch_kernel(const cp_conv_prepare_job* __restrict__ jobs, int njobs) {
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const cp_conv_prepare_job j = jobs[lo];
  const int nchunk = (j.Cin + KC - 1) / KC;
  const int total = ((j.Cout + 63) / 64 * 4) * nchunk * j.taps * 2 * 64;
  const int e = ((int)blockIdx.x - j.first_block) * 256 + threadIdx.x;
  if (e < total) wperm_element(j.weight, (bf16x8*)j.wperm, j.Cout, j.Cin, nchunk, j.taps, j.transposed, e);
}

// sum over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15), left in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});           // quad_perm [1, 0, 3, 2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});           // quad_perm [2, 3, 0, 1]
  v += dpp(v, std::integral_constant<int, 0x141>{});          // row_half_mirror
  v += dpp(v, std::integral_constant<int, 0x140>{});          // row_mirror
  return v;
}

constexpr int MAXSRC = 4;
struct CvArgs {
  const float* xsrc[MAXSRC];   // the input is the channel concatenation of up to 4 tensors [B][csrc[i]][H][W]
  int csrc[MAXSRC];            // (several sources: each a multiple of 32 channels, so no k-step straddles two)
  const bf16x8* wp;
  const float* bias;      // [Cout] or null
  const float* res;       // same shape as out, or null
  float* out;
  int Cin, H, W, Ho, Wo, Cout, nchunk, ncot, tiles_x, relu;   // H, W: input; Ho, Wo: output grid of the launch
  const float* mask;           // same shape as out, or null: out = mask > 0 ? value : 0 (gradient through a ReLU whose output was `mask`)
  float* colsum;               // null, or [B * tiles * 8][Cout] scratch: row (b, tile, wave, half) = sums of out per channel over RW x 16 pixels
  int Hf, Wf;                  // IG2: the gradient map (element (y, x) of class (py, px) goes to (2 y + py, 2 x + px)); else Ho, Wo
  // SPLIT activations (round 4): the tensor as two planes [hi | lo] of [B][C / 8][H][W][8 x bf16] -- what the staging
  // below builds from float32 anyway, written once by the PRODUCER's epilogue (out_split) and staged by the consumer
  // (template PRE, xpre) with two 16-byte loads per unit and no conversion arithmetic.  Same bytes as float32.
  const bf16x8* xpre = nullptr;   // PRE: the input's hi plane; lo plane at + pre_plane units
  long long pre_plane = 0;
  int out_split = 0;              // the output leaves as split planes (out = hi plane, lo at + out_plane units); Cout % 8 == 0
  long long out_plane = 0;
};

// KS > 1 (deep, small layers whose grid cannot fill the chip -- one wave per SIMD exposes every load latency): the
// workgroup is KS groups of 4 waves, group q takes the channel chunks q, q + KS, ... of the SAME output tile through
// its own staged tile, so the sequential chunk steps drop KS-fold; the partial accumulators meet in LDS in a fixed
// order (deterministic) and group 0 runs the epilogue.
// S = 2: stride 2 (the first convolution of DLA levels 2-5): the staged tile is 2 TH + 1 rows x 65 columns, the B
// fragments are read at twice the pixel stride (32-byte lane stride: two-way bank conflicts on those reads).
// ST = 2 with TAPS = 1 (the stride-2 1x1 skip convolutions of the Hourglass residuals, large_hourglass.py:55-81): a
// 1x1 convolution of every second pixel -- the tile geometry is the stride-1 one, the staging reads at twice the
// coordinates, nothing that is not multiplied is staged.
// IG2 (TAPS = 9, stride-1 geometry over grad_out): the whole input gradient of a stride-2 3x3 convolution in ONE pass --
// the four parity classes of the gradient's rows / columns keep their own accumulators (4 x MT x NT tiles), every one
// of the nine weight taps (ky, kx) feeds the class ((ky != 1), (kx != 1)) from the staged row / column offset
// (ky == 0 ? 2 : 1, kx == 0 ? 2 : 1): grad_out is staged once and the matrix cores do exactly the convolution's flops.
template <int MT, int RW, int TAPS, int KS = 1, int ST = 1, bool IG2 = false, bool PRE = false, bool OSPLIT = false>
__global__ __launch_bounds__(256 * KS, KS > 1 ? (ST > 1 && TAPS == 9 ? 1 : 4) : 2) void conv_mfma_kernel(CvArgs a) {
  static_assert(!IG2 || (TAPS == 9 && KS == 1 && ST == 1), "IG2: 3x3, stride-1 staging, no split-K");
  static_assert(!PRE || (TAPS == 9 && ST == 1 && !IG2), "PRE: 3x3 / stride 1 forward");
  static_assert(!OSPLIT || !IG2, "split planes out: forward forms");
  // orientation of the accumulator tiles: channels x pixels (lane = 4 channels of a pixel: the split-plane and the
  // stride-2 gradient epilogues) or pixels x channels (lane = 4 consecutive pixels of a channel: float32 NCHW out)
  constexpr bool TRD = !IG2 && !OSPLIT;
  constexpr int NCLS = IG2 ? 4 : 1;
#ifdef CP_CVSTAMP
  unsigned long long stamp[16];
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
  int sti = 1;
#endif
  CVSTAMP(0);
  constexpr int S = TAPS == 1 ? 1 : ST;          // stride of the staged tile's geometry
  constexpr int SUB = TAPS == 1 ? ST : 1;        // 1x1: input subsampling folded into the staging addresses
  constexpr int HALO = TAPS == 9 ? 1 : 0, LW = S * TW + (TAPS == 9 ? 3 : 1) - S;
  constexpr int TH = 4 * RW, LH = S * TH + (TAPS == 9 ? 3 : 1) - S, NT = 2 * RW, PLANE = 4 * LH * LW;   // fragments per half
  constexpr int UNITS = PLANE, ITERS = (UNITS + 255) / 256, SB = ITERS <= 6 ? ITERS : 5;
  constexpr int RED = (KS - 1) * 256 * MT * NT;                                     // f32x4 slots of the reduction
  constexpr int XS_F = KS * 2 * PLANE;
  __shared__ bf16x8 Xall[XS_F > RED ? XS_F : RED];

  const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  bf16x8* Xs = Xall + grp * 2 * PLANE;
  const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6, g = lane >> 4, c = lane & 15;
  // XCD-aware order: the hardware deals consecutive workgroups round-robin over the 8 XCDs (each with its own L2), so
  // the ncot workgroups that stage the SAME input tile (one per output-channel tile) would fetch it through 4 different
  // L2s -- 4.2x the input's bytes on the fabric side at 128 -> 128 (PMC FETCH_SIZE).  Workgroup w takes the logical index
  // (w % 8) * (n / 8) + w / 8: logical neighbours share an XCD, and with it the tile's rows in L2.
  // (Channel-tile-major order for the deep, small maps whose weights outweigh their input -- 512 -> 512 @32x64 -- was
  //  measured too: no gain.)
  int lw = blockIdx.x;
  if ((gridDim.x & 7) == 0) lw = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int cot = lw % a.ncot, tile = lw / a.ncot;
  const int x0 = (tile % a.tiles_x) * TW, y0 = (tile / a.tiles_x) * TH, b = blockIdx.y;
  const int HW = a.H * a.W;                                  // input plane size

  // staging units of this thread: (channel group, row, col) -> byte offset of channel 0 of the group, or OOB
  unsigned soff[ITERS];
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int u = tid + i * 256;
    const int col = u % LW, r = (u / LW) % LH, cg = u / (LW * LH);
    const int gy = SUB * (S * y0 - HALO + r), gx = SUB * (S * x0 - HALO + col);
    const bool ok = u < UNITS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    soff[i] = !ok ? OOB
              : PRE ? ((unsigned)cg * (unsigned)HW + (unsigned)(gy * a.W + gx)) * 16u
                    : ((unsigned)(cg * 8) * (unsigned)HW + (unsigned)(gy * a.W + gx)) * 4u;
  }
  const unsigned cstep = (unsigned)HW * 4u;

  f32x4 accs[NCLS][MT][NT];
#pragma unroll
  for (int q = 0; q < NCLS; ++q)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) accs[q][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 (&acc)[MT][NT] = accs[0];

  const long long tstride = (long long)a.nchunk * TAPS * 2 * 64;       // fragments per 16-row weight tile
  const bf16x8* wq = a.wp + (long long)cot * MT * tstride + lane;
  bf16x8 af[MT][2];
  {
    const bf16x8* fq = wq + (long long)(min(grp, a.nchunk - 1) * TAPS * 2) * 64;     // this group's first chunk
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      af[m][0] = fq[m * tstride];
      af[m][1] = fq[m * tstride + 64];
    }
  }

  // B fragment base of this lane: channel group g, wave's first row, col c (tile origin is (-HALO, -HALO))
  const int bbase = (g * LH + S * wid * RW) * LW + S * c;

  int src = 0, src_c0 = 0;                                    // source tensor of the current chunk, its first channel
  // staging of one chunk: `load` issues the unit loads i0 .. i0 + N - 1 into v, `store` splits them into the LDS planes
  auto load = [&](int chunk, int i0, auto& v) {
    while (src + 1 < MAXSRC && chunk * KC >= src_c0 + a.csrc[src] && a.csrc[src + 1] > 0) {
      src_c0 += a.csrc[src];
      ++src;
    }
    const int cs = a.csrc[src];
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.xsrc[src] + (long long)b * cs * HW), 0, (int)((unsigned)cs * (unsigned)HW * 4u), 0x00020000);
    const unsigned cb = (unsigned)(chunk * KC - src_c0) * cstep;
    constexpr int N = sizeof(v) / sizeof(v[0]);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const unsigned o = (i0 + i >= ITERS || soff[(i0 + i) % ITERS] == OOB) ? OOB : soff[(i0 + i) % ITERS] + cb;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[i][j] = (CP_CVABL & 2) ? (float)(o + j) : __builtin_bit_cast(
            float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, o == OOB ? OOB : o + j * cstep, 0, 0));
    }
  };
  auto store = [&](int i0, auto& v) {
    constexpr int N = sizeof(v) / sizeof(v[0]);
#pragma unroll
    for (int i = 0; i < N; ++i) {
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
  };
  // PRE: the unit is already [8 x bf16] in both planes -- two 16-byte loads, two 16-byte LDS stores
  const long long pre_img = PRE ? (long long)b * (a.Cin >> 3) * HW : 0;
  const int pre_bytes = PRE ? (int)((unsigned)(a.Cin >> 3) * (unsigned)HW * 16u) : 0;
  const __amdgpu_buffer_rsrc_t rs_ph = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16x8*>(PRE ? a.xpre + pre_img : a.wp), 0, pre_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_pl = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16x8*>(PRE ? a.xpre + a.pre_plane + pre_img : a.wp), 0, pre_bytes, 0x00020000);
  auto load_pre = [&](int chunk, int i0, auto& vh, auto& vl) {
    const unsigned cb = (unsigned)chunk * 4u * (unsigned)HW * 16u;
    constexpr int N = sizeof(vh) / sizeof(vh[0]);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const unsigned o = (i0 + i >= ITERS || soff[(i0 + i) % ITERS] == OOB) ? OOB : soff[(i0 + i) % ITERS] + cb;
      vh[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_ph, o, 0, 0);
      vl[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_pl, o, 0, 0);
    }
  };
  auto store_pre = [&](int i0, auto& vh, auto& vl) {
    constexpr int N = sizeof(vh) / sizeof(vh[0]);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int u = tid + (i0 + i) * 256;
      if (i0 + i < ITERS && u < UNITS) {
        Xs[u] = __builtin_bit_cast(bf16x8, vh[i]);
        Xs[PLANE + u] = __builtin_bit_cast(bf16x8, vl[i]);
      }
    }
  };
  auto bread = [&](int tap, bf16x8 (&bh)[NT], bf16x8 (&bl)[NT]) {
    const int dy = IG2 ? (tap / 3 == 0 ? 2 : 1) : tap / 3, dx = IG2 ? (tap % 3 == 0 ? 2 : 1) : tap % 3;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int idx = bbase + (S * (n >> 1) + dy) * LW + S * (n & 1) * 16 + dx;
      bh[n] = (CP_CVABL & 4) ? af[n % MT][0] : Xs[idx];
      bl[n] = (CP_CVABL & 4) ? af[n % MT][1] : Xs[PLANE + idx];
    }
  };

  for (int chunk = grp; chunk - grp < a.nchunk; chunk += KS) {
    const bool active = chunk < a.nchunk;                     // (wave-uniform; every wave takes every barrier)
    __syncthreads();                                          // the previous chunk's fragments have been read
#ifdef CP_CVSTAMP
    CVSTAMP(sti); ++sti;
#endif
    if (active && !(CP_CVABL & 16)) {
#pragma unroll
      for (int i0 = 0; i0 < ITERS; i0 += SB) {                // batches of SB units: 8 SB loads in flight per thread
        if constexpr (PRE) {
          u32x4 vh[SB], vl[SB];
          load_pre(chunk, i0, vh, vl);
          store_pre(i0, vh, vl);
        } else {
          float v[SB][8];
          load(chunk, i0, v);
          store(i0, v);
        }
      }
    }
#ifdef CP_CVSTAMP
    CVSTAMP(sti); ++sti;
#endif
    __syncthreads();
#ifdef CP_CVSTAMP
    CVSTAMP(sti); ++sti;
#endif
    if (!active) continue;

#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int cls = IG2 ? (tap / 3 != 1) * 2 + (tap % 3 != 1) : 0;
      // next tap's (or this group's next chunk's first) weight fragments; past the end: re-read the last
      bf16x8 an[MT][2];
      {
        const int nfrag = tap < TAPS - 1 ? chunk * TAPS + tap + 1
                                         : (chunk + KS < a.nchunk ? (chunk + KS) * TAPS : chunk * TAPS + tap);
        const bf16x8* nq = wq + (long long)(nfrag * 2) * 64;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          an[m][0] = (CP_CVABL & 1) ? af[m][0] : nq[m * tstride];
          an[m][1] = (CP_CVABL & 1) ? af[m][1] : nq[m * tstride + 64];
        }
      }
      {
        bf16x8 bh[NT], bl[NT];
        bread(tap, bh, bl);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            f32x4& d = accs[cls][m][n];
            if (CP_CVABL & 8) {
              d[0] += (float)bh[n][0] * (float)af[m][0][0] + (float)bl[n][1] * (float)af[m][1][1];
              continue;
            }
            if constexpr (TRD) {
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[n], af[m][0], d, 0, 0, 0);
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[n], af[m][0], d, 0, 0, 0);
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[n], af[m][1], d, 0, 0, 0);
            } else {
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][0], bh[n], d, 0, 0, 0);
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][0], bl[n], d, 0, 0, 0);
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][1], bh[n], d, 0, 0, 0);
            }
          }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        af[m][0] = an[m][0];
        af[m][1] = an[m][1];
      }
      __builtin_amdgcn_sched_barrier(0);                      // keep the taps' fragment reads from piling up
    }
#ifdef CP_CVSTAMP
    CVSTAMP(sti); ++sti;
#endif
  }

  if (KS > 1) {                                               // partial sums of groups 1 .. KS-1 -> LDS -> group 0
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(Xall);
    if (grp > 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) red[((grp - 1) * MT * NT + m * NT + n) * 256 + tid] = acc[m][n];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int q = 1; q < KS; ++q)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] += red[((q - 1) * MT * NT + m * NT + n) * 256 + tid];
  }

  // epilogue: D[row = 4 g + r (co)][col = c (pixel)]
  const int HWf = a.Hf * a.Wf;
  float* ob = a.out + (long long)b * a.Cout * HWf;
  const float* rb = a.res ? a.res + (long long)b * a.Cout * HWf : nullptr;
  if (IG2) {                                                   // class (py, px) of (y, x) -> (2 y + py, 2 x + px)
    const bool pair = (a.Wf & 1) == 0;                         // even width: the two column classes leave as one float2
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = (cot * MT + m) * 16 + 4 * g + r;
        if (co >= a.Cout) continue;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int y = y0 + wid * RW + (n >> 1), x = x0 + (n & 1) * 16 + c;
          if (y >= a.Ho || x >= a.Wo) continue;
#pragma unroll
          for (int py = 0; py < 2; ++py) {
            const int Y = 2 * y + py, X = 2 * x;
            if (Y >= a.Hf || X >= a.Wf) continue;
            const long long o = (long long)co * HWf + (long long)Y * a.Wf + X;
            float v0 = accs[NCLS > 1 ? 2 * py : 0][m][n][r], v1 = accs[NCLS > 1 ? 2 * py + 1 : 0][m][n][r];
            if (pair) {
              if (rb) {
                const float2 q = *reinterpret_cast<const float2*>(rb + o);
                v0 += q.x;
                v1 += q.y;
              }
              *reinterpret_cast<float2*>(ob + o) = make_float2(v0, v1);
            } else {
              ob[o] = rb ? v0 + rb[o] : v0;
              if (X + 1 < a.Wf) ob[o + 1] = rb ? v1 + rb[o + 1] : v1;
            }
          }
        }
      }
    }
    return;
  }
  if constexpr (OSPLIT) {
    // split planes out: the lane's four consecutive channels (4 g .. 4 g + 3) of its pixel are half a unit -- one
    // 8-byte store per plane; the lanes of a row pair (g, g | 1) x 16 pixels fill 256 contiguous bytes per store
    const long long img = (long long)b * (a.Cout >> 3) * HWf;
    const int obytes = (int)((unsigned)(a.Cout >> 3) * (unsigned)HWf * 16u);
    bf16x8* oh = reinterpret_cast<bf16x8*>(a.out) + img;
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(oh, 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc(oh + a.out_plane, 0, obytes, 0x00020000);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int co0 = (cot * MT + m) * 16 + 4 * g;
      float bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = (a.bias && co0 + r < a.Cout) ? a.bias[co0 + r] : 0.f;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int y = y0 + wid * RW + (n >> 1), xx = x0 + (n & 1) * 16 + c;
        const bool ok = co0 < a.Cout && y < a.Ho && xx < a.Wo;
        bf16x4 h, l;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[m][n][r] + bv[r];
          if (a.relu) v = fmaxf(v, 0.f);
          const __bf16 hh = (__bf16)v;
          h[r] = hh;
          l[r] = (__bf16)(v - (float)hh);
        }
        const unsigned off = ok ? ((unsigned)(co0 >> 3) * (unsigned)HWf + (unsigned)(y * a.Wf + xx)) * 16u + (unsigned)(g & 1) * 8u
                                : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h), rs_h, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, l), rs_l, off, 0, 0);
      }
    }
    return;
  }
  // Float32 output.  The matrix instruction runs TRANSPOSED here (pixels on the M side, channels on the N side -- the
  // two operands have the same register layout, so it is only their order): lane (g, c) holds the four consecutive
  // pixels 4 g .. 4 g + 3 of channel c, i.e. ONE 16-byte store / residual / mask access per accumulator tile, 16
  // channels x 64 contiguous bytes per wave instruction (round 4; before: dword accesses on 128-byte lines behind a
  // v_permlane16_swap, 4x the vector-memory instructions -- the split-plane epilogue above showed what they cost).
  // Branch-free: invalid elements get an offset past the descriptor's range (loads return 0, stores are dropped).
  // Maps whose width is not a multiple of 4 (or unaligned tensors) take dword accesses element by element.
  const int cot0 = cot * MT * 16;
  const unsigned span = (unsigned)min(a.Cout - cot0, MT * 16) * (unsigned)HWf * 4u;       // this workgroup's channels
  const long long tbase = (long long)b * a.Cout * HWf + (long long)cot0 * HWf;
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(a.out + tbase, 0, (int)span, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_r =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res + tbase : a.out), 0, a.res ? (int)span : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_m =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.mask ? a.mask + tbase : a.out), 0, a.mask ? (int)span : 0, 0x00020000);
  constexpr unsigned EOOB = 0x80000000u;
  const bool vec = (a.Wf & 3) == 0 && ((reinterpret_cast<unsigned long long>(a.out) | reinterpret_cast<unsigned long long>(a.res) |
                                        reinterpret_cast<unsigned long long>(a.mask)) & 15) == 0;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int cl = m * 16 + c;                                   // channel within the workgroup's tile
    const bool cok = cot0 + cl < a.Cout;
    const float bv = (a.bias && cok) ? a.bias[cot0 + cl] : 0.f;
    unsigned off[NT];
    f32x4 rv[NT], mv[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int y = y0 + wid * RW + (n >> 1), x = x0 + (n & 1) * 16 + 4 * g;
      off[n] = (cok && y < a.Ho && x < a.Wo) ? ((unsigned)cl * (unsigned)HWf + (unsigned)(y * a.Wf + x)) * 4u : EOOB;
      if (vec) {
        if (a.res) rv[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, off[n], 0, 0));
        if (a.mask) mv[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_m, off[n], 0, 0));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned o = (off[n] != EOOB && x + r < a.Wo) ? off[n] + 4u * r : EOOB;
          if (a.res) rv[n][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, o, 0, 0));
          if (a.mask) mv[n][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_m, o, 0, 0));
        }
      }
    }
    float csum[2] = {0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int x = x0 + (n & 1) * 16 + 4 * g;
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = acc[m][n][r] + bv;
        if (a.res) t += rv[n][r];
        if (a.relu) t = fmaxf(t, 0.f);
        if (a.mask && !(mv[n][r] > 0.f)) t = 0.f;
        v[r] = t;
        csum[n & 1] += (off[n] != EOOB && x + r < a.Wo) ? t : 0.f;
      }
      if (vec) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_o, off[n], 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)v[r]), rs_o,
                                                (off[n] != EOOB && x + r < a.Wo) ? off[n] + 4u * r : EOOB, 0, 0);
      }
    }
    if (a.colsum) {                                              // (wave-uniform) per wave and 16-pixel half: this channel's sum
      const long long row0 = (((long long)b * (gridDim.x / a.ncot) + tile) * 4 + wid) * 2;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float t = csum[h];
        t += __shfl_xor(t, 16);
        t += __shfl_xor(t, 32);
        if (g == 0 && cok) a.colsum[(row0 + h) * a.Cout + cot0 + cl] = t;
      }
    }
  }
#ifdef CP_CVSTAMP
  {
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
      float* o = a.out + (long long)gridDim.y * a.Cout * a.Hf * a.Wf + ((long long)b * gridDim.x + blockIdx.x) * 16;   // past the output
      for (int i = 0; i + 1 < 14; ++i) o[i] = i + 1 < sti ? (float)(stamp[i + 1] - stamp[i]) : 0.f;
      o[13] = (float)sti;
      o[14] = (float)(rt1 - rt0);          // 100 MHz ticks
      o[15] = (float)(rt0 & 0xffffff);
    }
  }
#endif
}

// out[c] += sum over rows of part[row][c] (the per-wave channel sums of a colsum launch): grid (ceil(C / 64), 32),
// workgroup = 64 channels x 4 row phases; one float atomic per workgroup and channel (32 per address)
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ part, int nrows, int C,
                                                            float* __restrict__ out) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  float s = 0.f;
  if (c < C)
    for (int r = blockIdx.y * 4 + ph; r < nrows; r += gridDim.y * 4) s += part[(long long)r * C + c];
  red[ph][cl] = s;
  __syncthreads();
  if (ph == 0 && c < C) atomicAdd(&out[c], (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]));
}

int tiles16(int Cout) { return (Cout + 63) / 64 * 4; }             // 16-row weight tiles, padded to whole groups of 4

}  // namespace

extern "C" {

int cp_conv3x3_mfma_supported(int32_t Cin, int32_t Cout, int32_t H, int32_t W) {
  // channels past Cin (the last k-step of a ragged Cin) lie beyond the buffer bound and read as zero
  return Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1 && (long long)Cin * H * W * 4 < 0x7FFFFFF0ll &&
         (long long)(Cin + KC) * H * W * 4 < 0xFFFFFFF0ll;
}

size_t cp_conv_mfma_weight_bytes(int32_t Cin, int32_t Cout, int32_t taps) {
  return (size_t)tiles16(Cout) * ((Cin + KC - 1) / KC) * taps * 2 * 64 * 16;
}

size_t cp_conv3x3_mfma_weight_bytes(int32_t Cin, int32_t Cout) { return cp_conv_mfma_weight_bytes(Cin, Cout, 9); }

// weight: [Cout][Cin][k][k], k*k = taps (transposed = 0), or -- for the input gradient of a convolution whose weight
// is [K][M][k][k] -- the same tensor read as Wsrc[m][k][tap] = W[k][m][taps - 1 - tap] (transposed = 1; Cin := K,
// Cout := M).
int cp_conv_mfma_prepare(const float* weight, int32_t Cin, int32_t Cout, int32_t taps, int32_t transposed, void* wperm,
                         void* stream) {
  hipStream_t st = (hipStream_t)stream;
  CP_CHECK_ARG(weight && wperm && Cin >= 1 && Cout >= 1 && (taps == 1 || taps == 9));
  const int nchunk = (Cin + KC - 1) / KC;
  const int total = tiles16(Cout) * nchunk * taps * 2 * 64;
  hipLaunchKernelGGL(conv_mfma_wperm_kernel, dim3((total + 255) / 256), dim3(256), 0, st, weight, (bf16x8*)wperm, Cout,
                     Cin, nchunk, taps, transposed, total);
  return cp_launch_status();
}

int32_t cp_conv_mfma_prepare_blocks(int32_t Cin, int32_t Cout, int32_t taps) {
  return (int32_t)((cp_conv_mfma_weight_bytes(Cin, Cout, taps) / 16 + 255) / 256);
}

int cp_conv_mfma_prepare_batch(const cp_conv_prepare_job* jobs_device, int32_t njobs, int32_t total_blocks, void* stream) {
  CP_CHECK_ARG(jobs_device && njobs >= 1 && total_blocks >= 1);
  hipLaunchKernelGGL(conv_mfma_wperm_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_device, njobs);
  return cp_launch_status();
}

int cp_conv3x3_mfma_prepare(const float* weight, int32_t Cin, int32_t Cout, int32_t transposed, void* wperm,
                            void* stream) {
  return cp_conv_mfma_prepare(weight, Cin, Cout, 9, transposed, wperm, stream);
}

// The input is the channel concatenation of `nsrc` (1..4) tensors xs[i] = [B][cs[i]][H][W] (what the reference
// builds with torch.cat before a 1x1 `Root` convolution, pose_dla_dcn.py:148-166) -- read in place, never
// materialised.  With several sources every cs[i] must be a multiple of 32.  taps = 9: 3x3 / pad 1; taps = 1: 1x1.
// Tile-variant choice and launch for a filled-in argument block (a.Ho x a.Wo: the launch's output grid).
static int conv_dispatch(CvArgs& a, int B, int Cout, int taps, int stride, hipStream_t st, int* tiles_out = nullptr) {
  const int Ho = a.Ho;
  // Tile variant by how many workgroups it yields (the chip wants >= 2 per CU): 64 output channels x 8 rows is the
  // most efficient (fewest fragment bytes per MFMA); layers that cannot fill the CUs with it take 32 channels
  // x 8 rows, then 32 x 4 rows.  <= 32 output channels: 32 x 16 rows, then the same narrow forms.
  auto wgs = [&](int mt, int th) { return (long long)a.tiles_x * ((Ho + th - 1) / th) * B * ((Cout + 16 * mt - 1) / (16 * mt)); };
  auto launch = [&](auto kernel, int mt, int th, int ks) {
    a.ncot = (Cout + 16 * mt - 1) / (16 * mt);
    const int tiles = a.tiles_x * ((Ho + th - 1) / th);
    if (tiles_out) *tiles_out = tiles;
    hipLaunchKernelGGL(kernel, dim3(tiles * a.ncot, B), dim3(256 * ks), 0, st, a);
  };
  if (stride == 2 && taps == 1) {                    // 1x1 over every second pixel: the 1x1 forms on the output grid
    if (Cout > 32 && wgs(4, 8) >= 448) launch(conv_mfma_kernel<4, 2, 1, 1, 2>, 4, 8, 1);
    else if (wgs(2, 8) >= 320) launch(conv_mfma_kernel<2, 2, 1, 1, 2>, 2, 8, 1);
    else if (wgs(2, 4) < 384 && a.nchunk >= 8) launch(conv_mfma_kernel<2, 1, 1, 4, 2>, 2, 4, 4);
    else launch(conv_mfma_kernel<2, 1, 1, 1, 2>, 2, 4, 1);
  } else if (stride == 2) {                          // 4-row tiles only (the staged tile is 9 x 65 pixels)
    const bool big = Cout > 32 && wgs(4, 4) >= 448;
    // deep, small maps (256 -> 512 @64x128: 256 workgroups, one per CU, eight sequential chunks): two groups of waves take
    // alternate chunks through their own staged tiles (2 x 75 KB of LDS: one workgroup per CU is all there is anyway)
    const bool ks2 = !big && wgs(2, 4) <= 256 && a.nchunk >= 4;
    if (a.out_split) {
      if (big) launch(conv_mfma_kernel<4, 1, 9, 1, 2, false, false, true>, 4, 4, 1);
      else if (ks2) launch(conv_mfma_kernel<2, 1, 9, 2, 2, false, false, true>, 2, 4, 2);
      else launch(conv_mfma_kernel<2, 1, 9, 1, 2, false, false, true>, 2, 4, 1);
    } else {
      if (big) launch(conv_mfma_kernel<4, 1, 9, 1, 2>, 4, 4, 1);
      else if (ks2) launch(conv_mfma_kernel<2, 1, 9, 2, 2>, 2, 4, 2);
      else launch(conv_mfma_kernel<2, 1, 9, 1, 2>, 2, 4, 1);
    }
  } else if (taps == 9) {                            // split-plane input / output: the same tile choice, other staging / epilogue
    auto pick = [&](auto pre, auto os) {
      constexpr bool PRE = decltype(pre)::value, OS = decltype(os)::value;
      if (Cout > 32 && wgs(4, 8) >= 448) launch(conv_mfma_kernel<4, 2, 9, 1, 1, false, PRE, OS>, 4, 8, 1);
      else if (Cout <= 32 && wgs(2, 16) >= 448) launch(conv_mfma_kernel<2, 4, 9, 1, 1, false, PRE, OS>, 2, 16, 1);
      else if (wgs(2, 8) >= 320) launch(conv_mfma_kernel<2, 2, 9, 1, 1, false, PRE, OS>, 2, 8, 1);
      else if (wgs(2, 4) < 384 && a.nchunk >= 8) launch(conv_mfma_kernel<2, 1, 9, 4, 1, false, PRE, OS>, 2, 4, 4);   // in-workgroup K split
      else if (wgs(2, 4) < 768 && a.nchunk >= 4) launch(conv_mfma_kernel<2, 1, 9, 2, 1, false, PRE, OS>, 2, 4, 2);
      else launch(conv_mfma_kernel<2, 1, 9, 1, 1, false, PRE, OS>, 2, 4, 1);
    };
    if (a.xpre && a.out_split) pick(std::true_type{}, std::true_type{});
    else if (a.xpre) pick(std::true_type{}, std::false_type{});
    else if (a.out_split) pick(std::false_type{}, std::true_type{});
    else pick(std::false_type{}, std::false_type{});
  } else {                                          // 1x1: bandwidth-bound, the grid only has to fill the chip
    if (Cout > 32 && wgs(4, 8) >= 448) launch(conv_mfma_kernel<4, 2, 1>, 4, 8, 1);
    else if (wgs(2, 8) >= 320) launch(conv_mfma_kernel<2, 2, 1>, 2, 8, 1);
    else if (wgs(2, 4) < 384 && a.nchunk >= 8) launch(conv_mfma_kernel<2, 1, 1, 4>, 2, 4, 4);
    else if (wgs(2, 4) < 768 && a.nchunk >= 4) launch(conv_mfma_kernel<2, 1, 1, 2>, 2, 4, 2);
    else launch(conv_mfma_kernel<2, 1, 1>, 2, 4, 1);
  }
  return cp_launch_status();
}

static int conv_forward_impl(const float* const* xs, const int32_t* cs, int32_t nsrc, const void* wperm, const float* bias,
                             const float* residual, const float* mask, float* colsum, float* out, int32_t B, int32_t H,
                             int32_t W, int32_t Cout, int32_t taps, int32_t stride, int32_t relu, void* stream,
                             int* tiles_out = nullptr, int x_split = 0, int out_split = 0) {
  hipStream_t st = (hipStream_t)stream;
  CP_CHECK_ARG(xs && cs && wperm && out && B >= 1 && nsrc >= 1 && nsrc <= MAXSRC && (taps == 1 || taps == 9));
  if (stride != 1 && stride != 2) return CP_EUNSUPPORTED;
  CvArgs a;
  int Cin = 0;
  for (int i = 0; i < MAXSRC; ++i) {
    a.xsrc[i] = i < nsrc ? xs[i] : nullptr;
    a.csrc[i] = i < nsrc ? cs[i] : 0;
    if (i < nsrc) {
      CP_CHECK_ARG(xs[i] && cs[i] >= 1);
      if (nsrc > 1 && cs[i] % KC != 0) return CP_EUNSUPPORTED;
      if (!cp_conv3x3_mfma_supported(cs[i], Cout, H, W)) return CP_EUNSUPPORTED;
      Cin += cs[i];
    }
  }
  a.wp = (const bf16x8*)wperm;
  a.bias = bias;
  a.res = residual;
  a.out = out;
  a.Cin = Cin;
  a.H = H;
  a.W = W;
  a.Cout = Cout;
  a.Ho = (H - 1) / stride + 1;                       // (pad = k / 2)
  a.Wo = (W - 1) / stride + 1;
  a.nchunk = (Cin + KC - 1) / KC;
  a.tiles_x = (a.Wo + TW - 1) / TW;
  a.relu = relu;
  a.Hf = a.Ho; a.Wf = a.Wo;
  a.mask = mask; a.colsum = colsum;
  if ((long long)a.Ho * a.Wo * 64 * 4 >= 0x7FFFFFF0ll) return CP_EUNSUPPORTED;     // (32-bit offsets within a channel tile)
  if (x_split) {                                     // xs[0] = the hi plane of a split tensor
    if (nsrc != 1 || taps != 9 || stride != 1 || Cin % KC != 0) return CP_EUNSUPPORTED;
    a.xpre = reinterpret_cast<const bf16x8*>(xs[0]);
    a.pre_plane = (long long)B * (Cin / 8) * H * W;
  }
  if (out_split) {
    if (Cout % 8 != 0 || residual || mask || colsum || taps != 9) return CP_EUNSUPPORTED;
    a.out_split = 1;
    a.out_plane = (long long)B * (Cout / 8) * a.Ho * a.Wo;
  }
  return conv_dispatch(a, B, Cout, taps, stride, st, tiles_out);
}

// float32 [B][C][H][W] <-> split planes [hi | lo][B][C / 8][H][W][8 x bf16] (hi = bf16(v), lo = bf16(v - hi): the
// staging's own split, so a convolution of the split tensor equals the convolution of the float32 one bit for bit)
__global__ __launch_bounds__(256) void activation_split_kernel(const float* __restrict__ x, bf16x8* __restrict__ out,
                                                               long long HW, long long plane) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= plane) return;
  const long long bc = e / HW, p = e % HW;
  bf16x8 h, l;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = x[(bc * 8 + j) * HW + p];
    const __bf16 hh = (__bf16)v;
    h[j] = hh;
    l[j] = (__bf16)(v - (float)hh);
  }
  out[e] = h;
  out[plane + e] = l;
}

__global__ __launch_bounds__(256) void activation_unsplit_kernel(const bf16x8* __restrict__ in, float* __restrict__ x,
                                                                 long long HW, long long plane) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= plane) return;
  const long long bc = e / HW, p = e % HW;
  const bf16x8 h = in[e], l = in[plane + e];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[(bc * 8 + j) * HW + p] = (float)h[j] + (float)l[j];
}

int cp_activation_split(const float* x, void* out, int32_t B, int32_t C, int32_t H, int32_t W, void* stream) {
  CP_CHECK_ARG(x && out && B >= 1 && C >= 8 && H >= 1 && W >= 1);
  if (C % 8 != 0) return CP_EUNSUPPORTED;
  const long long HW = (long long)H * W, plane = (long long)B * (C / 8) * HW;
  if (plane >= 0x7FFFFFFFll * 256) return CP_EUNSUPPORTED;
  hipLaunchKernelGGL(activation_split_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (bf16x8*)out, HW, plane);
  return cp_launch_status();
}

int cp_activation_unsplit(const void* in, float* x, int32_t B, int32_t C, int32_t H, int32_t W, void* stream) {
  CP_CHECK_ARG(x && in && B >= 1 && C >= 8 && H >= 1 && W >= 1);
  if (C % 8 != 0) return CP_EUNSUPPORTED;
  const long long HW = (long long)H * W, plane = (long long)B * (C / 8) * HW;
  if (plane >= 0x7FFFFFFFll * 256) return CP_EUNSUPPORTED;
  hipLaunchKernelGGL(activation_unsplit_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16x8*)in, x, HW, plane);
  return cp_launch_status();
}

// cp_conv_mfma_forward_strided for ONE source whose input and / or output are split planes (x_split: 3x3 / stride 1
// and Cin % 32 == 0 only; out_split: Cout % 8 == 0, no residual).  The planes hold exactly the halves the float32
// form's staging computes, so the results are bit-identical to the float32 route.
int cp_conv_mfma_forward_split(const void* x, int32_t x_split, const void* wperm, const float* bias, const float* residual,
                               void* out, int32_t out_split, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t Cout,
                               int32_t taps, int32_t stride, int32_t relu, void* stream) {
  const float* xs[1] = {reinterpret_cast<const float*>(x)};
  const int32_t cs[1] = {Cin};
  return conv_forward_impl(xs, cs, 1, wperm, bias, residual, nullptr, nullptr, reinterpret_cast<float*>(out), B, H, W, Cout,
                           taps, stride, relu, stream, nullptr, x_split ? 1 : 0, out_split ? 1 : 0);
}

int cp_conv_mfma_forward_strided(const float* const* xs, const int32_t* cs, int32_t nsrc, const void* wperm,
                                 const float* bias, const float* residual, float* out, int32_t B, int32_t H, int32_t W,
                                 int32_t Cout, int32_t taps, int32_t stride, int32_t relu, void* stream) {
  return conv_forward_impl(xs, cs, nsrc, wperm, bias, residual, nullptr, nullptr, out, B, H, W, Cout, taps, stride, relu,
                           stream);
}

// Input gradient of a stride-1 convolution (3x3 / pad 1 or 1x1) whose INPUT was the output `y` of a bias + ReLU
// epilogue (the heads' Conv2d(3x3, bias) -> ReLU -> Conv2d(1x1), pose_dla_dcn.py:445-462), with that ReLU's backward
// and its bias gradient in the epilogue:
//   grad_y[b][c][p] = [y[b][c][p] > 0] * sum_{co, taps} w[co][c][tap'] * grad_out[b][co][p + tap]
//   grad_bias[c]   += sum_{b, p} grad_y[b][c][p]
// instead of writing the unmasked gradient, re-reading it with y and writing it again in a separate pass.  The bias
// gradient: every 16-lane row leaves its pixels' channel sums in the workspace (plain stores), a second small kernel adds the
// rows up (an atomic per wave and channel straight into grad_bias ran 5x longer than the convolution: 8 192 same-address
// adds per channel at the training size).
// wperm_t: cp_conv_mfma_prepare(w, Cin := Cout of the convolution, Cout := its Cin, taps, transposed = 1).
size_t cp_conv_mfma_input_grad_relu_workspace_bytes(int32_t B, int32_t Cin, int32_t H, int32_t W) {
  if (B < 1 || Cin < 1 || H < 1 || W < 1) return 0;
  return (size_t)B * ((W + TW - 1) / TW) * ((H + 3) / 4) * 8 * Cin * sizeof(float);     // (the narrowest tile form)
}

int cp_conv_mfma_input_grad_relu(const float* grad_out, const void* wperm_t, const float* y, float* grad_y,
                                 float* grad_bias, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t Cout,
                                 int32_t taps, void* workspace, size_t workspace_bytes, void* stream) {
  CP_CHECK_ARG(grad_out && y && grad_y && (!grad_bias || workspace));
  if (grad_bias && workspace_bytes < cp_conv_mfma_input_grad_relu_workspace_bytes(B, Cin, H, W)) return CP_EWORKSPACE;
  const float* xs[1] = {grad_out};
  const int32_t cs[1] = {Cout};
  int tiles = 0;
  const int rc = conv_forward_impl(xs, cs, 1, wperm_t, nullptr, nullptr, y, grad_bias ? (float*)workspace : nullptr, grad_y,
                                   B, H, W, Cin, taps, 1, 0, stream, &tiles);
  if (rc != CP_OK || !grad_bias) return rc;
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((Cin + 63) / 64, 32), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, B * tiles * 8, Cin, grad_bias);
  return cp_launch_status();
}

// INPUT GRADIENT of a 3x3 / stride 2 / pad 1 convolution (the first convolution of DLA levels 2-5, pose_dla_dcn.py:32-40;
// what the reference gets from cuDNN's backward-data):
//   grad_in[b][ci][2 i + py][2 j + px] = sum_{co} sum_{(ky, kx) in class (py, px)} w[co][ci][ky][kx] * grad_out[b][co][i + dy(ky)][j + dx(kx)]
// -- per parity class (py, px) of the gradient's rows / columns a stride-1 convolution of grad_out with 1, 2, 2 or 4 of
// the nine taps (no multiplications by inserted zeros: the four classes together do exactly the forward's work).
// ONE launch: grad_out is staged once per workgroup and feeds the four
// parity classes' accumulators, tap by tap (kernel template IG2); the two column classes leave as one float2 per pixel.
//   grad_in = (residual ? residual : 0) + conv_transpose2d(grad_out, w, stride 2, pad 1)   cropped to H x W
// wperm_t: cp_conv_mfma_prepare(w, Cin := Cout of the convolution, Cout := its Cin, taps 9, transposed = 6).
int cp_conv3x3_s2_input_grad(const float* grad_out, const void* wperm_t, const float* residual, float* grad_in, int32_t B,
                             int32_t Cin, int32_t H, int32_t W, int32_t Cout, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  CP_CHECK_ARG(grad_out && wperm_t && grad_in && B >= 1 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1);
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  if (!cp_conv3x3_mfma_supported(Cout, Cin, Ho, Wo) || (long long)Cin * H * W * 4 >= 0x7FFFFFF0ll) return CP_EUNSUPPORTED;
  CvArgs a;
  for (int i = 0; i < MAXSRC; ++i) {
    a.xsrc[i] = i == 0 ? grad_out : nullptr;
    a.csrc[i] = i == 0 ? Cout : 0;
  }
  a.wp = (const bf16x8*)wperm_t;
  a.bias = nullptr; a.res = residual; a.out = grad_in;
  a.Cin = Cout; a.H = Ho; a.W = Wo; a.Cout = Cin;
  a.Ho = Ho; a.Wo = Wo;                                      // the launch's grid: one point per grad_out pixel
  a.nchunk = (Cout + KC - 1) / KC;
  a.tiles_x = (Wo + TW - 1) / TW;
  a.relu = 0;
  a.Hf = H; a.Wf = W;
  a.mask = nullptr; a.colsum = nullptr;
  auto wgs = [&](int mt, int th) { return (long long)a.tiles_x * ((Ho + th - 1) / th) * B * ((Cin + 16 * mt - 1) / (16 * mt)); };
  auto launch = [&](auto kernel, int mt, int th) {
    a.ncot = (Cin + 16 * mt - 1) / (16 * mt);
    hipLaunchKernelGGL(kernel, dim3(a.tiles_x * ((Ho + th - 1) / th) * a.ncot, B), dim3(256), 0, st, a);
  };
  if (Cin > 32 && wgs(4, 4) >= 448) launch(conv_mfma_kernel<4, 1, 9, 1, 1, true>, 4, 4);
  else if (wgs(2, 8) >= 320) launch(conv_mfma_kernel<2, 2, 9, 1, 1, true>, 2, 8);
  else launch(conv_mfma_kernel<2, 1, 9, 1, 1, true>, 2, 4);
  return cp_launch_status();
}

int cp_conv_mfma_forward(const float* const* xs, const int32_t* cs, int32_t nsrc, const void* wperm, const float* bias,
                         const float* residual, float* out, int32_t B, int32_t H, int32_t W, int32_t Cout,
                         int32_t taps, int32_t relu, void* stream) {
  return cp_conv_mfma_forward_strided(xs, cs, nsrc, wperm, bias, residual, out, B, H, W, Cout, taps, 1, relu, stream);
}

int cp_conv3x3_mfma_forward(const float* x, const void* wperm, const float* bias, const float* residual, float* out,
                            int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t relu, void* stream) {
  CP_CHECK_ARG(x);
  if (!cp_conv3x3_mfma_supported(Cin, Cout, H, W)) return CP_EUNSUPPORTED;
  const float* xs[1] = {x};
  const int32_t cs[1] = {Cin};
  return cp_conv_mfma_forward(xs, cs, 1, wperm, bias, residual, out, B, H, W, Cout, 9, relu, stream);
}

}  // extern "C"

// =====================================================================================================================
// Weight gradient of the same convolution, same arithmetic (split-bf16 x3 on the bf16 matrix cores):
//   gw[co][ci][dy][dx] += sum_{b, y, x} go[b][co][y][x] * in[b][ci][y + dy - 1][x + dx - 1]
// GEMM view: D[co][ci] (one per tap) += A[co][px] * B[px][ci], contraction over PIXELS, k-step = 32 pixels of one row.
//   * workgroup = (a run of 4 x 32 pixel tiles, 64 input channels, 32 MTW output channels), 8 waves; wave w owns
//     input-channel fragment w & 3 and MTW output-channel fragments, all 9 taps: 9 * MTW accumulator tiles.
//   * per pixel tile both operands are loaded as float4 (prefetched a tile ahead into registers), split once and stored
//     as bf16 hi / lo planes: go as [co][4 rows x 32 px] (pitch 136), in as [ci][6 rows][48 px] (pitch 296, the tile's
//     column 0 at element 8) -- both pitches are an odd number of 16-byte units, so the 16 lanes of a fragment read
//     hit 16 different bank groups.
//   * an A fragment (8 consecutive pixels of a go row) is one ds_read_b128 per half.  A B fragment of the centre column
//     tap is one ds_read_b128 too; the dx = -1 / +1 taps are the same 8 pixels shifted by one bf16: built in registers
//     from the centre fragment and one neighbouring dword each side with five v_alignbit per half (VALU work that
//     runs beside the matrix cores), instead of unaligned LDS reads or three shifted copies.
//   * flush: accumulators -> LDS [32 co][64 ci x 9] -> one coalesced float atomicAdd per element (each output row
//     of a workgroup is 576 contiguous floats of gw).
// =====================================================================================================================
namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct WgArgs {
  const float* x;
  const float* go;
  float* gw;
  int Cin, Cout, H, W, tiles_x, tiles_y, ntiles, tiles_per_wg, n_ci;
};

constexpr int WG_GP = 4 * 32 + 8;          // go pitch per output channel (bf16 elements)
constexpr int WG_XR = 48;                  // in: row pitch
constexpr int WG_XP = 6 * WG_XR + 8;       // in: channel pitch
constexpr int WG_CI = 64;

__device__ __forceinline__ void split4(const f32x4 v, u32x2& hi, u32x2& lo) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 h0, h1, l0, l1;
  h0[0] = (__bf16)v.x;
  h0[1] = (__bf16)v.y;
  h1[0] = (__bf16)v.z;
  h1[1] = (__bf16)v.w;
  l0[0] = (__bf16)(v.x - (float)h0[0]);
  l0[1] = (__bf16)(v.y - (float)h0[1]);
  l1[0] = (__bf16)(v.z - (float)h1[0]);
  l1[1] = (__bf16)(v.w - (float)h1[1]);
  hi = u32x2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
  lo = u32x2{__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1)};
}

// TAPS = 1: the 1x1 convolution's weight gradient -- the same tiles and staging, only the centre tap contracted.
template <int MTW, int TAPS>
__global__ __launch_bounds__(512, 2) void conv3x3_wgrad_kernel(WgArgs a) {
  constexpr int CO = 32 * MTW;
  constexpr int G_PLANE = CO * WG_GP, X_PLANE = WG_CI * WG_XP;                 // elements per half
  // staged input per channel: 6 rows x 10 float4 (halo for the 3x3 taps); the 1x1 form needs only the tile's own 4 x 8
  constexpr int XROWS = TAPS == 9 ? 6 : 4, XQ = TAPS == 9 ? 10 : 8;
  constexpr int G_ITERS = CO * 32 / 512, X_UNITS = WG_CI * XROWS * XQ, X_ITERS = (X_UNITS + 511) / 512;
  constexpr int STAGE_BYTES = (2 * G_PLANE + 2 * X_PLANE) * 2;
  constexpr int OP = WG_CI * TAPS + 1, OUT_BYTES = 32 * OP * 4;
  constexpr int SMEM = STAGE_BYTES > OUT_BYTES ? STAGE_BYTES : OUT_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  unsigned short* Gs = reinterpret_cast<unsigned short*>(smem);                // [2][G_PLANE]
  unsigned short* Xs = Gs + 2 * G_PLANE;                                       // [2][X_PLANE]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, c = lane & 15;
  const int nt = wid & 3, mtb = (wid >> 2) * MTW;
  const int ci0 = (blockIdx.y % a.n_ci) * WG_CI, co0 = (blockIdx.y / a.n_ci) * CO;
  const int HW = a.H * a.W;
  const int k_begin = blockIdx.x * a.tiles_per_wg;
  const int k_end = min(a.ntiles, k_begin + a.tiles_per_wg);

  f32x4 acc[TAPS][MTW];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int m = 0; m < MTW; ++m) acc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging units of this thread (tile-independent parts)
  int g_lds[G_ITERS], g_rc[G_ITERS];          // LDS element; packed (co, row, col)
#pragma unroll
  for (int i = 0; i < G_ITERS; ++i) {
    const int u = tid + i * 512, q4 = u & 7, r = (u >> 3) & 3, co = u >> 5;
    g_lds[i] = co * WG_GP + r * 32 + 4 * q4;
    g_rc[i] = (co << 16) | (r << 8) | (4 * q4);
  }
  int x_lds[X_ITERS], x_rc[X_ITERS];
#pragma unroll
  for (int i = 0; i < X_ITERS; ++i) {
    const int u = tid + i * 512;
    const int q = u % XQ + (TAPS == 9 ? 0 : 1), rr = (u / XQ) % XROWS + (TAPS == 9 ? 0 : 1), ci = u / (XROWS * XQ);
    x_lds[i] = u < X_UNITS ? ci * WG_XP + rr * WG_XR + 4 + 4 * q : -1;
    x_rc[i] = (ci << 16) | (rr << 8) | (4 * q);
  }

  f32x4 gv[G_ITERS], xv[X_ITERS];
  auto load_tile = [&](int k) __attribute__((always_inline)) {
    const int tx = k % a.tiles_x, ty = (k / a.tiles_x) % a.tiles_y, b = k / (a.tiles_x * a.tiles_y);
    const int x0 = tx * 32, y0 = ty * 4;
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.go + (long long)b * a.Cout * HW), 0, (int)((unsigned)a.Cout * (unsigned)HW * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x + (long long)b * a.Cin * HW), 0, (int)((unsigned)a.Cin * (unsigned)HW * 4u), 0x00020000);
#pragma unroll
    for (int i = 0; i < G_ITERS; ++i) {
      const int co = co0 + (g_rc[i] >> 16), y = y0 + ((g_rc[i] >> 8) & 255), x = x0 + (g_rc[i] & 255);
      const bool ok = co < a.Cout && y < a.H && x < a.W;
      const unsigned off = ok ? ((unsigned)co * (unsigned)HW + (unsigned)(y * a.W + x)) * 4u : OOB;
      gv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, off, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < X_ITERS; ++i) {
      const int ci = ci0 + (x_rc[i] >> 16), y = y0 - 1 + ((x_rc[i] >> 8) & 255), x = x0 - 4 + (x_rc[i] & 255);
      const bool ok = x_lds[i] >= 0 && ci < a.Cin && y >= 0 && y < a.H && x >= 0 && x < a.W;
      const unsigned off = ok ? ((unsigned)ci * (unsigned)HW + (unsigned)(y * a.W + x)) * 4u : OOB;
      xv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
    }
  };

  if (k_begin < k_end) load_tile(k_begin);

  // fragment bases of this lane
  const int a_base = c * WG_GP + 8 * g;                              // + (mtb + m) * 16 * GP + r * 32
  const int b_base = (nt * 16 + c) * WG_XP + 8 + 8 * g;              // + (r + dy) * XR

  for (int k = k_begin; k < k_end; ++k) {
    __syncthreads();                                                 // the previous tile's fragments have been read
#pragma unroll
    for (int i = 0; i < G_ITERS; ++i) {
      u32x2 hi, lo;
      split4(gv[i], hi, lo);
      *reinterpret_cast<u32x2*>(&Gs[g_lds[i]]) = hi;
      *reinterpret_cast<u32x2*>(&Gs[G_PLANE + g_lds[i]]) = lo;
    }
#pragma unroll
    for (int i = 0; i < X_ITERS; ++i) {
      if (x_lds[i] >= 0) {
        u32x2 hi, lo;
        split4(xv[i], hi, lo);
        *reinterpret_cast<u32x2*>(&Xs[x_lds[i]]) = hi;
        *reinterpret_cast<u32x2*>(&Xs[X_PLANE + x_lds[i]]) = lo;
      }
    }
    __syncthreads();
    if (k + 1 < k_end) load_tile(k + 1);                             // in flight during the matrix phase

#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bf16x8 ah[MTW], al[MTW];
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
        const int e = a_base + (mtb + m) * 16 * WG_GP + r * 32;
        ah[m] = *reinterpret_cast<const bf16x8*>(&Gs[e]);
        al[m] = *reinterpret_cast<const bf16x8*>(&Gs[G_PLANE + e]);
      }
#pragma unroll
      for (int dy = (TAPS == 9 ? 0 : 1); dy < (TAPS == 9 ? 3 : 2); ++dy) {
        const int e = b_base + (r + dy) * WG_XR;
        u32x4 bh[3], bl[3];                                          // dx = 0 (x - 1), 1 (x), 2 (x + 1)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) {
          const unsigned short* pl = Xs + hl * X_PLANE + e;
          const u32x4 q = *reinterpret_cast<const u32x4*>(pl);
          const unsigned pL = *reinterpret_cast<const unsigned*>(pl - 2);
          const unsigned pR = *reinterpret_cast<const unsigned*>(pl + 8);
          const unsigned s01 = __builtin_amdgcn_alignbit(q[1], q[0], 16), s12 = __builtin_amdgcn_alignbit(q[2], q[1], 16),
                         s23 = __builtin_amdgcn_alignbit(q[3], q[2], 16);
          u32x4* dst = hl ? bl : bh;
          dst[0] = u32x4{__builtin_amdgcn_alignbit(q[0], pL, 16), s01, s12, s23};
          dst[1] = q;
          dst[2] = u32x4{s01, s12, s23, __builtin_amdgcn_alignbit(pR, q[3], 16)};
        }
#pragma unroll
        for (int dx = (TAPS == 9 ? 0 : 1); dx < (TAPS == 9 ? 3 : 2); ++dx) {
          const bf16x8 fh = __builtin_bit_cast(bf16x8, bh[dx]), fl = __builtin_bit_cast(bf16x8, bl[dx]);
#pragma unroll
          for (int m = 0; m < MTW; ++m) {
            f32x4 v = acc[TAPS == 9 ? dy * 3 + dx : 0][m];
            v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], fh, v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], fl, v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m], fh, v, 0, 0, 0);
            acc[TAPS == 9 ? dy * 3 + dx : 0][m] = v;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- flush: 32 output channels per pass through LDS, then coalesced atomics ----
  float* O = reinterpret_cast<float*>(smem);
  const int ciw = min(WG_CI, a.Cin - ci0) * TAPS;
#pragma unroll
  for (int pass = 0; pass < CO / 32; ++pass) {
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      const int row0 = (mtb + m) * 16 + 4 * g;                       // rows row0 .. row0 + 3 lie in one pass
      if (row0 / 32 == pass) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) O[(row0 % 32 + r) * OP + (nt * 16 + c) * TAPS + t] = acc[t][m][r];
      }
    }
    __syncthreads();
    for (int idx = tid; idx < 32 * ciw; idx += 512) {
      const int row = idx / ciw, col = idx - row * ciw;
      const int co = co0 + pass * 32 + row;
      if (co < a.Cout) atomicAdd(&a.gw[((long long)co * a.Cin + ci0) * TAPS + col], O[row * OP + col]);
    }
  }
}

}  // namespace

extern "C" {

int cp_conv3x3_mfma_wgrad_supported(int32_t Cin, int32_t Cout, int32_t H, int32_t W) {
  return Cin >= 1 && Cout >= 1 && H >= 1 && W >= 4 && W % 4 == 0 && (long long)Cin * H * W * 4 < 0x7FFFFFF0ll &&
         (long long)Cout * H * W * 4 < 0x7FFFFFF0ll;
}

// gw [Cout][Cin][k][k] += the weight gradient, k*k = taps (9: 3x3 / pad 1; 1: 1x1); the caller zeroes gw or carries
// an accumulation.
int cp_conv_mfma_wgrad(const float* x, const float* go, float* gw, int32_t B, int32_t Cin, int32_t H, int32_t W,
                       int32_t Cout, int32_t taps, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  CP_CHECK_ARG(x && go && gw && B >= 1 && (taps == 1 || taps == 9));
  if (!cp_conv3x3_mfma_wgrad_supported(Cin, Cout, H, W)) return CP_EUNSUPPORTED;
  WgArgs a;
  a.x = x;
  a.go = go;
  a.gw = gw;
  a.Cin = Cin;
  a.Cout = Cout;
  a.H = H;
  a.W = W;
  a.tiles_x = (W + 31) / 32;
  a.tiles_y = (H + 3) / 4;
  a.ntiles = a.tiles_x * a.tiles_y * B;
  a.n_ci = (Cin + WG_CI - 1) / WG_CI;
  const int MTW = Cout <= 32 ? 1 : 2;
  const int n_co = (Cout + 32 * MTW - 1) / (32 * MTW);
  const int pairs = a.n_ci * n_co;
  // ~1 workgroup (8 waves) per CU: every workgroup adds its whole partial gradient to gw with float atomics at the end, so
  // the atomics scale with the workgroup count -- 512 workgroups (two per CU) ran the same contraction 14-18 % SLOWER
  // (4 x 128 -> 128 @128x256: 179 vs 150 us; tools/probe history in DESIGN 4.13), the second workgroup bought nothing
  int nsplit = (256 + pairs - 1) / pairs;
  if (nsplit > a.ntiles) nsplit = a.ntiles;
  a.tiles_per_wg = (a.ntiles + nsplit - 1) / nsplit;
  nsplit = (a.ntiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
  const dim3 grid(nsplit, pairs);
  if (taps == 9) {
    if (MTW == 2) hipLaunchKernelGGL((conv3x3_wgrad_kernel<2, 9>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_wgrad_kernel<1, 9>), grid, dim3(512), 0, st, a);
  } else {
    if (MTW == 2) hipLaunchKernelGGL((conv3x3_wgrad_kernel<2, 1>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_wgrad_kernel<1, 1>), grid, dim3(512), 0, st, a);
  }
  return cp_launch_status();
}

int cp_conv3x3_mfma_wgrad(const float* x, const float* go, float* gw, int32_t B, int32_t Cin, int32_t H, int32_t W,
                          int32_t Cout, void* stream) {
  return cp_conv_mfma_wgrad(x, go, gw, B, Cin, H, W, Cout, 9, stream);
}

}  // extern "C"
