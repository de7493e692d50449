// DCNv2 forward, region form (gfx950): the input tile of a block of output pixels is staged ONCE per 16-channel
// chunk into LDS with coalesced row loads, the bilinear samples are read back from LDS by the lane that needs them
// and go STRAIGHT into bf16 MFMA operand registers -- no per-lane global gathers, no column tile, no f32 MFMA.
//
// Replaces (together with dcn_fwd.hip) the native extension behind `from .DCNv2.dcn_v2 import DCN`
// (reference: src/lib/models/networks/pose_dla_dcn.py:16, call site :354).
//
// Why: the gather kernels of dcn_fwd.hip are bound by the texture addresser (two 8-byte gathers per channel, tap and
// pixel) and, on the 64-channel layers, by the f32 MFMA sharing the vector ALU with the sampling arithmetic.
//
//   * Workgroup = 4 waves = a tile of 8 rows x 32 columns of one image x 64 output channels.  Wave w owns rows
//     2w, 2w + 1; a row of 32 pixels is the N side of v_mfma_f32_32x32x16_bf16, 32 output channels the M side,
//     16 input channels of one tap the K side: lane (pixel = lane & 31, half = lane >> 5) supplies the 8 channels
//     8 half .. 8 half + 7 of its pixel -- exactly the 8 values it samples.
//   * Region: (8 + 8) x (32 + 8) input cells around the tile, channel-interleaved in groups of four
//     ([group][row][col] float4), so a corner of 4 channels is ONE ds_read_b128 and the x neighbour is the next
//     16 bytes.  Cells outside the image are staged as zeros (buffer loads past num_records), which is DCNv2's
//     per-corner bounds rule.  The region origin follows the tile's sampling positions: a block-wide min / max of
//     the integer corners picks the window, so a smooth offset field of any size costs nothing; samples that still
//     fall outside (white-noise offsets beyond +-2 rows / +-3 columns) take a wave-uniformly skipped cold path with
//     global gathers.  Two chunk buffers: the next chunk is staged (buffer loads -> ds_write_b128, spread over the
//     taps) while the current one is sampled; one barrier per chunk.
//   * Per (row, tap, chunk) step a lane reads 8 x 16 bytes, forms 8 bilinear values (fp32 fma chain), splits each
//     into bf16 hi + lo and feeds 2 x 3 MFMAs (hi*hi + hi*lo + lo*hi, fp32 accumulate: ~2^-16 relative, an fp32
//     emulation).  The weights come pre-split in A-fragment order (dcn_region_wperm_kernel) straight from global
//     memory (L1 / L2 resident, 4 KB per tap and chunk, shared by both rows), one tap ahead.
//   * Per-pixel sampling recipes (ly, lx, mask, region byte offset) of the wave's 2 x 9 (row, tap) pairs stay in
//     registers for the whole K loop.
// LDS: 2 x 40 KB = exactly half a CU's 160 KB, two workgroups per CU (8 waves, 256 registers each).
#include "cp_common.h"
#include "dcn_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 32;            // output tile
constexpr int RH = 16, RW = 40;           // staged region
constexpr int CELLS = RH * RW;            // 640
constexpr int PLANE = CELLS * 16;         // bytes of one 4-channel group
constexpr int CHB = 4 * PLANE;            // bytes of one 16-channel chunk buffer (40 960)
constexpr int TAPS = 9;
constexpr int ITEMS = CELLS * 4 / 256;    // float4 cells staged per thread and chunk (10)
// timing-only ablation builds (make libcp_rabl_<mask>.so, tools/probe_region_ablate.py; results are wrong, never shipped):
// 1 no weight re-loads, 2 no staging of later chunks, 4 no LDS sample reads, 8 no MFMA, 16 no bilinear / split arithmetic
#ifndef CP_RABL
#define CP_RABL 0
#endif
#ifndef CP_RPRIO
#define CP_RPRIO 0      // per-chunk priority alternation between the workgroups sharing a CU: OFF since round 4 (A/B on the
                        // final round-3 kernel: 50.7 us with, 50.8 without; libcp_rprio.so builds it with the alternation ON)
#endif
constexpr int RABL = CP_RABL;
// diagnostic build (make libcp_rstamp.so, tools/probe_region_stamp.py): wave 0 of every workgroup overwrites 8 floats of
// out[] with s_memtime deltas of its phases (the results are destroyed; never shipped)
#ifdef CP_RSTAMP
#define RSTAMP(i) do { stamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RSTAMP(i) do { } while (0)
#endif
constexpr unsigned OOB = 0x80000000u;     // voffset past every num_records (< 2^31, also with the scalar offset added): the load returns 0
static_assert(CELLS * 4 % 256 == 0, "staging items divide evenly");

struct RegionArgs {
  const float* x;
  const float* offset;
  const float* mask;
  const bf16x8* wp;
  const float* bias;
  const float* ep_scale;
  const float* ep_shift;
  float* out;
  long long offset_bstride, mask_bstride;
  int B, Cin, H, W, Cout;
  int mask_is_logit, relu;
  int tiles_x, tiles_y;
  // fused conv_offset_mask (kernel template FUSE): its weights in fragment order, its bias, optional copy-out of the
  // 27 channels it computes ([B][27][H][W]: the training backward reads them)
  const bf16x8* om_wp;
  const float* om_bias;
  float* om_out;
  // K split (deep, small maps whose tiles do not fill the chip): grid z = B * ksplit, slice z / B takes the input-channel
  // chunks [slice * chunks_per_split, ...) and writes its RAW partial sums to partial[slice][B][Cout][H][W]; the caller
  // adds them up and applies bias / folded BN / ReLU (dcn_splitk_reduce_kernel of dcn_fwd.hip).  ksplit = 1: off.
  float* partial;
  int ksplit, chunks_per_split;
};

// wp[(((cb * nchunk + chunk) * 9 + t) * 4 + ct * 2 + hl) * 64 + lane][j] =
//     half hl of W[co = 64 cb + 32 ct + (lane & 31)][ci = 16 chunk + 8 (lane >> 5) + j][t]   (0 for co >= Cout)
__global__ __launch_bounds__(256) void dcn_region_wperm_kernel(const float* __restrict__ w, bf16x8* __restrict__ wp,
                                                               int Cout, int Cin, int nchunk, int total) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int lane = e & 63;
  int r = e >> 6;
  const int hl = r & 1, ct = (r >> 1) & 1;
  r >>= 2;
  const int t = r % TAPS;
  r /= TAPS;
  const int chunk = r % nchunk, cb = r / nchunk;
  const int co = cb * 64 + ct * 32 + (lane & 31);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ci = chunk * 16 + 8 * (lane >> 5) + j;
    const float v = (co < Cout && ci < Cin) ? w[((long long)co * Cin + ci) * TAPS + t] : 0.f;
    const __bf16 h = (__bf16)v;
    o[j] = hl ? (__bf16)(v - (float)h) : h;
  }
  wp[e] = o;
}

// conv_offset_mask weights [27][Cin][3][3] -> A fragments (32 rows, rows >= 27 zero):
// wp[((chunk * 9 + t) * 2 + hl) * 64 + lane][j] = half hl of Wom[co = lane & 31][ci = 16 chunk + 8 (lane >> 5) + j][t]
__global__ __launch_bounds__(256) void dcn_region_omperm_kernel(const float* __restrict__ w, bf16x8* __restrict__ wp, int Cin,
                                                                int total) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int lane = e & 63;
  int r = e >> 6;
  const int hl = r & 1;
  r >>= 1;
  const int t = r % TAPS, chunk = r / TAPS;
  const int co = lane & 31;
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ci = chunk * 16 + 8 * (lane >> 5) + j;
    const float v = (co < 27 && ci < Cin) ? w[((long long)co * Cin + ci) * TAPS + t] : 0.f;
    const __bf16 h = (__bf16)v;
    o[j] = hl ? (__bf16)(v - (float)h) : h;
  }
  wp[e] = o;
}

__device__ __forceinline__ unsigned pk_min(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}

// fp32 pair -> bf16 hi pair + bf16 lo pair (v = hi + lo up to 2^-17 relative)
__device__ __forceinline__ void split2(float v0, float v1, unsigned& hi, unsigned& lo) {
  const bf16x2 h = __builtin_convertvector(f32x2{v0, v1}, bf16x2);
  const unsigned hb = __builtin_bit_cast(unsigned, h);
  const float h0 = __builtin_bit_cast(float, hb << 16), h1 = __builtin_bit_cast(float, hb & 0xffff0000u);
  const bf16x2 l = __builtin_convertvector(f32x2{v0 - h0, v1 - h1}, bf16x2);
  hi = hb;
  lo = __builtin_bit_cast(unsigned, l);
}

// Value held by lane half 0 -> r0 in every lane, value held by lane half 1 -> r1 in every lane (one v_permlane32_swap:
// lanes 32-63 of the first operand trade places with lanes 0-31 of the second).  The empty asm keeps hipcc (ROCm 7.2)
// from folding the two results into one when they are bit-cast to float (it then stores r0 twice).
__device__ __forceinline__ void bcast_rows(unsigned u, unsigned& r0, unsigned& r1) {
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  r0 = r[0];
  r1 = r[1];
  asm("" : "+v"(r0), "+v"(r1));
}

// FUSE: the 27-channel conv_offset_mask (3x3, pad 1, + bias; DCNv2/dcn_v2.py's `self.conv_offset_mask(x)`, reference call
// site pose_dla_dcn.py:354) is computed HERE for the workgroup's own tile, as a first pass over the input chunks on the
// same matrix cores (split-bf16 x3, the arithmetic of conv_mfma.hip), instead of by a separate launch whose 27-channel
// output is written to and read back from memory: the tile +-1 halo is staged pre-split (bf16 hi | lo, 8 channels =
// one B fragment per cell), 9 taps x 2 rows x 3 MFMAs per 16-channel chunk, the result lands in accumulator registers,
// one v_permlane32_swap per register hands every lane half its row's 27 values, and the recipes are built from them.
template <bool FUSE>
__global__ __launch_bounds__(256, 2) void dcn_fwd_region_kernel(RegionArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 31, kg = lane >> 5;
  const int H = a.H, W = a.W, HW = H * W;
#ifdef CP_RSTAMP
  unsigned long long stamp[8];
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  RSTAMP(0);

  // XCD-aware tile order: blocks b and b + 8 share an XCD (and its L2); give each XCD a contiguous band of tiles
  const int ntile = a.tiles_x * a.tiles_y;
  int tile = blockIdx.x;
  if ((ntile & 7) == 0) tile = (tile & 7) * (ntile >> 3) + (tile >> 3);
  const int ty = (tile / a.tiles_x) * TH, tx = (tile % a.tiles_x) * TW;
  const int b = (int)blockIdx.z % a.B, ksl = (int)blockIdx.z / a.B, cb = blockIdx.y;
  const int nchunk_all = a.Cin >> 4;
  const int c_begin = FUSE ? 0 : ksl * a.chunks_per_split;                     // (the fused form never splits)
  const int c_end = FUSE ? nchunk_all : min(nchunk_all, c_begin + a.chunks_per_split);
  const int nchunk = nchunk_all;                                               // (weight layout / the fused offset conv)
  const unsigned plane_bytes = (unsigned)HW * 4u;

  // Buffer descriptors span the whole tensor of the image; a channel plane rides in the scalar offset.  (Measured on
  // gfx950: the range check is on voffset + soffset, so num_records must cover the planes; OOB = 2^31 fails it under
  // either rule and the load returns 0.)
  const __amdgpu_buffer_rsrc_t rs_off = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.offset + (long long)b * a.offset_bstride), 0, (int)(18u * plane_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_msk = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.mask + (long long)b * a.mask_bstride), 0, (int)(9u * plane_bytes), 0x00020000);
  // x: voffset carries (group plane + pixel), soffset the chunk's channel plane
  const float* xb = a.x + (long long)b * a.Cin * HW;
  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (int)((unsigned)a.Cin * plane_bytes), 0x00020000);

  // ------------------------------------------------------------------ sampling recipes
  // Lane half h computes and KEEPS the 9 recipes of row 2w + h of its pixel column (both halves of a lane pair need
  // the recipe of the row being processed: it is broadcast per tap with one v_permlane32_swap per value, below):
  //   u0 = (1 - ly) * mask, u1 = ly * mask, lx, region byte offset (bit 31: cold sample, a global pixel index).
  float ru0[TAPS], ru1[TAPS], rlx[TAPS];
  unsigned roff[TAPS];                    // first (y0 + 1) << 16 | (x0 + 1), later the region byte offset
  unsigned bb_lo = 0xFFFFFFFFu, bb_hi = 0u;
  {
    const int y = ty + 2 * wid + kg, x = tx + px;
    const bool ok = y < H && x < W;
    const unsigned vo = ok ? (unsigned)(y * W + x) * 4u : OOB;
    float oy[TAPS], ox[TAPS], ml[TAPS];
    if constexpr (FUSE) {
      // ---- conv_offset_mask of this tile: pre-split staging tile [buffer][hi | lo][lane half][10 x 34 cells] x 16 B
      constexpr int OW = TW + 2, OCELLS = (TH + 2) * OW, OPL = OCELLS * 16, OBUF = 4 * OPL;     // 340 cells, 21 760 B
      constexpr int OITEMS = (2 * OCELLS + 255) / 256;                                              // (cell, half) per thread: 3
      static_assert(2 * OBUF <= 2 * CHB - 64, "staging tiles of the offset convolution fit below the bounding-box words");
      unsigned ovo[OITEMS], olds[OITEMS];
#pragma unroll
      for (int i = 0; i < OITEMS; ++i) {
        const int e = tid + 256 * i;
        const int hf = e / OCELLS, cell = e - hf * OCELLS;
        const int ry = cell / OW, rx = cell - ry * OW;
        const int gy = ty - 1 + ry, gx = tx - 1 + rx;
        const bool in = e < 2 * OCELLS && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        ovo[i] = in ? ((unsigned)(hf * 8) * (unsigned)HW + (unsigned)(gy * W + gx)) * 4u : OOB;
        olds[i] = e < 2 * OCELLS ? (unsigned)(hf * OPL + cell * 16) : 0xFFFFFFFFu;
      }
      auto oload = [&](int i, int c0, float (&v)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, ovo[i], (unsigned)(c0 + j) * plane_bytes, 0));
      };
      auto owrite = [&](int i, unsigned buf, const float (&v)[8]) __attribute__((always_inline)) {
        if (olds[i] == 0xFFFFFFFFu) return;
        unsigned hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split2(v[2 * j], v[2 * j + 1], hi[j], lo[j]);
        *reinterpret_cast<u32x4*>(smem + buf + olds[i]) = u32x4{hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<u32x4*>(smem + buf + 2 * OPL + olds[i]) = u32x4{lo[0], lo[1], lo[2], lo[3]};
      };
      f32x16 aom[2];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) aom[r][i] = 0.f;
      const bf16x8* oq = a.om_wp + lane;
      bf16x8 of[2] = {oq[0], oq[64]};
      {
        float v0[OITEMS][8];
#pragma unroll
        for (int i = 0; i < OITEMS; ++i) oload(i, 0, v0[i]);
#pragma unroll
        for (int i = 0; i < OITEMS; ++i) owrite(i, 0u, v0[i]);
      }
      __syncthreads();
      const unsigned obase = (unsigned)kg * OPL + (unsigned)((2 * wid) * OW + px) * 16u;
      for (int c = 0; c < nchunk; ++c) {
        const unsigned cur = (unsigned)(c & 1) * OBUF, nxt = OBUF - cur;
        const bool more = c + 1 < nchunk;
        float sv[8];                          // next chunk's tile: item t / 3 is loaded at tap 3 i and written at tap 3 i + 2
        // B fragments (hi, lo) of both rows for one tap, read one tap ahead of the MFMAs that use them
        auto bfrag = [&](int t, bf16x8 (&f)[2][2]) __attribute__((always_inline)) {
          const int ky = t / 3, kx = t - ky * 3;
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const unsigned ad = cur + obase + (unsigned)(((r + ky) * OW + kx) * 16);
            f[r][0] = *reinterpret_cast<const bf16x8*>(smem + ad);
            f[r][1] = *reinterpret_cast<const bf16x8*>(smem + ad + 2 * OPL);
          }
        };
        bf16x8 bc[2][2];
        bfrag(0, bc);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          bf16x8 on[2], bn[2][2];
          {
            const int nt = (t + 1 < TAPS) ? c * TAPS + t + 1 : (more ? (c + 1) * TAPS : c * TAPS);
            on[0] = oq[(long long)nt * 128];
            on[1] = oq[(long long)nt * 128 + 64];
          }
          if (t + 1 < TAPS) bfrag(t + 1, bn);
          static_assert(OITEMS <= 3, "three staging slots per chunk");
          if (more && t % 3 == 0 && t / 3 < OITEMS) oload(t / 3, (c + 1) * 16, sv);
          if (more && t % 3 == 2 && t / 3 < OITEMS) owrite(t / 3, nxt, sv);
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            aom[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of[0], bc[r][0], aom[r], 0, 0, 0);
            aom[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of[0], bc[r][1], aom[r], 0, 0, 0);
            aom[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of[1], bc[r][0], aom[r], 0, 0, 0);
          }
          of[0] = on[0];
          of[1] = on[1];
          if (t + 1 < TAPS) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              bc[r][0] = bn[r][0];
              bc[r][1] = bn[r][1];
            }
          }
        }
        __syncthreads();                      // next chunk's tile written, this one no longer read
      }
      // + bias (this lane's channels: (i & 3) + 8 (i >> 2) + 4 kg), then one swap per register: the lower lane half
      // ends up with all 32 rows of tile row 2w, the upper half with those of row 2w + 1 --
      //   channel co of this lane's row = (co >> 2) & 1 ? qv[i] : pv[i],  i = (co & 3) + 4 (co >> 3)
      const __amdgpu_buffer_rsrc_t rs_ob = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.om_bias), 0, 27 * 4, 0x00020000);
      float pv[16], qv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                       rs_ob, (unsigned)kg * 16u + (unsigned)((i & 3) + 8 * (i >> 2)) * 4u, 0, 0));
        unsigned p0 = __builtin_bit_cast(unsigned, aom[0][i] + bv), q0 = __builtin_bit_cast(unsigned, aom[1][i] + bv);
        const auto sw = __builtin_amdgcn_permlane32_swap(p0, q0, false, false);
        p0 = sw[0];
        q0 = sw[1];
        asm("" : "+v"(p0), "+v"(q0));
        pv[i] = __builtin_bit_cast(float, p0);
        qv[i] = __builtin_bit_cast(float, q0);
      }
      auto chan = [&](int co) __attribute__((always_inline)) -> float {
        const int i = (co & 3) + 4 * (co >> 3);
        return ((co >> 2) & 1) ? qv[i] : pv[i];
      };
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        oy[t] = chan(2 * t);
        ox[t] = chan(2 * t + 1);
        ml[t] = chan(18 + t);
      }
      if (a.om_out && ok) {                   // the training backward reads the 27 channels
        float* oo = a.om_out + (long long)b * 27 * HW + (long long)y * W + x;
#pragma unroll
        for (int co = 0; co < 27; ++co) oo[(long long)co * HW] = chan(co);
      }
    } else {
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        oy[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_off, vo, (unsigned)(2 * t) * plane_bytes, 0));
        ox[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_off, vo, (unsigned)(2 * t + 1) * plane_bytes, 0));
        ml[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_msk, vo, (unsigned)t * plane_bytes, 0));
      }
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int ky = t / 3, kx = t - ky * 3;
      float m = ml[t];
      if (a.mask_is_logit) m = __builtin_amdgcn_rcpf(1.f + __expf(-m));      // (v_rcp_f32: 1 ulp)
      const float py = (float)(y - 1 + ky) + oy[t];
      const float pxf = (float)(x - 1 + kx) + ox[t];
      const bool inside = ok && py > -1.f && pxf > -1.f && py < (float)H && pxf < (float)W;
      const float fy = floorf(py), fx = floorf(pxf);
      const float ly = inside ? py - fy : 0.f;   // (!inside: weights exactly +0, whatever the offsets hold)
      if (!inside) m = 0.f;
      ru0[t] = (1.f - ly) * m;
      ru1[t] = ly * m;
      rlx[t] = inside ? pxf - fx : 0.f;
      const unsigned pk = inside ? (((unsigned)((int)fy + 1)) << 16) | (unsigned)((int)fx + 1) : 0xFFFFFFFFu;
      roff[t] = pk;
      bb_lo = pk_min(bb_lo, pk);
      bb_hi = pk_max(bb_hi, inside ? pk : 0u);
    }
  }
  RSTAMP(1);
  // block-wide bounding box of the integer corners -> region origin
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    bb_lo = pk_min(bb_lo, (unsigned)__shfl_xor((int)bb_lo, o, 64));
    bb_hi = pk_max(bb_hi, (unsigned)__shfl_xor((int)bb_hi, o, 64));
  }
  unsigned* scratch = reinterpret_cast<unsigned*>(smem + 2 * CHB - 64);      // tail of buffer 1: free until chunk 0 runs
  if (lane == 0) {
    scratch[2 * wid] = bb_lo;
    scratch[2 * wid + 1] = bb_hi;
  }
  __syncthreads();
  int oy0, ox0;
  {
    unsigned lo = scratch[0], hi = scratch[1];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      lo = pk_min(lo, scratch[2 * w]);
      hi = pk_max(hi, scratch[2 * w + 1]);
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    const int ymin = (int)(lo >> 16) - 1, ymax = (int)(hi >> 16) - 1;
    const int xmin = (int)(lo & 0xffffu) - 1, xmax = (int)(hi & 0xffffu) - 1;
    const int spany = ymax + 2 - ymin, spanx = xmax + 2 - xmin;             // rows / columns the corners touch
    oy0 = (spany > 0 && spany <= RH) ? ymin - ((RH - spany) >> 1) : ty - 4;  // else: symmetric window, |dy| < 3
    ox0 = (spanx > 0 && spanx <= RW) ? xmin - ((RW - spanx) >> 1) : tx - 4;  //                         |dx| < 3
  }

  // recipes -> region byte offsets.  Samples outside the window are cold: the offset register holds
  // 0x80000000 | (y0' W + x0') and the step gathers the 2 x 2 corners from memory with the recipe's weights.  A cold
  // sample touching the image border (one row or column of its corners outside: zero there, DCNv2's per-corner rule)
  // is re-expressed on the in-image 2 x 2 block next to it: the row / column weight that survives moves into u0 / u1,
  // lx becomes 0 or 1 -- same value, no bounds checks and no second code path in the K loop.
  unsigned long long coldany[TAPS];       // per wave and tap: low word = lanes of row 0, high word = row 1
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    const unsigned pk = roff[t];
    int y0 = (int)(pk >> 16) - 1, x0 = (int)(pk & 0xffffu) - 1;
    const int iy = y0 - oy0, ix = x0 - ox0;
    const bool live = pk != 0xFFFFFFFFu;
    const bool warm = (unsigned)iy <= (unsigned)(RH - 2) && (unsigned)ix <= (unsigned)(RW - 2);
    const bool cold = live && !warm;
    roff[t] = (live && warm) ? (unsigned)(iy * RW + ix) * 16u : 0u;
    if (cold) {
      float u0 = ru0[t], u1 = ru1[t], lx = rlx[t];
      if (y0 < 0) {                         // rows (-1, 0) -> (0, 1): the weight of row 0 moves up
        u0 = u1;
        u1 = 0.f;
        y0 = 0;
      } else if (y0 + 1 >= H) {             // rows (H - 1, H) -> (H - 2, H - 1)
        u1 = u0;
        u0 = 0.f;
        y0 = H - 2;
      }
      if (x0 < 0) {                         // columns (-1, 0) -> (0, 1)
        u0 *= lx;
        u1 *= lx;
        lx = 0.f;
        x0 = 0;
      } else if (x0 + 1 >= W) {             // columns (W - 1, W) -> (W - 2, W - 1)
        u0 *= 1.f - lx;
        u1 *= 1.f - lx;
        lx = 1.f;
        x0 = W - 2;
      }
      ru0[t] = u0;
      ru1[t] = u1;
      rlx[t] = lx;
      roff[t] = 0x80000000u | (unsigned)(y0 * W + x0);
    }
    coldany[t] = __builtin_amdgcn_ballot_w64(cold);
  }

  // staging addresses: this thread's float4 cells e = tid + 256 i; cells i and i + 5 are the same pixel two channel
  // groups (8 planes) apart, so five offsets serve the ten items (the 8 planes ride in the scalar offset)
  unsigned svo[ITEMS / 2];
#pragma unroll
  for (int i = 0; i < ITEMS / 2; ++i) {
    const int e = tid + 256 * i;
    const int g = e / CELLS, cell = e - g * CELLS;
    const int ry = cell / RW, rx = cell - ry * RW;
    const int gy = oy0 + ry, gx = ox0 + rx;
    const bool in = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    svo[i] = in ? ((unsigned)(g * 4) * (unsigned)HW + (unsigned)(gy * W + gx)) * 4u : OOB;
  }
  auto stage_load = [&](int i, int c0, f32x4& v) __attribute__((always_inline)) {
    const int up = i >= ITEMS / 2 ? 8 : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      v[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, svo[i % (ITEMS / 2)],
                                                                          (unsigned)(c0 + q + up) * plane_bytes, 0));
  };
  auto stage_store = [&](int i, unsigned bufbase, const f32x4& v) __attribute__((always_inline)) {
    *reinterpret_cast<f32x4*>(smem + bufbase + (unsigned)(tid + 256 * i) * 16u) = v;
  };

  RSTAMP(2);
  // prologue: chunk 0 -> buffer 0
#pragma unroll
  for (int i0 = 0; i0 < ITEMS; i0 += 5) {
    f32x4 sv[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) stage_load(i0 + i, c_begin * 16, sv[i]);
#pragma unroll
    for (int i = 0; i < 5; ++i) stage_store(i0 + i, 0u, sv[i]);
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[r][ct][i] = 0.f;

  const bf16x8* wq = a.wp + (long long)cb * nchunk * (TAPS * 4 * 64) + lane;
  bf16x8 wf[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) wf[f] = wq[(long long)c_begin * TAPS * (4 * 64) + f * 64];
  __syncthreads();
  RSTAMP(3);

  const unsigned kgoff = (unsigned)kg * (2u * PLANE);
#if CP_RPRIO
  const int prio_half = (int)((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) >> 8);
#endif
  for (int c = c_begin; c < c_end; ++c) {
#if CP_RPRIO
    // The two workgroups of a CU share its SIMDs; VALU issue goes to the older wave, so one of them finishes its K loop
    // ~25 % later than the other and the launch waits for it.  Alternating the priority per chunk between the halves of
    // the grid (workgroups b and b + 256 of the dispatch order are the ones that meet on a CU under round-robin
    // placement; a wrong guess costs nothing) evened them out on an early form of this kernel (53.8 -> 49.8 us); on the
    // final kernel the A/B shows nothing (profiles/r03_dcn_fwd_region_ablations.txt), so it is compiled out by default.
    if (((prio_half) ^ c) & 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
#endif
    const unsigned cur = (unsigned)((c - c_begin) & 1) * CHB, nxt = CHB - cur;
    const unsigned curk = cur + kgoff;
    const bool more = c + 1 < c_end;
    const int cn = (c + 1) * 16;
    f32x4 sv[3];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      // weights of the next tap (or of tap 0 of the next chunk), a whole tap ahead of their first use
      bf16x8 wn[4];
      if (!(RABL & 1)) {
        const bf16x8* qn = wq + (long long)((t + 1 < TAPS) ? c * TAPS + t + 1 : (more ? (c + 1) * TAPS : c * TAPS)) * (4 * 64);
#pragma unroll
        for (int f = 0; f < 4; ++f) wn[f] = qn[f * 64];
      }
      // recipe of tap t for row 0 (x[0]) and row 1 (x[1]) in both lane halves
      unsigned bu0[2], bu1[2], blx[2], bof[2];
      bcast_rows(__builtin_bit_cast(unsigned, ru0[t]), bu0[0], bu0[1]);
      bcast_rows(__builtin_bit_cast(unsigned, ru1[t]), bu1[0], bu1[1]);
      bcast_rows(__builtin_bit_cast(unsigned, rlx[t]), blx[0], blx[1]);
      bcast_rows(roff[t], bof[0], bof[1]);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int s = t * 2 + r;
        // staging of the next chunk, spread over the steps: item s is issued here and written two steps later
        if (more && !(RABL & 2)) {
          if (s >= 2 && s < ITEMS + 2) stage_store(s - 2, nxt, sv[(s - 2) % 3]);
          if (s < ITEMS) stage_load(s, cn, sv[s % 3]);
        }
        // ---- sample 8 channels of (row r, tap t) for this lane's pixel
        const unsigned ad = (unsigned)max((int)bof[r], 0) + curk;   // (cold samples read cell 0)
        f32x4 a00, a01, a10, a11, b00, b01, b10, b11;
        if (RABL & 4) {
          const float z = __builtin_bit_cast(float, ad);
          a00 = a01 = a10 = a11 = f32x4{z, z, z, z};
          b00 = b01 = b10 = b11 = f32x4{z, z, z, z};
        } else {
          a00 = *reinterpret_cast<const f32x4*>(smem + ad);
          a01 = *reinterpret_cast<const f32x4*>(smem + ad + 16);
          a10 = *reinterpret_cast<const f32x4*>(smem + ad + RW * 16);
          a11 = *reinterpret_cast<const f32x4*>(smem + ad + RW * 16 + 16);
          b00 = *reinterpret_cast<const f32x4*>(smem + ad + PLANE);
          b01 = *reinterpret_cast<const f32x4*>(smem + ad + PLANE + 16);
          b10 = *reinterpret_cast<const f32x4*>(smem + ad + PLANE + RW * 16);
          b11 = *reinterpret_cast<const f32x4*>(smem + ad + PLANE + RW * 16 + 16);
        }
        const float u0 = __builtin_bit_cast(float, bu0[r]), u1 = __builtin_bit_cast(float, bu1[r]);
        const float lx = __builtin_bit_cast(float, blx[r]), hx = 1.f - lx;
        const float w00 = u0 * hx, w01 = u0 * lx, w10 = u1 * hx, w11 = u1 * lx;
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (RABL & 16) {
            v[j] = a00[j] + w00;
            v[4 + j] = b00[j] + w11;
            asm volatile("" ::"v"(a01[j]), "v"(a10[j]), "v"(a11[j]), "v"(b01[j]), "v"(b10[j]), "v"(b11[j]), "v"(w01), "v"(w10));
          } else {
            v[j] = w00 * a00[j] + w01 * a01[j] + w10 * a10[j] + w11 * a11[j];
            v[4 + j] = w00 * b00[j] + w01 * b01[j] + w10 * b10[j] + w11 * b11[j];
          }
        }
        if (__builtin_expect((unsigned)(coldany[t] >> (32 * r)) != 0u, 0)) {
          if ((int)bof[r] < 0) {            // cold sample: same weights, corners gathered from memory
            const float* p = xb + (long long)(c * 16 + kg * 8) * HW + (bof[r] & 0x7fffffffu);
            float g[8][4];
#pragma unroll
            for (int j = 0; j < 8; ++j, p += HW) {
              g[j][0] = p[0];
              g[j][1] = p[1];
              g[j][2] = p[W];
              g[j][3] = p[W + 1];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = w00 * g[j][0] + w01 * g[j][1] + w10 * g[j][2] + w11 * g[j][3];
          }
        }
        unsigned hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (RABL & 16) {
            hi[j] = __builtin_bit_cast(unsigned, v[2 * j]);
            lo[j] = __builtin_bit_cast(unsigned, v[2 * j + 1]);
          } else {
            split2(v[2 * j], v[2 * j + 1], hi[j], lo[j]);
          }
        }
        const bf16x8 bh = __builtin_bit_cast(bf16x8, hi), bl = __builtin_bit_cast(bf16x8, lo);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          if (RABL & 8) {
            asm volatile("" ::"v"(bh), "v"(bl), "v"(wf[2 * ct]), "v"(wf[2 * ct + 1]));
          } else {
            acc[r][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[2 * ct], bh, acc[r][ct], 0, 0, 0);
            acc[r][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[2 * ct], bl, acc[r][ct], 0, 0, 0);
            acc[r][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[2 * ct + 1], bh, acc[r][ct], 0, 0, 0);
          }
        }
      }
      if (!(RABL & 1)) {
#pragma unroll
        for (int f = 0; f < 4; ++f) wf[f] = wn[f];
      }
    }
    __syncthreads();                        // next chunk staged, this one no longer read
  }

#if CP_RPRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  RSTAMP(4);
  // ------------------------------------------------------------------ epilogue: D[co][pixel]
  // Accumulator layout: lane = pixel column (+ 4 channels per lane half), registers = channels.  Through LDS (the
  // region buffers are free now; 16 KB per wave, [channel][row][32 px]) so that a lane stores 16 bytes: a quarter
  // of the store instructions of the direct form, whose issue bounded the kernel's tail.
  {
    float* ot = reinterpret_cast<float*>(smem) + wid * (64 * 2 * 32);
    // per-channel epilogue constants: this lane's channels are 64 cb + 4 kg + a compile-time constant, so the loads
    // are buffer loads with one per-lane offset and immediates (channels past Cout read 0)
    const unsigned cbase = (unsigned)(cb * 64 + 4 * kg) * 4u;
    const unsigned cbytes = (unsigned)a.Cout * 4u;
    const float* shp = a.ep_shift ? a.ep_shift : a.bias;
    const __amdgpu_buffer_rsrc_t rs_sc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ep_scale), 0, a.ep_scale ? (int)cbytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_sh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(shp), 0, shp ? (int)cbytes : 0, 0x00020000);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int col = ct * 32 + (i & 3) + 8 * (i >> 2);    // + 4 kg
        const float sh = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_sh, cbase + (unsigned)col * 4u, 0, 0));
        const float sc = a.ep_scale ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_sc, cbase + (unsigned)col * 4u, 0, 0)) : 1.f;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          float o = acc[r][ct][i];
          if (a.ksplit <= 1) {                              // (a slice of a K split leaves its raw sums)
            o = o * sc + sh;
            if (a.relu) o = fmaxf(o, 0.f);
          }
          ot[((col + 4 * kg) * 2 + r) * 32 + px] = o;
        }
      }
    __syncthreads();
    // rows of 32 pixels back as float4: lane = (row pair index lane >> 3, 4 pixels lane & 7); 16 stores of 16 bytes.  The
    // channel stride rides in the scalar offset of a buffer store whose range check drops channels past Cout.
    const bool wide = (W & 3) == 0 && (reinterpret_cast<unsigned long long>(a.out) & 15ull) == 0 &&
                      (long long)a.Cout * HW * 4 < (1ll << 31);
    const int r = (lane >> 3) & 1, x4 = tx + (lane & 7) * 4, y = ty + 2 * wid + r;
    const int col0 = lane >> 4;                                // + 4 it
    float* const outp = a.ksplit > 1 ? a.partial + (long long)ksl * a.B * a.Cout * HW : a.out;
    if (wide) {
      float* ob = outp + (long long)b * a.Cout * HW;
      const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(ob, 0, (int)((unsigned)a.Cout * plane_bytes), 0x00020000);
      const unsigned vo = (y < H && x4 < W) ? ((unsigned)(cb * 64 + col0) * (unsigned)HW + (unsigned)(y * W + x4)) * 4u : OOB;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(ot + (it * 8 + (lane >> 3)) * 32 + (lane & 7) * 4);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_o, vo, (unsigned)(it * 4) * plane_bytes, 0);
      }
    } else {
#pragma unroll 2
      for (int it = 0; it < 16; ++it) {
        const int co = cb * 64 + col0 + it * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ot + (it * 8 + (lane >> 3)) * 32 + (lane & 7) * 4);
        if (co >= a.Cout || y >= H) continue;
        float* dst = outp + (((long long)b * a.Cout + co) * H + y) * W + x4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (x4 + j < W) dst[j] = v[j];
      }
    }
  }
#ifdef CP_RSTAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RSTAMP(5);
  const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) {
    float* o = a.out + ((long long)b * gridDim.x + blockIdx.x) * 8;
    for (int i = 0; i < 5; ++i) o[i] = (float)(stamp[i + 1] - stamp[i]);
    o[5] = (float)(rt1 - rt0);          // 100 MHz ticks
    o[6] = (float)(rt0 & 0xffffff);     // start, 100 MHz ticks (low bits)
    o[7] = (float)(__builtin_amdgcn_s_getreg((15 << 11) | 4) | (__builtin_amdgcn_s_getreg((3 << 11) | 20) << 16));   // HW_ID[15:0] | XCC_ID << 16
  }
#endif
}

}  // namespace

bool cp_dcn_region_supported(const cp_dcn_shape* s) {
  if (s->kh != 3 || s->kw != 3 || s->stride != 1 || s->pad != 1 || s->dil != 1 || s->deformable_groups != 1) return false;
  if (s->Cin < 16 || (s->Cin & 15)) return false;
  const long long HW = (long long)s->H * s->W;
  if (s->H >= 65534 || s->W >= 65534 || s->H < 2 || s->W < 2) return false;
  if (HW * 4 * 27 >= (1ll << 31) || (long long)s->Cin * HW * 4 >= (1ll << 31)) return false;
  if (s->B > 65535 || (s->Cout + 63) / 64 > 65535) return false;
  return true;
}

size_t cp_dcn_region_wperm_bytes(const cp_dcn_shape* s) {
  return (size_t)((s->Cout + 63) / 64) * (s->Cin / 16) * TAPS * 4 * 64 * 16;
}

int cp_dcn_region_prepare(const cp_dcn_shape* s, const float* weight, void* wp, hipStream_t st) {
  const int total = (int)(cp_dcn_region_wperm_bytes(s) / 16);
  hipLaunchKernelGGL(dcn_region_wperm_kernel, dim3((total + 255) / 256), dim3(256), 0, st, weight, (bf16x8*)wp, s->Cout,
                     s->Cin, s->Cin / 16, total);
  return cp_launch_status();
}

size_t cp_dcn_region_om_wperm_bytes(const cp_dcn_shape* s) { return (size_t)(s->Cin / 16) * TAPS * 2 * 64 * 16; }

int cp_dcn_region_prepare_om(const cp_dcn_shape* s, const float* om_weight, void* wp, hipStream_t st) {
  const int total = (int)(cp_dcn_region_om_wperm_bytes(s) / 16);
  hipLaunchKernelGGL(dcn_region_omperm_kernel, dim3((total + 255) / 256), dim3(256), 0, st, om_weight, (bf16x8*)wp, s->Cin, total);
  return cp_launch_status();
}

// om_wp != null: the fused form (conv_offset_mask computed in the kernel from om_wp / om_bias; offset / mask unused,
// om_out optional)
int cp_dcn_region_ksplit(const cp_dcn_shape* s) {
  // Tiles x 64-channel blocks x images that do not give every CU its two workgroups are topped up by splitting the
  // input channels: >= 2 chunks of 16 per slice, about 512 workgroups.
  const long long wgs = (long long)((s->H + TH - 1) / TH) * ((s->W + TW - 1) / TW) * ((s->Cout + 63) / 64) * s->B;
  const int nchunk = s->Cin / 16;
  if (wgs >= 384 || nchunk < 4) return 1;
  int k = (int)((512 + wgs - 1) / wgs);
  if (k > nchunk / 2) k = nchunk / 2;
  if (k > 16) k = 16;
  if (k < 1) k = 1;
  const int cps = (nchunk + k - 1) / k;
  return (nchunk + cps - 1) / cps;
}

int cp_dcn_region_forward(const cp_dcn_shape* s, const float* x, const float* offset, int64_t offset_bstride,
                          const float* mask, int64_t mask_bstride, int32_t mask_is_logit, const void* wp,
                          const float* bias, const float* ep_scale, const float* ep_shift, int32_t relu, float* out,
                          const void* om_wp, const float* om_bias, float* om_out, float* partial, int ksplit,
                          hipStream_t st) {
  RegionArgs a;
  a.x = x; a.offset = offset; a.mask = mask; a.wp = (const bf16x8*)wp; a.bias = bias;
  a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.out = out;
  a.offset_bstride = offset_bstride; a.mask_bstride = mask_bstride;
  a.B = s->B; a.Cin = s->Cin; a.H = s->H; a.W = s->W; a.Cout = s->Cout;
  a.mask_is_logit = mask_is_logit; a.relu = relu;
  a.tiles_x = (s->W + TW - 1) / TW;
  a.tiles_y = (s->H + TH - 1) / TH;
  a.om_wp = (const bf16x8*)om_wp; a.om_bias = om_bias; a.om_out = om_out;
  if (ksplit < 1 || (ksplit > 1 && (!partial || om_wp))) return CP_EINVAL;
  const int nchunk = s->Cin / 16;
  a.partial = partial;
  a.chunks_per_split = (nchunk + ksplit - 1) / ksplit;
  a.ksplit = (nchunk + a.chunks_per_split - 1) / a.chunks_per_split;
  if ((long long)s->B * a.ksplit > 65535) return CP_EUNSUPPORTED;
  const int lds = 2 * CHB;
  dim3 grid(a.tiles_x * a.tiles_y, (s->Cout + 63) / 64, s->B * a.ksplit);
  if (om_wp) {
    a.mask_is_logit = 1;                    // the convolution's mask channels are logits
    (void)hipFuncSetAttribute((const void*)dcn_fwd_region_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(dcn_fwd_region_kernel<true>, grid, dim3(256), lds, st, a);
  } else {
    (void)hipFuncSetAttribute((const void*)dcn_fwd_region_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(dcn_fwd_region_kernel<false>, grid, dim3(256), lds, st, a);
  }
  return cp_launch_status();
}
