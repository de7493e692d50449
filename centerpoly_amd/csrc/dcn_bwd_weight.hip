// DCNv2 backward, weight gradient for gfx950 -- columns sampled straight into the MFMA operand.
//
// grad_weight[co][ci][t] = sum_{b, p} grad_out[b][co][p] * col[b][p][ci][t],
// col = sigmoid(mask_t(p)) * bilinear(x[b][ci], p + p_t + offset_t(p))          (upstream dcn_v2_backward,
// reference call site src/lib/models/networks/pose_dla_dcn.py:354).
//
// One workgroup = (image, a run of T tiles of 8 rows x 16 pixels, a chunk of 16 input channels, a slab of
// 64 output channels); its partial gradient [slab][16 ci][9 taps] stays in accumulator registers over
// the whole run and is added to grad_weight once (float atomics: 9216 per workgroup).
// The contraction runs over PIXELS: per k-step of 4 pixels
//   A[co][px]  = grad_out, from an LDS copy of the tile's grad_out ([co][128 px], staged per tile);
//   B[px][ci]  = the column value of ONE tap, computed by the lane that owns (px = lane >> 4, ci = lane & 15)
//                right before the MFMA: the (px, tap) recipe from LDS (four corner weights with the mask folded in,
//                region index), four corner reads from the LDS-staged input region, four fmas -- no column tile in LDS, all
//                16 columns of every n-tile useful (the round-1 kernel padded 36 columns to 48 and wrote /
//                re-read a column tile per chunk);
//   D[co][ci]  += one 16x16 tile per (tap, 16 output channels).
// Wave w owns tap w for all pixels; tap 8 is split over the waves by k-step (wave w takes k-steps w, w+8, ...).
// The channel planes of the input region are an odd number of banks apart, so the 16 channels of a lane group
// read 16 different banks.  Taps whose corners leave the region gather from memory (cold, wave-uniformly
// skipped).
#include <stdlib.h>

#include "cp_common.h"

namespace {

constexpr int TAPS = 9;
constexpr int TW = 16, TH = 8, NPX = TW * TH;      // tile: 8 rows x 16 pixels, one k-step = 4 pixels of a row
constexpr int KC = 16;                             // input channels per workgroup
constexpr int HALO = 3, HALO_L = 4, RWD = 24, RH = TH + 2 * HALO, RSZ = RH * RWD;
constexpr int RSZP = RSZ + 1;                      // 337: odd channel stride
constexpr int KSTEPS = NPX / 4;                    // 32
constexpr unsigned OOBW = 0x80000000u;

struct W2Args {
  const float* x;
  const float* offset;
  const float* mask;
  const float* go;
  float* gw;
  float* gb;                  // grad_bias (may be null): summed by the workgroups of the first input-channel chunk
  long long offset_bstride, mask_bstride;
  int B, Cin, H, W, Cout;
  int pad, dil, mask_is_logit;
  int tpr, ntiles;            // tiles per row, tiles per image
  int T, runs_per_image;      // tiles per workgroup run
  int chunks, slabs;          // input-channel chunks of 16, output-channel slabs of 64 (1-D grid: see the kernels)
};

template <int SLAB>             // output channels per workgroup (64; wider layers take several slabs in grid z)
__global__ __launch_bounds__(512, 4) void dcn_bwd_weight2_kernel(W2Args a) {
  constexpr int MT = SLAB / 16;
  constexpr int LDG = NPX + 2;                     // grad_out row pitch 130 = 2 (mod 32): lanes (co, px) hit 32 banks
  __shared__ float goT[SLAB * KC * TAPS];          // [co][px] (pitch LDG); reused as [co][ci * 9 + tap] for the flush
  static_assert(KC * TAPS >= LDG, "flush tile fits the grad_out buffer");
  static_assert(SLAB * 8 == 512 && NPX == 128, "bias sum: one thread per (output channel, 16 pixels)");
  __shared__ __attribute__((aligned(16))) float xreg[KC * RSZP];
  __shared__ float4 rec[TAPS * NPX];               // per (tap, pixel): the four corner weights times the mask
  __shared__ int recb[TAPS * NPX];                 // ... and the region index (>= 0), -2 (nothing) or the cold-path code

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lci = lane & 15, g = lane >> 4;
  // XCD-aware order (1-D grid): workgroups L and L + 8 share an XCD and its L2.  The Cin / 16 channel-chunk workgroups
  // of one tile run all read the run's grad_out tiles and offsets: they take consecutive dispatch slots on ONE XCD
  // (L = ((run group * nchunkslab + chunkslab) * 8 + run % 8), so three of their four fetches hit that L2 -- the y / z
  // grid of round 3 dispatched a run's chunks 256 workgroups apart and fetched everything per chunk from memory.
  int runb, cs;
  {
    const int L = blockIdx.x, nrb = a.runs_per_image * a.B, ncs = a.chunks * a.slabs;
    if ((nrb & 7) == 0) {
      const int q = L >> 3;
      cs = q % ncs;
      runb = (q / ncs) * 8 + (L & 7);
    } else {
      runb = L % nrb;
      cs = L / nrb;
    }
  }
  const int b = runb / a.runs_per_image, run = runb - b * a.runs_per_image;
  const int chunk_i = cs % a.chunks;
  const int c0 = chunk_i * KC;
  const int co0 = (cs / a.chunks) * SLAB;
  const int HW = a.H * a.W;
  const float* xb = a.x + (long long)b * a.Cin * HW;
  const float* gob = a.go + (long long)b * a.Cout * HW;
  const float* off = a.offset + (long long)b * a.offset_bstride;
  const float* msk = a.mask + (long long)b * a.mask_bstride;
  const unsigned plane_bytes = (unsigned)HW * 4u;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xb), 0, (int)((unsigned)a.Cin * plane_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_go = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(gob), 0, (int)((unsigned)a.Cout * plane_bytes), 0x00020000);

  f32x4 acc[MT], acc8[MT];                         // tap `wid`, and this wave's share of tap 8
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = acc8[m] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int t_begin = run * a.T, t_end = min(a.ntiles, t_begin + a.T);
  // The next tile's global data is fetched into registers while this tile is contracted (offsets / mask
  // of this thread's recipe entries, its 16-byte chunks of the input region, its grad_out values), so a
  // tile only pays the recipe arithmetic and the LDS stores between two barriers.
  constexpr int NREC = (TAPS * NPX + 511) / 512;             // recipe entries per thread (3)
  constexpr int NXQ = (KC * RH * (RWD / 4) + 511) / 512;     // region chunks per thread (3)
  constexpr int NGO = SLAB * NPX / 512;                      // grad_out values per thread (16)
  const bool vec_x = (a.W & 3) == 0;
  float raw[NREC][3];
  float gq[NGO];
  auto tile_origin = [&](int tile, int& ty0, int& tx0) {
    const int tyi = tile / a.tpr;
    ty0 = tyi * TH;
    tx0 = (tile - tyi * a.tpr) * TW;
  };
  auto fetch = [&](int tile) {
    int ty0, tx0;
    tile_origin(tile, ty0, tx0);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                   // keep the per-thread index math out of the loop-invariant set
#pragma unroll
    for (int i = 0; i < NREC; ++i) {
      const int e = tid + 512 * i;
      const int t = min(e / NPX, TAPS - 1), q = e & (NPX - 1);
      const int py = ty0 + (q >> 4), pxx = tx0 + (q & 15);
      const bool ok = e < TAPS * NPX && py < a.H && pxx < a.W;
      const int p = ok ? py * a.W + pxx : 0;
      raw[i][0] = ok ? off[(long long)(2 * t) * HW + p] : 0.f;
      raw[i][1] = ok ? off[(long long)(2 * t + 1) * HW + p] : 0.f;
      raw[i][2] = ok ? msk[(long long)t * HW + p] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NGO; ++i) {
      const int e = tid + 512 * i;
      const int co = e / NPX, q = e - co * NPX;
      const int py = ty0 + (q >> 4), pxx = tx0 + (q & 15);
      const bool ok = py < a.H && pxx < a.W && co0 + co < a.Cout;
      const unsigned o = ok ? (unsigned)(co0 + co) * plane_bytes + 4u * (unsigned)(py * a.W + pxx) : OOBW;
      gq[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_go, o, 0, 0));
    }
  };
  float gbsum = 0.f;                                 // thread (co = tid >> 3, part = tid & 7): 16 pixels of a grad_out row
  const bool do_bias = a.gb != nullptr && chunk_i == 0;
  if (t_begin < t_end) fetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    int ty0, tx0;
    tile_origin(tile, ty0, tx0);
    const int ry0 = ty0 - HALO, rx0 = tx0 - HALO_L;
    __syncthreads();                               // the previous tile's reads of goT / xreg / rec are done
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                   // (as in fetch)

    // ---- recipes of the tile: thread -> (tap, pixel), 9 * 128 entries over 512 threads ----
    bool lane_fb = false;
#pragma unroll
    for (int i = 0; i < NREC; ++i) {
      const int e = tid + 512 * i;
      if (e < TAPS * NPX) {
        const int t = e / NPX, q = e - t * NPX;
        const int py = ty0 + (q >> 4), pxx = tx0 + (q & 15);
        const bool p_ok = py < a.H && pxx < a.W;
        const int ky = t / 3, kx = t - ky * 3;
        float m = raw[i][2];
        if (a.mask_is_logit) m = 1.f / (1.f + __expf(-m));
        const float sy = (float)(py - a.pad + ky * a.dil) + raw[i][0];
        const float sx = (float)(pxx - a.pad + kx * a.dil) + raw[i][1];
        const bool inside = p_ok && sy > -1.f && sx > -1.f && sy < (float)a.H && sx < (float)a.W;
        const float fy = floorf(sy), fx = floorf(sx);
        const int y0 = (int)fy, x0 = (int)fx;
        const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= a.H - 1;
        const bool x0ok = x0 >= 0, x1ok = x0 + 1 <= a.W - 1;
        const int y0c = min(max(y0, 0), a.H - 1), x0c = min(max(x0, 0), a.W - 1);
        const int vb = (y0ok && x0ok ? 1 : 0) | (y0ok && x1ok ? 2 : 0) | (y1ok && x0ok ? 4 : 0) | (y1ok && x1ok ? 8 : 0);
        const int ry = y0 - ry0, rx = x0 - rx0;
        const bool in_region = inside && ry >= 0 && ry + 1 < RH && rx >= 0 && rx + 1 < RWD;
        const int rb = in_region ? ry * RWD + rx : (inside ? -(3 + (((y0c * a.W + x0c) << 4) | vb)) : -2);
        lane_fb |= rb <= -3;
        const float ly = sy - fy, lx = sx - fx, hy = 1.f - ly, hx = 1.f - lx, mm = p_ok ? m : 0.f;
        rec[e] = make_float4(hy * hx * mm, hy * lx * mm, ly * hx * mm, ly * lx * mm);
        recb[e] = rb;
      }
    }
    // ---- input region of the chunk's 16 channels: 16-byte chunks (W % 4 == 0) or scalars ----
    if (vec_x) {
      f32x4 xq[NXQ];
#pragma unroll
      for (int i = 0; i < NXQ; ++i) {
        const int q = tid + 512 * i;
        const int c = q / (RH * (RWD / 4)), r = q - c * (RH * (RWD / 4));
        const int ry = r / (RWD / 4), c4 = r - ry * (RWD / 4);
        const int gy = ry0 + ry, gx = rx0 + 4 * c4;
        const bool ok = q < KC * RH * (RWD / 4) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const unsigned o = ok ? (unsigned)(c0 + c) * plane_bytes + 4u * (unsigned)(gy * a.W + gx) : OOBW;
        xq[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, o, 0, 0));
      }
#pragma unroll
      for (int i = 0; i < NXQ; ++i) {
        const int q = tid + 512 * i;
        if (q < KC * RH * (RWD / 4)) {
          const int c = q / (RH * (RWD / 4)), r = q - c * (RH * (RWD / 4));
          const int ry = r / (RWD / 4), c4 = r - ry * (RWD / 4);
          float* d = &xreg[c * RSZP + ry * RWD + 4 * c4];
          d[0] = xq[i].x; d[1] = xq[i].y; d[2] = xq[i].z; d[3] = xq[i].w;
        }
      }
    } else {
      for (int e = tid; e < KC * RSZ; e += 512) {
        const int c = e / RSZ, cell = e - c * RSZ;
        const int ry = cell / RWD, rx = cell - ry * RWD;
        const int gy = ry0 + ry, gx = rx0 + rx;
        const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const unsigned o = ok ? (unsigned)(c0 + c) * plane_bytes + 4u * (unsigned)(gy * a.W + gx) : OOBW;
        xreg[c * RSZP + cell] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, o, 0, 0));
      }
    }
    // ---- grad_out tile: [co][128 px] ----
#pragma unroll
    for (int i = 0; i < NGO; ++i) {
      const int e = tid + 512 * i;
      const int co = e / NPX, q = e - co * NPX;
      goT[co * LDG + q] = gq[i];
    }
    const bool any_fallback = __syncthreads_or(lane_fb ? 1 : 0) != 0;   // also publishes rec / xreg / goT
    if (tile + 1 < t_end) fetch(tile + 1);          // lands during the contraction below
    if (do_bias) {                                   // grad_bias[co] += sum over the tile's pixels of grad_out
      const float* row = goT + (tid >> 3) * LDG + (tid & 7) * 16;
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) sm += row[i];
      gbsum += sm;
    }

    // ---- contraction over the tile's 128 pixels ----
    const float* xw = xreg + lci * RSZP;
    auto column = [&](int t, int q) -> float {      // col value of (tap t, pixel slot q) for channel c0 + lci
      const float4 w = rec[t * NPX + q];
      const int rb = recb[t * NPX + q];
      const int rbc = max(rb, 0);
      const float val = w.x * xw[rbc] + w.y * xw[rbc + 1] + w.z * xw[rbc + RWD] + w.w * xw[rbc + RWD + 1];
      return rb >= 0 ? val : 0.f;
    };
#pragma unroll 4
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int q = 4 * ks + g;
      const float bv = column(wid, q);
      float av[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) av[m] = goT[(16 * m + lci) * LDG + q];
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv, acc[m], 0, 0, 0);
      if ((ks & 7) == wid) {                        // (wave-uniform) this wave's share of tap 8
        const float b8 = column(8, q);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc8[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], b8, acc8[m], 0, 0, 0);
      }
    }
    // cold path: taps whose corners leave the region gather their corners from memory
    if (any_fallback) {
      const bool ci_ok = c0 + lci < a.Cin;
      const float* xc = xb + (long long)min(c0 + lci, a.Cin - 1) * HW;
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const int q = 4 * ks + g;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
          if (which == 1 && (ks & 7) != wid) continue;
          const int t = which == 0 ? wid : 8;
          const float4 w = rec[t * NPX + q];
          const int rb = recb[t * NPX + q];
          float bv = 0.f;
          if (rb <= -3 && ci_ok) {
            const int code = -rb - 3;
            const unsigned vb = (unsigned)(code & 15);
            const int fbase = code >> 4;
            const int dx = ((vb & 3u) == 3u || (vb & 12u) == 12u) ? 1 : 0;
            const int dy = ((vb & 5u) == 5u || (vb & 10u) == 10u) ? a.W : 0;
            const float* p = xc + fbase;
            const float v00 = (vb & 1u) ? p[0] : 0.f, v01 = (vb & 2u) ? p[dx] : 0.f;
            const float v10 = (vb & 4u) ? p[dy] : 0.f, v11 = (vb & 8u) ? p[dy + dx] : 0.f;
            bv = w.x * v00 + w.y * v01 + w.z * v10 + w.w * v11;
          }
          if (__builtin_amdgcn_ballot_w64(rb <= -3) == 0ull) continue;     // (wave-uniform skip)
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float av = goT[(16 * m + lci) * LDG + q];
            if (which == 0) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[m], 0, 0, 0);
            else acc8[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc8[m], 0, 0, 0);
          }
        }
      }
    }
  }

  if (do_bias) {
    gbsum += __shfl_xor(gbsum, 1, 64);
    gbsum += __shfl_xor(gbsum, 2, 64);
    gbsum += __shfl_xor(gbsum, 4, 64);
    if ((tid & 7) == 0 && co0 + (tid >> 3) < a.Cout) atomicAdd(&a.gb[co0 + (tid >> 3)], gbsum);
  }
  // ---- flush: the partial gradient goes through LDS so that grad_weight receives ONE coalesced atomic add
  // per element and workgroup (rows of 16 ci x 9 taps = 144 contiguous floats per output channel); tap 8's
  // eight per-wave partial sums meet in LDS first.  D[co = 16 m + 4 g + reg][ci = lci].
  __syncthreads();
  float* outT = goT;                                // [SLAB][144]
  for (int e = tid; e < SLAB * KC; e += 512) outT[e * TAPS + 8] = 0.f;
  __syncthreads();
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* q = &outT[(16 * m + 4 * g + r) * (KC * TAPS) + lci * TAPS];
      q[wid] = acc[m][r];
      atomicAdd(&q[8], acc8[m][r]);
    }
  __syncthreads();
  const int Ktot = a.Cin * TAPS;
  for (int e = tid; e < SLAB * KC * TAPS; e += 512) {
    const int co = e / (KC * TAPS), k = e - co * (KC * TAPS);
    if (co0 + co < a.Cout && c0 * TAPS + k < Ktot) atomicAdd(&a.gw[(long long)(co0 + co) * Ktot + c0 * TAPS + k], outT[e]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Split-bf16 form of the same kernel (W % 4 == 0; default): the contraction runs on the bf16 matrix cores as three
// products of bf16 halves (v = hi + lo; hi*hi + hi*lo + lo*hi, fp32 accumulate, ~2^-16 relative error per product).
// The f32-input MFMA executes on the SIMD's vector ALUs, where it competed with the sampling arithmetic for the
// same issue slots (4608 of ~7200 issue cycles per wave and tile); the bf16 cores run beside the VALU.
// One k-step is now 32 pixels = two rows of the 8 x 16 tile: lane (g, ci) samples the 8 consecutive pixels
// 8 g .. 8 g + 7 of the step for its channel -- the B fragment of v_mfma_f32_16x16x32_bf16 -- splits them, and the A
// fragments (8 consecutive pixels of a grad_out row) come from bf16 hi / lo planes of the tile (pitch 136: an odd
// number of 16-byte units, conflict-free ds_read_b128), split once when the tile is staged.  Same work split over
// the waves (wave w owns tap w; tap 8: k-step w & 3, output-channel half w >> 2), same flush.  Taps whose corners
// leave the LDS region gather from memory inside the sampling function (rare, exec-masked).
typedef __bf16 w3_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned w3_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void w3_split4(const f32x4 v, w3_u32x2& hi, w3_u32x2& lo) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 h0, h1, l0, l1;
  h0[0] = (__bf16)v.x; h0[1] = (__bf16)v.y; h1[0] = (__bf16)v.z; h1[1] = (__bf16)v.w;
  l0[0] = (__bf16)(v.x - (float)h0[0]); l0[1] = (__bf16)(v.y - (float)h0[1]);
  l1[0] = (__bf16)(v.z - (float)h1[0]); l1[1] = (__bf16)(v.w - (float)h1[1]);
  hi = w3_u32x2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
  lo = w3_u32x2{__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1)};
}

__global__ __launch_bounds__(512, 4) void dcn_bwd_weight3_kernel(W2Args a) {
  constexpr int SLAB = 64, MT = 4;
  constexpr int GP = NPX + 8;                      // grad_out plane pitch per output channel (bf16 elements)
  constexpr int G_PLANE = SLAB * GP;
  constexpr int STAGE_BYTES = 2 * G_PLANE * 2, OUT_BYTES = 32 * KC * TAPS * 4;
  __shared__ __attribute__((aligned(16))) unsigned char gbuf[STAGE_BYTES > OUT_BYTES ? STAGE_BYTES : OUT_BYTES];
  unsigned short* Gs = reinterpret_cast<unsigned short*>(gbuf);          // [hi | lo][co][136]
  __shared__ __attribute__((aligned(16))) float xreg[KC * RSZP];
  __shared__ float4 rec[TAPS * NPX];
  __shared__ int recb[TAPS * NPX];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lci = lane & 15, g = lane >> 4;
  // XCD-aware order (1-D grid): workgroups L and L + 8 share an XCD and its L2.  The Cin / 16 channel-chunk workgroups
  // of one tile run all read the run's grad_out tiles and offsets: they take consecutive dispatch slots on ONE XCD
  // (L = ((run group * nchunkslab + chunkslab) * 8 + run % 8), so three of their four fetches hit that L2 -- the y / z
  // grid of round 3 dispatched a run's chunks 256 workgroups apart and fetched everything per chunk from memory.
  int runb, cs;
  {
    const int L = blockIdx.x, nrb = a.runs_per_image * a.B, ncs = a.chunks * a.slabs;
    if ((nrb & 7) == 0) {
      const int q = L >> 3;
      cs = q % ncs;
      runb = (q / ncs) * 8 + (L & 7);
    } else {
      runb = L % nrb;
      cs = L / nrb;
    }
  }
  const int b = runb / a.runs_per_image, run = runb - b * a.runs_per_image;
  const int chunk_i = cs % a.chunks;
  const int c0 = chunk_i * KC;
  const int co0 = (cs / a.chunks) * SLAB;
  const int HW = a.H * a.W;
  const float* xb = a.x + (long long)b * a.Cin * HW;
  const float* gob = a.go + (long long)b * a.Cout * HW;
  const float* off = a.offset + (long long)b * a.offset_bstride;
  const float* msk = a.mask + (long long)b * a.mask_bstride;
  const unsigned plane_bytes = (unsigned)HW * 4u;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xb), 0, (int)((unsigned)a.Cin * plane_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_go = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(gob), 0, (int)((unsigned)a.Cout * plane_bytes), 0x00020000);

  f32x4 acc[MT], acc8[2];                          // tap `wid` (all 64 co), and this wave's share of tap 8
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  acc8[0] = acc8[1] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int t_begin = run * a.T, t_end = min(a.ntiles, t_begin + a.T);
  constexpr int NREC = (TAPS * NPX + 511) / 512;             // recipe entries per thread (3)
  constexpr int NXQ = (KC * RH * (RWD / 4) + 511) / 512;     // region chunks per thread (3)
  constexpr int NGO = SLAB * NPX / 4 / 512;                  // grad_out float4 per thread (4)
  float raw[NREC][3];
  f32x4 gq[NGO];
  auto tile_origin = [&](int tile, int& ty0, int& tx0) {
    const int tyi = tile / a.tpr;
    ty0 = tyi * TH;
    tx0 = (tile - tyi * a.tpr) * TW;
  };
  auto fetch = [&](int tile) {
    int ty0, tx0;
    tile_origin(tile, ty0, tx0);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                   // keep the per-thread index math out of the loop-invariant set
#pragma unroll
    for (int i = 0; i < NREC; ++i) {
      const int e = tid + 512 * i;
      const int t = min(e / NPX, TAPS - 1), q = e & (NPX - 1);
      const int py = ty0 + (q >> 4), pxx = tx0 + (q & 15);
      const bool ok = e < TAPS * NPX && py < a.H && pxx < a.W;
      const int p = ok ? py * a.W + pxx : 0;
      raw[i][0] = ok ? off[(long long)(2 * t) * HW + p] : 0.f;
      raw[i][1] = ok ? off[(long long)(2 * t + 1) * HW + p] : 0.f;
      raw[i][2] = ok ? msk[(long long)t * HW + p] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NGO; ++i) {                 // unit = (co, row, 4-pixel group): 32 units per output channel
      const int u = tid + 512 * i;
      const int co = u >> 5, r = (u >> 2) & 7, c4 = u & 3;
      const int py = ty0 + r, pxx = tx0 + 4 * c4;
      const bool ok = py < a.H && pxx < a.W && co0 + co < a.Cout;      // W % 4 == 0: a group is all in or all out
      const unsigned o = ok ? (unsigned)(co0 + co) * plane_bytes + 4u * (unsigned)(py * a.W + pxx) : OOBW;
      gq[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_go, o, 0, 0));
    }
  };
  float gbsum[NGO];                                  // grad_bias partial sums of this thread's NGO output channels
#pragma unroll
  for (int i = 0; i < NGO; ++i) gbsum[i] = 0.f;
  const bool do_bias = a.gb != nullptr && chunk_i == 0;
  const bool ci_ok = c0 + lci < a.Cin;
  const float* xc = xb + (long long)min(c0 + lci, a.Cin - 1) * HW;

  if (t_begin < t_end) fetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    int ty0, tx0;
    tile_origin(tile, ty0, tx0);
    const int ry0 = ty0 - HALO, rx0 = tx0 - HALO_L;
    __syncthreads();                               // the previous tile's reads of Gs / xreg / rec are done
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));

    // ---- recipes of the tile (as in dcn_bwd_weight2_kernel) ----
#pragma unroll
    for (int i = 0; i < NREC; ++i) {
      const int e = tid + 512 * i;
      if (e < TAPS * NPX) {
        const int t = e / NPX, q = e - t * NPX;
        const int py = ty0 + (q >> 4), pxx = tx0 + (q & 15);
        const bool p_ok = py < a.H && pxx < a.W;
        const int ky = t / 3, kx = t - ky * 3;
        float m = raw[i][2];
        if (a.mask_is_logit) m = 1.f / (1.f + __expf(-m));
        const float sy = (float)(py - a.pad + ky * a.dil) + raw[i][0];
        const float sx = (float)(pxx - a.pad + kx * a.dil) + raw[i][1];
        const bool inside = p_ok && sy > -1.f && sx > -1.f && sy < (float)a.H && sx < (float)a.W;
        const float fy = floorf(sy), fx = floorf(sx);
        const int y0 = (int)fy, x0 = (int)fx;
        const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= a.H - 1;
        const bool x0ok = x0 >= 0, x1ok = x0 + 1 <= a.W - 1;
        const int y0c = min(max(y0, 0), a.H - 1), x0c = min(max(x0, 0), a.W - 1);
        const int vb = (y0ok && x0ok ? 1 : 0) | (y0ok && x1ok ? 2 : 0) | (y1ok && x0ok ? 4 : 0) | (y1ok && x1ok ? 8 : 0);
        const int ry = y0 - ry0, rx = x0 - rx0;
        const bool in_region = inside && ry >= 0 && ry + 1 < RH && rx >= 0 && rx + 1 < RWD;
        const int rb = in_region ? ry * RWD + rx : (inside ? -(3 + (((y0c * a.W + x0c) << 4) | vb)) : -2);
        const float ly = sy - fy, lx = sx - fx, hy = 1.f - ly, hx = 1.f - lx, mm = p_ok ? m : 0.f;
        rec[e] = make_float4(hy * hx * mm, hy * lx * mm, ly * hx * mm, ly * lx * mm);
        recb[e] = rb;
      }
    }
    // ---- input region of the chunk's 16 channels, 16-byte chunks ----
    {
      f32x4 xq[NXQ];
#pragma unroll
      for (int i = 0; i < NXQ; ++i) {
        const int q = tid + 512 * i;
        const int c = q / (RH * (RWD / 4)), r = q - c * (RH * (RWD / 4));
        const int ry = r / (RWD / 4), c4 = r - ry * (RWD / 4);
        const int gy = ry0 + ry, gx = rx0 + 4 * c4;
        const bool ok = q < KC * RH * (RWD / 4) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const unsigned o = ok ? (unsigned)(c0 + c) * plane_bytes + 4u * (unsigned)(gy * a.W + gx) : OOBW;
        xq[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, o, 0, 0));
      }
#pragma unroll
      for (int i = 0; i < NXQ; ++i) {
        const int q = tid + 512 * i;
        if (q < KC * RH * (RWD / 4)) {
          const int c = q / (RH * (RWD / 4)), r = q - c * (RH * (RWD / 4));
          const int ry = r / (RWD / 4), c4 = r - ry * (RWD / 4);
          float* d = &xreg[c * RSZP + ry * RWD + 4 * c4];
          d[0] = xq[i].x; d[1] = xq[i].y; d[2] = xq[i].z; d[3] = xq[i].w;
        }
      }
    }
    // ---- grad_out tile: split once, bf16 hi / lo planes [co][row * 16 + col] ----
#pragma unroll
    for (int i = 0; i < NGO; ++i) {
      const int u = tid + 512 * i;
      const int co = u >> 5, q = (u & 31) * 4;
      w3_u32x2 hi, lo;
      w3_split4(gq[i], hi, lo);
      *reinterpret_cast<w3_u32x2*>(&Gs[co * GP + q]) = hi;
      *reinterpret_cast<w3_u32x2*>(&Gs[G_PLANE + co * GP + q]) = lo;
      if (do_bias) gbsum[i] += (gq[i].x + gq[i].y) + (gq[i].z + gq[i].w);
    }
    __syncthreads();                                // publishes rec / xreg / Gs
    if (tile + 1 < t_end) fetch(tile + 1);          // lands during the contraction below

    // ---- contraction over the tile's 128 pixels: 4 k-steps of 32 ----
    const float* xw = xreg + lci * RSZP;
    auto column = [&](int t, int q) __attribute__((always_inline)) -> float {
      const float4 w = rec[t * NPX + q];
      const int rb = recb[t * NPX + q];
      const int rbc = max(rb, 0);
      float val = w.x * xw[rbc] + w.y * xw[rbc + 1] + w.z * xw[rbc + RWD] + w.w * xw[rbc + RWD + 1];
      val = rb >= 0 ? val : 0.f;
      if (rb <= -3) {                               // corners outside the region: gather from memory (rare)
        val = 0.f;
        if (ci_ok) {
          const int code = -rb - 3;
          const unsigned vb = (unsigned)(code & 15);
          const int fbase = code >> 4;
          const int dx = ((vb & 3u) == 3u || (vb & 12u) == 12u) ? 1 : 0;
          const int dy = ((vb & 5u) == 5u || (vb & 10u) == 10u) ? a.W : 0;
          const float* p = xc + fbase;
          const float v00 = (vb & 1u) ? p[0] : 0.f, v01 = (vb & 2u) ? p[dx] : 0.f;
          const float v10 = (vb & 4u) ? p[dy] : 0.f, v11 = (vb & 8u) ? p[dy + dx] : 0.f;
          val = w.x * v00 + w.y * v01 + w.z * v10 + w.w * v11;
        }
      }
      return val;
    };
    auto sample8 = [&](int t, int s, w3_bf16x8& fh, w3_bf16x8& fl) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = column(t, 32 * s + 8 * g + j);
        const __bf16 h = (__bf16)v;
        fh[j] = h;
        fl[j] = (__bf16)(v - (float)h);
      }
    };
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      w3_bf16x8 bh, bl;
      sample8(wid, s, bh, bl);
      w3_bf16x8 ah[MT], al[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int e = (16 * m + lci) * GP + 32 * s + 8 * g;
        ah[m] = *reinterpret_cast<const w3_bf16x8*>(&Gs[e]);
        al[m] = *reinterpret_cast<const w3_bf16x8*>(&Gs[G_PLANE + e]);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bh, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bl, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m], bh, acc[m], 0, 0, 0);
      if ((wid & 3) == s) {                         // (wave-uniform) this wave's share of tap 8: output channels
        w3_bf16x8 b8h, b8l;                         // 32 (wid >> 2) .. + 31 of this k-step
        sample8(8, s, b8h, b8l);
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          const int m = 2 * (wid >> 2) + mm;
          const int e = (16 * m + lci) * GP + 32 * s + 8 * g;
          const w3_bf16x8 a8h = *reinterpret_cast<const w3_bf16x8*>(&Gs[e]);
          const w3_bf16x8 a8l = *reinterpret_cast<const w3_bf16x8*>(&Gs[G_PLANE + e]);
          acc8[mm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8h, b8h, acc8[mm], 0, 0, 0);
          acc8[mm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8h, b8l, acc8[mm], 0, 0, 0);
          acc8[mm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8l, b8h, acc8[mm], 0, 0, 0);
        }
      }
    }
  }

  if (do_bias) {                                     // the 32 threads tid & ~31 .. share an output channel per unit slot
#pragma unroll
    for (int i = 0; i < NGO; ++i) {
      float v = gbsum[i];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      const int co = (tid + 512 * i) >> 5;
      if ((tid & 31) == 0 && co0 + co < a.Cout) atomicAdd(&a.gb[co0 + co], v);
    }
  }
  // ---- flush (as dcn_bwd_weight2_kernel, in two passes of 32 output channels so that the staging buffer bounds
  // the LDS footprint: two workgroups per CU): through LDS, one coalesced atomic add per element ----
  float* outT = reinterpret_cast<float*>(gbuf);     // [32][144]
  const int Ktot = a.Cin * TAPS;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
    for (int e = tid; e < 32 * KC; e += 512) outT[e * TAPS + 8] = 0.f;
    __syncthreads();
#pragma unroll
    for (int mm = 0; mm < 2; ++mm) {
      const int m = 2 * pass + mm;
#pragma unroll
      for (int r = 0; r < 4; ++r) outT[(16 * mm + 4 * g + r) * (KC * TAPS) + lci * TAPS + wid] = acc[m][r];
    }
    if ((wid >> 2) == pass) {
#pragma unroll
      for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          atomicAdd(&outT[(16 * mm + 4 * g + r) * (KC * TAPS) + lci * TAPS + 8], acc8[mm][r]);
    }
    __syncthreads();
    for (int e = tid; e < 32 * KC * TAPS; e += 512) {
      const int co = 32 * pass + e / (KC * TAPS), k = e % (KC * TAPS);
      if (co0 + co < a.Cout && c0 * TAPS + k < Ktot)
        atomicAdd(&a.gw[(long long)(co0 + co) * Ktot + c0 * TAPS + k], outT[e]);
    }
  }
}

}  // namespace

bool cp_dcn_bwd_weight2_supported(const cp_dcn_shape* s) {
  if (s->stride != 1 || s->pad != s->dil || s->kh != 3 || s->kw != 3) return false;
  const unsigned long long hw = (unsigned long long)s->H * s->W;
  if (hw >= (1ull << 27)) return false;
  if ((unsigned long long)(s->Cin + KC) * hw * 4ull >= 0x70000000ull) return false;
  if ((unsigned long long)s->Cout * hw * 4ull >= 0x70000000ull) return false;
  return true;
}

int cp_dcn_bwd_weight2(const cp_dcn_shape* s, const float* x, const float* offset, int64_t offset_bstride,
                       const float* mask, int64_t mask_bstride, int32_t mask_is_logit, const float* grad_out,
                       float* grad_weight, float* grad_bias, int32_t flags, hipStream_t st) {
  W2Args a;
  a.x = x; a.offset = offset; a.mask = mask; a.go = grad_out; a.gw = grad_weight; a.gb = grad_bias;
  a.offset_bstride = offset_bstride; a.mask_bstride = mask_bstride;
  a.B = s->B; a.Cin = s->Cin; a.H = s->H; a.W = s->W; a.Cout = s->Cout;
  a.pad = s->pad; a.dil = s->dil; a.mask_is_logit = mask_is_logit;
  a.tpr = (s->W + TW - 1) / TW;
  a.ntiles = a.tpr * ((s->H + TH - 1) / TH);
  const int chunks = (s->Cin + KC - 1) / KC;
  const int slab = 64;      // 128-wide slabs halve the sampling per MFMA but leave one workgroup per CU: measured
                            // slower (0.72 vs 0.65 ms at 128->128 @128x256 x4)
  const int slabs = (s->Cout + slab - 1) / slab;
  // tiles per run: ONE round of workgroups on 256 CUs x 2 (every extra tile amortises the flush, whose float atomics all
  // land in the same Cout x Cin x 9 block; round 4 sweep at the six layer shapes: 1024 / 768 / 512 / 256 workgroups =
  // 0.453 / 0.542 / 0.449 / 0.533 ms at 64 -> 64 @256x512 x4, 0.428 / 0.448 / 0.417 / 0.499 at 128 -> 128 @128x256 x4)
  const long long units = (long long)a.ntiles * s->B * chunks * slabs;
  int T = (int)((units + 511) / 512);
  if (T < 1) T = 1;
  if (T > a.ntiles) T = a.ntiles;
  a.T = T;
  a.runs_per_image = (a.ntiles + T - 1) / T;
  a.chunks = chunks;
  a.slabs = slabs;
  const dim3 grid(a.runs_per_image * s->B * chunks * slabs, 1, 1);
  // split-bf16 form by default (needs whole float4 groups of grad_out: W % 4 == 0); CP_DCN_BWD_EXACT_F32: exact f32
  const bool exact = (flags & CP_DCN_BWD_EXACT_F32) != 0;
  if (!exact && (s->W & 3) == 0)
    hipLaunchKernelGGL(dcn_bwd_weight3_kernel, grid, dim3(512), 0, st, a);
  else
    hipLaunchKernelGGL((dcn_bwd_weight2_kernel<64>), grid, dim3(512), 0, st, a);
  return cp_launch_status();
}
