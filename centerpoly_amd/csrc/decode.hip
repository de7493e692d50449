// Fused 3x3-max NMS + exact top-K + polygon decode for gfx950.
//
// Replaces _nms (src/lib/models/decode.py:13-19), _topk (:117-133),
// polydet_decode (:512-670) and the NHWC-copy gather helpers
// (src/lib/models/utils.py:12-26) of the reference.
//
// Order of selection = score descending, ties by flat (class, y, x) index
// ascending.  That is exactly what the reference's two-level top-k gives when
// torch.topk is replaced by a stable sort, so indices are bit-exact against the
// oracle for ANY input (no NaNs), not only tie-free ones.
//
// Stage 1 (grid = tiles x B): every workgroup reads one 4096-element slice of the
//   image's [C*H*W] heat, applies the NMS test against its 8 neighbours (L2
//   resident), and keeps the slice's K best by an in-register 4x8-bit radix
//   select; ties at the threshold are resolved in index order with a block scan.
//   The zero bin (almost everything after NMS) is wave-aggregated so LDS
//   histogram atomics never serialise on it.
// Stage 2 (grid = B): one 1024-thread workgroup radix-selects the K best of the
//   tiles' candidates on the 64-bit (value, ~index) key, bitonic-sorts them and
//   decodes: gather reg/poly/depth with direct strided reads, polar->cartesian in
//   double (the reference uses math.cos on a Python float), bbox = min/max.
#include "cp_common.h"

namespace {

constexpr int S1_THREADS = 256;
constexpr int S1_EPT = 16;                       // elements per thread
constexpr int S1_TILE = S1_THREADS * S1_EPT;     // 4096
constexpr int S2_THREADS = 1024;
constexpr int KMAX = 256;
constexpr uint32_t OZ = 0x80000000u;             // orderable(+0.0f)

__device__ __forceinline__ uint32_t f2ord(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}

// Suffix scan of a 256-bin histogram by wave 0: finds digit d with
//   count(bins > d) < need <= count(bins >= d);  sel[0] = d, sel[1] = need - count(bins > d).
__device__ __forceinline__ void pick_digit(const uint32_t* hist, uint32_t need, uint32_t* sel) {
  const int lane = threadIdx.x & 63;
  if ((threadIdx.x >> 6) != 0) return;
  uint32_t h[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) h[i] = hist[lane * 4 + i];
  const uint32_t s = h[0] + h[1] + h[2] + h[3];
  uint32_t incl = s;                       // inclusive suffix sum over lanes >= lane
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_down(incl, o, 64);
    if (lane + o < 64) incl += t;
  }
  uint32_t above = incl - s;               // bins owned by higher lanes
  if (above < need && need <= incl) {
#pragma unroll
    for (int i = 3; i >= 0; --i) {
      if (above < need && need <= above + h[i]) {
        sel[0] = lane * 4 + i;
        sel[1] = need - above;
      }
      above += h[i];
    }
  }
}

// Block-wide exclusive scan of one uint per thread (NT threads), result via LDS scratch.
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wsum) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wid; ++w) base += wsum[w];
  __syncthreads();
  return base + incl - v;
}

// ----------------------------------------------------------------- stage 1 ---
__global__ __launch_bounds__(S1_THREADS) void nms_tile_topk_kernel(
    const float* __restrict__ heat, unsigned long long* __restrict__ cand, int C, int H, int W,
    int K, int tiles) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t sel[2];
  __shared__ uint32_t wsum[S1_THREADS / 64];
  __shared__ uint32_t slot;
  const int b = blockIdx.y, tile = blockIdx.x;
  const int HW = H * W;
  const long long total = (long long)C * HW;
  const float* hb = heat + (long long)b * total;
  const int tid = threadIdx.x, lane = tid & 63;
  const long long e0 = (long long)tile * S1_TILE + (long long)tid * S1_EPT;

  uint32_t ov[S1_EPT];
#pragma unroll
  for (int i = 0; i < S1_EPT; ++i) {
    const long long e = e0 + i;
    ov[i] = 0;                                   // 0 = "no element"
    if (e < total) {
      const int c = (int)(e / HW);
      const int r = (int)(e - (long long)c * HW);
      const int y = r / W, x = r - y * W;
      const float* pl = hb + (long long)c * HW;
      const float v = pl[r];
      float m = v;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int yy = y + dy, xx = x + dx;
          if (yy >= 0 && yy < H && xx >= 0 && xx < W) m = fmaxf(m, pl[yy * W + xx]);
        }
      const float kept = (m == v) ? v : 0.f;      // heat * keep, -0 canonicalised to +0
      ov[i] = f2ord(kept + 0.f);
    }
  }
  if (tid == 0) slot = 0;
  // valid elements in the tile (same for all threads)
  const long long rem = total - (long long)tile * S1_TILE;
  const uint32_t n_tile = rem >= S1_TILE ? S1_TILE : (uint32_t)rem;
  unsigned long long* out = cand + ((long long)b * tiles + tile) * K;

  uint32_t vstar = 0, need = 0;
  const bool take_all = n_tile <= (uint32_t)K;
  if (!take_all) {
    uint32_t prefix = 0;
    need = K;
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      hist[tid] = 0;                               // S1_THREADS == 256 bins
      __syncthreads();
      uint32_t zeros = 0;
#pragma unroll
      for (int i = 0; i < S1_EPT; ++i) {
        const uint32_t o = ov[i];
        if (o == 0) continue;
        if (pass > 0 && (o >> (shift + 8)) != prefix) continue;
        if (o == OZ) ++zeros;
        else atomicAdd(&hist[(o >> shift) & 255u], 1u);
      }
      // wave-aggregate the zero bin
      for (int off = 32; off > 0; off >>= 1) zeros += __shfl_xor(zeros, off, 64);
      if (lane == 0 && zeros) atomicAdd(&hist[(OZ >> shift) & 255u], zeros);
      __syncthreads();
      pick_digit(hist, need, sel);
      __syncthreads();
      prefix = (prefix << 8) | sel[0];
      need = sel[1];
      __syncthreads();
    }
    vstar = prefix;
  }
  // ties at vstar are taken in index order: rank them with an ordered block scan
  uint32_t my_ties = 0;
#pragma unroll
  for (int i = 0; i < S1_EPT; ++i) my_ties += (!take_all && ov[i] == vstar) ? 1u : 0u;
  uint32_t tie_rank = block_excl_scan<S1_THREADS>(my_ties, wsum);
#pragma unroll
  for (int i = 0; i < S1_EPT; ++i) {
    const uint32_t o = ov[i];
    if (o == 0) continue;
    bool take = take_all || o > vstar;
    if (!take_all && o == vstar) {
      take = tie_rank < need;
      ++tie_rank;
    }
    if (take) {
      const uint32_t s = atomicAdd(&slot, 1u);
      const uint32_t e = (uint32_t)(e0 + i);
      out[s] = ((unsigned long long)o << 32) | (unsigned long long)(~e);
    }
  }
  __syncthreads();
  // pad with the empty key
  for (uint32_t s = slot + tid; s < (uint32_t)K; s += S1_THREADS) out[s] = 0ull;
}

// ----------------------------------------------------------------- stage 2 ---
struct DecodeArgs {
  const unsigned long long* cand;
  const float* polys;
  const float* depth;
  const float* reg;
  float* dets;
  long long* inds;
  int* clses;
  int C, H, W, N2, K, rep, ncand, cat_spec;
};

template <int EPT>
__global__ __launch_bounds__(S2_THREADS) void select_decode_kernel(DecodeArgs a) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t sel[2];
  __shared__ uint32_t slot;
  __shared__ unsigned long long top[KMAX];
  extern __shared__ float rows[];               // [K][N2] gathered polygon rows
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const unsigned long long* cb = a.cand + (long long)b * a.ncand;

  unsigned long long key[EPT];
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int q = tid + i * S2_THREADS;
    key[i] = q < a.ncand ? cb[q] : 0ull;
  }
  if (tid < KMAX) top[tid] = 0ull;
  if (tid == 0) slot = 0;

  // 8 x 8-bit radix select of the K-th largest 64-bit key (keys are distinct)
  unsigned long long prefix = 0;
  uint32_t need = a.K;
#pragma unroll 1
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = 56 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const uint32_t zdig = (uint32_t)((((unsigned long long)OZ << 32) >> shift) & 255ull);
    uint32_t zeros = 0;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const unsigned long long k = key[i];
      if (k == 0ull) continue;
      if (pass > 0 && (k >> (shift + 8)) != prefix) continue;
      const uint32_t d = (uint32_t)((k >> shift) & 255ull);
      // in the value passes nearly all keys share the +0.0 digit: aggregate it
      if (pass < 4 && d == zdig) ++zeros;
      else atomicAdd(&hist[d], 1u);
    }
    if (pass < 4) {
      for (int off = 32; off > 0; off >>= 1) zeros += __shfl_xor(zeros, off, 64);
      if (lane == 0 && zeros) atomicAdd(&hist[zdig], zeros);
    }
    __syncthreads();
    pick_digit(hist, need, sel);
    __syncthreads();
    prefix = (prefix << 8) | (unsigned long long)sel[0];
    need = sel[1];
    const uint32_t in_bin = hist[sel[0]];                 // keys that carry the prefix so far
    __syncthreads();
    // the four value passes are done and EVERY key with the K-th value is wanted (no tie straddles the cut, the
    // normal case for float scores): the index passes would only reproduce the all-zero suffix
    if (pass == 3 && in_bin == need) {
      prefix <<= 32;
      break;
    }
  }
  const unsigned long long kth = prefix;
#pragma unroll
  for (int i = 0; i < EPT; ++i)
    if (key[i] != 0ull && key[i] >= kth) {
      const uint32_t s = atomicAdd(&slot, 1u);
      if (s < KMAX) top[s] = key[i];
    }
  __syncthreads();
  // bitonic sort, descending, KMAX entries (empty keys sink to the end)
  for (int k2 = 2; k2 <= KMAX; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      if (tid < KMAX) {
        const int ixj = tid ^ j;
        if (ixj > tid) {
          const unsigned long long x = top[tid], y = top[ixj];
          const bool desc = (tid & k2) == 0;
          if (desc ? (x < y) : (x > y)) {
            top[tid] = y;
            top[ixj] = x;
          }
        }
      }
      __syncthreads();
    }

  // ---- decode the K winners ----
  const int K = a.K, N2 = a.N2, HW = a.H * a.W;
  // (--cat_spec_poly, decode.py:534-537: one polygon per class, the detection's class picks its channel block)
  const float* pb = a.polys + (long long)b * N2 * (a.cat_spec ? a.C : 1) * HW;
  for (int q = tid; q < K * N2; q += S2_THREADS) {
    const int k = q / N2, j = q - k * N2;
    const uint32_t e = ~(uint32_t)(top[k] & 0xffffffffull);
    const int sp = (int)(e % (uint32_t)HW);
    const int cbase = a.cat_spec ? (int)(e / (uint32_t)HW) * N2 : 0;
    rows[q] = pb[(long long)(cbase + j) * HW + sp];
  }
  __syncthreads();
  if (a.rep != CP_REP_CARTESIAN) {
    for (int q = tid; q < K * (N2 / 2); q += S2_THREADS) {
      const int k = q / (N2 / 2), v = q - k * (N2 / 2);
      const float r = rows[k * N2 + 2 * v];
      double ang;
      if (a.rep == CP_REP_POLAR_FIXED) ang = 2 * 3.14 - 2 * 3.14 / (double)N2 * (double)(2 * v);
      else ang = (double)rows[k * N2 + 2 * v + 1];
      // decode.py:597-614: r * math.cos(theta): double trig, result rounded to fp32, fp32 multiply
      const float cs = (float)cos(ang), sn = (float)sin(ang);
      rows[k * N2 + 2 * v] = __fmul_rn(r, cs);
      rows[k * N2 + 2 * v + 1] = __fmul_rn(r, sn);
    }
    __syncthreads();
  }
  if (tid < K) {
    const int k = tid;
    const unsigned long long kk = top[k];
    const uint32_t e = ~(uint32_t)(kk & 0xffffffffull);
    const int c = (int)(e / (uint32_t)HW);
    const int sp = (int)(e - (uint32_t)c * (uint32_t)HW);
    const float score = ord2f((uint32_t)(kk >> 32));
    // decode.py:122-123: (ind / w).int().float() with true division in fp32
    float ys = truncf(__fdiv_rn((float)sp, (float)a.W));
    float xs = (float)(sp - (sp / a.W) * a.W);
    if (a.reg) {
      const float* rb = a.reg + (long long)b * 2 * HW;
      xs = __fadd_rn(xs, rb[sp]);
      ys = __fadd_rn(ys, rb[HW + sp]);
    } else {
      xs = __fadd_rn(xs, 0.5f);
      ys = __fadd_rn(ys, 0.5f);
    }
    float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    float* d = a.dets + ((long long)b * K + k) * (N2 + 7);
    for (int v = 0; v < N2 / 2; ++v) {
      const float px = __fadd_rn(rows[k * N2 + 2 * v], xs);
      const float py = __fadd_rn(rows[k * N2 + 2 * v + 1], ys);
      xmin = fminf(xmin, px); xmax = fmaxf(xmax, px);
      ymin = fminf(ymin, py); ymax = fmaxf(ymax, py);
      d[6 + 2 * v] = px;
      d[6 + 2 * v + 1] = py;
    }
    d[0] = xmin; d[1] = ymin; d[2] = xmax; d[3] = ymax;
    d[4] = score;
    d[5] = (float)c;
    d[6 + N2] = a.depth[(long long)b * HW + sp];
    if (a.inds) a.inds[(long long)b * K + k] = sp;
    if (a.clses) a.clses[(long long)b * K + k] = c;
  }
}

inline int tiles_per_image(int C, int H, int W) {
  const long long total = (long long)C * H * W;
  return (int)((total + S1_TILE - 1) / S1_TILE);
}

}  // namespace

extern "C" size_t cp_polydet_decode_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t W,
                                                    int32_t K) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0) return 0;
  return (size_t)B * tiles_per_image(C, H, W) * K * sizeof(unsigned long long);
}

extern "C" int cp_polydet_decode(const float* heat, const float* polys, const float* depth,
                                 const float* reg, int32_t B, int32_t C, int32_t H, int32_t W,
                                 int32_t N2, int32_t K, int32_t rep, float* dets, int64_t* inds,
                                 int32_t* clses, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  return cp_polydet_decode_ex(heat, polys, depth, reg, B, C, H, W, N2, K, rep, 0, dets, inds, clses, workspace,
                              workspace_bytes, stream);
}

extern "C" int cp_polydet_decode_ex(const float* heat, const float* polys, const float* depth,
                                    const float* reg, int32_t B, int32_t C, int32_t H, int32_t W,
                                    int32_t N2, int32_t K, int32_t rep, int32_t cat_spec_poly, float* dets,
                                    int64_t* inds, int32_t* clses, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  CP_CHECK_ARG(heat && polys && depth && dets && workspace);
  CP_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && N2 > 0 && (N2 & 1) == 0 && K > 0);
  CP_CHECK_ARG(rep >= CP_REP_CARTESIAN && rep <= CP_REP_POLAR_FIXED);
  const long long total = (long long)C * H * W;
  if (K > KMAX || total >= (1ll << 31) || B > 65535) return CP_EUNSUPPORTED;
  CP_CHECK_ARG((long long)K <= total);
  if (workspace_bytes < cp_polydet_decode_workspace_bytes(B, C, H, W, K)) return CP_EWORKSPACE;
  const int tiles = tiles_per_image(C, H, W);
  const long long ncand = (long long)tiles * K;
  if (ncand > 64ll * S2_THREADS) return CP_EUNSUPPORTED;
  const size_t row_lds = (size_t)K * N2 * sizeof(float);
  if (row_lds > 60 * 1024) return CP_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* cand = (unsigned long long*)workspace;
  hipLaunchKernelGGL(nms_tile_topk_kernel, dim3(tiles, B), dim3(S1_THREADS), 0, st, heat, cand, C,
                     H, W, K, tiles);
  DecodeArgs a;
  a.cand = cand; a.polys = polys; a.depth = depth; a.reg = reg; a.dets = dets;
  a.inds = (long long*)inds; a.clses = clses;
  a.C = C; a.H = H; a.W = W; a.N2 = N2; a.K = K; a.rep = rep; a.ncand = (int)ncand;
  a.cat_spec = cat_spec_poly ? 1 : 0;
  const int ept = (int)((ncand + S2_THREADS - 1) / S2_THREADS);
  if (ept <= 4) hipLaunchKernelGGL(select_decode_kernel<4>, dim3(B), dim3(S2_THREADS), row_lds, st, a);
  else if (ept <= 16) hipLaunchKernelGGL(select_decode_kernel<16>, dim3(B), dim3(S2_THREADS), row_lds, st, a);
  else if (ept <= 32) hipLaunchKernelGGL(select_decode_kernel<32>, dim3(B), dim3(S2_THREADS), row_lds, st, a);
  else hipLaunchKernelGGL(select_decode_kernel<64>, dim3(B), dim3(S2_THREADS), row_lds, st, a);
  return cp_launch_status();
}
