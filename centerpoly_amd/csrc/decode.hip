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
//   image's [C*H*W] heat, applies the NMS test against its 8 neighbours (rows as
//   float4 runs when W % 16 == 0: 18 loads per thread instead of 144), and keeps the
//   slice's K best by an in-register 4x8-bit radix select; ties at the threshold are
//   resolved in index order with a block scan.  The zero bin (almost everything after
//   NMS) is wave-aggregated so LDS histogram atomics never serialise on it.
// Stage 2 (round 4, grid = groups x B): the K best of 4096 candidates (32 tiles at K = 128) per
//   workgroup, in parallel across the map -- the top K of a union is the top K of the subsets'
//   top Ks, so any grouping is exact -- by an 8-bit radix select on the 64-bit (value, ~index)
//   key with PER-WAVE histograms (scores of a heat map share their top byte: one 256-bin
//   histogram took thousands of same-address LDS atomics per pass, the 67 us of the round-3
//   single-workgroup kernel).  Repeated while more than 1024 candidates remain.
// Stage 3 (grid = B): <= 1024 candidates, one per thread: rank = number of greater keys (a
//   broadcast LDS sweep, no histogram, no sort network), winners land sorted; then decode: gather
//   reg/poly/depth with direct strided reads, polar->cartesian in double (the reference uses
//   math.cos on a Python float), bbox = min/max.
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
    int K, int tiles, int vec) {
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
  if (vec && e0 + S1_EPT <= total) {
    // the thread's 16 elements are one aligned run of a row: three rows of 4 float4 + the two edge columns
    const int c = (int)(e0 / HW);
    const int r0 = (int)(e0 - (long long)c * HW);
    const int y = r0 / W, x0 = r0 - y * W;
    const float* pl = hb + (long long)c * HW;
    float hmax[S1_EPT], ctr[S1_EPT];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = y + dy;
      if (yy < 0 || yy >= H) continue;                       // (wave-divergent only at the top / bottom rows)
      const float* row = pl + (long long)yy * W + x0;
      float v[S1_EPT + 2];
#pragma unroll
      for (int q = 0; q < S1_EPT / 4; ++q) {
        const float4 f = *reinterpret_cast<const float4*>(row + 4 * q);
        v[1 + 4 * q] = f.x; v[2 + 4 * q] = f.y; v[3 + 4 * q] = f.z; v[4 + 4 * q] = f.w;
      }
      const bool hl = x0 > 0, hr = x0 + S1_EPT < W;
      v[0] = hl ? row[-1] : v[1];                            // (a missing neighbour: repeat an element of the window)
      v[S1_EPT + 1] = hr ? row[S1_EPT] : v[S1_EPT];
#pragma unroll
      for (int i = 0; i < S1_EPT; ++i) {
        const float m3 = fmaxf(fmaxf(v[i], v[i + 1]), v[i + 2]);
        if (dy == -1 || (dy == 0 && y == 0)) hmax[i] = m3;   // first row visited
        else hmax[i] = fmaxf(hmax[i], m3);
        if (dy == 0) ctr[i] = v[i + 1];
      }
    }
#pragma unroll
    for (int i = 0; i < S1_EPT; ++i) {
      const float kept = (hmax[i] == ctr[i]) ? ctr[i] : 0.f;   // heat * keep, -0 canonicalised to +0
      ov[i] = f2ord(kept + 0.f);
    }
  } else {
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

// ----------------------------------------------------------------- stage 2 ---
// K best of up to MERGE_KEYS candidate keys (one contiguous segment of the candidate list per workgroup), written
// unsorted to out[K] (padded with the empty key 0).  8-bit radix select on the 64-bit key, most significant digit
// first, with one histogram PER WAVE (16 x 256 bins): scores of one heat map share their leading byte, and a single
// histogram serialised thousands of LDS atomics on one address per pass.
constexpr int SEL_THREADS = 256;                       // 4 waves: a workgroup barrier is cheap, 4 histogram copies
constexpr int MERGE_EPT = 16;
constexpr int MERGE_KEYS = SEL_THREADS * MERGE_EPT;    // 4096
constexpr int FINAL_EPT = S2_THREADS / SEL_THREADS;    // the last stage takes <= S2_THREADS = 1024 keys

// Histogram copies: one per wave and per lane & 7.  The scores of a heat map share their leading digits, and LDS atomics on
// one address serialise (~4 cycles per lane: with one copy per wave a pass over 16 keys per thread took ~1.7 us); eight
// copies per wave, 257 entries apart so that equal digits of different copies fall into different banks, cut that by 8.
constexpr int SEL_SUB = 8;
struct SelectScratch {
  uint32_t whist[SEL_THREADS / 64][SEL_SUB][257];
  uint32_t hist[256];
};

// The K-th largest of the workgroup's EPT x SEL_THREADS keys (0 = empty slot; live keys are distinct): every key >= the
// returned threshold is one of the K best.  All SEL_THREADS threads call it.  Two barriers per 8-bit pass: the per-wave
// histograms are zeroed and filled by their own wave, summed by thread = bin, and EVERY wave scans the sum for the digit
// (no broadcast through LDS).
template <int EPT>
__device__ __forceinline__ unsigned long long block_kth_key(const unsigned long long (&key)[EPT], int K, SelectScratch& sc) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  unsigned long long prefix = 0;
  uint32_t need = K;
#pragma unroll 1
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = 56 - 8 * pass;
    const int sub = lane & (SEL_SUB - 1);
    for (int i = lane; i < SEL_SUB * 257; i += 64) (&sc.whist[wid][0][0])[i] = 0;
    const uint32_t zdig = (uint32_t)((((unsigned long long)OZ << 32) >> shift) & 255ull);
    uint32_t zeros = 0;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const unsigned long long k = key[i];
      if (k == 0ull) continue;
      if (pass > 0 && (k >> (shift + 8)) != prefix) continue;
      const uint32_t d = (uint32_t)((k >> shift) & 255ull);
      if (pass < 4 && d == zdig) ++zeros;                  // (value passes: nearly every key carries the +0.0 digit)
      else atomicAdd(&sc.whist[wid][sub][d], 1u);
    }
    if (pass < 4) {
      for (int off = 32; off > 0; off >>= 1) zeros += __shfl_xor(zeros, off, 64);
      if (lane == 0 && zeros) atomicAdd(&sc.whist[wid][0][zdig], zeros);
    }
    __syncthreads();
    {
      uint32_t t = 0;
#pragma unroll
      for (int w = 0; w < SEL_THREADS / 64; ++w)
#pragma unroll
        for (int u = 0; u < SEL_SUB; ++u) t += sc.whist[w][u][tid];       // SEL_THREADS == 256 bins
      sc.hist[tid] = t;
    }
    __syncthreads();
    // suffix scan of the 256 bins in every wave: digit d with count(bins > d) < need <= count(bins >= d)
    uint32_t h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = sc.hist[lane * 4 + i];
    const uint32_t sum4 = h[0] + h[1] + h[2] + h[3];
    uint32_t incl = sum4;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_down(incl, o, 64);
      if (lane + o < 64) incl += t;
    }
    uint32_t above = incl - sum4, dig = 0, rem = 0, inbin = 0;
    const bool owner = above < need && need <= incl;
    if (owner) {
#pragma unroll
      for (int i = 3; i >= 0; --i) {
        if (above < need && need <= above + h[i]) {
          dig = lane * 4 + i;
          rem = need - above;
          inbin = h[i];
        }
        above += h[i];
      }
    }
    const int src = __builtin_ctzll(__builtin_amdgcn_ballot_w64(owner) | (1ull << 63));   // exactly one owner lane
    dig = (uint32_t)__shfl((int)dig, src, 64);
    rem = (uint32_t)__shfl((int)rem, src, 64);
    inbin = (uint32_t)__shfl((int)inbin, src, 64);
    prefix = (prefix << 8) | (unsigned long long)dig;
    need = rem;
    // value passes done and every key with the K-th value wanted (no tie straddles the cut): the index passes would
    // only reproduce the all-zero suffix
    if (pass == 3 && inbin == need) {
      prefix <<= 32;
      break;
    }
  }
  return prefix;
}

__global__ __launch_bounds__(SEL_THREADS) void merge_topk_kernel(const unsigned long long* __restrict__ in,
                                                                unsigned long long* __restrict__ out, int n_per_image,
                                                                int seg, int K, int groups) {
  __shared__ SelectScratch sc;
  __shared__ uint32_t slot;
  const int b = blockIdx.y, grp = blockIdx.x, tid = threadIdx.x;
  const long long base = (long long)b * n_per_image + (long long)grp * seg;
  const int n = min(seg, n_per_image - grp * seg);
  unsigned long long key[MERGE_EPT];
#pragma unroll
  for (int i = 0; i < MERGE_EPT; ++i) {
    const int q = tid + i * SEL_THREADS;
    key[i] = q < n ? in[base + q] : 0ull;
  }
  if (tid == 0) slot = 0;
  unsigned long long* o = out + ((long long)b * groups + grp) * K;
  const unsigned long long kth = block_kth_key<MERGE_EPT>(key, K, sc);
#pragma unroll
  for (int i = 0; i < MERGE_EPT; ++i)
    if (key[i] != 0ull && key[i] >= kth) {
      const uint32_t sl = atomicAdd(&slot, 1u);
      if (sl < (uint32_t)K) o[sl] = key[i];
    }
  __syncthreads();
  for (uint32_t sl = slot + tid; sl < (uint32_t)K; sl += SEL_THREADS) o[sl] = 0ull;
}

// ----------------------------------------------------------------- stage 3 ---
// <= S2_THREADS candidate keys, one per thread: the K winners by the radix select above, then each winner's rank among
// the winners by counting (K x K comparisons, keys are distinct) -- they land sorted, no sort network.  (Ranking all
// 1024 keys against each other is 1 M 64-bit comparisons on one CU: 20 us; measured.)
__global__ __launch_bounds__(SEL_THREADS) void rank_decode_kernel(DecodeArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned long long win[KMAX];
  __shared__ unsigned long long top[KMAX];
  __shared__ float aux[KMAX][3];                // reg x, reg y, depth of a winner: gathered beside its polygon row
  __shared__ uint32_t slot;
  // one dynamic buffer: the selection's histograms first, then (they are dead by then) the gathered polygon rows [K][N2]
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  SelectScratch& sc = *reinterpret_cast<SelectScratch*>(dyn);
  float* rows = reinterpret_cast<float*>(dyn);
  const int b = blockIdx.x, tid = threadIdx.x;
  unsigned long long key[FINAL_EPT];
#pragma unroll
  for (int i = 0; i < FINAL_EPT; ++i) {
    const int q = tid + i * SEL_THREADS;
    key[i] = q < a.ncand ? a.cand[(long long)b * a.ncand + q] : 0ull;
  }
  static_assert(KMAX <= SEL_THREADS, "one thread per winner");
  top[tid] = 0ull;
  win[tid] = 0ull;
  if (tid == 0) slot = 0;
  const unsigned long long kth = block_kth_key<FINAL_EPT>(key, a.K, sc);
#pragma unroll
  for (int i = 0; i < FINAL_EPT; ++i)
    if (key[i] != 0ull && key[i] >= kth) {
      const uint32_t sl = atomicAdd(&slot, 1u);
      if (sl < (uint32_t)KMAX) win[sl] = key[i];
    }
  __syncthreads();
  if (tid < KMAX) {
    const unsigned long long mine = win[tid];
    uint32_t rank = 0;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const u64x2* kp = reinterpret_cast<const u64x2*>(win);
#pragma unroll 8
    for (int j = 0; j < KMAX / 2; ++j) {        // (same address in every lane: LDS broadcast reads)
      const u64x2 kk = kp[j];
      rank += (kk.x > mine) ? 1u : 0u;
      rank += (kk.y > mine) ? 1u : 0u;
    }
    if (mine != 0ull && rank < (uint32_t)a.K) top[rank] = mine;
  }
  __syncthreads();

  // ---- decode the K winners ----
  const int K = a.K, N2 = a.N2, HW = a.H * a.W;
  // (--cat_spec_poly, decode.py:534-537: one polygon per class, the detection's class picks its channel block)
  const float* pb = a.polys + (long long)b * N2 * (a.cat_spec ? a.C : 1) * HW;
  for (int q = tid; q < K * N2; q += SEL_THREADS) {
    const int k = q / N2, j = q - k * N2;
    const uint32_t e = ~(uint32_t)(top[k] & 0xffffffffull);
    const int sp = (int)(e % (uint32_t)HW);
    const int cbase = a.cat_spec ? (int)(e / (uint32_t)HW) * N2 : 0;
    rows[q] = pb[(long long)(cbase + j) * HW + sp];
  }
  if (tid < K) {                                // (issued with the row gathers: one memory latency for all of them)
    const uint32_t e = ~(uint32_t)(top[tid] & 0xffffffffull);
    const int sp = (int)(e % (uint32_t)HW);
    aux[tid][0] = a.reg ? a.reg[(long long)b * 2 * HW + sp] : 0.5f;
    aux[tid][1] = a.reg ? a.reg[(long long)b * 2 * HW + HW + sp] : 0.5f;
    aux[tid][2] = a.depth[(long long)b * HW + sp];
  }
  __syncthreads();
  if (a.rep != CP_REP_CARTESIAN) {
    for (int q = tid; q < K * (N2 / 2); q += SEL_THREADS) {
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
    xs = __fadd_rn(xs, aux[k][0]);               // reg, or the + 0.5 of decode.py:527-528
    ys = __fadd_rn(ys, aux[k][1]);
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
    d[6 + N2] = aux[k][2];
    if (a.inds) a.inds[(long long)b * K + k] = sp;
    if (a.clses) a.clses[(long long)b * K + k] = c;
  }
}

inline int tiles_per_image(int C, int H, int W) {
  const long long total = (long long)C * H * W;
  return (int)((total + S1_TILE - 1) / S1_TILE);
}

}  // namespace

// Candidate lists: level 0 = tiles x K keys per image; every merge level turns segments of MERGE_KEYS keys (a whole
// number of K-key lists) into one K-key list, until <= S2_THREADS keys remain.  Two buffers, ping-pong.
struct DecodePlan {
  int tiles, lists_per_seg, levels;
  long long n0, n1;                       // keys per image in buffer 0 (level 0) and the largest later level
};

static DecodePlan decode_plan(int C, int H, int W, int K) {
  DecodePlan p;
  p.tiles = tiles_per_image(C, H, W);
  p.lists_per_seg = MERGE_KEYS / K;       // K <= KMAX = 256: >= 16 lists per segment
  p.n0 = (long long)p.tiles * K;
  p.n1 = 0;
  p.levels = 0;
  long long lists = p.tiles;
  while (lists * K > S2_THREADS) {
    lists = (lists + p.lists_per_seg - 1) / p.lists_per_seg;
    if (p.levels == 0) p.n1 = lists * K;
    ++p.levels;
  }
  return p;
}

extern "C" size_t cp_polydet_decode_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t W,
                                                    int32_t K) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0 || K > KMAX) return 0;
  const DecodePlan p = decode_plan(C, H, W, K);
  return (size_t)B * (size_t)(p.n0 + p.n1) * sizeof(unsigned long long);
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
  const DecodePlan p = decode_plan(C, H, W, K);
  size_t row_lds = (size_t)K * N2 * sizeof(float);
  if (row_lds > 56 * 1024) return CP_EUNSUPPORTED;
  if (row_lds < sizeof(SelectScratch)) row_lds = sizeof(SelectScratch);
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* buf0 = (unsigned long long*)workspace;
  unsigned long long* buf1 = buf0 + (size_t)B * p.n0;
  const int vec = (W % 16 == 0) && ((reinterpret_cast<uintptr_t>(heat) & 15) == 0) ? 1 : 0;
  hipLaunchKernelGGL(nms_tile_topk_kernel, dim3(p.tiles, B), dim3(S1_THREADS), 0, st, heat, buf0, C,
                     H, W, K, p.tiles, vec);
  const unsigned long long* cur = buf0;
  unsigned long long* nxt = buf1;
  long long lists = p.tiles;
  for (int l = 0; l < p.levels; ++l) {
    const long long groups = (lists + p.lists_per_seg - 1) / p.lists_per_seg;
    hipLaunchKernelGGL(merge_topk_kernel, dim3((unsigned)groups, B), dim3(SEL_THREADS), 0, st, cur, nxt,
                       (int)(lists * K), p.lists_per_seg * K, K, (int)groups);
    lists = groups;
    cur = nxt;
    nxt = (nxt == buf1) ? buf0 : buf1;      // (level l + 1 writes at most as much as level l read)
  }
  DecodeArgs a;
  a.cand = cur; a.polys = polys; a.depth = depth; a.reg = reg; a.dets = dets;
  a.inds = (long long*)inds; a.clses = clses;
  a.C = C; a.H = H; a.W = W; a.N2 = N2; a.K = K; a.rep = rep; a.ncand = (int)(lists * K);
  a.cat_spec = cat_spec_poly ? 1 : 0;
  hipLaunchKernelGGL(rank_decode_kernel, dim3(B), dim3(SEL_THREADS), row_lds, st, a);
  return cp_launch_status();
}
