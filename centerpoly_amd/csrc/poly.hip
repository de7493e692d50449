// Polygon-IoU (literal Weiler-Atherton) and vertex-order losses for gfx950,
// forward and analytic backward.  One wavefront (= one workgroup) per object.
//
// Replaces the Python object loop of PolyLoss.forward
// (reference: src/lib/models/losses.py:868-909), WeilPolygonClipper (:373-628)
// and area (:25-41), which cost 41-173 ms PER OBJECT in the reference.
//
// Literal semantics kept: every point is read as (r, theta);
// subject = prediction sorted by theta with |r|; side codes 1/2/0 for R<0/R==0/R>0;
// OUT iff ts==0 && te in {1,2}, IN iff te==0 && ts in {1,2}, both only when the clip
// edge's end points lie on different sides of the subject edge; slope/intercept
// intersection with the vertical-edge branches; traversal that stops once
// len(used) >= len(remaining inbounds); shoelace with the k=0 term counted twice;
// intersection = [A_clip == 0] * min(A_s, A_c) + A_clip.  Where the reference would
// raise or spin (no outbound for an inbound, inbounds exhausted mid-walk, runaway
// output) the result is DEFINED exactly as in oracle/losses.py.
//
// The intersection's polar round trip (sqrt / atan + quadrant fix, losses.py:469-479)
// followed by area()'s r*cos, r*sin is the identity up to 1e-8 offsets; it is applied
// as the identity here, for values and gradients.
//
// Numerics: side tests and intersections use individually rounded fp32 operations
// (no FMA contraction) so that branch decisions follow the reference's op sequence;
// cos/sin are evaluated in double and rounded to fp32.
#include "cp_common.h"

namespace {

constexpr int NMAX = 64;            // vertices per polygon supported
constexpr float TWO_PI_314 = 6.28f; // the reference's 2*3.14

struct PolyArgs {
  const float* feat;
  const long long* ind;
  const unsigned char* mask;
  const float* target;
  int B, N, HW, M, flags;
};

__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fsub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fdiv(float a, float b) { return __fdiv_rn(a, b); }

// losses.py:378-397
__device__ __forceinline__ int side_code(float ax, float ay, float bx, float by, float qx, float qy) {
  const float R = fsub(fmul(fsub(bx, ax), fsub(qy, ay)), fmul(fsub(by, ay), fsub(qx, ax)));
  return R < 0.f ? 1 : (R == 0.f ? 2 : 0);
}

// losses.py:401-466 (cartesian part).  branch: 0 = subject edge vertical, 1 = clip edge
// vertical, 2 = general.
__device__ __forceinline__ int intersect(float x1, float y1, float x2, float y2, float x3, float y3,
                                         float x4, float y4, float& x, float& y) {
  if (fsub(x2, x1) == 0.f) {
    x = x1;
    const float m2 = fdiv(fsub(y4, y3), fsub(x4, x3));
    const float b2 = fsub(y3, fmul(m2, x3));
    y = fadd(fmul(m2, x), b2);
    return 0;
  }
  if (fsub(x4, x3) == 0.f) {
    x = x3;
    const float m1 = fdiv(fsub(y2, y1), fsub(x2, x1));
    const float b1 = fsub(y1, fmul(m1, x1));
    y = fadd(fmul(m1, x), b1);
    return 1;
  }
  const float m1 = fdiv(fsub(y2, y1), fsub(x2, x1));
  const float b1 = fsub(y1, fmul(m1, x1));
  const float m2 = fdiv(fsub(y4, y3), fsub(x4, x3));
  const float b2 = fsub(y3, fmul(m2, x3));
  x = fdiv(fsub(b2, b1), fsub(m1, m2));
  y = fadd(fmul(m1, x), b1);
  return 2;
}

// Reverse mode of `intersect` w.r.t. the subject edge (x1,y1)-(x2,y2).
__device__ __forceinline__ void intersect_bwd(int branch, float x1, float y1, float x2, float y2,
                                              float x3, float y3, float x4, float y4, float gx,
                                              float gy, float& g1x, float& g1y, float& g2x,
                                              float& g2y) {
  g1x = g1y = g2x = g2y = 0.f;
  if (branch == 0) {
    const float m2 = (y4 - y3) / (x4 - x3);
    g1x = gx + gy * m2;
    return;
  }
  const float dxs = x2 - x1;
  const float m1 = (y2 - y1) / dxs;
  const float b1 = y1 - m1 * x1;
  float g_m1, g_b1;
  if (branch == 1) {
    g_m1 = gy * x3;
    g_b1 = gy;
  } else {
    const float m2 = (y4 - y3) / (x4 - x3);
    const float b2 = y3 - m2 * x3;
    const float d = m1 - m2;
    const float x = (b2 - b1) / d;
    const float gxt = gx + gy * m1;
    g_m1 = gy * x - gxt * x / d;
    g_b1 = gy - gxt / d;
  }
  g1y += g_b1;
  g_m1 += -g_b1 * x1;
  g1x += -g_b1 * m1;
  g2y += g_m1 / dxs;
  g1y -= g_m1 / dxs;
  g2x += -g_m1 * m1 / dxs;
  g1x += g_m1 * m1 / dxs;
}

struct Lds {
  float* sr;     // [NMAX] sorted |r|
  float* sth;    // [NMAX] sorted theta
  float* ssg;    // [NMAX] sign of the raw r
  int* sperm;    // [NMAX] source vertex of sorted slot
  float* sx;     // [NMAX]
  float* sy;
  float* cx;
  float* cy;
  int* out_first;   // [NMAX] first OUT record on subject edge j, -1 if none
  int* in_head;     // [NMAX] first alive IN record on clip edge i, -1 if none
  int* in_tail;     // [NMAX] last IN record on clip edge i (list building only)
  int* rec;         // [maxrec] (type << 16) | (j << 8) | i   type: 0 out, 1 in
  float* rix;       // [maxrec]
  float* riy;
  int* in_next;     // [maxrec] next IN record on the same clip edge (scan order)
  int* in_list;     // [maxrec] IN records in scan order
  int* desc;        // [maxpts] (kind << 16) | idx   kind: 0 S, 1 C, 2 I
  float* vx;        // [maxpts]
  float* vy;
  float* gvx;       // [maxpts] (backward)
  float* gvy;
};

__host__ __device__ inline int max_rec(int N) { return N * N; }
__host__ __device__ inline int max_pts(int N) { return 8 * (N + N); }

__host__ __device__ inline size_t lds_bytes(int N) {
  return (size_t)(11 * NMAX + 5 * max_rec(N) + 5 * max_pts(N)) * 4;
}

__device__ __forceinline__ Lds carve(float* base, int N) {
  Lds L;
  float* p = base;
  L.sr = p; p += NMAX;
  L.sth = p; p += NMAX;
  L.ssg = p; p += NMAX;
  L.sperm = (int*)p; p += NMAX;
  L.sx = p; p += NMAX;
  L.sy = p; p += NMAX;
  L.cx = p; p += NMAX;
  L.cy = p; p += NMAX;
  L.out_first = (int*)p; p += NMAX;
  L.in_head = (int*)p; p += NMAX;
  L.in_tail = (int*)p; p += NMAX;
  const int mr = max_rec(N), mp = max_pts(N);
  L.rec = (int*)p; p += mr;
  L.rix = p; p += mr;
  L.riy = p; p += mr;
  L.in_next = (int*)p; p += mr;
  L.in_list = (int*)p; p += mr;
  L.desc = (int*)p; p += mp;
  L.vx = p; p += mp;
  L.vy = p; p += mp;
  L.gvx = p; p += mp;
  L.gvy = p;
  return L;
}

// Shoelace of losses.py:25-41 on K cartesian vertices (wave-parallel).  Returns R - L.
__device__ __forceinline__ float shoelace_rl(const float* X, const float* Y, int K, int lane) {
  if (K == 0) return 0.f;
  float l = 0.f, r = 0.f;
  for (int k = lane; k < K; k += 64) {
    const int k1 = (k + 1 == K) ? 0 : k + 1;
    l += X[k] * Y[k1];
    r += Y[k] * X[k1];
  }
  l = cp_wave_sum(l);
  r = cp_wave_sum(r);
  if (K >= 2) {            // the k = K term of the doubled array repeats the k = 0 term
    l += X[0] * Y[1];
    r += Y[0] * X[1];
  } else {
    l += X[0] * Y[0];
    r += Y[0] * X[0];
  }
  return r - l;
}

// d area / d vertices for area = |0.5 (R - L)|; ga = upstream gradient on the area.
__device__ __forceinline__ void shoelace_bwd(const float* X, const float* Y, int K, float rl,
                                             float ga, float* GX, float* GY, int lane,
                                             bool accumulate) {
  const float sgn = rl > 0.f ? 1.f : (rl < 0.f ? -1.f : 0.f);
  const float gR = 0.5f * sgn * ga, gL = -gR;
  for (int v = lane; v < K; v += 64) {
    const int vn = (v + 1 == K) ? 0 : v + 1, vp = (v == 0) ? K - 1 : v - 1;
    float gx = gL * Y[vn] + gR * Y[vp];
    float gy = gL * X[vp] + gR * X[vn];
    if (K >= 2) {
      if (v == 0) { gx += gL * Y[1]; gy += gR * X[1]; }
      if (v == 1) { gx += gR * Y[0]; gy += gL * X[0]; }
    }
    if (accumulate) { GX[v] += gx; GY[v] += gy; }
    else { GX[v] = gx; GY[v] = gy; }
  }
}

struct ObjResult {
  float a_clip_rl, a_s_rl, a_c_rl;   // R - L of the three shoelaces
  float a_clip, a_s, a_c, inter, uni, iou;
  int K;
};

// Everything up to the IoU of one object.  pr/tg: the 2N raw prediction / target values in
// registers of lanes (lane v < N holds vertex v).  All lanes must call.
__device__ void clip_object(const Lds& L, int N, int lane, float p_r, float p_t, float t_r,
                            float t_t, ObjResult& res) {
  const int n = N, m = N;
  // ---- sort subject by theta (stable), |r| ----
  __shared__ float tmp_t[NMAX];
  __shared__ float tmp_r[NMAX];
  if (lane < n) { tmp_t[lane] = p_t; tmp_r[lane] = p_r; }
  __syncthreads();
  if (lane < n) {
    int rank = 0;
    for (int k = 0; k < n; ++k) {
      const float tk = tmp_t[k];
      rank += (tk < p_t || (tk == p_t && k < lane)) ? 1 : 0;
    }
    L.sth[rank] = p_t;
    L.sr[rank] = fabsf(p_r);
    L.ssg[rank] = p_r > 0.f ? 1.f : (p_r < 0.f ? -1.f : 0.f);
    L.sperm[rank] = lane;
    const float cs = (float)cos((double)t_t), sn = (float)sin((double)t_t);
    L.cx[lane] = fmul(t_r, cs);
    L.cy[lane] = fmul(t_r, sn);
    L.out_first[lane] = -1;
    L.in_head[lane] = -1;
  }
  __syncthreads();
  if (lane < n) {
    const float th = L.sth[lane], r = L.sr[lane];
    L.sx[lane] = fmul(r, (float)cos((double)th));
    L.sy[lane] = fmul(r, (float)sin((double)th));
  }
  __syncthreads();

  // ---- crossing scan: clip edges outer (sequential), subject edges on lanes ----
  int nrec = 0, nin = 0;
  const int j = lane, j0 = (lane == 0) ? n - 1 : lane - 1;
  float sxe = 0.f, sye = 0.f, sxs = 0.f, sys = 0.f;
  if (lane < n) { sxe = L.sx[j]; sye = L.sy[j]; sxs = L.sx[j0]; sys = L.sy[j0]; }
  for (int i = 0; i < m; ++i) {
    const int i0 = (i == 0) ? m - 1 : i - 1;
    const float cxs = L.cx[i0], cys = L.cy[i0], cxe = L.cx[i], cye = L.cy[i];
    int kind = -1;           // 0 out, 1 in
    float ix = 0.f, iy = 0.f;
    if (lane < n) {
      const int te = side_code(cxs, cys, cxe, cye, sxe, sye);
      const int ts = side_code(cxs, cys, cxe, cye, sxs, sys);
      const bool is_out = ts == 0 && (te == 1 || te == 2);
      const bool is_in = te == 0 && (ts == 1 || ts == 2);
      if (is_out || is_in) {
        const int a = side_code(sxs, sys, sxe, sye, cxe, cye);
        const int b = side_code(sxs, sys, sxe, sye, cxs, cys);
        if (a != b) {
          kind = is_out ? 0 : 1;
          intersect(sxs, sys, sxe, sye, cxs, cys, cxe, cye, ix, iy);
        }
      }
    }
    const unsigned long long bal = __ballot(kind >= 0);
    if (kind >= 0) {
      const int k = nrec + __popcll(bal & ((1ull << lane) - 1ull));
      L.rec[k] = (kind << 16) | (j << 8) | i;
      L.rix[k] = ix;
      L.riy[k] = iy;
      L.in_next[k] = -1;
    }
    nrec += __popcll(bal);
  }
  __syncthreads();

  // ---- index structures + traversal (lane 0; pointer chasing) ----
  __shared__ int sh_K;
  if (lane == 0) {
    int nout = 0;
    // first OUT per subject edge; IN lists per clip edge, both in scan order
    for (int k = 0; k < nrec; ++k) {
      const int rc = L.rec[k];
      const int rj = (rc >> 8) & 255, ri = rc & 255;
      if ((rc >> 16) == 0) {
        ++nout;
        if (L.out_first[rj] < 0) L.out_first[rj] = k;
      } else {
        L.in_list[nin++] = k;
        if (L.in_head[ri] < 0) L.in_head[ri] = k;
        else L.in_next[L.in_tail[ri]] = k;
        L.in_tail[ri] = k;
      }
    }
    const int cap = max_pts(N);
    int cnt = 0;
    auto push = [&](int kind, int idx) {
      if (cnt < cap) L.desc[cnt] = (kind << 16) | idx;
      ++cnt;
    };
    if (nin > 0 && nout > 0) {
      int alive = nin, used = 0, first = 0;   // in_list[first..] : first alive in scan order
      // alive flag: rec type field is rewritten to 2 when an inbound is consumed
      bool done = false;
      while (!done && used < alive) {
        while ((L.rec[L.in_list[first]] >> 16) != 1) ++first;
        const int r0 = L.rec[L.in_list[first]];
        const int stop_j = (r0 >> 8) & 255, stop_i = r0 & 255;
        int jj = stop_j, ii = stop_i;
        bool start = true;
        while (jj != stop_j || ii != stop_i || start) {
          start = false;
          while (L.out_first[jj] < 0) {
            push(0, jj);
            jj = (jj + 1 == n) ? 0 : jj + 1;
            if (cnt >= cap) { done = true; break; }
          }
          if (done) break;
          const int ko = L.out_first[jj];
          push(2, ko);
          ii = L.rec[ko] & 255;
          if (alive == 0) { done = true; break; }
          while (L.in_head[ii] < 0) {
            push(1, ii);
            ii = (ii + 1 == m) ? 0 : ii + 1;
            if (cnt >= cap) { done = true; break; }
          }
          if (done) break;
          const int ki = L.in_head[ii];
          jj = (L.rec[ki] >> 8) & 255;
          push(2, ki);
          L.in_head[ii] = L.in_next[ki];
          L.rec[ki] = (2 << 16) | (L.rec[ki] & 0xffff);
          --alive;
          ++used;
          if (cnt >= cap) { done = true; break; }
        }
      }
    }
    sh_K = cnt < cap ? cnt : cap;
  }
  __syncthreads();
  const int K = sh_K;
  res.K = K;
  for (int k = lane; k < K; k += 64) {
    const int d = L.desc[k], kind = d >> 16, idx = d & 0xffff;
    L.vx[k] = kind == 0 ? L.sx[idx] : (kind == 1 ? L.cx[idx] : L.rix[idx]);
    L.vy[k] = kind == 0 ? L.sy[idx] : (kind == 1 ? L.cy[idx] : L.riy[idx]);
  }
  __syncthreads();
  res.a_clip_rl = shoelace_rl(L.vx, L.vy, K, lane);
  res.a_s_rl = shoelace_rl(L.sx, L.sy, n, lane);
  res.a_c_rl = shoelace_rl(L.cx, L.cy, m, lane);
  res.a_clip = fabsf(0.5f * res.a_clip_rl);
  res.a_s = fabsf(0.5f * res.a_s_rl);
  res.a_c = fabsf(0.5f * res.a_c_rl);
  res.inter = (res.a_clip == 0.f ? fminf(res.a_s, res.a_c) : 0.f) + res.a_clip;
  res.uni = res.a_c + res.a_s - res.inter;
  res.iou = res.inter / (res.uni + 1e-6f);
}

// order term helpers (losses.py:891-904): adjusted angle of lane j and its hinge share.
__device__ __forceinline__ float order_adjust(float ang, int lane, int N, bool& adj) {
  const unsigned long long pos = __ballot(lane < N && ang > 0.f);
  const int first_pos = pos ? __ffsll((long long)pos) - 1 : 64;
  adj = lane < N && ang < 0.f && lane > first_pos;
  return adj ? fadd(ang, TWO_PI_314) : ang;
}

__global__ __launch_bounds__(64) void poly_fwd_kernel(PolyArgs a, float* per_obj_iou,
                                                      float* per_obj_order, float* pred_add) {
  extern __shared__ float dyn[];
  const int o = blockIdx.x, lane = threadIdx.x;
  if (!a.mask[o]) {
    if (lane == 0) { per_obj_iou[o] = 0.f; per_obj_order[o] = 0.f; }
    return;
  }
  const int b = o / a.M, N = a.N;
  const long long sp = a.ind[o];
  const float* fb = a.feat + (long long)b * 2 * N * a.HW + sp;
  float p_r = 0.f, p_t = 0.f, t_r = 0.f, t_t = 0.f;
  if (lane < N) {
    p_r = fb[(long long)(2 * lane) * a.HW];
    p_t = fb[(long long)(2 * lane + 1) * a.HW];
    t_r = a.target[(long long)o * 2 * N + 2 * lane];
    t_t = a.target[(long long)o * 2 * N + 2 * lane + 1];
  }
  if (a.flags & 1) {
    Lds L = carve(dyn, N);
    ObjResult res;
    clip_object(L, N, lane, p_r, p_t, t_r, t_t, res);
    if (lane == 0) per_obj_iou[o] = res.iou;
  } else if (lane == 0) per_obj_iou[o] = 0.f;
  if (a.flags & 2) {
    bool adj;
    const float aj = order_adjust(p_t, lane, N, adj);
    if (lane < N && pred_add) pred_add[(long long)o * 2 * N + 2 * lane + 1] = adj ? TWO_PI_314 : 0.f;
    float h = 0.f;
    for (int k = 0; k < N; ++k) {
      const float ak = __shfl(aj, k, 64);
      if (lane < N - 1 && k >= lane) {
        const float d = fsub(aj, ak);
        if (d > 0.f) h += d;
      }
    }
    h = cp_wave_sum(h);
    if (lane == 0) per_obj_order[o] = h;
  } else if (lane == 0) per_obj_order[o] = 0.f;
}

// sums the per-object values in object order (deterministic) and forms the two losses
__global__ __launch_bounds__(256) void poly_finalize_kernel(const unsigned char* mask, int BM,
                                                            const float* per_obj_iou,
                                                            const float* per_obj_order,
                                                            float* cnt_out, float* iou_loss,
                                                            float* order_loss, int flags) {
  __shared__ double r0[4], r1[4];
  __shared__ int rc[4];
  double s0 = 0, s1 = 0;
  int c = 0;
  for (int o = threadIdx.x; o < BM; o += 256)
    if (mask[o]) {
      ++c;
      s0 += (double)per_obj_iou[o];
      s1 += (double)per_obj_order[o];
    }
  s0 = cp_wave_sum_d(s0);
  s1 = cp_wave_sum_d(s1);
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0) { r0[threadIdx.x >> 6] = s0; r1[threadIdx.x >> 6] = s1; rc[threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t0 = r0[0] + r0[1] + r0[2] + r0[3], t1 = r1[0] + r1[1] + r1[2] + r1[3];
    const int n = rc[0] + rc[1] + rc[2] + rc[3];
    cnt_out[0] = (float)n;
    if (iou_loss) iou_loss[0] = (flags & 1) ? 1.f - (float)t0 / ((float)n + 1e-6f) : 0.f;
    if (order_loss) order_loss[0] = (flags & 2) ? (float)t1 / (10.f * (float)n + 1e-4f) : 0.f;
  }
}

__global__ __launch_bounds__(64) void poly_bwd_kernel(PolyArgs a, const float* cnt,
                                                      const float* g_iou_loss,
                                                      const float* g_order_loss, float* grad_feat) {
  extern __shared__ float dyn[];
  const int o = blockIdx.x, lane = threadIdx.x;
  if (!a.mask[o]) return;
  const int b = o / a.M, N = a.N;
  const long long sp = a.ind[o];
  const float* fb = a.feat + (long long)b * 2 * N * a.HW + sp;
  float* gb = grad_feat + (long long)b * 2 * N * a.HW + sp;
  float p_r = 0.f, p_t = 0.f, t_r = 0.f, t_t = 0.f;
  if (lane < N) {
    p_r = fb[(long long)(2 * lane) * a.HW];
    p_t = fb[(long long)(2 * lane + 1) * a.HW];
    t_r = a.target[(long long)o * 2 * N + 2 * lane];
    t_t = a.target[(long long)o * 2 * N + 2 * lane + 1];
  }
  const float nobj = cnt[0];
  if (a.flags & 1) {
    Lds L = carve(dyn, N);
    ObjResult r;
    clip_object(L, N, lane, p_r, p_t, t_r, t_t, r);
    // loss_iou = 1 - sum(iou)/(n + 1e-6)
    const float g_iou = -g_iou_loss[0] / (nobj + 1e-6f);
    const float U = r.uni + 1e-6f;
    const float g_inter = g_iou * (U + r.inter) / (U * U);
    float g_as = g_iou * (-r.inter) / (U * U);
    float g_aclip = g_inter;
    if (r.a_clip == 0.f) {                       // containment fallback: min(A_s, A_c)
      g_as += g_inter * (r.a_s < r.a_c ? 1.f : (r.a_s == r.a_c ? 0.5f : 0.f));
    }
    // grads on the clip polygon's vertices, then on the subject's own shoelace
    __shared__ float gsx[NMAX], gsy[NMAX];
    if (lane < N) { gsx[lane] = 0.f; gsy[lane] = 0.f; }
    __syncthreads();
    shoelace_bwd(L.sx, L.sy, N, r.a_s_rl, g_as, gsx, gsy, lane, true);
    shoelace_bwd(L.vx, L.vy, r.K, r.a_clip_rl, g_aclip, L.gvx, L.gvy, lane, false);
    __syncthreads();
    // scatter vertex grads to the subject (LDS float atomics; a subject vertex or edge can
    // appear several times in the output polygon)
    for (int k = lane; k < r.K; k += 64) {
      const int d = L.desc[k], kind = d >> 16, idx = d & 0xffff;
      const float gx = L.gvx[k], gy = L.gvy[k];
      if (kind == 0) {
        atomicAdd(&gsx[idx], gx);
        atomicAdd(&gsy[idx], gy);
      } else if (kind == 2) {
        const int rc = L.rec[idx];
        const int rj = (rc >> 8) & 255, ri = rc & 255;
        const int rj0 = rj == 0 ? N - 1 : rj - 1, ri0 = ri == 0 ? N - 1 : ri - 1;
        const float x1 = L.sx[rj0], y1 = L.sy[rj0], x2 = L.sx[rj], y2 = L.sy[rj];
        const float x3 = L.cx[ri0], y3 = L.cy[ri0], x4 = L.cx[ri], y4 = L.cy[ri];
        const int branch = (x2 - x1 == 0.f) ? 0 : ((x4 - x3 == 0.f) ? 1 : 2);
        float g1x, g1y, g2x, g2y;
        intersect_bwd(branch, x1, y1, x2, y2, x3, y3, x4, y4, gx, gy, g1x, g1y, g2x, g2y);
        atomicAdd(&gsx[rj0], g1x);
        atomicAdd(&gsy[rj0], g1y);
        atomicAdd(&gsx[rj], g2x);
        atomicAdd(&gsy[rj], g2y);
      }
    }
    __syncthreads();
    // cartesian -> (|r|, theta) -> raw prediction slots (undo the sort)
    if (lane < N) {
      const float th = L.sth[lane], rr = L.sr[lane];
      const float cs = (float)cos((double)th), sn = (float)sin((double)th);
      const float g_r = gsx[lane] * cs + gsy[lane] * sn;
      const float g_t = -gsx[lane] * rr * sn + gsy[lane] * rr * cs;
      const int src = L.sperm[lane];
      const float gr_raw = g_r * L.ssg[lane];
      if (gr_raw != 0.f) atomicAdd(&gb[(long long)(2 * src) * a.HW], gr_raw);
      if (g_t != 0.f) atomicAdd(&gb[(long long)(2 * src + 1) * a.HW], g_t);
    }
  }
  if (a.flags & 2) {
    bool adj;
    const float aj = order_adjust(p_t, lane, N, adj);
    const float g = g_order_loss[0] / (10.f * nobj + 1e-4f);
    float gs = 0.f;
    for (int k = 0; k < N; ++k) {
      const float ak = __shfl(aj, k, 64);
      if (lane < N) {
        if (lane < N - 1 && k >= lane && fsub(aj, ak) > 0.f) gs += 1.f;   // lane as j
        if (k < N - 1 && lane >= k && fsub(ak, aj) > 0.f) gs -= 1.f;     // lane as k
      }
    }
    if (lane < N && gs != 0.f) atomicAdd(&gb[(long long)(2 * lane + 1) * a.HW], g * gs);
  }
}

int fill_args(PolyArgs& a, const float* feat, const int64_t* ind, const uint8_t* mask,
              const float* target, int32_t B, int32_t N, int32_t H, int32_t W, int32_t M,
              int32_t flags) {
  CP_CHECK_ARG(feat && ind && mask && target);
  CP_CHECK_ARG(B > 0 && N > 0 && H > 0 && W > 0 && M > 0 && (flags & ~3) == 0 && flags != 0);
  if (N > NMAX || N < 3) return CP_EUNSUPPORTED;
  if ((long long)H * W >= (1ll << 31) || (long long)B * M >= (1ll << 24)) return CP_EUNSUPPORTED;
  if (lds_bytes(N) > 150 * 1024) return CP_EUNSUPPORTED;
  a.feat = feat; a.ind = (const long long*)ind; a.mask = mask; a.target = target;
  a.B = B; a.N = N; a.HW = H * W; a.M = M; a.flags = flags;
  return CP_OK;
}

}  // namespace

extern "C" size_t cp_poly_iou_order_workspace_bytes(int32_t B, int32_t M, int32_t N) {
  if (B <= 0 || M <= 0 || N <= 0) return 0;
  return ((size_t)2 * B * M + 4) * sizeof(float);
}

extern "C" int cp_poly_iou_order_forward(const float* feat, const int64_t* ind,
                                         const uint8_t* mask, const float* target, int32_t B,
                                         int32_t N, int32_t H, int32_t W, int32_t M, int32_t flags,
                                         float* iou_loss_out, float* order_loss_out,
                                         float* pred_add_out, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  PolyArgs a;
  const int rc = fill_args(a, feat, ind, mask, target, B, N, H, W, M, flags);
  if (rc != CP_OK) return rc;
  CP_CHECK_ARG(workspace && iou_loss_out && order_loss_out);
  CP_CHECK_ARG(!(flags & 2) || pred_add_out);
  if (workspace_bytes < cp_poly_iou_order_workspace_bytes(B, M, N)) return CP_EWORKSPACE;
  float* ws = (float*)workspace;
  float* per_iou = ws;
  float* per_ord = ws + (size_t)B * M;
  float* cnt = ws + (size_t)2 * B * M;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (flags & 1) ? lds_bytes(N) : 0;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)poly_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(poly_fwd_kernel, dim3(B * M), dim3(64), lds, st, a, per_iou, per_ord,
                     pred_add_out);
  hipLaunchKernelGGL(poly_finalize_kernel, dim3(1), dim3(256), 0, st, mask, B * M, per_iou,
                     per_ord, cnt, iou_loss_out, order_loss_out, flags);
  return cp_launch_status();
}

extern "C" int cp_poly_iou_order_backward(const float* feat, const int64_t* ind,
                                          const uint8_t* mask, const float* target, int32_t B,
                                          int32_t N, int32_t H, int32_t W, int32_t M,
                                          int32_t flags, const float* grad_iou_loss,
                                          const float* grad_order_loss, float* grad_feat,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  PolyArgs a;
  const int rc = fill_args(a, feat, ind, mask, target, B, N, H, W, M, flags);
  if (rc != CP_OK) return rc;
  CP_CHECK_ARG(workspace && grad_iou_loss && grad_order_loss && grad_feat);
  if (workspace_bytes < cp_poly_iou_order_workspace_bytes(B, M, N)) return CP_EWORKSPACE;
  const float* cnt = (const float*)workspace + (size_t)2 * B * M;   // written by the forward
  const size_t lds = (flags & 1) ? lds_bytes(N) : 0;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)poly_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(poly_bwd_kernel, dim3(B * M), dim3(64), lds, (hipStream_t)stream, a, cnt,
                     grad_iou_loss, grad_order_loss, grad_feat);
  return cp_launch_status();
}
