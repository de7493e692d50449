// Output stage of the detection heads at inference:
//     out[b][o][p] = b1[o] + sum_c W1[o][c] * act(y[b][c][p] + b3[c])
// i.e. the bias + ReLU that follows a head's 3x3 convolution and the head's 1x1 convolution
// (reference: the `fc` Sequential of DLASeg, src/lib/models/networks/pose_dla_dcn.py:445-462:
// Conv2d(3x3, bias) -> ReLU -> Conv2d(1x1, bias)) in ONE streaming pass over the 3x3
// convolution's raw output.  The library path reads / writes that 256-channel tensor three times
// (bias+ReLU pass, then a GEMM with 1..32 output rows); this kernel reads it once.
//
// HBM-bound (Cout <= 8) or VALU-bound (Cout = 32): algorithmic bytes = 4*Cin*HW read +
// 4*Cout*HW written per image.  One workgroup = 256 pixels (float4 per lane); its four waves
// split the input channels, partial sums meet in LDS in a fixed order (deterministic).  The
// 1x1 weights arrive transposed ([Cin][Cout]) so a channel's Cout weights are contiguous scalar
// loads (the channel index is made wave-uniform with readfirstlane).
#include "cp_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct HeadArgs {
  const float* y;
  const float* in_bias;
  const float* wt;
  const float* bias;
  float* out;
  long long y_bstride, HW;
  int Cin, Cout, relu_in;
};

template <int CO>
__global__ __launch_bounds__(256) void conv1x1_act_kernel(HeadArgs a) {
  constexpr int RO = CO < 8 ? CO : 8;              // outputs reduced per LDS round
  __shared__ f32x4 red[3][RO][64];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y;
  const long long p = ((long long)blockIdx.x * 64 + lane) * 4;
  const bool ok = p < a.HW;
  const float* yb = a.y + (long long)b * a.y_bstride + (ok ? p : 0);
  f32x4 acc[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) acc[o] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int cpw = (a.Cin + 3) / 4;
  const int c_begin = wid * cpw, c_end = min(a.Cin, c_begin + cpw);
  // Wide heads: the wave's [cpw][CO] weight block is staged once in LDS (wave-private, zero
  // padded) and read back as broadcasts; 32 scalar weights per channel through the scalar cache
  // left this kernel latency-bound (91 us for the 32-output head, SGPR spills).
  constexpr bool WLDS = CO >= 16;
  extern __shared__ float wl_all[];
  float* wl = wl_all + (WLDS ? wid * cpw * CO : 0);
  if (WLDS) {
    for (int e = lane; e < cpw * CO; e += 64) {
      const int cc = e / CO, o = e - cc * CO;
      wl[e] = (c_begin + cc < c_end && o < a.Cout) ? a.wt[(long long)(c_begin + cc) * a.Cout + o] : 0.f;
    }
  }
  constexpr int U = CO >= 32 ? 2 : 4;              // channel rows in flight (32 x U scalar weights live)
  for (int c0 = c_begin; c0 < c_end; c0 += U) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = min(c0 + u, c_end - 1);
      v[u] = *reinterpret_cast<const f32x4*>(yb + (long long)c * a.HW);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {                  // (no `break`: the loop must unroll fully, a
      const int c = min(c0 + u, c_end - 1);        //  runtime-indexed v[] / acc[] goes to scratch)
      const bool live = c0 + u < c_end;            // wave-uniform
      const float ib = a.in_bias ? a.in_bias[c] : 0.f;
      f32x4 x = v[u];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        x[k] += ib;
        if (a.relu_in) x[k] = fmaxf(x[k], 0.f);
      }
      const float* wr = a.wt + (long long)c * a.Cout;
      // <2 x float> arithmetic -> v_pk_fma_f32: two FMAs per instruction (the 32-output head is
      // VALU-bound: 128 FMAs per channel per lane)
      const f32x2 xl = f32x2{x[0], x[1]}, xh = f32x2{x[2], x[3]};
#pragma unroll
      for (int o = 0; o < CO; ++o) {
        const float w = WLDS ? wl[(c - c_begin) * CO + o] * (live ? 1.f : 0.f)
                             : ((live && o < a.Cout) ? wr[o] : 0.f);
        const f32x2 wv = f32x2{w, w};
        const f32x2 lo = wv * xl + f32x2{acc[o][0], acc[o][1]};
        const f32x2 hi = wv * xh + f32x2{acc[o][2], acc[o][3]};
        acc[o] = f32x4{lo[0], lo[1], hi[0], hi[1]};
      }
    }
  }
#pragma unroll
  for (int r0 = 0; r0 < CO; r0 += RO) {
    if (wid > 0) {
#pragma unroll
      for (int o = 0; o < RO; ++o) red[wid - 1][o][lane] = acc[r0 + o];
    }
    __syncthreads();
    if (wid == 0) {
#pragma unroll
      for (int o = 0; o < RO; ++o) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          acc[r0 + o][k] = ((acc[r0 + o][k] + red[0][o][lane][k]) + red[1][o][lane][k]) + red[2][o][lane][k];
      }
    }
    __syncthreads();
  }
  if (wid == 0 && ok) {
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      if (o < a.Cout) {
        const float bo = a.bias ? a.bias[o] : 0.f;
        f32x4 r = acc[o];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] += bo;
        *reinterpret_cast<f32x4*>(a.out + ((long long)b * a.Cout + o) * a.HW + p) = r;
      }
    }
  }
}

template <int CO>
int launch_head(const HeadArgs& a, int B, hipStream_t st) {
  const unsigned gx = (unsigned)((a.HW / 4 + 63) / 64);
  const size_t lds = CO >= 16 ? (size_t)4 * ((a.Cin + 3) / 4) * CO * sizeof(float) : 0;
  if (lds > 96 * 1024) return CP_EUNSUPPORTED;
  if (lds > 32 * 1024)
    (void)hipFuncSetAttribute((const void*)conv1x1_act_kernel<CO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(conv1x1_act_kernel<CO>, dim3(gx, B), dim3(256), lds, st, a);
  return cp_launch_status();
}

}  // namespace

extern "C" int cp_conv1x1_act_forward(const float* y, int64_t y_bstride, const float* in_bias,
                                      int32_t relu_in, const float* w_t, const float* bias,
                                      float* out, int32_t B, int32_t Cin, int32_t Cout, int64_t HW,
                                      void* stream) {
  CP_CHECK_ARG(y && w_t && out && B > 0 && Cin > 0 && Cout > 0 && HW > 0);
  if (Cout > 32 || (HW & 3) != 0 || B > 65535) return CP_EUNSUPPORTED;
  if ((((uintptr_t)y) & 15) != 0 || (((uintptr_t)out) & 15) != 0 || (y_bstride & 3) != 0) return CP_EUNSUPPORTED;
  HeadArgs a;
  a.y = y; a.in_bias = in_bias; a.wt = w_t; a.bias = bias; a.out = out;
  a.y_bstride = y_bstride; a.HW = HW; a.Cin = Cin; a.Cout = Cout; a.relu_in = relu_in ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  if (Cout <= 1) return launch_head<1>(a, B, st);
  if (Cout <= 2) return launch_head<2>(a, B, st);
  if (Cout <= 4) return launch_head<4>(a, B, st);
  if (Cout <= 8) return launch_head<8>(a, B, st);
  if (Cout <= 16) return launch_head<16>(a, B, st);
  return launch_head<32>(a, B, st);
}
