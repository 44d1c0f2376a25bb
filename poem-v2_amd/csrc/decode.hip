// Row N1 of the scope table: the convolutional glue between the backbone's multi-level features and the head
// (lib/models/POEM.py:167-222 upstream, HRNet branch) --
//   feat_decode : three stride-2 ConvBlocks (3x3 conv + BatchNorm(eval) + ReLU) with lateral adds, bilinear x2, 1x1 conv
//   uv_decode   : three [bilinear x2 -> concat -> 3x3 ConvBlock] stages, 2x2 max-pool, 1x1 conv + sigmoid (21 heat maps)
// as four kernels:
//   upcat_pad_kernel      bilinear x2 of A (align_corners=False) | channel-concat with B | zero border of `pad` pixels
//   conv3x3_kernel        implicit GEMM on the fp32 matrix cores over a zero-bordered input (no boundary predicates:
//                         every tap is base + scalar offset), epilogue = per-channel affine (conv bias and BatchNorm
//                         folded on the host), ReLU, lateral add; output strides let it write straight into the next
//                         conv's bordered input
//   pool_head_kernel      2x2 max-pool + 1x1 conv (40 -> 21) + sigmoid
// (the 1x1 feat_in conv reuses conv1x1_kernel of sample.hip; the heat-map read-out is poem_heatmap_uv of dlt.hip).
//
// conv3x3 as a GEMM: D[co][pixel] = sum_{tap, ci} W[co][ci][tap] * X[ci][pixel + tap offset].
//   A = packed weights  WP[((cot * 9 + tap) * Cin/8 + cc) * 64 + lane] = float4( W[32cot + (lane&31)][8cc + 4(lane>>5) + 0..3][tap] )
//       (one coalesced 1 KiB wave load = four k-steps of a 32-channel tile; rows >= Cout are zero)
//   B = the input itself: lane = output pixel (32 consecutive pixels of the raster), k-step t of chunk cc reads channel
//       8cc + 4(lane>>5) + t at that pixel's tap position -- a dword load whose lane offset is computed once per tile
//       and whose (chunk, tap, k-step) part is a scalar offset of the buffer descriptor.
//   D layout: lane = pixel, registers = output channels -> the epilogue's stores are 128-byte row segments.
// A wave owns CT channel tiles x PT pixel tiles; waves are independent (no LDS, no barriers).
#include "common.h"
#include <algorithm>
#include <type_traits>

__global__ void pack_conv3x3_kernel(const float* __restrict__ w, int Cout, int Cin, float4* __restrict__ out, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int lane = i & 63;
  int f = i >> 6;
  const int cc = f % (Cin / 8);
  f /= (Cin / 8);
  const int tap = f % 9, cot = f / 9;
  const int co = cot * 32 + (lane & 31), ci = 8 * cc + 4 * (lane >> 5);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (co < Cout) {
    const float* p = w + ((size_t)co * Cin + ci) * 9 + tap;
    v = make_float4(p[0], p[9], p[18], p[27]);
  }
  out[i] = v;
}

// The same weights for v_mfma_f32_16x16x4_f32 (16 output channels per tile: Cout = 40 pads to 48 instead of 64, Cout = 80
// not at all instead of to 96): P16[((cot * 9 + tap) * Cin/8 + cc) * 64 + lane] = float2( W[16cot + (lane&15)][8cc + (lane>>4)][tap],
// W[16cot + (lane&15)][8cc + 4 + (lane>>4)][tap] ) -- one 512-byte wave load = the A operands of a chunk's two k-steps.
__global__ void pack_conv3x3_m16_kernel(const float* __restrict__ w, int Cout, int Cin, float2* __restrict__ out, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int lane = i & 63;
  int f = i >> 6;
  const int cc = f % (Cin / 8);
  f /= (Cin / 8);
  const int tap = f % 9, cot = f / 9;
  const int co = cot * 16 + (lane & 15), ci = 8 * cc + (lane >> 4);
  float2 v = make_float2(0.f, 0.f);
  if (co < Cout) {
    const float* p = w + ((size_t)co * Cin + ci) * 9 + tap;
    v = make_float2(p[0], p[36]);
  }
  out[i] = v;
}

// packed image = [32-row fragment image | 16-row fragment image]: which one a launch reads depends on the kernel that takes
// its shape (conv3x3_m16: the 16-row one when it pads fewer output channels)
static size_t conv3x3_floats32(int Cout, int Cin) { return (size_t)((Cout + 31) / 32) * 9 * (Cin / 8) * 64 * 4; }
static size_t conv3x3_floats16(int Cout, int Cin) { return (size_t)((Cout + 15) / 16) * 9 * (Cin / 8) * 64 * 2; }
extern "C" size_t poem_conv3x3_packed_floats(int Cout, int Cin) { return conv3x3_floats32(Cout, Cin) + conv3x3_floats16(Cout, Cin); }
static bool conv3x3_m16(int Cout) { return (Cout + 15) / 16 * 16 < (Cout + 31) / 32 * 32 && (Cout + 15) / 16 <= 5; }

extern "C" hipError_t poem_launch_pack_conv3x3(const float* w, int Cout, int Cin, void* out, hipStream_t s) {
  if (Cin % 8) return hipErrorInvalidValue;
  const int total = ((Cout + 31) / 32) * 9 * (Cin / 8) * 64;
  hipLaunchKernelGGL(pack_conv3x3_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, Cout, Cin, (float4*)out, total);
  const int total16 = ((Cout + 15) / 16) * 9 * (Cin / 8) * 64;
  hipLaunchKernelGGL(pack_conv3x3_m16_kernel, dim3((total16 + 255) / 256), dim3(256), 0, s, w, Cout, Cin,
                     (float2*)((float*)out + conv3x3_floats32(Cout, Cin)), total16);
  return hipGetLastError();
}

struct Conv3Args {
  const float* in;      // (views, Cin, H+2, W+2) zero-bordered
  const float4* wp;     // packed weights (32-row fragment image)
  const float2* wp16;   // ... the 16-row image behind it (pack_conv3x3_m16_kernel)
  const float* scale;   // (cot*32) per-channel multiplier   (BatchNorm folded; 1 without norm)
  const float* shift;   // (cot*32) per-channel offset       (conv bias + BatchNorm folded)
  const float* res;     // optional lateral input (views, Cout, Ho, Wo), added after the activation
  float* out;           // element (n, co, y, x) at out[n * out_ns + co * out_cs + y * out_rs + x + out_off]
  int Cin, Cout, H, W, stride, relu;
  long out_ns;
  int out_cs, out_rs, out_off;
  int views;
  // fused input (conv3x3_lds_kernel<CT, true>): in = [bilinear x2 of up_a (views, Ca, H/2, W/2) | skip_b (views, Cb, H, W)]
  // with the zero border applied while the halo is staged -- the concatenated tensor never exists
  const float* up_a;
  const float* skip_b;
  int Ca, Cb;
  // fused read-out head (conv3x3_lds16_kernel<.., POOL>): instead of `out`, the block's rows go through max_pool2d(2, 2) ->
  // 1x1 conv (J x Cout, bias) -> sigmoid -> hmap (views, J, H/2, W/2); the convolution's own output never reaches HBM
  const float* head_w;
  const float* head_b;
  float* hmap;
  int J;
};

template <int CT, int PT>
__global__ __launch_bounds__(256) void conv3x3_kernel(Conv3Args A) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  const int Ho = A.H / A.stride, Wo = A.W / A.stride, Hp = A.H + 2, Wp = A.W + 2;
  const int ptiles = Ho * Wo / 32, pgroups = ptiles / PT;
  const int cogroups = ((A.Cout + 31) / 32) / CT;
  const long item = (long)blockIdx.x * 4 + wv;
  if (item >= (long)A.views * cogroups * pgroups) return;
  const int pg = (int)(item % pgroups);
  const int cg = (int)((item / pgroups) % cogroups);
  const int n = (int)(item / ((long)pgroups * cogroups));
  const int plane = Hp * Wp, KC = A.Cin / 8;
  const __amdgpu_buffer_rsrc_t xrs = frag_rsrc(A.in + (size_t)n * A.Cin * plane, (unsigned)((size_t)A.Cin * plane * 4));
  const __amdgpu_buffer_rsrc_t wrs = frag_rsrc(A.wp, 0xffffffffu);

  int xoff[PT];      // lane byte offset: channel 4h of the pixel's top-left tap
  int py[PT], px[PT];
#pragma unroll
  for (int p = 0; p < PT; ++p) {
    const int pix = (pg * PT + p) * 32 + j;
    py[p] = pix / Wo;
    px[p] = pix % Wo;
    xoff[p] = (4 * h * plane + py[p] * A.stride * Wp + px[p] * A.stride) * 4;
  }
  f32x16 acc[CT][PT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int p = 0; p < PT; ++p) acc[c][p] = zero16();
  const int wbase = (cg * CT) * 9 * KC * 1024;        // bytes
  // Two-stage software pipeline over the 9 * KC (tap, 8-channel chunk) steps with pinned order (sched_barrier): the
  // operands of step q+1 are in flight while the 4*CT*PT MFMAs of step q issue.
  const int total = 9 * KC;
  int tap = 0, cc = 0, q = 0;
  float4 a0[CT], a1[CT];
  float b0[PT][4], b1[PT][4];
#define POEM_CLOAD(A, B)                                                                                          \
  {                                                                                                               \
    const int woff_ = wbase + (tap * KC + cc) * 1024;                                                             \
    const int soff_ = cc * 8 * plane * 4 + ((tap / 3) * Wp + (tap % 3)) * 4;                                      \
    _Pragma("unroll") for (int c = 0; c < CT; ++c) A[c] = frag_load(wrs, lane * 16, woff_ + c * 9 * KC * 1024);   \
    _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                                 \
      _Pragma("unroll") for (int p = 0; p < PT; ++p)                                                              \
        B[p][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xoff[p], soff_ + t * plane * 4, 0)); \
    if (q + 1 < total) { ++q; if (++cc == KC) { cc = 0; ++tap; } }   /* saturates on the last step */              \
  }
#define POEM_CMMA(A, B)                                                                                           \
  _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                                   \
    _Pragma("unroll") for (int c = 0; c < CT; ++c)                                                                \
      _Pragma("unroll") for (int p = 0; p < PT; ++p) acc[c][p] = mfma32((&A[c].x)[t], B[p][t], acc[c][p]);
  POEM_CLOAD(a0, b0)
  int done = 0;
  for (; done + 1 < total; done += 2) {
    POEM_CLOAD(a1, b1)
    __builtin_amdgcn_sched_barrier(0);
    POEM_CMMA(a0, b0)
    __builtin_amdgcn_sched_barrier(0);
    POEM_CLOAD(a0, b0)
    __builtin_amdgcn_sched_barrier(0);
    POEM_CMMA(a1, b1)
    __builtin_amdgcn_sched_barrier(0);
  }
  if (done < total) { POEM_CMMA(a0, b0) }   // odd step count: the last step is already in (a0, b0)
#undef POEM_CLOAD
#undef POEM_CMMA
  // epilogue: affine (conv bias + BatchNorm), ReLU, lateral add; lane = pixel, register e = channel 8(e>>2) + 4h + (e&3)
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const int cbase = (cg * CT + c) * 32 + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 sc = *reinterpret_cast<const float4*>(A.scale + cbase + 8 * g);
      const float4 sh = *reinterpret_cast<const float4*>(A.shift + cbase + 8 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = cbase + 8 * g + e;
        if (co >= A.Cout) continue;
#pragma unroll
        for (int p = 0; p < PT; ++p) {
          float v = fmaf(acc[c][p][4 * g + e], (&sc.x)[e], (&sh.x)[e]);
          if (A.relu) v = fmaxf(v, 0.f);
          if (A.res) v += A.res[((size_t)n * A.Cout + co) * (Ho * Wo) + py[p] * Wo + px[p]];
          A.out[(size_t)n * A.out_ns + (size_t)co * A.out_cs + py[p] * A.out_rs + px[p] + A.out_off] = v;
        }
      }
    }
  }
}

// Staging of one 8-channel chunk of a convolution's input tile -- (TR + 2) x (W + 2) per channel, zero border included --
// into LDS, shared by the LDS-staged kernels below.  Per staged float everything that does not depend on the chunk is
// worked out once (init): its LDS slot, and for the plain input its byte offset; for the fused [bilinear x2 of up_a | skip_b]
// input the four tap offsets and the four tap weights of F.interpolate(scale_factor=2, align_corners=False) -- the weights
// of an out-of-image position are zero, so the border costs no select -- or the skip tensor's offset.  The chunk's part of
// every address is a scalar offset of a per-view buffer descriptor: `load` is loads only (no VALU), `finish` one multiply and
// three fmas per interpolated float.  (Round 2 unpacked bit fields and formed 64-bit addresses per float and chunk: a fifth of
// the fused kernels' time went into staging arithmetic that the fp32 MFMAs do not overlap with.)
template <bool UPCAT, int MAXLD_ = 7>
struct Conv3Stager {
  static constexpr int MAXLD = MAXLD_;     // staged floats per thread and chunk (stride-1 halo: 8 * 396 / 512 -> 7)
  float st[MAXLD], r1[MAXLD], r2[MAXLD], r3[MAXLD];
  float q0[MAXLD], q1[MAXLD], q2[MAXLD], q3[MAXLD];       // bilinear tap weights
  int o0[MAXLD], o1[MAXLD], o2[MAXLD], o3[MAXLD], ob[MAXLD], sdst[MAXLD];
  __amdgpu_buffer_rsrc_t rs_a, rs_b;
  int Ca, up_plane4, sk_plane4, in_plane4, count, so;

  __device__ __forceinline__ void init(const Conv3Args& A, int n, int y0, int tid, int tplane, int tstride) {
    const int W = A.W, Wp = A.W + 2, h2 = A.H / 2, w2 = A.W / 2;
    count = 8 * tplane;
    Ca = A.Ca;
    up_plane4 = h2 * w2 * 4; sk_plane4 = A.H * W * 4; in_plane4 = (A.H + 2) * Wp * 4;
    if (UPCAT) {
      rs_a = frag_rsrc(A.up_a + (size_t)n * A.Ca * (h2 * w2), (unsigned)((size_t)A.Ca * up_plane4));
      rs_b = frag_rsrc(A.skip_b + (size_t)n * A.Cb * (A.H * W), (unsigned)((size_t)A.Cb * sk_plane4));
    } else {
      rs_a = frag_rsrc(A.in + (size_t)n * A.Cin * ((A.H + 2) * Wp) + (size_t)y0 * Wp, (unsigned)((size_t)A.Cin * in_plane4));
      rs_b = rs_a;
    }
#pragma unroll
    for (int u = 0; u < MAXLD; ++u) {
      const int i = min(tid + 512 * u, count - 1);
      const int chl = i / tplane, o = i % tplane;
      sdst[u] = chl * tstride + o;
      if (!UPCAT) {
        o0[u] = (chl * (A.H + 2) * Wp + o) * 4;
        continue;
      }
      const int y = y0 + o / Wp - 1, x = o % Wp - 1;
      const bool valid = y >= 0 && y < A.H && x >= 0 && x < W;
      const float sy = fmaxf((y + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf((x + 0.5f) * 0.5f - 0.5f, 0.f);
      const int yy0 = (int)sy, xx0 = (int)sx;
      const int yy1 = min(yy0 + 1, h2 - 1), xx1 = min(xx0 + 1, w2 - 1);
      const float ly = sy - (float)yy0, lx = sx - (float)xx0, hy = 1.f - ly, hx = 1.f - lx;
      q0[u] = valid ? hy * hx : 0.f; q1[u] = valid ? hy * lx : 0.f; q2[u] = valid ? ly * hx : 0.f; q3[u] = valid ? ly * lx : 0.f;
      const int base = valid ? chl * up_plane4 : 0;
      o0[u] = valid ? base + (yy0 * w2 + xx0) * 4 : 0; o1[u] = valid ? base + (yy0 * w2 + xx1) * 4 : 0;
      o2[u] = valid ? base + (yy1 * w2 + xx0) * 4 : 0; o3[u] = valid ? base + (yy1 * w2 + xx1) * 4 : 0;
      ob[u] = valid ? chl * sk_plane4 + (y * W + x) * 4 : -1;
    }
  }
  // Stride-2 convolution (padding 1) of an UNBORDERED input (views, C, H, W) = A.skip_b: output rows [yo0, yo0 + 8) need input
  // rows 2 yo0 - 1 .. 2 yo0 + 15 (17) and columns -1 .. W - 1; staged per channel as [row][column parity][column / 2] (row
  // stride 2 RS, RS = W / 2 + 1) so that the 16 consecutive output columns of a unit read consecutive LDS words whatever the
  // tap (input column 2 xo + tx - 1 -> parity tx & 1, half index xo + (tx >> 1)).  Every chunk is of kind 1; positions
  // outside the image get offset -1 -> zero.
  __device__ __forceinline__ void init_s2(const Conv3Args& A, int n, int yo0, int tid, int tstride, int nthreads) {
    const int W = A.W, RS = W / 2 + 1, tplane = 17 * 2 * RS;
    count = 8 * tplane;
    Ca = 0;
    up_plane4 = 0; in_plane4 = 0; sk_plane4 = A.H * W * 4;
    rs_b = frag_rsrc(A.skip_b + (size_t)n * A.Cin * (A.H * W), (unsigned)((size_t)A.Cin * sk_plane4));
    rs_a = rs_b;
#pragma unroll
    for (int u = 0; u < MAXLD; ++u) {
      const int i = min(tid + nthreads * u, count - 1);
      const int chl = i / tplane, o = i % tplane;
      const int r = o / (2 * RS), pp = (o / RS) & 1, xh = o % RS;
      const int y = 2 * yo0 - 1 + r, x = 2 * xh + pp - 1;
      const bool valid = y >= 0 && y < A.H && x >= 0 && x < W;
      sdst[u] = chl * tstride + o;
      ob[u] = valid ? chl * sk_plane4 + (y * W + x) * 4 : -1;
    }
  }
  __device__ __forceinline__ static float ld(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
  }
  // The same per staged float, with the chunk's kind a compile-time constant (KIND 1: plain input or skip tensor, 2:
  // interpolated part) for the kernels with a pinned tap pipeline.  They spread a chunk's requests over its taps: vmcnt
  // retires in order, so a weight fragment requested BEHIND a burst of 28 staging gathers is not usable before all of them
  // are back -- one float's loads per tap keeps the wait in front of every tap at "all but the youngest four"; and a run-time
  // branch on the kind inside that pipeline made the compiler merge the two paths through scratch memory.
  template <int KIND>
  __device__ __forceinline__ void begin_k(int cc) {
    so = __builtin_amdgcn_readfirstlane(!UPCAT ? 8 * cc * in_plane4 : (KIND == 2 ? 8 * cc * up_plane4 : (8 * cc - Ca) * sk_plane4));
  }
  template <int KIND, int u>
  __device__ __forceinline__ void load_k() {
    if (!UPCAT) st[u] = ld(rs_a, o0[u], so);
    else if (KIND == 2) { st[u] = ld(rs_a, o0[u], so); r1[u] = ld(rs_a, o1[u], so); r2[u] = ld(rs_a, o2[u], so); r3[u] = ld(rs_a, o3[u], so); }
    else st[u] = ld(rs_b, max(ob[u], 0), so);
  }
  template <int KIND, int U = 0>
  __device__ __forceinline__ void load_all() {
    if constexpr (U < MAXLD) { load_k<KIND, U>(); load_all<KIND, U + 1>(); }
  }
  // staged floats [LO, HI) (a tap's share of the chunk in the pinned pipelines)
  template <int KIND, int LO, int HI>
  __device__ __forceinline__ void load_range() {
    if constexpr (LO < HI && LO < MAXLD) { load_k<KIND, LO>(); load_range<KIND, LO + 1, HI>(); }
  }
  template <int KIND>
  __device__ __forceinline__ void finish_k() {
    if (!UPCAT) return;
#pragma unroll
    for (int u = 0; u < MAXLD; ++u)
      st[u] = KIND == 2 ? fmaf(q3[u], r3[u], fmaf(q2[u], r2[u], fmaf(q1[u], r1[u], q0[u] * st[u]))) : (ob[u] >= 0 ? st[u] : 0.f);
  }
  __device__ __forceinline__ void store(float* buf, int tid, int nthreads = 512) const {
#pragma unroll
    for (int u = 0; u < MAXLD; ++u)
      if (tid + nthreads * u < count) buf[sdst[u]] = st[u];
  }
};

// The fused [bilinear x2 of up_a | skip_b] staging of the 64^2 and 32^2 levels (uv_decode's last two convolutions, 120 -> 40 and
// 240 -> 80: two thirds of the stage's time), organised by rows instead of by staged float: wave w of the block stages channel w
// of the chunk, a lane one image column of some of the tile's rows.  A lane's column pair (xx0, xx1) and its weights (hx, lx)
// never change; tile row r (image row y0 - 1 + r) interpolates between the source rows (L_{r/2}, L_{r/2+1}) of the TR / 2 + 2
// rows L_i = clamp(y0 / 2 - 1 + i) the tile needs, with the row weight ly of F.interpolate's own formula (wave-uniform at
// W = 64; hy = 1 - ly; both 0 outside the image).  At the top edge the formula's second source row differs from the pattern's,
// with weight ly = 0 there: the value is the same.  8 / 12 coalesced row loads per chunk instead of 28 scattered taps, and ~20
// registers of staging state instead of 98 -- which is what lets TWO blocks share a CU (175 / 202 -> <= 128 VGPRs): with one,
// every block's set-up, first-chunk latency and epilogue were exposed (982 -> 790 us and 739 -> 672 us at 256 views).
// Every staged value is the same expression as Conv3Stager's: fmaf(q3, r3, fmaf(q2, r2, fmaf(q1, r1, q0 * st))), q = (hy hx, hy lx, ly hx, ly lx).
template <int W>
struct Conv3RowStager {
  // lane = (row group rsub = lane / W, column x = lane % W): R = 64 / W groups; a thread's NRT rows are r_k = rsub + R k.  W = 64:
  // six rows from four source rows, everything about a row wave-uniform; W = 32 (eight-row tiles): five rows from six source
  // rows (row r_k between L_k and L_{k+1}), the row weights per lane.
  static constexpr int R = 64 / W, TR = 256 / W, NRT = (TR + 2) / R, NL = (TR + 2) / 2 + 1;
  static_assert((TR + 2) % R == 0 && (W == 64 || W == 32), "tile rows deal evenly to the row groups");
  static constexpr int MAXLD = (2 * NL <= 8) ? 2 * NL : NL;     // load slots per chunk (<= 8: one per tap); two loads a slot when 2 NL > 8
  static constexpr int LPS = (2 * NL <= 8) ? 1 : 2;
  static_assert(MAXLD <= 8 && NRT <= MAXLD, "slots");
  float a[2 * NL], st[NRT];
  float lx, hx;
  int xo0, xo1, xo, dst, Wp;
  float ly[NRT];                           // lower-row weight of tile row r_k, -1 outside the image (wave-uniform at W = 64)
  int Lb[NL], ybase, H;                    // byte offsets of the source rows (kind 2, wave-uniform); image row of r_0
  __amdgpu_buffer_rsrc_t rs_a, rs_b;
  int Ca, up_plane4, sk_plane4, wv, so;

  __device__ __forceinline__ void init(const Conv3Args& A, int n, int y0, int tid, int /*tplane*/, int tstride) {
    const int h2 = A.H / 2, w2 = W / 2, x = (tid & 63) % W, rsub = (tid & 63) / W;
    wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    Wp = W + 2;
    Ca = A.Ca;
    up_plane4 = h2 * w2 * 4; sk_plane4 = A.H * W * 4;
    rs_a = frag_rsrc(A.up_a + (size_t)n * A.Ca * (h2 * w2), (unsigned)((size_t)A.Ca * up_plane4));
    rs_b = frag_rsrc(A.skip_b + (size_t)n * A.Cb * (A.H * W), (unsigned)((size_t)A.Cb * sk_plane4));
    const float sx = fmaxf((x + 0.5f) * 0.5f - 0.5f, 0.f);
    const int xx0 = (int)sx, xx1 = min(xx0 + 1, w2 - 1);
    lx = sx - (float)xx0; hx = 1.f - lx;
    xo0 = xx0 * 4; xo1 = xx1 * 4; xo = x * 4;
    dst = wv * tstride + rsub * Wp + x + 1;
#pragma unroll
    for (int i = 0; i < NL; ++i) Lb[i] = min(max(y0 / 2 - 1 + i, 0), h2 - 1) * w2 * 4;
#pragma unroll
    for (int k = 0; k < NRT; ++k) {
      const int y = y0 - 1 + rsub + R * k;
      const bool valid = y >= 0 && y < A.H;
      const float sy = fmaxf((y + 0.5f) * 0.5f - 0.5f, 0.f);
      ly[k] = valid ? sy - (float)(int)sy : -1.f;
    }
    ybase = y0 - 1 + rsub;
    H = A.H;
  }
  // the two border columns of every staged row are zero for the whole kernel (both buffers): written once
  __device__ __forceinline__ void zero_borders(float* tile, int tid, int tstride) const {
    for (int i = tid; i < 2 * 8 * (TR + 2) * 2; i += 512) {
      const int side = i & 1, r = (i >> 1) % (TR + 2), chl = (i / (2 * (TR + 2))) % 8, buf = i / (16 * (TR + 2));
      tile[buf * 8 * tstride + chl * tstride + r * Wp + side * (Wp - 1)] = 0.f;
    }
  }
  __device__ __forceinline__ static float ld(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
  }
  template <int KIND>
  __device__ __forceinline__ void begin_k(int cc) {
    so = __builtin_amdgcn_readfirstlane(KIND == 2 ? (8 * cc + wv) * up_plane4 : (8 * cc - Ca + wv) * sk_plane4);
  }
  template <int KIND, int u>
  __device__ __forceinline__ void load_k() {
    if constexpr (KIND == 2) {
      if constexpr (LPS == 1) a[u] = ld(rs_a, (u & 1) ? xo1 : xo0, so + Lb[u >> 1]);
      else { a[2 * u] = ld(rs_a, xo0, so + Lb[u]); a[2 * u + 1] = ld(rs_a, xo1, so + Lb[u]); }
    } else if constexpr (u < NRT) {
      a[u] = ld(rs_b, xo + min(max(ybase + R * u, 0), H - 1) * (W * 4), so);
    }
  }
  template <int KIND, int U = 0>
  __device__ __forceinline__ void load_all() {
    if constexpr (U < MAXLD) { load_k<KIND, U>(); load_all<KIND, U + 1>(); }
  }
  template <int KIND>
  __device__ __forceinline__ void finish_k() {
#pragma unroll
    for (int k = 0; k < NRT; ++k) {
      if constexpr (KIND == 2) {
        const int i0 = R == 1 ? (k >> 1) : k, i1 = i0 + 1;      // r_k >> 1
        const float l = fmaxf(ly[k], 0.f), hy = ly[k] < 0.f ? 0.f : 1.f - ly[k];
        const float q0 = hy * hx, q1 = hy * lx, q2 = l * hx, q3 = l * lx;
        st[k] = fmaf(q3, a[2 * i1 + 1], fmaf(q2, a[2 * i1], fmaf(q1, a[2 * i0 + 1], q0 * a[2 * i0])));
      } else {
        st[k] = (unsigned)(ybase + R * k) < (unsigned)H ? a[k] : 0.f;
      }
    }
  }
  __device__ __forceinline__ void store(float* buf, int /*tid*/) const {
#pragma unroll
    for (int k = 0; k < NRT; ++k) buf[dst + R * k * Wp] = st[k];
  }
};

// Stride-1 variant with the input staged in LDS (the three uv_decode convolutions: 9 x Cin x Cout x H x W = 177 M
// multiply-adds per view each).  The direct kernel above re-reads its input for every tap and every channel-tile group
// through caches that do not hold it (PMC: 5.2 GB fetched for the 0.5 GB input of the 120 -> 40 layer).  Here a block of 8
// waves owns 256 raster-consecutive output pixels (256 / W rows) of one view and ALL output channels: per 8-channel chunk
// the (rows + 2) x (W + 2) halo of the zero-bordered input -- contiguous in memory per channel -- is copied to LDS once
// (double-buffered: the next chunk's loads are in flight during the MFMAs), every tap of every wave reads it from there
// (lane = pixel: conflict-free), and the input leaves HBM (rows + 2) / rows times instead of nine times per channel-tile
// group.  Chunk-major accumulation (for each chunk the nine taps) -- the direct kernel sums tap-major; both are within the
// fp32 round-off the tests allow against the reference.
// PIN (round 4; the fused-input 480 -> 160 convolution of the 16^2 level): the taps as the pinned two-stage pipeline of
// conv3x3_lds16_kernel -- tap t + 1's weight fragments and operands requested before tap t's MFMAs, the next chunk's staging
// loads one float per tap behind them.  Left to the compiler the chunk began with the burst of 28 staging gathers and tap 0's
// weights queued behind it (vmcnt retires in order): every chunk of every wave waited an L2 round trip at the barrier's far side.
template <int CT, bool UPCAT, bool PIN = false>
__global__ __launch_bounds__(512) void conv3x3_lds_kernel(Conv3Args A) {
  extern __shared__ __attribute__((aligned(16))) float tile[];      // 2 x 8 x trows x Wp
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 31, h = lane >> 5;
  const int W = A.W, Wp = A.W + 2, KC = A.Cin / 8;
  const int TR = 256 / W, rblocks = A.H / TR;
  const int rb = (int)(blockIdx.x % rblocks), n = (int)(blockIdx.x / rblocks);
  const int y0 = rb * TR, tplane = (TR + 2) * Wp, chunk_floats = 8 * tplane;
  const int pix = wv * 32 + j, py = pix / W, px = pix % W;
  const __amdgpu_buffer_rsrc_t wrs = frag_rsrc(A.wp, 0xffffffffu);
  Conv3Stager<UPCAT> sg;
  sg.init(A, n, y0, tid, tplane, tplane);
  f32x16 acc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) acc[c] = zero16();
  // chunk 0 (each kind stages and stores inside its own branch: no register merges behind it)
  if (UPCAT && A.Ca > 0) {
    sg.template begin_k<2>(0);
    sg.template load_all<2>();
    sg.template finish_k<2>();
    sg.store(tile, tid);
  } else {
    sg.template begin_k<1>(0);
    sg.template load_all<1>();
    sg.template finish_k<1>();
    sg.store(tile, tid);
  }
  __syncthreads();
  const int boff = (4 * h) * tplane + py * Wp + px;
  using SG = Conv3Stager<UPCAT>;
  float4 wa[PIN ? CT : 1], wb[PIN ? CT : 1];
  float ba[4], bb[4];
#define POEM_T32_LOADW(AW, TAP, CC)                                                                             \
  _Pragma("unroll") for (int c = 0; c < CT; ++c) AW[c] = frag_load(wrs, lane * 16, ((c * 9 + (TAP)) * KC + (CC)) * 1024);
#define POEM_T32_LOADB(B, TAP)                                                                                  \
  {                                                                                                             \
    const int toff_ = ((TAP) / 3) * Wp + ((TAP) % 3);                                                           \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) B[t] = tb[t * tplane + toff_];                                \
  }
#define POEM_T32_MMA(AW, B)                                                                                     \
  _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                                 \
    _Pragma("unroll") for (int c = 0; c < CT; ++c) acc[c] = mfma32((&AW[c].x)[t], B[t], acc[c]);
#define POEM_T32_STEP(CUR_A, CUR_B, NXT_A, NXT_B, TAP)                                                          \
  POEM_T32_LOADW(NXT_A, (TAP) + 1, cc) POEM_T32_LOADB(NXT_B, (TAP) + 1)                                         \
  if constexpr (NEXT != 0 && (TAP) < SG::MAXLD) sg.template load_k<NEXT, ((TAP) < SG::MAXLD ? (TAP) : 0)>();    \
  __builtin_amdgcn_sched_barrier(0);                                                                            \
  POEM_T32_MMA(CUR_A, CUR_B)                                                                                    \
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (PIN) { POEM_T32_LOADW(wa, 0, 0) }
  // one chunk; NEXT = kind of the chunk staged meanwhile (0: none, 1: plain input / skip tensor, 2: interpolated part)
  auto chunk = [&](auto next_tag, const int cc) {
    constexpr int NEXT = decltype(next_tag)::value;
    if constexpr (PIN) {
      if constexpr (NEXT != 0) sg.template begin_k<NEXT>(cc + 1);
      const float* tb = tile + (cc & 1) * chunk_floats + boff;
      POEM_T32_LOADB(ba, 0)
      __builtin_amdgcn_sched_barrier(0);
      POEM_T32_STEP(wa, ba, wb, bb, 0) POEM_T32_STEP(wb, bb, wa, ba, 1) POEM_T32_STEP(wa, ba, wb, bb, 2) POEM_T32_STEP(wb, bb, wa, ba, 3)
      POEM_T32_STEP(wa, ba, wb, bb, 4) POEM_T32_STEP(wb, bb, wa, ba, 5) POEM_T32_STEP(wa, ba, wb, bb, 6) POEM_T32_STEP(wb, bb, wa, ba, 7)
      { const int ccn = min(cc + 1, KC - 1); POEM_T32_LOADW(wb, 0, ccn) }
      __builtin_amdgcn_sched_barrier(0);
      POEM_T32_MMA(wa, ba)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < CT; ++c) wa[c] = wb[c];
      if constexpr (NEXT != 0) { sg.template finish_k<NEXT>(); sg.store(tile + ((cc + 1) & 1) * chunk_floats, tid); }
      __syncthreads();
      return;
    }
    if constexpr (NEXT != 0) { sg.template begin_k<NEXT>(cc + 1); sg.template load_all<NEXT>(); }
    const float* tb = tile + (cc & 1) * chunk_floats + boff;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      float4 a[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) a[c] = frag_load(wrs, lane * 16, ((c * 9 + tap) * KC + cc) * 1024);
      const int toff = (tap / 3) * Wp + (tap % 3);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float b = tb[t * tplane + toff];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = mfma32((&a[c].x)[t], b, acc[c]);
      }
    }
    if constexpr (NEXT != 0) { sg.template finish_k<NEXT>(); sg.store(tile + ((cc + 1) & 1) * chunk_floats, tid); }
    __syncthreads();
  };
  {
    const int n_up = UPCAT ? A.Ca / 8 : 0;         // chunks [0, n_up) are interpolated, [n_up, KC) plain / skip tensor
    int cc = 0;
    for (; cc + 1 < n_up; ++cc) chunk(std::integral_constant<int, 2>{}, cc);
    for (; cc + 1 < KC; ++cc) chunk(std::integral_constant<int, 1>{}, cc);
    chunk(std::integral_constant<int, 0>{}, cc);
  }
#undef POEM_T32_LOADW
#undef POEM_T32_LOADB
#undef POEM_T32_MMA
#undef POEM_T32_STEP
  // epilogue: affine (conv bias + BatchNorm), ReLU, lateral add; lane = pixel, register e = channel 8(e>>2) + 4h + (e&3)
  const int Ho = A.H, Wo = A.W, oy = y0 + py;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const int cbase = c * 32 + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 sc = *reinterpret_cast<const float4*>(A.scale + cbase + 8 * g);
      const float4 sh = *reinterpret_cast<const float4*>(A.shift + cbase + 8 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = cbase + 8 * g + e;
        if (co >= A.Cout) continue;
        float v = fmaf(acc[c][4 * g + e], (&sc.x)[e], (&sh.x)[e]);
        if (A.relu) v = fmaxf(v, 0.f);
        if (A.res) v += A.res[((size_t)n * A.Cout + co) * (Ho * Wo) + oy * Wo + px];
        A.out[(size_t)n * A.out_ns + (size_t)co * A.out_cs + oy * A.out_rs + px + A.out_off] = v;
      }
    }
  }
}

// The LDS-staged kernel above on v_mfma_f32_16x16x4_f32, for output widths that are not multiples of 32: Cout = 40 (the last
// uv_decode convolution, 40 % of the stage's FLOPs) multiplied 37 % zeros as two 32-channel tiles, Cout = 80 17 % as three.
// A wave still owns 32 raster-consecutive pixels -- two units of 16 -- and all CT16 channel tiles; lane (j = lane & 15,
// g = lane >> 4) feeds pixel j of a unit with channel 8cc + g (then 8cc + 4 + g) of the chunk.  The staged planes are padded
// to a stride == 16 mod 32 floats so that the two channel planes a 32-lane group reads fall on disjoint banks.
// Result layout: lane (g, j) holds output channels 16c + 4g .. + 3 of pixel j.
typedef float f32x2v __attribute__((ext_vector_type(2)));
template <int CT16, bool UPCAT, int RW = 0, bool POOL = false>
__global__ __launch_bounds__(512, RW ? 4 : 2) void conv3x3_lds16_kernel(Conv3Args A) {
  constexpr bool ROWSG = RW != 0;
  extern __shared__ __attribute__((aligned(16))) float tile[];      // 2 x 8 x tstride
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 15, g = lane >> 4;
  const int W = A.W, Wp = A.W + 2, KC = A.Cin / 8;
  const int TR = 256 / W, rblocks = A.H / TR;
  const int rb = (int)(blockIdx.x % rblocks), n = (int)(blockIdx.x / rblocks);
  const int y0 = rb * TR, tplane = (TR + 2) * Wp, tstride = ((tplane + 15) & ~31) + 16;
  const __amdgpu_buffer_rsrc_t wrs = frag_rsrc(A.wp16, 0xffffffffu);
  static_assert(!ROWSG || UPCAT, "the row stager stages the fused input");
  using SG = std::conditional_t<ROWSG, Conv3RowStager<ROWSG ? RW : 64>, Conv3Stager<UPCAT>>;
  SG sg;
  sg.init(A, n, y0, tid, tplane, tstride);
  if constexpr (ROWSG) sg.zero_borders(tile, tid, tstride);
  f32x4 acc[CT16][2];
#pragma unroll
  for (int c = 0; c < CT16; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) acc[c][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  // chunk 0 (each kind stages and stores inside its own branch: no register merges behind it)
  if (UPCAT && A.Ca > 0) {
    sg.template begin_k<2>(0);
    sg.template load_all<2>();
    sg.template finish_k<2>();
    sg.store(tile, tid);
  } else {
    sg.template begin_k<1>(0);
    sg.template load_all<1>();
    sg.template finish_k<1>();
    sg.store(tile, tid);
  }
  __syncthreads();
  int boff[2], opy[2], opx[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int pix = wv * 32 + 16 * u + j;
    opy[u] = pix / W;
    opx[u] = pix % W;
    boff[u] = g * tstride + opy[u] * Wp + opx[u];
  }
  // Per chunk the nine taps run as a two-stage software pipeline with a pinned order (left alone the compiler issued each
  // tap's LDS reads right in front of the MFMAs that need them and waited: 62 % of the pipe; pinned: see DESIGN): the weight
  // fragments and the four B operands of tap t + 1 are requested before tap t's 4 * CT16 MFMAs issue.
  f32x2v wa[CT16], wb[CT16];
  float ba[4], bb[4];
#define POEM_T16_LOADW(A, TAP, CC)                                                                              \
  _Pragma("unroll") for (int c = 0; c < CT16; ++c)                                                              \
    A[c] = __builtin_bit_cast(f32x2v, __builtin_amdgcn_raw_buffer_load_b64(wrs, lane * 8, ((c * 9 + (TAP)) * KC + (CC)) * 512, 0));
#define POEM_T16_LOADB(B, TAP)                                                                                  \
  {                                                                                                             \
    const int toff_ = ((TAP) / 3) * Wp + ((TAP) % 3);                                                           \
    B[0] = tb[boff[0] + toff_]; B[1] = tb[boff[1] + toff_];                                                     \
    B[2] = tb[boff[0] + 4 * tstride + toff_]; B[3] = tb[boff[1] + 4 * tstride + toff_];                         \
  }
#define POEM_T16_MMA(A, B)                                                                                      \
  _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                                 \
    _Pragma("unroll") for (int c = 0; c < CT16; ++c) acc[c][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c][0], B[u], acc[c][u], 0, 0, 0); \
  _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                                 \
    _Pragma("unroll") for (int c = 0; c < CT16; ++c) acc[c][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c][1], B[2 + u], acc[c][u], 0, 0, 0);
#define POEM_T16_STEP(CUR_A, CUR_B, NXT_A, NXT_B, TAP)                                                          \
  POEM_T16_LOADW(NXT_A, (TAP) + 1, cc) POEM_T16_LOADB(NXT_B, (TAP) + 1)                                         \
  if constexpr (NEXT != 0 && (TAP) < SG::MAXLD) sg.template load_k<NEXT, ((TAP) < SG::MAXLD ? (TAP) : 0)>();    \
  __builtin_amdgcn_sched_barrier(0);                                                                            \
  POEM_T16_MMA(CUR_A, CUR_B)                                                                                    \
  __builtin_amdgcn_sched_barrier(0);
  // one chunk; NEXT = kind of the chunk staged meanwhile (0: none, 1: plain input / skip tensor, 2: interpolated part)
  auto chunk = [&](auto next_tag, const int cc) {
    constexpr int NEXT = decltype(next_tag)::value;
    if constexpr (NEXT != 0) sg.template begin_k<NEXT>(cc + 1);
    const float* tb = tile + (cc & 1) * 8 * tstride;
    POEM_T16_LOADB(ba, 0)
    __builtin_amdgcn_sched_barrier(0);
    POEM_T16_STEP(wa, ba, wb, bb, 0) POEM_T16_STEP(wb, bb, wa, ba, 1) POEM_T16_STEP(wa, ba, wb, bb, 2) POEM_T16_STEP(wb, bb, wa, ba, 3)
    POEM_T16_STEP(wa, ba, wb, bb, 4) POEM_T16_STEP(wb, bb, wa, ba, 5) POEM_T16_STEP(wa, ba, wb, bb, 6) POEM_T16_STEP(wb, bb, wa, ba, 7)
    // last tap: the next chunk's first weight fragments ride behind it (its B operands wait for the barrier)
    { const int ccn = min(cc + 1, KC - 1); POEM_T16_LOADW(wb, 0, ccn) }
    __builtin_amdgcn_sched_barrier(0);
    POEM_T16_MMA(wa, ba)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < CT16; ++c) wa[c] = wb[c];
    if constexpr (NEXT != 0) { sg.template finish_k<NEXT>(); sg.store(tile + ((cc + 1) & 1) * 8 * tstride, tid); }
    __syncthreads();
  };
  POEM_T16_LOADW(wa, 0, 0)
  {
    const int n_up = UPCAT ? A.Ca / 8 : 0;         // chunks [0, n_up) are interpolated, [n_up, KC) plain / skip tensor
    int cc = 0;
    for (; cc + 1 < n_up; ++cc) chunk(std::integral_constant<int, 2>{}, cc);
    for (; cc + 1 < KC; ++cc) chunk(std::integral_constant<int, 1>{}, cc);
    chunk(std::integral_constant<int, 0>{}, cc);
  }
#undef POEM_T16_LOADW
#undef POEM_T16_LOADB
#undef POEM_T16_MMA
#undef POEM_T16_STEP
  // epilogue: lane (g, j) holds channels 16c + 4g + e of pixel (unit u, j)
  const int Ho = A.H, Wo = A.W;
  if constexpr (POOL) {
    // uv_decode's last convolution (POEM.py:203-207 upstream): its (views, 40, 64, 64) output is only ever read by
    // max_pool2d + uv_out + sigmoid.  The block's TR = 4 rows hold two rows of 2 x 2 windows: horizontal maxima by a lane
    // exchange (pixels j, j ^ 1 of a unit), into the idle staging tile as PH[channel][row][column pair]; then (pooled pixel,
    // joint) pairs over the threads: vertical maximum, the J x Cout contraction as pool_head_kernel's fma chain (channels in
    // order from 0), bias, sigmoid -- the same bits as the two-launch form.
    static_assert(RW == 64, "four-row tiles of a 64-pixel-wide map");
    constexpr int PW = 32, PR = 4;
    float* PH = tile;                                 // Cout x PR x PW
    float* HW_ = tile + A.Cout * PR * PW;             // J x Cout weights | J biases  (launcher: fits the staging tile)
    for (int i = tid; i < A.J * A.Cout; i += 512) HW_[i] = A.head_w[i];
    for (int i = tid; i < A.J; i += 512) HW_[A.J * A.Cout + i] = A.head_b[i];
#pragma unroll
    for (int c = 0; c < CT16; ++c) {
      const int cbase = c * 16 + 4 * g;
      const float4 sc = *reinterpret_cast<const float4*>(A.scale + cbase);
      const float4 sh = *reinterpret_cast<const float4*>(A.shift + cbase);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float v = fmaf(acc[c][u][e], (&sc.x)[e], (&sh.x)[e]);
          if (A.relu) v = fmaxf(v, 0.f);
          v = fmaxf(v, __shfl_xor(v, 1, 64));
          if (!(j & 1) && cbase + e < A.Cout) PH[((cbase + e) * PR + opy[u]) * PW + (opx[u] >> 1)] = v;
        }
    }
    __syncthreads();
    const int h2 = Ho / 2, w2 = Wo / 2;
    for (int o = tid; o < A.J * 2 * PW; o += 512) {
      const int jn = o / (2 * PW), pp = o % (2 * PW), prow = pp / PW, pcol = pp % PW;
      const float* ph = PH + (2 * prow) * PW + pcol;
      const float* wj = HW_ + jn * A.Cout;
      float a = 0.f;
      for (int c = 0; c < A.Cout; ++c) a = fmaf(fmaxf(ph[c * PR * PW], ph[c * PR * PW + PW]), wj[c], a);
      const float v = a + HW_[A.J * A.Cout + jn];
      A.hmap[(((size_t)n * A.J + jn) * h2 + (y0 / 2 + prow)) * w2 + pcol] = 1.0f / (1.0f + expf(-v));
    }
    return;
  }
#pragma unroll
  for (int c = 0; c < CT16; ++c) {
    const int cbase = c * 16 + 4 * g;
    const float4 sc = *reinterpret_cast<const float4*>(A.scale + cbase);
    const float4 sh = *reinterpret_cast<const float4*>(A.shift + cbase);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = cbase + e;
      if (co >= A.Cout) continue;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int oy = y0 + opy[u];
        float v = fmaf(acc[c][u][e], (&sc.x)[e], (&sh.x)[e]);
        if (A.relu) v = fmaxf(v, 0.f);
        if (A.res) v += A.res[((size_t)n * A.Cout + co) * (Ho * Wo) + oy * Wo + opx[u]];
        A.out[(size_t)n * A.out_ns + (size_t)co * A.out_cs + oy * A.out_rs + opx[u] + A.out_off] = v;
      }
    }
  }
}

// The three stride-2 ConvBlocks of feat_decode (POEM.py:183-189: 40 -> 80 at 64^2, 80 -> 160 at 32^2, 160 -> 320 at 16^2; each
// 59 MFLOP per view) on the LDS staging, 16x16x4 tiles.  Round 2 ran them on the direct kernel (every tap of every wave a
// strided global gather: 40 TFLOP/s) over zero-bordered copies.  Here a block of 8 waves owns 8 output rows of a view =
// PXB pixel tiles of 32, and CG groups of five 16-channel tiles (wave = one pixel tile x one group): (PXB, CG) = (8, 1),
// (4, 2), (2, 4) for the three layers -- the input is read unbordered (Conv3Stager::init_s2), the output written plain.
template <int PXB, int CG, int MAXLD>
__global__ __launch_bounds__(512, 4) void conv3x3_s2_kernel(Conv3Args A) {
  constexpr int CT16 = 5;
  static_assert(PXB * CG == 8, "eight waves");
  extern __shared__ __attribute__((aligned(16))) float tile[];      // 2 x 8 x tstride
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 15, g = lane >> 4;
  const int W = A.W, Wo = W / 2, Ho = A.H / 2, RS = W / 2 + 1, KC = A.Cin / 8;
  // staged plane of a channel: 17 rows x 2 column parities x RS (see Conv3Stager::init_s2), padded to whole waves of 64 floats
  // so that one wave-wide LDS-DMA request (lane l -> LDS base + 4 l) never straddles two channels; + 16: bank offset of planes
  const int tplane = 17 * 2 * RS, tpad = (tplane + 63) & ~63, tstride = tpad + 16;
  const int rblocks = Ho / 8;
  const int rb = (int)(blockIdx.x % rblocks), n = (int)(blockIdx.x / rblocks);
  const int yo0 = rb * 8;
  const int ptile = wv % PXB, cgrp = wv / PXB;
  const __amdgpu_buffer_rsrc_t wrs = frag_rsrc(A.wp16 + (size_t)cgrp * CT16 * 9 * KC * 64, 0xffffffffu);
  // Staging by LDS-DMA (buffer_load_dword ... lds): the loaded dwords go straight from the memory pipeline into LDS, no
  // registers, no ds_write, and a position outside the image is simply an offset beyond the descriptor's range (the hardware
  // writes zero).  Staged float i = 512 u + tid of a chunk (u < MAXLD): channel i / tpad, plane position i % tpad.
  const int plane4 = A.H * W * 4;
  const __amdgpu_buffer_rsrc_t xrs = frag_rsrc(A.skip_b + (size_t)n * A.Cin * (A.H * W), (unsigned)((size_t)A.Cin * plane4));
  int voff[MAXLD];
#pragma unroll
  for (int u = 0; u < MAXLD; ++u) {
    const int i = tid + 512 * u;
    const int chl = i / tpad, o = i % tpad;
    const int r = o / (2 * RS), pp = (o / RS) & 1, xh = o % RS;
    const int y = 2 * yo0 - 1 + r, x = 2 * xh + pp - 1;
    const bool valid = chl < 8 && o < tplane && y >= 0 && y < A.H && x >= 0 && x < W;
    voff[u] = valid ? chl * plane4 + (y * W + x) * 4 : 0x7ffffff0;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr;
  // request staged floats [LO, HI) of chunk cc into buffer `buf` (all 64 lanes of a wave share the LDS base: wave-uniform)
#define POEM_S2_DMA(U, CC, BUF)                                                                                  \
  if constexpr ((U) < MAXLD) {                                                                                  \
    const int i0_ = __builtin_amdgcn_readfirstlane(512 * (U) + 64 * wv);                                        \
    if (i0_ < 8 * tpad)                                                                                         \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr)(tile + (BUF) * 8 * tstride + (i0_ / tpad) * tstride + i0_ % tpad), 4, \
                                               voff[(U) < MAXLD ? (U) : 0], __builtin_amdgcn_readfirstlane(8 * (CC) * plane4), 0, 0); \
  }
  f32x4 acc[CT16][2];
#pragma unroll
  for (int c = 0; c < CT16; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) acc[c][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    POEM_S2_DMA(0, 0, 0) POEM_S2_DMA(1, 0, 0) POEM_S2_DMA(2, 0, 0) POEM_S2_DMA(3, 0, 0) POEM_S2_DMA(4, 0, 0) POEM_S2_DMA(5, 0, 0)
    POEM_S2_DMA(6, 0, 0) POEM_S2_DMA(7, 0, 0) POEM_S2_DMA(8, 0, 0) POEM_S2_DMA(9, 0, 0) POEM_S2_DMA(10, 0, 0) POEM_S2_DMA(11, 0, 0)
    POEM_S2_DMA(12, 0, 0) POEM_S2_DMA(13, 0, 0) POEM_S2_DMA(14, 0, 0) POEM_S2_DMA(15, 0, 0) POEM_S2_DMA(16, 0, 0) POEM_S2_DMA(17, 0, 0)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int boff[2], opy[2], opx[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int pix = ptile * 32 + 16 * u + j;          // within the block's 8 x Wo output pixels
    opy[u] = pix / Wo;
    opx[u] = pix % Wo;
    boff[u] = g * tstride + (2 * opy[u]) * 2 * RS + opx[u];
  }
  f32x2v wa[CT16], wb[CT16];
  float ba[4], bb[4];
#define POEM_S2_LOADW(AW, TAP, CC)                                                                              \
  _Pragma("unroll") for (int c = 0; c < CT16; ++c)                                                              \
    AW[c] = __builtin_bit_cast(f32x2v, __builtin_amdgcn_raw_buffer_load_b64(wrs, lane * 8, ((c * 9 + (TAP)) * KC + (CC)) * 512, 0));
#define POEM_S2_LOADB(B, TAP)                                                                                   \
  {                                                                                                             \
    const int toff_ = ((TAP) / 3) * 2 * RS + (((TAP) % 3) & 1) * RS + (((TAP) % 3) >> 1);                       \
    B[0] = tb[boff[0] + toff_]; B[1] = tb[boff[1] + toff_];                                                     \
    B[2] = tb[boff[0] + 4 * tstride + toff_]; B[3] = tb[boff[1] + 4 * tstride + toff_];                         \
  }
#define POEM_S2_MMA(AW, B)                                                                                      \
  _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                                 \
    _Pragma("unroll") for (int c = 0; c < CT16; ++c) acc[c][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(AW[c][0], B[u], acc[c][u], 0, 0, 0); \
  _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                                 \
    _Pragma("unroll") for (int c = 0; c < CT16; ++c) acc[c][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(AW[c][1], B[2 + u], acc[c][u], 0, 0, 0);
  // staging requests of the next chunk: two per tap (vmcnt retires in order: a burst in front of the taps would stand
  // between every tap and its weight fragments)
#define POEM_S2_STEP(CUR_A, CUR_B, NXT_A, NXT_B, TAP)                                                           \
  POEM_S2_LOADW(NXT_A, (TAP) + 1, cc) POEM_S2_LOADB(NXT_B, (TAP) + 1)                                           \
  if constexpr (NEXT != 0) { POEM_S2_DMA(2 * (TAP), cc + 1, (cc + 1) & 1) POEM_S2_DMA(2 * (TAP) + 1, cc + 1, (cc + 1) & 1) }  \
  __builtin_amdgcn_sched_barrier(0);                                                                            \
  POEM_S2_MMA(CUR_A, CUR_B)                                                                                     \
  __builtin_amdgcn_sched_barrier(0);
  auto chunk = [&](auto next_tag, const int cc) {
    constexpr int NEXT = decltype(next_tag)::value;
    const float* tb = tile + (cc & 1) * 8 * tstride;
    POEM_S2_LOADB(ba, 0)
    __builtin_amdgcn_sched_barrier(0);
    POEM_S2_STEP(wa, ba, wb, bb, 0) POEM_S2_STEP(wb, bb, wa, ba, 1) POEM_S2_STEP(wa, ba, wb, bb, 2) POEM_S2_STEP(wb, bb, wa, ba, 3)
    POEM_S2_STEP(wa, ba, wb, bb, 4) POEM_S2_STEP(wb, bb, wa, ba, 5) POEM_S2_STEP(wa, ba, wb, bb, 6) POEM_S2_STEP(wb, bb, wa, ba, 7)
    { const int ccn = min(cc + 1, KC - 1); POEM_S2_LOADW(wb, 0, ccn) }
    if constexpr (NEXT != 0) { POEM_S2_DMA(16, cc + 1, (cc + 1) & 1) POEM_S2_DMA(17, cc + 1, (cc + 1) & 1) }
    __builtin_amdgcn_sched_barrier(0);
    POEM_S2_MMA(wa, ba)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < CT16; ++c) wa[c] = wb[c];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's LDS-DMA of the next chunk has landed
    __syncthreads();
  };
  POEM_S2_LOADW(wa, 0, 0)
  {
    int cc = 0;
    for (; cc + 1 < KC; ++cc) chunk(std::integral_constant<int, 1>{}, cc);
    chunk(std::integral_constant<int, 0>{}, cc);
  }
#undef POEM_S2_DMA
#undef POEM_S2_LOADW
#undef POEM_S2_LOADB
#undef POEM_S2_MMA
#undef POEM_S2_STEP
  // epilogue: lane (g, j) holds channels 80 cgrp + 16c + 4g + e of output pixel (unit u, j)
#pragma unroll
  for (int c = 0; c < CT16; ++c) {
    const int cbase = cgrp * (CT16 * 16) + c * 16 + 4 * g;
    const float4 sc = *reinterpret_cast<const float4*>(A.scale + cbase);
    const float4 sh = *reinterpret_cast<const float4*>(A.shift + cbase);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = cbase + e;
      if (co >= A.Cout) continue;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int oy = yo0 + opy[u];
        float v = fmaf(acc[c][u][e], (&sc.x)[e], (&sh.x)[e]);
        if (A.relu) v = fmaxf(v, 0.f);
        if (A.res) v += A.res[((size_t)n * A.Cout + co) * (Ho * Wo) + oy * Wo + opx[u]];
        A.out[(size_t)n * A.out_ns + (size_t)co * A.out_cs + oy * A.out_rs + opx[u] + A.out_off] = v;
      }
    }
  }
}

// The stride-2 ConvBlocks with the staging on a wave of its own (round 4).  What held conv3x3_s2_kernel at 0.43-0.52 of the pipe:
// vmcnt retires IN ORDER, and every wave had its weight fragments (L2 hits) queued behind its own LDS-DMA requests (HBM:
// microseconds) -- the MFMAs of tap t + 1 could not start before the staging requests issued two taps earlier were back, so a tap
// took half an HBM round trip instead of 640 cycles (measured: 24 us per chunk for 5.2 us of MFMAs at the 64^2 level).  Here a
// block is four MFMA waves -- ROWS = 4 output rows of a view = PXB pixel tiles of 32 x CG groups of five 16-channel tiles,
// (PXB, CG) = (4, 1), (2, 2), (1, 4) -- plus a fifth wave that does nothing but stage: all LDS-DMA requests of the next chunk
// at the top of the current one, then s_waitcnt vmcnt(0) and the block barrier.  The MFMA waves have only weight fragments in
// their queues, requested two taps ahead (ring of three) across chunk and tile boundaries, and never wait for them at a barrier.
// Three persistent blocks per CU (one MFMA wave of each per SIMD, barriers and epilogues out of phase) walk their tiles; the
// next tile's first chunk is staged during the last chunk and the epilogue of the current one.
// Staged layout: per channel the 2 ROWS + 1 input rows as they lie in memory (W floats each, no pad column), fetched 16 bytes
// per lane (gfx950's buffer_load_dwordx4 ... lds: a wave request = 1 KB of LDS, 18 / 9 / 5 requests per chunk instead of 80 / 40
// / 24 of dwords at stride 2; every 128-byte line is used whole).  The column left of the image (tap column 0 of output
// column 0) is a select on the operand; rows above / below the image are offsets beyond the descriptor's range (zero).  The
// stride-2 operand reads are 2- to 4-way bank conflicts on 4 reads per 20 MFMAs: not on the critical path.
template <int PXB, int CG, int MAXLD, int ROWS>
__global__ __launch_bounds__(64 * (PXB * CG + 1)) void conv3x3_s2p_kernel(Conv3Args A, int tiles) {
  constexpr int CT16 = 5, NW = PXB * CG, NR = 2 * ROWS + 1;
  extern __shared__ __attribute__((aligned(16))) float tile[];      // 4 + 2 x bstride
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, g = lane >> 4;
  const int W = A.W, Wo = W / 2, Ho = A.H / 2, KC = A.Cin / 8;
  const int cplane = NR * W, cfl = 8 * cplane, bstride = (cfl + 255) & ~255;
  const int rblocks = Ho / ROWS;
  const int plane4 = A.H * W * 4;
  const int my_tiles = (tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  if (wv == NW) {
    // ---- staging wave ----
    int voff[MAXLD];
    auto stage = [&](int t, int cc, int buf) {
      const int n = t / rblocks, yo0 = (t % rblocks) * ROWS;
      const __amdgpu_buffer_rsrc_t xrs = frag_rsrc(A.skip_b + (size_t)n * A.Cin * (A.H * W), (unsigned)((size_t)A.Cin * plane4));
      if (cc == 0) {
#pragma unroll
        for (int u = 0; u < MAXLD; ++u) {
          const int i = 256 * u + 4 * lane;                   // first of this lane's four staged floats
          const int chl = i / cplane, rem = i % cplane;
          const int y = 2 * yo0 - 1 + rem / W, x = rem % W;
          const bool valid = i < cfl && y >= 0 && y < A.H;
          voff[u] = valid ? chl * plane4 + (y * W + x) * 4 : 0x7ffffff0;     // beyond the descriptor's range: zeros
        }
      }
      float* base = tile + 4 + buf * bstride;
      const int soff = __builtin_amdgcn_readfirstlane(8 * cc * plane4);
#if defined(__HIP_DEVICE_COMPILE__)     // (the host pass checks the 16-byte form against the host target and drops the kernel's stub)
#pragma unroll
      for (int u = 0; u < MAXLD; ++u)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(base + 256 * u), 16, voff[u], soff, 0, 0);
#else
      (void)xrs; (void)base; (void)soff;
#endif
    };
    int t = (int)blockIdx.x, cc = 0;
    stage(t, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    const int total = my_tiles * KC;
    for (int f = 1; f <= total; ++f) {
      if (++cc == KC) { cc = 0; t += (int)gridDim.x; }
      if (f < total) stage(t, cc, f & 1);
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    return;
  }
  // ---- MFMA waves ----
  const int ptile = wv % PXB, cgrp = wv / PXB;
  const __amdgpu_buffer_rsrc_t wrs = frag_rsrc(A.wp16 + (size_t)cgrp * CT16 * 9 * KC * 64, 0xffffffffu);
  int boff[2], opy[2], opx[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int pix = ptile * 32 + 16 * u + j;          // within the block's ROWS x Wo output pixels
    opy[u] = pix / Wo;
    opx[u] = pix % Wo;
    boff[u] = g * cplane + (2 * opy[u]) * W + 2 * opx[u] - 1;        // tap (0, 0) of channel g: input (2 y - 1, 2 x - 1)
  }
  f32x2v w0[CT16], w1[CT16], w2[CT16];
  float ba[4], bb[4];
  f32x4 acc[CT16][2];
#define POEM_S2P_LOADW(AW, TAP, CC)                                                                             \
  _Pragma("unroll") for (int c = 0; c < CT16; ++c)                                                              \
    AW[c] = __builtin_bit_cast(f32x2v, __builtin_amdgcn_raw_buffer_load_b64(wrs, lane * 8, ((c * 9 + (TAP)) * KC + (CC)) * 512, 0));
#define POEM_S2P_LOADB(B, TAP)                                                                                  \
  {                                                                                                             \
    const int toff_ = ((TAP) / 3) * W + (TAP) % 3;                                                              \
    B[0] = tb[boff[0] + toff_]; B[1] = tb[boff[1] + toff_];                                                     \
    B[2] = tb[boff[0] + 4 * cplane + toff_]; B[3] = tb[boff[1] + 4 * cplane + toff_];                           \
    if constexpr ((TAP) % 3 == 0) {                                                                             \
      B[0] = opx[0] ? B[0] : 0.f; B[1] = opx[1] ? B[1] : 0.f; B[2] = opx[0] ? B[2] : 0.f; B[3] = opx[1] ? B[3] : 0.f; \
    }                                                                                                           \
  }
#define POEM_S2P_MMA(AW, B)                                                                                     \
  _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                                 \
    _Pragma("unroll") for (int c = 0; c < CT16; ++c) acc[c][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(AW[c][0], B[u], acc[c][u], 0, 0, 0); \
  _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                                 \
    _Pragma("unroll") for (int c = 0; c < CT16; ++c) acc[c][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(AW[c][1], B[2 + u], acc[c][u], 0, 0, 0);
  // tap TAP on (CUR_W, CUR_B); the weights of tap TAP + 2 (of the next chunk past tap 6) and the operands of tap TAP + 1 requested
#define POEM_S2P_STEP(CUR_W, CUR_B, NXT_W, NXT_B, TAP)                                                          \
  POEM_S2P_LOADW(NXT_W, ((TAP) + 2) % 9, ((TAP) + 2 >= 9 ? ccn : cc))                                           \
  if constexpr ((TAP) < 8) POEM_S2P_LOADB(NXT_B, (TAP) + 1)                                                     \
  __builtin_amdgcn_sched_barrier(0);                                                                            \
  POEM_S2P_MMA(CUR_W, CUR_B)                                                                                    \
  __builtin_amdgcn_sched_barrier(0);
  POEM_S2P_LOADW(w0, 0, 0)
  POEM_S2P_LOADW(w1, 1, 0)
  asm volatile("s_barrier" ::: "memory");                   // chunk 0 of the first tile has landed
  int f = 0, t = (int)blockIdx.x;
  for (int it = 0; it < my_tiles; ++it, t += (int)gridDim.x) {
#pragma unroll
    for (int c = 0; c < CT16; ++c)
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[c][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int cc = 0; cc < KC; ++cc, ++f) {
      const float* tb = tile + 4 + (f & 1) * bstride;
      const int ccn = cc + 1 == KC ? 0 : cc + 1;
      POEM_S2P_LOADB(ba, 0)
      __builtin_amdgcn_sched_barrier(0);
      POEM_S2P_STEP(w0, ba, w2, bb, 0) POEM_S2P_STEP(w1, bb, w0, ba, 1) POEM_S2P_STEP(w2, ba, w1, bb, 2)
      POEM_S2P_STEP(w0, bb, w2, ba, 3) POEM_S2P_STEP(w1, ba, w0, bb, 4) POEM_S2P_STEP(w2, bb, w1, ba, 5)
      POEM_S2P_STEP(w0, ba, w2, bb, 6) POEM_S2P_STEP(w1, bb, w0, ba, 7) POEM_S2P_STEP(w2, ba, w1, bb, 8)
      // this chunk's operands have been read (the MFMAs consumed them); the weights in flight stay in flight
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // epilogue: lane (g, j) holds channels 80 cgrp + 16c + 4g + e of output pixel (unit u, j); per channel tile the eight
    // residual values first, then the eight stores
    const int n = t / rblocks, yo0 = (t % rblocks) * ROWS;
    int ey[2] = {opy[0], opy[1]}, ex[2] = {opx[0], opx[1]}, eg = g;
    // (opaque to the optimiser: the 40 store addresses are invariant over the tile loop and would be hoisted into 80 registers)
    asm volatile("" : "+v"(ey[0]), "+v"(ey[1]), "+v"(ex[0]), "+v"(ex[1]), "+v"(eg));
    // residual and result through buffer descriptors of the view: a lane offset per unit, the channel in the scalar offset -- all
    // 40 residual loads in flight at once (one HBM round trip per tile; 64-bit addresses would cost 80 registers)
    const int po4 = Ho * Wo * 4, cb0 = cgrp * (CT16 * 16) + 4 * eg;     // Cout % 80 == 0: every channel tile of a group exists
    const __amdgpu_buffer_rsrc_t ors = frag_rsrc(A.out + (size_t)n * A.out_ns + A.out_off, 0xffffffffu);
    int rvo[2], ovo[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      rvo[u] = cb0 * po4 + ((yo0 + ey[u]) * Wo + ex[u]) * 4;
      ovo[u] = (cb0 * A.out_cs + (yo0 + ey[u]) * A.out_rs + ex[u]) * 4;
    }
    float rv[CT16][4][2];
    if (A.res) {
      const __amdgpu_buffer_rsrc_t rrs = frag_rsrc(A.res + (size_t)n * A.Cout * (Ho * Wo), (unsigned)(A.Cout * po4));
#pragma unroll
      for (int c = 0; c < CT16; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int u = 0; u < 2; ++u)
            rv[c][e][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, rvo[u], (16 * c + e) * po4, 0));
    } else {
#pragma unroll
      for (int c = 0; c < CT16; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) rv[c][e][0] = rv[c][e][1] = 0.f;
    }
#pragma unroll
    for (int c = 0; c < CT16; ++c) {
      const float4 sc = *reinterpret_cast<const float4*>(A.scale + cb0 + 16 * c);
      const float4 sh = *reinterpret_cast<const float4*>(A.shift + cb0 + 16 * c);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float v = fmaf(acc[c][u][e], (&sc.x)[e], (&sh.x)[e]);
          if (A.relu) v = fmaxf(v, 0.f);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + rv[c][e][u]), ors, ovo[u], (16 * c + e) * A.out_cs * 4, 0);
        }
    }
  }
#undef POEM_S2P_LOADW
#undef POEM_S2P_LOADB
#undef POEM_S2P_MMA
#undef POEM_S2P_STEP
}

// Shapes conv3x3_s2_kernel takes: 80 / 160 / 320 output channels on 8-row output tiles of 32 / 16 / 8 columns.
static int g_row_stager = 3;            // A/B switch: 0 = Conv3Stager for every fused-input convolution
extern "C" void poem_decode_row_stager(int on) { g_row_stager = on; }      // bit 0: at W = 64, bit 1: at W = 32
static int g_pool_fused = 1;            // A/B switch: 0 = uv_decode's last convolution and the read-out head as two launches
extern "C" void poem_decode_pool_fused(int on) { g_pool_fused = on != 0; }
static int g_pin32 = 1;                 // A/B switch: 0 = compiler-scheduled taps in conv3x3_lds_kernel<5, true>
extern "C" void poem_decode_pin32(int on) { g_pin32 = on != 0; }
static int g_s2_staging_wave = 1;       // A/B switch: 0 = conv3x3_s2_kernel (every wave stages and multiplies)
static int g_s2_blocks_per_cu = 0;       // 0: by the tile count (below); 2..4 forced (A/B)
extern "C" void poem_decode_s2_staging_wave(int on) {
  if (on >= 2) { g_s2_blocks_per_cu = on; return; }
  g_s2_staging_wave = on != 0;
  g_s2_blocks_per_cu = 0;
}
// Persistent blocks per CU of conv3x3_s2p_kernel.  Three fit (LDS, registers); but with 4 tiles per CU (the 32^2 level at 256
// views) three blocks walk {2, 1, 1} tiles and the long one finishes alone -- one MFMA wave per SIMD, 0.75 of the pipe --
// where two blocks walk {2, 2}: measured 157 vs 140 us; with 8 tiles per CU {3, 3, 2} beats {4, 4} (153 vs 158 us).
static int s2_blocks_per_cu(int tiles, int cus) {
  if (g_s2_blocks_per_cu) return g_s2_blocks_per_cu;
  const int tpc = (tiles + cus - 1) / cus;
  return tpc >= 2 && tpc % 3 == 1 ? 2 : 3;
}
static int conv3x3_s2_shape(int Cout, int H, int W) {
  if (H != W || H % 16) return 0;
  if (Cout == 80 && W == 64) return 1;
  if (Cout == 160 && W == 32) return 2;
  if (Cout == 320 && W == 16) return 3;
  return 0;
}

extern "C" hipError_t poem_launch_conv3x3_down2(const float* in, const void* wp, const float* scale, const float* shift,
                                                const float* res, float* out, int views, int Cin, int Cout, int H, int W,
                                                int relu, long out_ns, int out_cs, int out_rs, int out_off, hipStream_t s) {
  const int shape = conv3x3_s2_shape(Cout, H, W);
  if (!shape || Cin % 8) return hipErrorNotSupported;
  Conv3Args a{nullptr, (const float4*)wp, (const float2*)((const float*)wp + conv3x3_floats32(Cout, Cin)), scale, shift, res, out, Cin, Cout,
              H, W, 2, relu, out_ns, out_cs, out_rs, out_off, views, nullptr, in, 0, Cin};
  const int RS = W / 2 + 1, tplane = 17 * 2 * RS, tstride = ((tplane + 63) & ~63) + 16;
  const size_t lds = (size_t)2 * 8 * tstride * sizeof(float);
  static std::atomic<unsigned long long> optin[6];
  if (g_s2_staging_wave) {
    // three persistent 5-wave blocks per CU (four MFMA waves + the staging wave), tiles of four output rows
    constexpr int ROWS = 4;
    const int tiles = views * (H / 2 / ROWS);
    const int cfl = 8 * (2 * ROWS + 1) * W;
    const size_t lds_p = (size_t)(4 + 2 * ((cfl + 255) & ~255)) * sizeof(float);
    const dim3 grid((unsigned)std::min(tiles, s2_blocks_per_cu(tiles, poem_device_cus()) * poem_device_cus())), block(320);
    if (shape == 1) {
      auto k = conv3x3_s2p_kernel<4, 1, 18, ROWS>;
      if (hipError_t e = poem_optin_lds(reinterpret_cast<const void*>(k), lds_p, optin[3]); e != hipSuccess) return e;
      hipLaunchKernelGGL(k, grid, block, lds_p, s, a, tiles);
    } else if (shape == 2) {
      auto k = conv3x3_s2p_kernel<2, 2, 9, ROWS>;
      if (hipError_t e = poem_optin_lds(reinterpret_cast<const void*>(k), lds_p, optin[4]); e != hipSuccess) return e;
      hipLaunchKernelGGL(k, grid, block, lds_p, s, a, tiles);
    } else {
      auto k = conv3x3_s2p_kernel<1, 4, 5, ROWS>;
      if (hipError_t e = poem_optin_lds(reinterpret_cast<const void*>(k), lds_p, optin[5]); e != hipSuccess) return e;
      hipLaunchKernelGGL(k, grid, block, lds_p, s, a, tiles);
    }
    return hipGetLastError();
  }
  const dim3 grid((unsigned)(views * (H / 2 / 8))), block(512);
  if (shape == 1) {
    auto k = conv3x3_s2_kernel<8, 1, 18>;
    if (hipError_t e = poem_optin_lds(reinterpret_cast<const void*>(k), lds, optin[0]); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, block, lds, s, a);
  } else if (shape == 2) {
    auto k = conv3x3_s2_kernel<4, 2, 10>;
    if (hipError_t e = poem_optin_lds(reinterpret_cast<const void*>(k), lds, optin[1]); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, block, lds, s, a);
  } else {
    auto k = conv3x3_s2_kernel<2, 4, 5>;
    if (hipError_t e = poem_optin_lds(reinterpret_cast<const void*>(k), lds, optin[2]); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, block, lds, s, a);
  }
  return hipGetLastError();
}

static bool conv3x3_lds_ok(int Cout, int H, int W) {
  return (Cout + 31) / 32 <= 5 && W <= 64 && 256 % W == 0 && H % (256 / W) == 0 && 8 * (256 / W + 2) * (W + 2) <= 7 * 512;
}

template <bool UPCAT>
static hipError_t launch_conv3x3_lds(const Conv3Args& a, hipStream_t s) {
  const int TR = 256 / a.W, cot = (a.Cout + 31) / 32;
  const dim3 grid((unsigned)(a.views * (a.H / TR))), block(512);
  if (conv3x3_m16(a.Cout)) {
    const int tplane = (TR + 2) * (a.W + 2), tstride = ((tplane + 15) & ~31) + 16;
    const size_t lds16 = (size_t)2 * 8 * tstride * sizeof(float);
    if constexpr (UPCAT) {
      // W == 64 / 32 (four- / eight-row tiles), whole chunks of either kind: staging by rows, two blocks per CU (Conv3RowStager)
      if (a.Ca % 8 == 0 && a.Cb % 8 == 0 && ((a.W == 64 && (g_row_stager & 1)) || (a.W == 32 && (g_row_stager & 2)))) {
#define POEM_CONVL16R(CTV)                                                                                    \
        if (a.W == 64) hipLaunchKernelGGL((conv3x3_lds16_kernel<CTV, true, 64>), grid, block, lds16, s, a);   \
        else hipLaunchKernelGGL((conv3x3_lds16_kernel<CTV, true, 32>), grid, block, lds16, s, a)
        switch ((a.Cout + 15) / 16) {
          case 1: POEM_CONVL16R(1); break;
          case 3: POEM_CONVL16R(3); break;
          default: POEM_CONVL16R(5); break;
        }
#undef POEM_CONVL16R
        return hipGetLastError();
      }
    }
#define POEM_CONVL16(CTV) hipLaunchKernelGGL((conv3x3_lds16_kernel<CTV, UPCAT>), grid, block, lds16, s, a)
    switch ((a.Cout + 15) / 16) {
      case 1: POEM_CONVL16(1); break;
      case 3: POEM_CONVL16(3); break;
      default: POEM_CONVL16(5); break;
    }
#undef POEM_CONVL16
    return hipGetLastError();
  }
  const size_t lds = (size_t)2 * 8 * (TR + 2) * (a.W + 2) * sizeof(float);
#define POEM_CONVL(CTV) hipLaunchKernelGGL((conv3x3_lds_kernel<CTV, UPCAT>), grid, block, lds, s, a)
  switch (cot) {
    case 1: POEM_CONVL(1); break;
    case 2: POEM_CONVL(2); break;
    case 3: POEM_CONVL(3); break;
    case 4: POEM_CONVL(4); break;
    default:
      if (UPCAT && g_pin32) hipLaunchKernelGGL((conv3x3_lds_kernel<5, UPCAT, true>), grid, block, lds, s, a);
      else POEM_CONVL(5);
      break;
  }
#undef POEM_CONVL
  return hipGetLastError();
}

// conv3x3(stride 1) of [bilinear x2 of a | b] without materialising the concatenated, bordered input.
// hipErrorNotSupported: the caller runs poem_launch_upcat_pad + poem_launch_conv3x3 instead.
extern "C" hipError_t poem_launch_upcat_conv3x3(const float* a_half, int Ca, const float* b_full, int Cb, const void* wp,
                                                const float* scale, const float* shift, float* out, int views, int Cout,
                                                int H, int W, int relu, long out_ns, int out_cs, int out_rs, int out_off,
                                                hipStream_t s) {
  if (Ca % 8 || Cb % 8 || Ca + Cb <= 0 || H % 2 || W % 2 || !conv3x3_lds_ok(Cout, H, W) || (long)(H / 2) * (W / 2) >= (1 << 20))
    return hipErrorNotSupported;
  Conv3Args a{nullptr, (const float4*)wp, (const float2*)((const float*)wp + conv3x3_floats32(Cout, Ca + Cb)), scale, shift, nullptr, out,
              Ca + Cb, Cout, H, W, 1, relu, out_ns, out_cs, out_rs, out_off, views, a_half, b_full, Ca, Cb};
  return launch_conv3x3_lds<true>(a, s);
}

// The same with the read-out head in the epilogue (Conv3Args::hmap): W == 64, Cout <= 48 on 16-row tiles, whole chunks of both
// input kinds.  hipErrorNotSupported: the caller runs poem_launch_upcat_conv3x3 + poem_launch_pool_head.
extern "C" hipError_t poem_launch_upcat_conv3x3_pool_head(const float* a_half, int Ca, const float* b_full, int Cb, const void* wp,
                                                          const float* scale, const float* shift, const float* head_w,
                                                          const float* head_b, float* hmap, int views, int Cout, int J, int H, int W,
                                                          int relu, hipStream_t s) {
  if (Ca % 8 || Cb % 8 || Ca <= 0 || Cb <= 0 || W != 64 || H % 4 || !conv3x3_lds_ok(Cout, H, W) || !conv3x3_m16(Cout) ||
      (Cout + 15) / 16 != 3 || J < 1 || J > 32 || !(g_row_stager & 1) || !g_pool_fused)
    return hipErrorNotSupported;
  const int TR = 256 / W, tplane = (TR + 2) * (W + 2), tstride = ((tplane + 15) & ~31) + 16;
  const size_t lds16 = (size_t)2 * 8 * tstride * sizeof(float);
  if ((size_t)(Cout * 4 * 32 + J * Cout + J) * sizeof(float) > lds16) return hipErrorNotSupported;
  Conv3Args a{nullptr, (const float4*)wp, (const float2*)((const float*)wp + conv3x3_floats32(Cout, Ca + Cb)), scale, shift, nullptr, nullptr,
              Ca + Cb, Cout, H, W, 1, relu, 0, 0, 0, 0, views, a_half, b_full, Ca, Cb, head_w, head_b, hmap, J};
  hipLaunchKernelGGL((conv3x3_lds16_kernel<3, true, 64, true>), dim3((unsigned)(views * (H / TR))), dim3(512), lds16, s, a);
  return hipGetLastError();
}

// in (views, Cin, H+2, W+2) zero-bordered; out element strides as in Conv3Args.  stride 1 or 2, H, W even,
// (H/stride)*(W/stride) % 32 == 0, Cin % 8 == 0.
extern "C" hipError_t poem_launch_conv3x3(const float* in, const void* wp, const float* scale, const float* shift,
                                          const float* res, float* out, int views, int Cin, int Cout, int H, int W,
                                          int stride, int relu, long out_ns, int out_cs, int out_rs, int out_off,
                                          hipStream_t s) {
  if (Cin % 8 || (stride != 1 && stride != 2) || H % stride || W % stride) return hipErrorInvalidValue;
  const int Ho = H / stride, Wo = W / stride;
  if ((Ho * Wo) % 32) return hipErrorInvalidValue;
  if ((size_t)Cin * (H + 2) * (W + 2) * 4 >= (1ull << 31)) return hipErrorInvalidValue;
  Conv3Args a{in, (const float4*)wp, (const float2*)((const float*)wp + conv3x3_floats32(Cout, Cin)), scale, shift, res, out, Cin, Cout, H, W,
              stride, relu, out_ns, out_cs, out_rs, out_off, views, nullptr, nullptr, 0, 0};
  const int cot = (Cout + 31) / 32, ptiles = Ho * Wo / 32;
  if (stride == 1 && conv3x3_lds_ok(Cout, H, W)) return launch_conv3x3_lds<false>(a, s);
  const int pt = (ptiles % 2 == 0) ? 2 : 1;
  const int ct = (cot % 5 == 0) ? 5 : (cot % 3 == 0) ? 3 : (cot % 2 == 0) ? 2 : 1;
  const long items = (long)views * (cot / ct) * (ptiles / pt);
  const dim3 grid((unsigned)((items + 3) / 4)), block(256);
#define POEM_CONV(CTV, PTV) hipLaunchKernelGGL((conv3x3_kernel<CTV, PTV>), grid, block, 0, s, a)
  if (pt == 2) {
    if (ct == 5) POEM_CONV(5, 2); else if (ct == 3) POEM_CONV(3, 2); else if (ct == 2) POEM_CONV(2, 2); else POEM_CONV(1, 2);
  } else {
    if (ct == 5) POEM_CONV(5, 1); else if (ct == 3) POEM_CONV(3, 1); else if (ct == 2) POEM_CONV(2, 1); else POEM_CONV(1, 1);
  }
#undef POEM_CONV
  return hipGetLastError();
}

// out (views, Ca + Cb, H + 2 pad, W + 2 pad): channels [0, Ca) = bilinear x2 (align_corners=False) of a (views, Ca, H/2, W/2),
// channels [Ca, Ca + Cb) = b (views, Cb, H, W), border = 0.  Ca or Cb may be 0 (pure upsample / pure pad).
__global__ void upcat_pad_kernel(const float* __restrict__ a, int Ca, const float* __restrict__ b, int Cb,
                                 float* __restrict__ out, int H, int W, int pad, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int Wp = W + 2 * pad, Hp = H + 2 * pad, C = Ca + Cb;
  const int xp = (int)(i % Wp);
  long t = i / Wp;
  const int yp = (int)(t % Hp);
  t /= Hp;
  const int c = (int)(t % C);
  const long n = t / C;
  const int x = xp - pad, y = yp - pad;
  float v = 0.f;
  if (x >= 0 && x < W && y >= 0 && y < H) {
    if (c < Ca) {
      // F.interpolate(scale_factor=2, mode='bilinear', align_corners=False): src = (dst + 0.5) / 2 - 0.5, clamped at 0
      const int h2 = H / 2, w2 = W / 2;
      const float sy = fmaxf((y + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf((x + 0.5f) * 0.5f - 0.5f, 0.f);
      const int y0 = (int)sy, x0 = (int)sx;
      const int y1 = min(y0 + 1, h2 - 1), x1 = min(x0 + 1, w2 - 1);
      const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
      const float* p = a + ((size_t)n * Ca + c) * h2 * w2;
      v = hy * (hx * p[y0 * w2 + x0] + lx * p[y0 * w2 + x1]) + ly * (hx * p[y1 * w2 + x0] + lx * p[y1 * w2 + x1]);
    } else {
      v = b[(((size_t)n * Cb + (c - Ca)) * H + y) * W + x];
    }
  }
  out[i] = v;
}

extern "C" hipError_t poem_launch_upcat_pad(const float* a, int Ca, const float* b, int Cb, float* out, int views, int H,
                                            int W, int pad, hipStream_t s) {
  if ((Ca && (H % 2 || W % 2)) || pad < 0 || pad > 1 || Ca + Cb <= 0) return hipErrorInvalidValue;
  const long total = (long)views * (Ca + Cb) * (H + 2 * pad) * (W + 2 * pad);
  hipLaunchKernelGGL(upcat_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, Ca, b, Cb, out, H, W, pad, total);
  return hipGetLastError();
}

// feat_decode's tail (POEM.py:190-193 upstream: F.interpolate(x, scale_factor=2, bilinear) then feat_in, a 1x1 ConvBlock
// without norm / activation) in one launch, for the 8 x 8 level of the HRNet pyramid: the 1x1 convolution commutes with the
// interpolation (both linear, the interpolation weights sum to one, so the bias passes through), so a block takes one view,
// stages its K x 64 input in LDS, runs the (C x K) . (K x 64) product on the fp32 matrix cores (wave = one 32-channel tile
// x one 32-pixel tile; packed Linear weights as A, four chunks ahead), leaves the 8 x 8 result in LDS and writes its
// bilinear x2 -- upcat_pad_kernel's expression -- as (C, 16, 16).  Round 3 before: conv1x1_kernel<1> (49 us, dependent
// 4-byte operand loads from L2) + upcat_pad_kernel (35 us) with the 8 x 8 tensor through HBM.
template <int NW>
__global__ __launch_bounds__(NW * 64) void conv1x1_up2_kernel(const float* __restrict__ in, const float4* __restrict__ Wp,
                                                             const float* __restrict__ bias, float* __restrict__ out, int K,
                                                             int C) {
  constexpr int HW = 64, SIDE = 8;
  extern __shared__ __attribute__((aligned(16))) float ftile[];    // K * 64 input, then C * 64 result
  float* ytile = ftile + (size_t)K * HW;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
  const int v = blockIdx.x, KC = K >> 3, ctiles = C / 32;
  {
    const float4* src = reinterpret_cast<const float4*>(in + (size_t)v * K * HW);
    for (int i = tid; i < K * (HW / 4); i += NW * 64) reinterpret_cast<float4*>(ftile)[i] = src[i];
  }
  __syncthreads();
  for (int job = wv; job < ctiles * 2; job += NW) {
    const int ct = job >> 1, pt = job & 1;
    const float4* wp = Wp + (size_t)ct * KC * 64 + lane;
    const float* fb = ftile + (4 * h) * HW + pt * 32 + r;
    f32x16 acc = zero16();
    float4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = wp[(size_t)min(u, KC - 1) * 64];
    for (int kc0 = 0; kc0 < KC; kc0 += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kc = kc0 + u;
        if (kc < KC) {
          const float4 ac = a[u];
          a[u] = wp[(size_t)min(kc + 4, KC - 1) * 64];
#pragma unroll
          for (int t = 0; t < 4; ++t) acc = mfma32((&ac.x)[t], fb[(kc * 8 + t) * HW], acc);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = ct * 32 + mfma_row(i, h);
      ytile[c * HW + pt * 32 + r] = acc[i] + (bias ? bias[c] : 0.f);
    }
  }
  __syncthreads();
  float* dst = out + (size_t)v * C * (4 * HW);
  for (int i = tid; i < C * 4 * HW; i += NW * 64) {
    const int c = i >> 8, y = (i >> 4) & 15, x = i & 15;
    // F.interpolate(scale_factor=2, mode='bilinear', align_corners=False): src = (dst + 0.5) / 2 - 0.5, clamped at 0
    const float sy = fmaxf((y + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf((x + 0.5f) * 0.5f - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, SIDE - 1), x1 = min(x0 + 1, SIDE - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* p = ytile + c * HW;
    dst[i] = hy * (hx * p[y0 * SIDE + x0] + lx * p[y0 * SIDE + x1]) + ly * (hx * p[y1 * SIDE + x0] + lx * p[y1 * SIDE + x1]);
  }
}

// in (views, K, 8, 8), packed Linear weights (C x K, pack_linear order), out (views, C, 16, 16)
extern "C" hipError_t poem_launch_conv1x1_up2(const float* in, const void* wp, const float* bias, float* out, int views, int K,
                                              int C, int h, int w, hipStream_t s) {
  const size_t lds = (size_t)(K + C) * 64 * sizeof(float);
  if (h != 8 || w != 8 || K % 8 || C % 32 || lds > 160 * 1024 || ((uintptr_t)in & 15)) return hipErrorNotSupported;
  constexpr int NW = 10;
  auto kern = conv1x1_up2_kernel<NW>;
  static std::atomic<unsigned long long> optin{0};
  if (hipError_t e = poem_optin_lds(reinterpret_cast<const void*>(kern), 160 * 1024, optin); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)views), dim3(NW * 64), lds, s, in, (const float4*)wp, bias, out, K, C);
  return hipGetLastError();
}

// x (views, C, H, W) -> 2x2 max-pool -> 1x1 conv (J x C, bias) -> sigmoid -> hmap (views, J, H/2, W/2).
// One thread per pooled pixel; weights (J*C + J floats) staged in LDS.  C <= 64, J <= 32.
__global__ __launch_bounds__(256) void pool_head_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ hmap, int C,
                                                        int J, int H, int W, long total) {
  __shared__ float ws[32 * 64 + 32];
  for (int i = threadIdx.x; i < J * C; i += blockDim.x) ws[i] = w[i];
  for (int i = threadIdx.x; i < J; i += blockDim.x) ws[32 * 64 + i] = bias[i];
  __syncthreads();
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int h2 = H / 2, w2 = W / 2;
  const int xo = (int)(i % w2), yo = (int)((i / w2) % h2);
  const long n = i / ((long)w2 * h2);
  float acc[32];
#pragma unroll
  for (int jn = 0; jn < 32; ++jn) acc[jn] = 0.f;
  for (int c = 0; c < C; ++c) {
    const float* p = x + (((size_t)n * C + c) * H + 2 * yo) * W + 2 * xo;
    const float2 r0 = *reinterpret_cast<const float2*>(p), r1 = *reinterpret_cast<const float2*>(p + W);
    const float pooled = fmaxf(fmaxf(r0.x, r0.y), fmaxf(r1.x, r1.y));
#pragma unroll
    for (int jn = 0; jn < 32; ++jn)
      if (jn < J) acc[jn] = fmaf(pooled, ws[jn * C + c], acc[jn]);
  }
#pragma unroll
  for (int jn = 0; jn < 32; ++jn)
    if (jn < J) {
      const float v = acc[jn] + ws[32 * 64 + jn];
      hmap[(((size_t)n * J + jn) * h2 + yo) * w2 + xo] = 1.0f / (1.0f + expf(-v));
    }
}

extern "C" hipError_t poem_launch_pool_head(const float* x, const float* w, const float* bias, float* hmap, int views, int C,
                                            int J, int H, int W, hipStream_t s) {
  if (C > 64 || J > 32 || H % 2 || W % 2) return hipErrorInvalidValue;
  const long total = (long)views * (H / 2) * (W / 2);
  hipLaunchKernelGGL(pool_head_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, w, bias, hmap, C, J, H, W, total);
  return hipGetLastError();
}
