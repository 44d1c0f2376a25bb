// Sampling stage: positional table, 1x1 input projection, BPS projection into every view, bilinear sampling of the
// view feature volumes (LDS-staged planes, coalesced writes along the point axis) and the cross-view merge glue.
#include "common.h"

// ---------------------------------------------------------------------------------------------------------
// Sine positional encoding (petr_transformer.py:434-469 upstream) for every view count N = 1..max_views:
// sine[(pv, ch, y, x)], pv enumerates (N, n) pairs in order N=1:(0) N=2:(0,1) ..., ch = [p(e_n) | p(e_y) | p(e_x)],
// p(e) = [sin(e/d_0), sin(e/d_2), ... | cos(e/d_1), cos(e/d_3), ...] (two concatenated halves of F/2).
// normalize = 0 (POSITIONAL_ENCODING.NORMALIZE false, :451 upstream): the embeddings are the plain cumulative counts n + 1, y + 1, x + 1.
__global__ void sine_pe_kernel(float* __restrict__ out, int F, int H, int W, int max_views, int normalize) {
  const int hw = H * W;
  const long total = (long)(max_views * (max_views + 1) / 2) * 3 * F * hw;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int p = (int)(i % hw);
  const int ch = (int)((i / hw) % (3 * F));
  int pv = (int)(i / ((long)hw * 3 * F));
  int N = 1;
  while (pv >= N) { pv -= N; ++N; }
  const int n = pv;
  const int axis = ch / F, f = ch % F;
  const int y = p / W, x = p % W;
  const float scale = 6.283185307179586f;  // 2*pi as fp32
  const float eps = 1e-6f;
  float e;
  if (!normalize) e = (float)((axis == 0 ? n : axis == 1 ? y : x) + 1);
  else if (axis == 0) e = (float)(n + 1) / ((float)N + eps) * scale;
  else if (axis == 1) e = (float)(y + 1) / ((float)H + eps) * scale;
  else e = (float)(x + 1) / ((float)W + eps) * scale;
  const bool is_cos = f >= F / 2;
  const int dim = is_cos ? 2 * (f - F / 2) + 1 : 2 * f;          // original channel index inside the axis block
  const float expo = (float)(2 * (dim / 2)) / (float)F;
  const float dim_t = powf(10000.0f, expo);
  const float v = e / dim_t;
  out[i] = is_cos ? cosf(v) : sinf(v);
}

extern "C" hipError_t poem_launch_sine_pe(float* out, int F, int H, int W, int max_views, int normalize, hipStream_t s) {
  const long total = (long)(max_views * (max_views + 1) / 2) * 3 * F * H * W;
  hipLaunchKernelGGL(sine_pe_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, out, F, H, W, max_views, normalize);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// 1x1 convolution as an MFMA GEMM with the weights as the A operand (packed fragment order) and the feature planes
// as the B operand read straight from global (lanes along the pixel axis -> coalesced):
//   x[v, c, p] = act(sum_k W[c, k] * feat[v, k, p] + bias[c] (+ table[pe_index ? pe_index[v] : v, c, p])),  act = ReLU or none
// One wave: 32 channels x (32*PT) pixels of one view.
template <int PT>
__global__ __launch_bounds__(256) void conv1x1_kernel(const float* __restrict__ feat, const float4* __restrict__ Wp,
                                                      const float* __restrict__ bias, const float* __restrict__ table,
                                                      const int* __restrict__ pe_index, float* __restrict__ x,
                                                      float* __restrict__ xt, int views, int K, int C, int hw, int relu) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int pgroups = hw / (32 * PT);
  const int ctiles = (C + 31) / 32;
  const int pg = wave % pgroups;
  const int ct = (wave / pgroups) % ctiles;
  const int v = wave / (pgroups * ctiles);
  if (v >= views) return;
  const int KC = K >> 3;
  const float* fv = feat + (size_t)v * K * hw + pg * 32 * PT + r;
  const float4* wp = Wp + (size_t)ct * KC * 64 + lane;
  f32x16 acc[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) acc[t] = zero16();
  for (int kc = 0; kc < KC; ++kc) {
    const float4 a = wp[(size_t)kc * 64];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float* row = fv + (size_t)(kc * 8 + 4 * h + t) * hw;
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) acc[pt] = mfma32((&a.x)[t], row[pt * 32], acc[pt]);
    }
  }
  const float* tab = table ? table + (size_t)(pe_index ? pe_index[v] : v) * C * hw : nullptr;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = ct * 32 + mfma_row(i, h);
    if (c >= C) continue;
    const float bv = bias ? bias[c] : 0.f;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int p = pg * 32 * PT + pt * 32 + r;
      float val = acc[pt][i] + bv;
      if (tab) val += tab[(size_t)c * hw + p];
      if (relu) val = relu_nan(val);
      acc[pt][i] = val;
      if (x) x[((size_t)v * C + c) * hw + p] = val;
    }
  }
  // channel-last copy xt[v][p][c] for the fused sampling kernel (merge.hip): registers 4g .. 4g+3 are four consecutive channels
  if (xt && C % 32 == 0) {
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      float* dst = xt + ((size_t)v * hw + pg * 32 * PT + pt * 32 + r) * C + ct * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(dst + 8 * g) = make_float4(acc[pt][4 * g], acc[pt][4 * g + 1], acc[pt][4 * g + 2], acc[pt][4 * g + 3]);
    }
  }
}

// The same product for the path's input_proj shape (hw % 128 == 0, K * 512 B of LDS): a block owns 128 pixels of one view
// and stages their K feature rows in LDS ONCE (80 KB at K = 160) instead of every channel tile's wave streaming them from
// L2 with dependent 4-byte loads; wave w computes channel tiles w, w + NW, ... x 128 pixels with the packed weight
// fragments as A (1 KiB wave loads) and conflict-free LDS row reads as B.  Same k-ordered fma chain per output element.
// PT = 32-pixel tiles per block: 4 (128 pixels, 80 KB of LDS at K = 160) or, when that leaves most CUs without a block -- a
// forward of one or two samples is 16-32 such blocks --, 1 (32 pixels, 20 KB: four times the blocks, a quarter of the latency).
template <int NW, int PT>
__global__ __launch_bounds__(NW * 64) void conv1x1_lds_kernel(const float* __restrict__ feat, const float4* __restrict__ Wp,
                                                             const float* __restrict__ bias, const float* __restrict__ table,
                                                             const int* __restrict__ pe_index, float* __restrict__ x,
                                                             float* __restrict__ xt, int views, int K, int C, int hw, int relu) {
  extern __shared__ __attribute__((aligned(16))) float ftile[];    // K * 32 * PT
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
  constexpr int PX = 32 * PT, F4 = PX / 4;      // pixels per block, float4 per staged feature row
  const int pgroups = hw / PX;
  const int v = blockIdx.x / pgroups, pg = blockIdx.x % pgroups;
  const int KC = K >> 3, ctiles = C / 32;
  {
    // all of a thread's loads in flight before the first LDS store (a load-store loop waited for every load in turn: ten
    // trips to HBM in a row at the head of every forward)
    const float* src = feat + (size_t)v * K * hw + pg * PX;
    constexpr int STG = 6;
    for (int i0 = tid; i0 < K * F4; i0 += STG * NW * 64) {
      float4 buf[STG];
#pragma unroll
      for (int u = 0; u < STG; ++u) {
        const int i = min(i0 + u * NW * 64, K * F4 - 1);
        buf[u] = *reinterpret_cast<const float4*>(src + (size_t)(i / F4) * hw + 4 * (i % F4));
      }
#pragma unroll
      for (int u = 0; u < STG; ++u) {
        const int i = i0 + u * NW * 64;
        if (i < K * F4) reinterpret_cast<float4*>(ftile)[i] = buf[u];
      }
    }
  }
  __syncthreads();
  const float* tab = table ? table + (size_t)(pe_index ? pe_index[v] : v) * C * hw : nullptr;
  const float* fb = ftile + (4 * h) * PX + r;
  for (int ct = wv; ct < ctiles; ct += NW) {
    const float4* wp = Wp + (size_t)ct * KC * 64 + lane;
    f32x16 acc[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) acc[t] = zero16();
    // weight fragments four chunks ahead (a chunk is 16 MFMAs = 1024 cycles: less than a trip to HBM / Infinity Cache, which
    // is where the first kernel of a forward finds them)
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
          for (int t = 0; t < 4; ++t) {
            const float* row = fb + (kc * 8 + t) * PX;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) acc[pt] = mfma32((&ac.x)[t], row[pt * 32], acc[pt]);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = ct * 32 + mfma_row(i, h);
      const float bv = bias ? bias[c] : 0.f;
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int p = pg * PX + pt * 32 + r;
        float val = acc[pt][i] + bv;
        if (tab) val += tab[(size_t)c * hw + p];
        if (relu) val = relu_nan(val);
        acc[pt][i] = val;
        if (x) x[((size_t)v * C + c) * hw + p] = val;
      }
    }
    if (xt) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        float* dst = xt + ((size_t)v * hw + pg * PX + pt * 32 + r) * C + ct * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(dst + 8 * g) = make_float4(acc[pt][4 * g], acc[pt][4 * g + 1], acc[pt][4 * g + 2], acc[pt][4 * g + 3]);
      }
    }
  }
}

extern "C" hipError_t poem_launch_conv1x1_ex(const float* feat, const void* Wp, const float* bias, const float* table,
                                             const int* pe_index, float* x, float* xt, int views, int K, int C, int hw,
                                             int relu, hipStream_t s) {
  const int ctiles = (C + 31) / 32;
  if (hw % 128 == 0 && C % 32 == 0 && K % 8 == 0 && (size_t)K * 512 <= 96 * 1024 && ((uintptr_t)feat & 15) == 0) {
    const int nw = ctiles >= 8 ? 8 : 4;
    if (views * (hw / 128) * 2 <= poem_device_cus()) {      // few views: 32-pixel blocks (same fma chain per output element)
      auto kern = nw == 8 ? conv1x1_lds_kernel<8, 1> : conv1x1_lds_kernel<4, 1>;
      hipLaunchKernelGGL(kern, dim3((unsigned)(views * (hw / 32))), dim3(nw * 64), (size_t)K * 128, s, feat, (const float4*)Wp, bias, table,
                         pe_index, x, xt, views, K, C, hw, relu);
      return hipGetLastError();
    }
    const size_t lds = (size_t)K * 512;
    auto kern = nw == 8 ? conv1x1_lds_kernel<8, 4> : conv1x1_lds_kernel<4, 4>;
    static std::atomic<unsigned long long> optin[2];
    if (hipError_t e = poem_optin_lds(reinterpret_cast<const void*>(kern), 96 * 1024, optin[nw == 8]); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)(views * (hw / 128))), dim3(nw * 64), lds, s, feat, (const float4*)Wp, bias, table,
                       pe_index, x, xt, views, K, C, hw, relu);
    return hipGetLastError();
  }
  if (hw % 128 == 0) {
    const long waves = (long)views * ctiles * (hw / 128);
    hipLaunchKernelGGL((conv1x1_kernel<4>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, feat, (const float4*)Wp,
                       bias, table, pe_index, x, xt, views, K, C, hw, relu);
  } else {
    const long waves = (long)views * ctiles * (hw / 32);
    hipLaunchKernelGGL((conv1x1_kernel<1>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, feat, (const float4*)Wp,
                       bias, table, pe_index, x, xt, views, K, C, hw, relu);
  }
  return hipGetLastError();
}

extern "C" hipError_t poem_launch_conv1x1(const float* feat, const void* Wp, const float* bias, const float* table,
                                          const int* pe_index, float* x, float* xt, int views, int K, int C, int hw,
                                          hipStream_t s) {
  return poem_launch_conv1x1_ex(feat, Wp, bias, table, pe_index, x, xt, views, K, C, hw, 0, s);
}

// ---------------------------------------------------------------------------------------------------------
// PETR position embedding, its input (BasePointEmbedHead.position_embeding, ptEmb_head.py:113-181 upstream; live when
// PETR_EMBEDDING is set): the frustum of every view -- feature-map pixel (x, y) at image coordinates (x * img1 / W,
// y * img0 / H), D depths -- lifted with the view's intrinsics, moved to the master frame with its extrinsics (camera ->
// master, NOT inverted), normalised by position_range and passed through inverse_sigmoid (transform.py:1145-1161):
//   out[v, 3 d + axis, y, x].  One thread per (view, pixel); the D depths are a loop (3 D stores, coalesced along x).
struct FrustumArgs {
  const float* intr;      // (views, 3, 3)
  const float* extr;      // (views, 4, 4)
  float* out;             // (views, 3 D, H, W)
  int views, H, W, D, lid;
  float img0, img1;       // inp_img_shape[0], [1] -- position_embeding unpacks them as (h, w), forward() as (w, h): followed as written
  float depth_start, bin_size;      // coords_d = depth_start + bin_size * i  (LID: bin_size * i * (i + 1)); fp32 roundings of the doubles
  float lo[3], span[3];   // (float)position_range[c], (float)(position_range[c + 3] - position_range[c])
};

__global__ void frustum_features_kernel(FrustumArgs A) {
  const int hw = A.H * A.W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)A.views * hw) return;
  const int v = (int)(i / hw), p = (int)(i % hw), y = p / A.W, x = p % A.W;
  const float* K = A.intr + (size_t)v * 9;
  const float* E = A.extr + (size_t)v * 16;
  const float u = (float)x * A.img1 / (float)A.W, w = (float)y * A.img0 / (float)A.H;      // :118-119
  const float cu = (u - K[2]) / K[0], cv = (w - K[5]) / K[4];                             // :153
  float* o = A.out + (size_t)v * 3 * A.D * hw + p;
  for (int d = 0; d < A.D; ++d) {
    const float fi = (float)d;
    const float dep = A.lid ? A.depth_start + A.bin_size * fi * (fi + 1.0f) : A.depth_start + A.bin_size * fi;   // :122-130
    const float X = cu * dep, Y = cv * dep;                                                // :154
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* r = E + 4 * c;
      const float wc = fmaf(r[2], dep, fmaf(r[1], Y, r[0] * X)) + r[3];                   // :164 (row c of extr . [X Y Z 1])
      float t = (wc - A.lo[c]) / A.span[c];                                                // :170-175
      t = fminf(fmaxf(t, 0.f), 1.f);                                                       // inverse_sigmoid, eps 1e-5
      o[(size_t)(3 * d + c) * hw] = logf(fmaxf(t, 1e-5f) / fmaxf(1.0f - t, 1e-5f));
    }
  }
}

extern "C" hipError_t poem_launch_frustum_features(const float* intr, const float* extr, float* out, int views, int H, int W, int D,
                                                   int lid, double depth_start, double depth_end, const double* position_range,
                                                   int img0, int img1, hipStream_t s) {
  FrustumArgs a{intr, extr, out, views, H, W, D, lid, (float)img0, (float)img1, (float)depth_start, 0.f, {}, {}};
  // the Python scalars of the reference are doubles; a tensor-scalar operation rounds the scalar to fp32 first
  a.bin_size = (float)(lid ? (depth_end - depth_start) / ((double)D * (1 + D)) : (depth_end - depth_start) / (double)D);
  for (int c = 0; c < 3; ++c) { a.lo[c] = (float)position_range[c]; a.span[c] = (float)(position_range[c + 3] - position_range[c]); }
  const long total = (long)views * H * W;
  hipLaunchKernelGGL(frustum_features_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Projection of the basis point set into each view (collation.py:48-65, transform.py:898-930, ptEmb_head.py:873-883
// upstream) down to the un-normalised sampling coordinates of grid_sample(align_corners=False):
//   T = inv(cam_extr[v]) (fp64 Gauss-Jordan, rounded to fp32);  p = T_R (bps + centre) + T_t;  q = K p;
//   z = |q_z| < 1e-7 ? 1e-7 : q_z;  uv = q_xy / z;  grid = uv * (1/res) * 2 - 1;  ix = ((grid_x + 1) * W - 1) / 2.
__global__ void invert_extr_kernel(const float* __restrict__ extr, float* __restrict__ inv, int views) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < views) invert4x4(extr + (size_t)v * 16, inv + (size_t)v * 16);
}

__global__ void project_kernel(const float* __restrict__ bps, const float* __restrict__ centre,
                               const int* __restrict__ view_sample, const float* __restrict__ intr,
                               const float* __restrict__ inv_extr, float* __restrict__ uv, int views, int S, int fw,
                               int fh, float inv_w, float inv_h) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  const int v = blockIdx.y;
  if (s >= S) return;
  const float* c = centre + (size_t)view_sample[v] * 3;
  const float px = bps[s * 3 + 0] + c[0], py = bps[s * 3 + 1] + c[1], pz = bps[s * 3 + 2] + c[2];
  const float* T = inv_extr + (size_t)v * 16;
  const float cx = fmaf(T[2], pz, fmaf(T[1], py, T[0] * px)) + T[3];
  const float cy = fmaf(T[6], pz, fmaf(T[5], py, T[4] * px)) + T[7];
  const float cz = fmaf(T[10], pz, fmaf(T[9], py, T[8] * px)) + T[11];
  const float* K = intr + (size_t)v * 9;
  const float qx = fmaf(K[2], cz, fmaf(K[1], cy, K[0] * cx));
  const float qy = fmaf(K[5], cz, fmaf(K[4], cy, K[3] * cx));
  float qz = fmaf(K[8], cz, fmaf(K[7], cy, K[6] * cx));
  if (fabsf(qz) < 1e-7f) qz = 1e-7f;
  const float u = qx / qz, w = qy / qz;
  const float gx = u * inv_w * 2.0f - 1.0f, gy = w * inv_h * 2.0f - 1.0f;
  const float ix = ((gx + 1.0f) * (float)fw - 1.0f) / 2.0f;
  const float iy = ((gy + 1.0f) * (float)fh - 1.0f) / 2.0f;
  reinterpret_cast<float2*>(uv)[(size_t)v * S + s] = make_float2(ix, iy);
}

// Bilinear sampling, zero padding.  Block = (view, group of CG channels): the CG planes (fh*fw floats each) are
// staged in LDS once, every thread then walks the point axis (consecutive threads -> consecutive s: coalesced
// reads of uv and coalesced writes of g[v, c, s]).
template <int CG>
__global__ __launch_bounds__(256) void grid_sample_kernel(const float* __restrict__ x, const float* __restrict__ uv,
                                                          float* __restrict__ g, int C, int fh, int fw, int S) {
  extern __shared__ float planes[];  // CG * fh * fw
  const int hw = fh * fw;
  const int v = blockIdx.y;
  const int c0 = blockIdx.x * CG;
  const float* xv = x + ((size_t)v * C + c0) * hw;
  for (int i = threadIdx.x; i < CG * hw; i += blockDim.x) planes[i] = (c0 + i / hw < C) ? xv[i] : 0.f;
  __syncthreads();
  const float2* uvv = reinterpret_cast<const float2*>(uv) + (size_t)v * S;
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    const float2 p = uvv[s];
    const float fx0 = floorf(p.x), fy0 = floorf(p.y);
    const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = p.x - fx0, wy1 = p.y - fy0;
    const float wx0 = (fx0 + 1.0f) - p.x, wy0 = (fy0 + 1.0f) - p.y;
    const bool vx0 = x0 >= 0 && x0 < fw, vx1 = x1 >= 0 && x1 < fw, vy0 = y0 >= 0 && y0 < fh, vy1 = y1 >= 0 && y1 < fh;
    const float w_nw = (vx0 && vy0) ? wx0 * wy0 : 0.f, w_ne = (vx1 && vy0) ? wx1 * wy0 : 0.f;
    const float w_sw = (vx0 && vy1) ? wx0 * wy1 : 0.f, w_se = (vx1 && vy1) ? wx1 * wy1 : 0.f;
    const int cx0 = min(max(x0, 0), fw - 1), cx1 = min(max(x1, 0), fw - 1);
    const int cy0 = min(max(y0, 0), fh - 1), cy1 = min(max(y1, 0), fh - 1);
    const int i_nw = cy0 * fw + cx0, i_ne = cy0 * fw + cx1, i_sw = cy1 * fw + cx0, i_se = cy1 * fw + cx1;
#pragma unroll
    for (int c = 0; c < CG; ++c) {
      if (c0 + c >= C) break;
      const float* pl = planes + c * hw;
      // same accumulation order as ATen's grid_sampler_2d: nw, ne, sw, se
      float acc = pl[i_nw] * w_nw;
      acc += pl[i_ne] * w_ne;
      acc += pl[i_sw] * w_sw;
      acc += pl[i_se] * w_se;
      g[((size_t)v * C + c0 + c) * S + s] = acc;
    }
  }
}

extern "C" hipError_t poem_launch_invert_extr(const float* extr, float* inv, int views, hipStream_t s) {
  hipLaunchKernelGGL(invert_extr_kernel, dim3((views + 63) / 64), dim3(64), 0, s, extr, inv, views);
  return hipGetLastError();
}

// The two halves of poem_launch_project_sample for the whole-path sequence (forward.cpp): the projection reads the CALLER's
// camera tensors and stays outside the captured body; the sampling reads workspace memory only.
extern "C" hipError_t poem_launch_project_uv(const float* bps, const float* centre, const int* view_sample, const float* intr,
                                             const float* extr, float* inv_scratch, float* uv, int views, int fh, int fw, int S,
                                             int img_w, int img_h, hipStream_t s) {
  hipLaunchKernelGGL(invert_extr_kernel, dim3((views + 63) / 64), dim3(64), 0, s, extr, inv_scratch, views);
  hipLaunchKernelGGL(project_kernel, dim3((S + 255) / 256, views), dim3(256), 0, s, bps, centre, view_sample, intr,
                     inv_scratch, uv, views, S, fw, fh, 1.0f / (float)img_w, 1.0f / (float)img_h);
  return hipGetLastError();
}
extern "C" hipError_t poem_launch_grid_sample(const float* x, const float* uv, float* g, int views, int C, int fh, int fw, int S,
                                              hipStream_t s) {
  constexpr int CG = 8;
  hipLaunchKernelGGL((grid_sample_kernel<CG>), dim3((C + CG - 1) / CG, views), dim3(256), CG * fh * fw * sizeof(float),
                     s, x, uv, g, C, fh, fw, S);
  return hipGetLastError();
}

extern "C" hipError_t poem_launch_project_sample(const float* x, const float* bps, const float* centre,
                                                 const int* view_sample, const float* intr, const float* extr,
                                                 float* inv_scratch, float* uv, float* g, int views, int C, int fh,
                                                 int fw, int S, int img_w, int img_h, hipStream_t s) {
  hipLaunchKernelGGL(invert_extr_kernel, dim3((views + 63) / 64), dim3(64), 0, s, extr, inv_scratch, views);
  // NB upstream names the pair (w, h) = inp_img_shape; the x coordinate is divided by the first entry.
  hipLaunchKernelGGL(project_kernel, dim3((S + 255) / 256, views), dim3(256), 0, s, bps, centre, view_sample, intr,
                     inv_scratch, uv, views, S, fw, fh, 1.0f / (float)img_w, 1.0f / (float)img_h);
  constexpr int CG = 8;
  hipLaunchKernelGGL((grid_sample_kernel<CG>), dim3((C + CG - 1) / CG, views), dim3(256), CG * fh * fw * sizeof(float),
                     s, x, uv, g, C, fh, fw, S);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Cross-view merge glue on the Q1 rows.  h2: (sum_i N_i * S, HALF) rows in Q1 order (row = off_i*S + s*N_i + n).
//   N_i > 1: w_n = <h_n, h_0>, m = sum_{n>=1} w_n h_n   (merge_features_mv, ptEmb_head.py:745-762 upstream)
//   N_i = 1: m = h_0                                    (merge_features_sv feeds net1 directly, :764-771)
// One wave per (sample, point).
__global__ __launch_bounds__(256) void merge_reduce_kernel(const float* __restrict__ h2, const int* __restrict__ offs,
                                                           float* __restrict__ m, int B, int S, int HALF) {
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (long)B * S) return;
  const int b = (int)(wid / S), s = (int)(wid % S);
  const int off = offs[b], N = offs[b + 1] - off;
  const float* base = h2 + ((size_t)off * S + (size_t)s * N) * HALF;
  float* out = m + (size_t)wid * HALF;
  if (N == 1) {
    for (int c = lane; c < HALF; c += 64) out[c] = base[c];
    return;
  }
  // HALF <= 512 -> up to 8 values per lane
  float acc[8], mast[8];
  const int per = (HALF + 63) / 64;
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc[i] = 0.f; mast[i] = (i < per && lane + 64 * i < HALF) ? base[lane + 64 * i] : 0.f; }
  for (int n = 1; n < N; ++n) {
    const float* hn = base + (size_t)n * HALF;
    float v[8], dot = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] = (i < per && lane + 64 * i < HALF) ? hn[lane + 64 * i] : 0.f;
      dot = fmaf(v[i], mast[i], dot);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fmaf(dot, v[i], acc[i]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) if (i < per && lane + 64 * i < HALF) out[lane + 64 * i] = acc[i];
}

extern "C" hipError_t poem_launch_merge_reduce(const float* h2, const int* offs, float* m, int B, int S, int HALF,
                                               hipStream_t s) {
  const long waves = (long)B * S;
  hipLaunchKernelGGL(merge_reduce_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, h2, offs, m, B, S, HALF);
  return hipGetLastError();
}

// out[b, s, c] = q1 + y / N  with q1 = Q1 row (s*N + 0) of sample b = g_flat[off_b*C*S + (s*N)*C + c]   (N = 1: q + y)
__global__ void merge_finalize_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                      const int* __restrict__ offs, float* __restrict__ out, int B, int S, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * S * C) return;
  const int c = (int)(i % C);
  const int s = (int)((i / C) % S);
  const int b = (int)(i / ((long)C * S));
  const int off = offs[b], N = offs[b + 1] - off;
  const float q1 = g[(size_t)off * C * S + ((size_t)s * N) * C + c];
  const float yv = y[i];
  out[i] = q1 + (N == 1 ? yv : yv / (float)N);
}

extern "C" hipError_t poem_launch_merge_finalize(const float* g, const float* y, const int* offs, float* out, int B,
                                                 int S, int C, hipStream_t s) {
  const long total = (long)B * S * C;
  hipLaunchKernelGGL(merge_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, y, offs, out, B,
                     S, C);
  return hipGetLastError();
}
