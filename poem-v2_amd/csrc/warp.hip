// Input side: crop / warp / normalise of the raw camera images on the device.
// Replaces, per view, the host chain of SimpleTransform2D.__call__ (lib/utils/transform.py:153-170):
//   cv2.warpAffine(image, affine[:2], (W, H), INTER_LINEAR, BORDER_CONSTANT)      uint8 HxWx3 -> uint8 OHxOWx3
//   colour jitter  image[:, :, c] = clip(image[:, :, c] * gain_c, 0, 255)         (training augmentation only)
//   tvF.to_tensor + tvF.normalize(mean 0.5, std 1)                                 -> fp32 (3, OH, OW) = p / 255 - 0.5
// and the optional mirror warp of MultiviewWebDataset.process_data_item (lib/data_wds/multiview_wds.py:112-118,
// uint8 output).  One launch handles every view of a batch; the raw images sit back to back in one byte blob (ragged
// sizes, CSR-style offsets).
//
// Arithmetic = OpenCV's fixed-point path for 8-bit INTER_LINEAR (modules/imgproc/src/imgwarp.cpp, 4.5.x: WarpAffineInvoker
// + remapBilinear; OpenCV itself is NOT in /root/reference -- requirements.txt pins opencv-python 4.5.1.48 -- so this is a
// restatement of the published algorithm, "parity unpinned"):
//   the caller inverts the 2x3 matrix in fp64 (as cv::warpAffine does); per destination pixel (x, y)
//     X = (rint((M1 y + M2) 2^10) + 16 + rint(M0 x 2^10)) >> 5,   Y likewise with M4, M5, M3       [AB_BITS 10, INTER_BITS 5]
//     sx = X >> 5, ax = X & 31 (1/32-pixel fraction), sy, ay likewise
//     out = (sum_k w_k tap_k + 2^14) >> 15 with w = (32-ax | ax)(32-ay | ay) * 32                  [INTER_REMAP_COEF_BITS 15]
//   taps outside the source read the constant border 0.  All integer: bit-exact against the oracle by construction.
// HBM-bound and tiny: per view it writes 3 OH OW fp32 (786 KB at 256x256) and reads the source footprint once.
#include "common.h"

__device__ inline int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

// One thread = 4 consecutive destination pixels of a row: the fp32 planes are written as 16-byte stores (a wave writes
// 1 KiB contiguous per plane), the uint8 image as three dwords.  Rows whose width is not a multiple of 4 finish scalar.
__device__ __forceinline__ void warp_pixel(const unsigned char* __restrict__ img, int sh, int sw, const double* __restrict__ M,
                                           int x, int X0, int Y0, const double* __restrict__ g3, int (&p)[3]) {
  // WarpAffineInvoker: adelta / bdelta per column; saturate_cast<int>(double) = round half to even.
  const int adelta = __double2int_rn(__dmul_rn(__dmul_rn(M[0], (double)x), 1024.0));
  const int bdelta = __double2int_rn(__dmul_rn(__dmul_rn(M[3], (double)x), 1024.0));
  const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
  const int sx = sat_short(X >> 5), sy = sat_short(Y >> 5), ax = X & 31, ay = Y & 31;
  const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
  const bool x0in = (unsigned)sx < (unsigned)sw, x1in = (unsigned)(sx + 1) < (unsigned)sw;
  const bool y0in = (unsigned)sy < (unsigned)sh, y1in = (unsigned)(sy + 1) < (unsigned)sh;
  const unsigned char* r0 = img + ((long long)sy * sw + sx) * 3;
  const unsigned char* r1 = r0 + (long long)sw * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int t00 = (y0in && x0in) ? r0[c] : 0, t01 = (y0in && x1in) ? r0[3 + c] : 0;
    const int t10 = (y1in && x0in) ? r1[c] : 0, t11 = (y1in && x1in) ? r1[3 + c] : 0;
    p[c] = (w00 * t00 + w01 * t01 + w10 * t10 + w11 * t11 + (1 << 14)) >> 15;
  }
  if (g3) {                                         // transform.py:159-164: float64 product, clip, C cast to uint8
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double g = __dmul_rn((double)p[c], g3[c]);
      g = g < 0.0 ? 0.0 : (g > 255.0 ? 255.0 : g);
      p[c] = (int)g;
    }
  }
}

__global__ __launch_bounds__(256) void warp_affine_kernel(const unsigned char* __restrict__ src,
                                                          const long long* __restrict__ src_off,
                                                          const int* __restrict__ src_hw, const double* __restrict__ minv,
                                                          const double* __restrict__ gain, float* __restrict__ out_f32,
                                                          unsigned char* __restrict__ out_u8, int OH, int OW) {
  const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y, v = blockIdx.z;
  if (x4 >= OW || y >= OH) return;
  const double* M = minv + (size_t)v * 6;
  const int sh = src_hw[2 * v], sw = src_hw[2 * v + 1];
  const unsigned char* img = src + src_off[v];
  const double* g3 = gain ? gain + 3 * v : nullptr;
  // X0 / Y0 per row.  __dmul_rn / __dadd_rn keep the compiler from contracting M1*y + M2 into an fma (OpenCV's build
  // does not).
  const int X0 = __double2int_rn(__dmul_rn(__dadd_rn(__dmul_rn(M[1], (double)y), M[2]), 1024.0)) + 16;
  const int Y0 = __double2int_rn(__dmul_rn(__dadd_rn(__dmul_rn(M[4], (double)y), M[5]), 1024.0)) + 16;
  int p[4][3];
  const int n = min(4, OW - x4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < n) warp_pixel(img, sh, sw, M, x4 + i, X0, Y0, g3, p[i]);
    else p[i][0] = p[i][1] = p[i][2] = 0;
  }
  const size_t plane = (size_t)OH * OW;
  if (n == 4 && (OW & 3) == 0) {
    if (out_u8) {                                   // 12 bytes, 4-byte aligned ((v*OH + y)*OW + x4) * 3 with OW, x4 % 4 == 0
      unsigned int* o = reinterpret_cast<unsigned int*>(out_u8 + (((size_t)v * OH + y) * OW + x4) * 3);
      o[0] = (unsigned)p[0][0] | ((unsigned)p[0][1] << 8) | ((unsigned)p[0][2] << 16) | ((unsigned)p[1][0] << 24);
      o[1] = (unsigned)p[1][1] | ((unsigned)p[1][2] << 8) | ((unsigned)p[2][0] << 16) | ((unsigned)p[2][1] << 24);
      o[2] = (unsigned)p[2][2] | ((unsigned)p[3][0] << 8) | ((unsigned)p[3][1] << 16) | ((unsigned)p[3][2] << 24);
    }
    if (out_f32) {                                  // to_tensor: fp32 p / 255 (correctly rounded); normalize: - 0.5, / 1
      float* o = out_f32 + ((size_t)v * 3 * OH + y) * OW + x4;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float4 t;
        t.x = __fsub_rn(__fdiv_rn((float)p[0][c], 255.0f), 0.5f);
        t.y = __fsub_rn(__fdiv_rn((float)p[1][c], 255.0f), 0.5f);
        t.z = __fsub_rn(__fdiv_rn((float)p[2][c], 255.0f), 0.5f);
        t.w = __fsub_rn(__fdiv_rn((float)p[3][c], 255.0f), 0.5f);
        *reinterpret_cast<float4*>(o + c * plane) = t;
      }
    }
    return;
  }
  for (int i = 0; i < n; ++i) {
    if (out_u8) {
      unsigned char* o = out_u8 + (((size_t)v * OH + y) * OW + x4 + i) * 3;
      o[0] = (unsigned char)p[i][0]; o[1] = (unsigned char)p[i][1]; o[2] = (unsigned char)p[i][2];
    }
    if (out_f32) {
      float* o = out_f32 + ((size_t)v * 3 * OH + y) * OW + x4 + i;
      for (int c = 0; c < 3; ++c) o[c * plane] = __fsub_rn(__fdiv_rn((float)p[i][c], 255.0f), 0.5f);
    }
  }
}

extern "C" hipError_t poem_launch_warp_affine(const unsigned char* src, const long long* src_off, const int* src_hw,
                                   const double* minv, const double* gain, float* out_f32, unsigned char* out_u8, int views,
                                   int OH, int OW, hipStream_t s) {
  dim3 grid((OW + 255) / 256, (OH + 3) / 4, views);
  warp_affine_kernel<<<grid, dim3(64, 4), 0, s>>>(src, src_off, src_hw, minv, gain, out_f32, out_u8, OH, OW);
  return hipGetLastError();
}
