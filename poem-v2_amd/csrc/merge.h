// Arguments of the fused sampling + merge kernels (merge.hip); shared with the host sequence in api.cpp.
#pragma once
#include <hip/hip_runtime.h>

struct SampleMergeArgs {
  const float* xt;            // (views, hw, C): input_proj + positional table, channel-LAST
  const float4* tab;          // (views, S, 2): per projected point the four bilinear weights | four tap pixels, 16 bits each, in two dwords (+ two unused)
  const int* view_sample;     // (views)  view -> sample
  const int* offs;            // (B + 1)  sample -> first view
  const float4* w0; const float* b0;   // merge_net_feature.0.0, packed (C x C)
  const float4* w1; const float* b1;   // merge_net_feature.0.2, packed (C/2 x C)
  float* h2;                  // (views * S, C/2): merge_net[0] of every Q1 row, in the layout `h2_tiled` selects
  float* q1;                  // (B * S, C): the Q1 rows with n == 0 (the residual of merge_features_mv / _sv)
  int views, S, hw;
  int h2_tiled;               // 0: row-major Q1 rows; 1: tile-major (see merge.hip)
};

struct MergeTailArgs {
  const float* h2;            // as written by the fused kernel
  const float* q1;            // (B * S, C)
  const int* offs;            // (B + 1)
  const float4* w0; const float* b0;   // merge_net_feature.1.0, packed (C/2 x C/2)
  const float4* w1; const float* b1;   // merge_net_feature.1.2, packed (C x C/2)
  float* out;                 // (B * S, C) bps_feat
  int B, S;
  int h2_tiled;
};
