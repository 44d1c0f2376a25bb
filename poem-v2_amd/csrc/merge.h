// Arguments of the fused sampling + merge kernels (merge.hip); shared with the host sequence in api.cpp.
#pragma once
#include <hip/hip_runtime.h>

struct SampleMergeArgs {
  const float* xt;            // (views, hw, C): input_proj + positional table, channel-LAST
  const float4* tab;          // (views, S): per projected point the four bilinear weights
  const uint2* tabo;          // (views, S): ... and its four tap pixels, 16 bits each, in two dwords
  const int* view_sample;     // (views)  view -> sample
  const int* offs;            // (B + 1)  sample -> first view
  const float4* w0; const float* b0;   // merge_net_feature.0.0, packed (C x C)
  const float4* w1; const float* b1;   // merge_net_feature.0.2, packed (C/2 x C)
  float* h2;                  // (views * S, C/2): merge_net[0] of every Q1 row, in the layout `h2_tiled` selects
  float* q1;                  // (B * S, C): the Q1 rows with n == 0 (the residual of merge_features_mv / _sv)
  int views, S, hw;
  int B;                      // samples (offs has B + 1 entries)
  int h2_tiled;               // 0: row-major Q1 rows; 1: tile-major (see merge.hip)
  // Layout-independent launches (forward.cpp, hipGraph replay): when set, the kernel reads the batch's view count from device
  // memory (= view_offsets[B]) and `views` is only the capacity the grid was sized for -- the captured launch then serves every
  // view layout of a batch size.
  const int* views_dev;
  // Grouped kernel (sample_group_kernel): samples whose view count N divides 8 run there when the batch has at least
  // `group_min_views` views (device-side count); this kernel and merge_tail_kernel then skip them.  INT_MAX: never (everything here).
  int group_min_views;
  int xcd_order;              // sample_group_kernel: XCD-aware unit order (merge.hip)
};

// sample_group_kernel: the whole sampling stage of the samples with N in {1, 2, 4, 8} in one kernel (merge.hip).
struct SampleGroupArgs {
  SampleMergeArgs sm;         // h2 unused; q1: the master rows, written at a group's first tile, read back at its last
  const float4* w2; const float* b2;   // merge_net_feature.1.0, packed (C/2 x C/2)
  const float4* w3; const float* b3;   // merge_net_feature.1.2, packed (C x C/2)
  float* out;                 // (B * S, C) bps_feat
};

// The per-view index arrays of a ragged batch, built on the device from offsets that travel in the KERNEL-ARGUMENT segment
// (misc.hip view_layout_kernel): no host buffer whose lifetime must outlast the copy, no pageable H2D copy that would block
// the host behind the previous forward.
#define POEM_LAYOUT_MAX_BATCH 1663
struct ViewLayoutArgs {
  int* offs;                  // (B + 1) out
  int* view_sample;           // (views) out: view -> sample
  int* pe_index;              // (views) out: slot in the folded positional table, n (n - 1) / 2 + k for view k of an n-view sample
  int B;
  unsigned short off16[POEM_LAYOUT_MAX_BATCH + 1];   // view_offsets (total views <= 65535)
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
  int group_min_views;        // as SampleMergeArgs (with views_dev): the samples the grouped kernel takes are skipped here
  const int* views_dev;
  int views;                  // the view capacity the launch was sized for (the device count is clamped to it, as in the other two kernels)
};
