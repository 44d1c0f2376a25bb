// The handle of libpoem_hip.so (include/poem_hip.h): tensor table, weight packing and composition at creation, the constant
// tables folded there (positional table, block-0 anchor tables), option switches, debug taps and the HIP-event profile.
#include "engine.h"

#include <atomic>
thread_local int g_last_hip_error = 0;

// The launcher switches that are PROCESS-wide (gemm_xcd_map | gemm_kslab << 1 | xattn_half << 2 | xattn_tail << 3: statics of gemm.hip / attn.hip,
// scheduling only, bit-identical results): whichever handle sets one sets it for every handle, so the launch-graph key carries the
// process's current values, not a per-handle copy that the launches would not follow.
static std::atomic<int> g_proc_switches{15};
int poem_process_switches() { return g_proc_switches.load(); }

// Everything of a handle that takes part in a stream capture -- the capture stream, the two side streams, the fork / join
// events -- comes from a process-wide pool per device ("capture kit") and goes back to it at poem_destroy instead of being
// destroyed, and the graph execs of a destroyed handle are parked, not destroyed: on ROCm 7.0's runtime hipGraphLaunch of a
// LATER, freshly instantiated exec crashes inside libamdhip64 (SIGSEGV, host side) after earlier handles' capture objects
// were destroyed.  Observed: with streams destroyed after five handle life cycles (round 3, first session: streams pooled
// since); with streams pooled but events and execs destroyed, in the 204-test GPU suite right after Python's collector
// released ten heads at once (POEM_TRACE: the second capture of the next head instantiates, its launch crashes) --
// reproducible on a box whose MIOpen cache is warm, never with one handle at a time (tools/lab/lifecycle_probe.py, 120
// cycles).  A kit's objects are idle when its handle is destroyed, so the next handle takes them over as they are.
//   Parked execs are RE-USED (round 4): the next capture of the same shape (forward.cpp graph_shape) takes one over through
// hipGraphExecUpdate instead of instantiating a new exec, so the number of execs alive in the process is bounded by the largest
// number ever cached at once per shape, not by the number of handles created or layouts met (poem_graph_stats reports both).
#include <mutex>
namespace {
struct CaptureKit {
  hipStream_t cap = nullptr, bps = nullptr, knn = nullptr;
  hipEvent_t ev[5 + 32] = {};
};
std::mutex g_pool_mutex;
std::map<int, std::vector<CaptureKit>> g_kit_pool;         // device id -> idle kits
// A parked exec belongs to the DEVICE it was instantiated on (its kernel nodes, and the internal branch streams the runtime
// gave it, live there) and may still be EXECUTING when it is parked -- an LRU-evicted exec's last hipGraphLaunch is only
// enqueued, the forward never syncs.  `done` is recorded behind that last launch; the exec is offered for an in-place update
// (hipGraphExecUpdate rewrites node parameters) only on its own device and only once the event has completed.
struct ParkedExec { hipGraphExec_t exec; uint64_t shape; int device; hipEvent_t done; };
std::vector<ParkedExec> g_parked_execs;                    // never destroyed (see above); re-used by poem_reuse_graph_exec
std::map<int, std::vector<hipEvent_t>> g_idle_done_events; // device -> events of re-used execs, kept for the next parking (never destroyed either)
int64_t g_exec_reuses = 0, g_exec_update_failures = 0, g_exec_busy_skips = 0;

// (g_pool_mutex held) an event recorded on `last_stream` behind the exec's last launch; nullptr when it never ran.  An event
// belongs to the device it was created on and records only on that device's streams: the pool is per device and creation runs
// with the exec's device current (poem_destroy may be called -- from a Python finaliser -- with another device current).
// `*ok` = false: the exec DID run and no event could be recorded behind it -- nobody can tell when its last launch ends, so the
// caller parks it with shape 0 (never offered for an update again) instead of as "never ran".
hipEvent_t done_event_locked(hipStream_t last_stream, bool launched, int device, bool* ok) {
  *ok = true;
  if (!launched) return nullptr;
  int cur = -1;
  const bool switched = hipGetDevice(&cur) == hipSuccess && cur != device && hipSetDevice(device) == hipSuccess;
  hipEvent_t ev = nullptr;
  auto& idle = g_idle_done_events[device];
  if (!idle.empty()) { ev = idle.back(); idle.pop_back(); }
  else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ev = nullptr; }
  if (ev && hipEventRecord(ev, last_stream) != hipSuccess) {
    (void)hipGetLastError();
    idle.push_back(ev);
    ev = nullptr;
  }
  if (!ev) {
    // (a destroyed caller stream is the common cause: drain the device once -- rare path -- and the exec is idle after all)
    *ok = hipDeviceSynchronize() == hipSuccess;
    (void)hipGetLastError();
  }
  if (switched) (void)hipSetDevice(cur);
  return ev;
}

hipEvent_t** kit_event_slots(poem_handle_t h, hipEvent_t** out) {
  int n = 0;
  out[n++] = &h->ev_fork; out[n++] = &h->ev_join_bps; out[n++] = &h->ev_join_knn; out[n++] = &h->ev_tab; out[n++] = &h->ev_fork0;
  for (int i = 0; i < 8; ++i) { out[n++] = &h->ev_bps[i]; out[n++] = &h->ev_xyz[i]; out[n++] = &h->ev_knn[i]; out[n++] = &h->ev_def[i]; }
  return out;
}

bool take_kit(poem_handle_t h) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  h->stream_device = dev;
  hipEvent_t* slots[5 + 32];
  kit_event_slots(h, slots);
  {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    auto& pool = g_kit_pool[dev];
    if (!pool.empty()) {
      const CaptureKit k = pool.back();
      pool.pop_back();
      h->cap_stream = k.cap; h->bps_stream = k.bps; h->knn_stream = k.knn;
      for (int i = 0; i < 5 + 32; ++i) *slots[i] = k.ev[i];
      return true;
    }
  }
  bool ok = hipStreamCreateWithFlags(&h->bps_stream, hipStreamNonBlocking) == hipSuccess &&
            hipStreamCreateWithFlags(&h->knn_stream, hipStreamNonBlocking) == hipSuccess &&
            hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) == hipSuccess;
  for (int i = 0; i < 5 + 32; ++i) ok = ok && hipEventCreateWithFlags(slots[i], hipEventDisableTiming) == hipSuccess;
  return ok;
}

void return_kit(poem_handle_t h) {
  hipEvent_t* slots[5 + 32];
  kit_event_slots(h, slots);
  bool whole = h->cap_stream && h->bps_stream && h->knn_stream;
  for (int i = 0; i < 5 + 32; ++i) whole = whole && *slots[i] != nullptr;
  if (!whole) {                             // a partly created kit: nothing was captured on it
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    if (h->bps_stream) (void)hipStreamDestroy(h->bps_stream);
    if (h->knn_stream) (void)hipStreamDestroy(h->knn_stream);
    for (int i = 0; i < 5 + 32; ++i) if (*slots[i]) (void)hipEventDestroy(*slots[i]);
  } else {
    CaptureKit k;
    k.cap = h->cap_stream; k.bps = h->bps_stream; k.knn = h->knn_stream;
    for (int i = 0; i < 5 + 32; ++i) k.ev[i] = *slots[i];
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    g_kit_pool[h->stream_device].push_back(k);
  }
  h->cap_stream = h->bps_stream = h->knn_stream = nullptr;
  for (int i = 0; i < 5 + 32; ++i) *slots[i] = nullptr;
}

void park_execs(poem_handle_t h) {
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  for (auto& g : h->graph_cache) {
    bool ok = true;
    hipEvent_t ev = done_event_locked(g.last_stream, g.launched, h->stream_device, &ok);
    g_parked_execs.push_back({g.exec, ok ? g.shape : 0, h->stream_device, ev});
  }
  h->graph_cache.clear();
}
}  // namespace

void poem_park_graph_exec(hipGraphExec_t e, uint64_t shape, int device, hipStream_t last_stream, bool launched) {
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  bool ok = true;
  hipEvent_t ev = done_event_locked(last_stream, launched, device, &ok);
  g_parked_execs.push_back({e, ok ? shape : 0, device, ev});
}

// A parked exec of the same shape, updated in place to the freshly captured graph (kernel arguments, grids and functions of
// every node are rewritten; earlier launches of the exec that are still in flight keep what they were launched with).  An
// update the runtime refuses leaves the exec parked.
hipGraphExec_t poem_reuse_graph_exec(hipGraph_t graph, uint64_t shape, int device) {
  ParkedExec cand{nullptr, 0, 0, nullptr};
  {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (size_t i = g_parked_execs.size(); i-- > 0;) {
      ParkedExec& pe = g_parked_execs[i];
      if (pe.shape != shape || pe.device != device) continue;
      if (pe.done) {
        if (hipEventQuery(pe.done) != hipSuccess) { (void)hipGetLastError(); ++g_exec_busy_skips; continue; }   // still running its last launch
        g_idle_done_events[pe.device].push_back(pe.done);
        pe.done = nullptr;
      }
      cand = pe;
      g_parked_execs.erase(g_parked_execs.begin() + i);
      break;
    }
  }
  if (!cand.exec) return nullptr;
  hipGraphNode_t err_node = nullptr;
  hipGraphExecUpdateResult res = hipGraphExecUpdateSuccess;
  const hipError_t e = hipGraphExecUpdate(cand.exec, graph, &err_node, &res);
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  if (e == hipSuccess && res == hipGraphExecUpdateSuccess) {
    ++g_exec_reuses;
    return cand.exec;
  }
  (void)hipGetLastError();
  POEM_TRACE("exec update refused e=%d res=%d", (int)e, (int)res);
  ++g_exec_update_failures;
  // shape 0 never matches a capture: an exec the runtime would not update is kept out of further attempts
  g_parked_execs.insert(g_parked_execs.begin(), {cand.exec, 0, cand.device, nullptr});
  return nullptr;
}

std::vector<TensorSpec> tensor_table(const poem_config_t& c) {
  const int C = c.embed, Q = c.nquery;
  std::vector<TensorSpec> t;
  auto lin = [&](int o, int i, bool pack, bool bias = true) {
    t.push_back({o, i, pack});
    if (bias) t.push_back({o, 1, false});
  };
  lin(C, c.in_channels, true);
  lin(C, 3 * C / 2, true);
  lin(C, C, true);
  lin(C / 2, C, true);
  lin(C / 2, C / 2, true);
  lin(C, C / 2, true);
  t.push_back({Q, C, false});
  for (int b = 0; b < c.nblocks; ++b) {
    lin(C, C, true);
    for (int a = 0; a < 2; ++a) {
      lin(C, C, true); lin(C, C, true); lin(C, C, true); lin(C, C, true);
      t.push_back({C, 1, false}); t.push_back({C, 1, false});
    }
    for (int a = 0; a < 2; ++a) {
      lin(C, C, true); lin(C, C, true);
      lin(C, 3, false);
      lin(C, C, true); lin(C, C, true); lin(C, C, true);
      lin(C, C, true, false); lin(C, C, true, false); lin(C, C, true, false);
    }
    lin(C, C, true);
    lin(3, C, false);
    lin(4 * C, C, true);
    lin(C, 4 * C, true);
    t.push_back({C, 1, false}); t.push_back({C, 1, false});
    if (c.parametric) {
      lin(1, Q, false);
      lin(106, C, false);
    }
  }
  if (c.petr_embedding) {      // position_encoder.0 / .2 (ptEmb_head.py:101-105), listed last: petr_slot()
    lin(2 * C, 3 * c.depth_num, true);
    lin(C, 2 * C, true);
  }
  return t;
}

int check_config(const poem_config_t* c) {
  if (!c) return POEM_E_ARG;
  const int C = c->embed;
  if (C < 32 || C > 1024 || (C & (C - 1))) return POEM_E_UNSUPPORTED;       // 32,64,...,1024
  if (c->knn < 1 || c->knn > 32) return POEM_E_UNSUPPORTED;       // the attention tile holds 32 neighbour columns (vecattn.hip)
  if (c->in_channels % 8 || c->nsample % 32 || c->nsample % C) return POEM_E_UNSUPPORTED;
  if (c->heads <= 0 || C % c->heads) return POEM_E_UNSUPPORTED;
  const int dh = C / c->heads;
  if (!(dh == 8 || dh == 16 || dh == 32 || dh == 64 || dh == 128 || dh == 256)) return POEM_E_UNSUPPORTED;
  if (c->nsample > 4096 || c->nquery > 4096 || c->nquery < 33) return POEM_E_UNSUPPORTED;
  if ((c->feat_h * c->feat_w) % 32) return POEM_E_UNSUPPORTED;
  if (c->max_views < 1 || c->max_views > 64 || c->nblocks < 1) return POEM_E_ARG;
  if (c->petr_embedding) {
    if (c->depth_num < 8 || c->depth_num > 1024 || (3 * c->depth_num) % 8) return POEM_E_UNSUPPORTED;      // 8-deep weight fragments
    if (!(c->depth_end > c->depth_start)) return POEM_E_ARG;
    for (int k = 0; k < 3; ++k)
      if (!(c->position_range[k + 3] > c->position_range[k])) return POEM_E_ARG;
  }
  return POEM_OK;
}

// ---- workspace plan -------------------------------------------------------------------------------------------------
extern "C" {

int poem_abi_version(void) { return 3; }
int poem_last_hip_error(void) { return g_last_hip_error; }
const char* poem_error_string(int code) {
  switch (code) {
    case POEM_OK: return "ok";
    case POEM_E_ARG: return "bad argument";
    case POEM_E_WORKSPACE: return "workspace too small";
    case POEM_E_LAUNCH: return "HIP runtime/launch error";
    case POEM_E_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown";
  }
}

int poem_num_weight_tensors(const poem_config_t* cfg) {
  if (check_config(cfg) != POEM_OK) return POEM_E_UNSUPPORTED;
  return (int)tensor_table(*cfg).size();
}

int64_t poem_weight_tensor_numel(const poem_config_t* cfg, int index) {
  if (check_config(cfg) != POEM_OK) return POEM_E_UNSUPPORTED;
  auto t = tensor_table(*cfg);
  if (index < 0 || index >= (int)t.size()) return POEM_E_ARG;
  return (int64_t)t[index].rows * t[index].cols;
}

size_t poem_packed_bytes(const poem_config_t* cfg) {
  if (check_config(cfg) != POEM_OK) return 0;
  size_t total = 0;
  for (auto& s : tensor_table(*cfg))
    if (s.pack) total += align_up(packed_bytes_linear(s.rows, s.cols), 256);
  const size_t hw = (size_t)cfg->feat_h * cfg->feat_w;
  {   // fused images F1..F4 per block + their concatenated biases (F1, F4)
    const size_t C = cfg->embed;
    const size_t per = align_up(packed_bytes_linear(6 * C, C), 256) + align_up(packed_bytes_linear(2 * C, C), 256) +
                       align_up(packed_bytes_linear(3 * C, C), 256) + align_up(packed_bytes_linear(5 * C, C), 256) +
                       align_up(6 * C * 4, 256) + align_up(2 * C * 4, 256) + align_up(3 * C * 4, 256) + align_up(5 * C * 4, 256) +
                       3 * align_up(packed_bytes_linear(C, C), 256) + 2 * align_up(C * 4, 256);
    total += per * cfg->nblocks;
    total += align_up((6 * C * C + 2 * C * C + 8 * C) * 4, 256) + 256;   // raw composites (init-time scratch, at the arena's end)
  }
  total += align_up((size_t)poem_handle_s::IDX_CAP * 4, 256);                              // per-view index arrays
  total += align_up(pe_views(cfg->max_views) * cfg->embed * hw * 4, 256);              // folded positional table
  total += align_up(pe_views(cfg->max_views) * (3 * cfg->embed / 2) * hw * 4, 256);    // sine scratch (init only)
  return total;
}

// The part of the arena that holds packed weight images (everything in front of the index arrays): what the native-image mirror
// of the chain kernels has to cover -- not the folded positional table, the sine scratch or the init-time composites behind it
// (tens of MB per engine at max_views = 10).
static size_t packed_weight_extent(const poem_config_t* cfg) {
  const size_t hw = (size_t)cfg->feat_h * cfg->feat_w, C = cfg->embed;
  const size_t tail = align_up((size_t)poem_handle_s::IDX_CAP * 4, 256) + align_up(pe_views(cfg->max_views) * C * hw * 4, 256) +
                      align_up(pe_views(cfg->max_views) * (3 * C / 2) * hw * 4, 256) + align_up((6 * C * C + 2 * C * C + 8 * C) * 4, 256) + 256;
  return poem_packed_bytes(cfg) - tail;
}

size_t poem_packed_linear_bytes(int out_features, int in_features) {
  if (in_features % 8) return 0;
  return packed_bytes_linear(out_features, in_features);
}

int poem_pack_linear(const float* w, int out_features, int in_features, void* packed, void* stream) {
  if (!w || !packed || in_features % 8 || out_features <= 0) return POEM_E_ARG;
  HIPCHK(poem_launch_pack_linear(w, out_features, in_features, packed, (hipStream_t)stream));
  return POEM_OK;
}

int poem_create(const poem_config_t* cfg, const void* const* raw_host, int n, const float* bps, const float* anchor,
                const int32_t* anchor_idx, const float* template_xyz, void* packed, size_t packed_bytes, void* stream,
                poem_handle_t* out) {
  int rc = check_config(cfg);
  if (rc != POEM_OK) return rc;
  if (!raw_host || !bps || !anchor || !anchor_idx || !template_xyz || !packed || !out) return POEM_E_ARG;
  auto* h = new poem_handle_s();
  h->cfg = *cfg;
  h->specs = tensor_table(*cfg);
  if (n != (int)h->specs.size() || packed_bytes < poem_packed_bytes(cfg)) { poem_destroy(h); return POEM_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  char* cur = (char*)packed;
  h->packed_base = (const char*)packed;
  h->packed_size = packed_bytes;
  if (cfg->embed >= 128) {
    if (hipMalloc((void**)&h->gemm_split, packed_bytes) != hipSuccess ||
        hipMalloc((void**)&h->gemm_scales, (packed_bytes / 256 + 1) * sizeof(float)) != hipSuccess) {
      g_last_hip_error = (int)hipGetLastError(); poem_destroy(h); return POEM_E_LAUNCH;
    }
  }
  if (poem_chain16_wants_native(cfg->embed) && hipMalloc((void**)&h->native16, packed_weight_extent(cfg)) != hipSuccess) {
    g_last_hip_error = (int)hipGetLastError(); poem_destroy(h); return POEM_E_LAUNCH;
  }
  // mirror of a freshly packed fp32 image at `at`: the split image of the same row-major weight, and (same offset in
  // h->native16) the image's native 16x16x4 form for the chain kernels' one-unit tiles
  auto mirror = [&](const float* w_rows, int rows, int cols, const char* at) -> hipError_t {
    if (h->native16)
      if (hipError_t e = poem_launch_native16(at, h->native16 + (at - (const char*)packed), packed_bytes_linear(rows, cols), s); e != hipSuccess) return e;
    if (!h->gemm_split || cols % 16) return hipSuccess;
    const size_t off = (size_t)(at - (const char*)packed);
    return poem_launch_pack_split_tiles(w_rows, rows, cols, h->gemm_split + off, h->gemm_scales + off / 256, cols / 2, s);
  };
  h->raw.resize(n);
  h->packed.assign(n, nullptr);
  for (int i = 0; i < n; ++i) {
    if (!raw_host[i]) { poem_destroy(h); return POEM_E_ARG; }
    h->raw[i] = (const float*)raw_host[i];
    if (h->specs[i].pack) {
      hipError_t e = poem_launch_pack_linear(h->raw[i], h->specs[i].rows, h->specs[i].cols, cur, s);
      if (e == hipSuccess) e = mirror(h->raw[i], h->specs[i].rows, h->specs[i].cols, cur);
      if (e != hipSuccess) { g_last_hip_error = (int)e; poem_destroy(h); return POEM_E_LAUNCH; }
      h->packed[i] = cur;
      cur += align_up(packed_bytes_linear(h->specs[i].rows, h->specs[i].cols), 256);
    }
  }
  h->bps = bps; h->anchor = anchor; h->anchor_idx = anchor_idx; h->tmpl = template_xyz;
  const int C = cfg->embed, hw = cfg->feat_h * cfg->feat_w;
  if (C >= 128) {
    h->split.resize(2 * cfg->nblocks);
    if (hipMalloc(&h->split_mem, (size_t)2 * cfg->nblocks * ((size_t)3 * C * C * 4 + 256)) != hipSuccess) {
      g_last_hip_error = (int)hipGetLastError(); poem_destroy(h); return POEM_E_LAUNCH;
    }
  }
  {
    h->fused.resize(cfg->nblocks);
    // init-time scratch for raw composites: rows (<= 6C x C), T (C x C), t2 (C x C), bias vectors
    float* raw_rows = (float*)((char*)packed + ((packed_bytes - align_up((size_t)(6 * C * C + 2 * C * C + 8 * C) * 4, 256)) & ~(size_t)255));
    float* raw_T = raw_rows + (size_t)6 * C * C;
    float* raw_tb = raw_T + (size_t)2 * C * C;       // C floats (+ spare)
    bool ok = true;
    auto LOK = [&](hipError_t e) { ok = ok && e == hipSuccess; };
    // rows [slot*C, (slot+1)*C) of raw_rows = A . Bm ; bias slot likewise = A . b1 + b2
    auto comp = [&](int slot, const float* A, const float* Bm, const float* b1, const float* b2, float* bias_out) {
      LOK(poem_launch_compose_weight(A, Bm, raw_rows + (size_t)slot * C * C, C, C, C, s));
      LOK(poem_launch_compose_bias(A, b1, b2, bias_out + (size_t)slot * C, C, C, s));
    };
    auto pack_rows = [&](int nslots, const void** wout) {
      *wout = cur;
      LOK(poem_launch_pack_linear(raw_rows, nslots * C, C, cur, s));
      LOK(mirror(raw_rows, nslots * C, C, cur));
      cur += align_up(packed_bytes_linear(nslots * C, C), 256);
    };
    for (int b = 0; b < cfg->nblocks && ok; ++b) {
      const int bb = h->block_base(b);
      const int a1 = bb + B_A1, a2 = bb + B_A2, vs = bb + B_VS, vc = bb + B_VC;
      auto& f = h->fused[b];
      const float* We = h->raw[bb + B_EMB_W];
      const float* be = h->raw[bb + B_EMB_B];
      // ---- F1: basis-point side
      float* b0 = (float*)cur; cur += align_up((size_t)6 * C * 4, 256);
      comp(0, h->raw[a1 + 2], We, be, h->raw[a1 + 3], b0);
      comp(1, h->raw[a1 + 4], We, be, h->raw[a1 + 5], b0);
      comp(2, h->raw[a2 + 2], We, be, h->raw[a2 + 3], b0);
      comp(3, h->raw[a2 + 4], We, be, h->raw[a2 + 5], b0);
      // T = fc1 . embedding, tb = fc1 . be + b_fc1 ; then (W_g1 w_ks) . T and w_vs . T (no bias of their own)
      float* raw_T2 = raw_T + (size_t)C * C;
      LOK(poem_launch_compose_weight(h->raw[vc + 0], We, raw_T, C, C, C, s));
      LOK(poem_launch_compose_bias(h->raw[vc + 0], be, h->raw[vc + 1], raw_tb, C, C, s));
      LOK(poem_launch_compose_weight(h->raw[vc + 8], h->raw[vc + 13], raw_T2, C, C, C, s));      // W_g1 w_ks
      comp(4, raw_T2, raw_T, raw_tb, nullptr, b0);
      comp(5, h->raw[vc + 14], raw_T, raw_tb, nullptr, b0);
      f.b[0] = b0;
      pack_rows(6, &f.w[0]);
      // ---- F2: query side, embedding | attn.query o embedding
      float* b1v = (float*)cur; cur += align_up((size_t)2 * C * 4, 256);
      LOK(hipMemcpyAsync(raw_rows, We, (size_t)C * C * 4, hipMemcpyDeviceToDevice, s));
      LOK(hipMemcpyAsync(b1v, be, (size_t)C * 4, hipMemcpyDeviceToDevice, s));
      comp(1, h->raw[a1 + 0], We, be, h->raw[a1 + 1], b1v);
      f.b[1] = b1v;
      pack_rows(2, &f.w[1]);
      // ---- F3: (W_g1 w_qs | W_g1 w_ks | w_vs) o fc1 of the vector self attention; the query part carries
      //      cvec = W_g1 b_d2 + b_g1 (vecattn.hip, composed form)
      float* b2v = (float*)cur; cur += align_up((size_t)3 * C * 4, 256);
      float* cvec = raw_tb + 2 * C;
      LOK(poem_launch_compose_bias(h->raw[vs + 8], h->raw[vs + 7], h->raw[vs + 9], cvec, C, C, s));
      LOK(poem_launch_compose_weight(h->raw[vs + 8], h->raw[vs + 12], raw_T, C, C, C, s));       // W_g1 w_qs
      comp(0, raw_T, h->raw[vs + 0], h->raw[vs + 1], cvec, b2v);
      LOK(poem_launch_compose_weight(h->raw[vs + 8], h->raw[vs + 13], raw_T, C, C, C, s));       // W_g1 w_ks
      comp(1, raw_T, h->raw[vs + 0], h->raw[vs + 1], nullptr, b2v);
      comp(2, h->raw[vs + 14], h->raw[vs + 0], h->raw[vs + 1], nullptr, b2v);
      f.b[2] = b2v;
      pack_rows(3, &f.w[2]);
      // ---- [4]: query of the vector cross attention, (W_g1 w_qs) f_self + (W_g1 b_d2 + b_g1)
      float* b4v = (float*)cur; cur += align_up((size_t)C * 4, 256);
      LOK(poem_launch_compose_bias(h->raw[vc + 8], h->raw[vc + 7], h->raw[vc + 9], b4v, C, C, s));
      LOK(poem_launch_compose_weight(h->raw[vc + 8], h->raw[vc + 12], raw_rows, C, C, C, s));
      f.b[4] = b4v;
      pack_rows(1, &f.w[4]);
      // ---- [5], [6]: W_g1 W_d2 of the two vector attentions
      LOK(poem_launch_compose_weight(h->raw[vs + 8], h->raw[vs + 6], raw_rows, C, C, C, s));
      pack_rows(1, &f.w[5]);
      LOK(poem_launch_compose_weight(h->raw[vc + 8], h->raw[vc + 6], raw_rows, C, C, C, s));
      pack_rows(1, &f.w[6]);
      f.b[5] = f.b[6] = nullptr;
      if (h->split_mem) {                      // split images: W_d2, W_g1 W_d2 (re-composed into raw_rows), W_g2
        const size_t img = (size_t)C * C * 4;
        for (int a = 0; a < 2; ++a) {
          const int vb = a == 0 ? vs : vc;
          char* base = (char*)h->split_mem + ((size_t)(2 * b + a)) * (3 * img + 256);
          float* sc = (float*)(base + 3 * img);
          LOK(poem_launch_pack_split(h->raw[vb + 6], C, base, sc + 0, s));
          LOK(poem_launch_compose_weight(h->raw[vb + 8], h->raw[vb + 6], raw_rows, C, C, C, s));
          LOK(poem_launch_pack_split(raw_rows, C, base + img, sc + 1, s));
          LOK(poem_launch_pack_split(h->raw[vb + 10], C, base + 2 * img, sc + 2, s));
          h->split[2 * b + a] = {{base, base + img, base + 2 * img}, sc};
        }
      }
      // F4: reg_branch.0 | intermediate.dense share f_cross; intermediate.dense is (4C, C): four C-row slabs of the raw tensor
      f.w[3] = cur;
      ok = ok && poem_launch_pack_linear(h->raw[bb + B_REG0_W], C, C, cur, s) == hipSuccess;
      LOK(mirror(h->raw[bb + B_REG0_W], C, C, cur));
      cur += packed_bytes_linear(C, C);
      ok = ok && poem_launch_pack_linear(h->raw[bb + B_INT_W], 4 * C, C, cur, s) == hipSuccess;
      LOK(mirror(h->raw[bb + B_INT_W], 4 * C, C, cur));
      cur += packed_bytes_linear(4 * C, C);
      cur = (char*)packed + align_up((size_t)(cur - (char*)packed), 256);
      f.b[3] = (const float*)cur;
      ok = ok && hipMemcpyAsync(cur, h->raw[bb + B_REG0_B], (size_t)C * 4, hipMemcpyDeviceToDevice, s) == hipSuccess;
      ok = ok && hipMemcpyAsync(cur + (size_t)C * 4, h->raw[bb + B_INT_B], (size_t)4 * C * 4, hipMemcpyDeviceToDevice, s) == hipSuccess;
      cur += align_up((size_t)5 * C * 4, 256);
    }
    if (!ok) { g_last_hip_error = (int)hipGetLastError(); poem_destroy(h); return POEM_E_LAUNCH; }
  }
  if ((size_t)(cur - (char*)packed) > packed_weight_extent(cfg)) { poem_destroy(h); return POEM_E_ARG; }      // (the mirror's extent)
  h->idx_dev = (int32_t*)cur;
  cur += align_up((size_t)poem_handle_s::IDX_CAP * 4, 256);
  h->pe_table = (float*)cur;
  cur += align_up(pe_views(cfg->max_views) * C * hw * 4, 256);
  float* sine = (float*)cur;
  rc = poem_pe_table_ex(h->P(T_ADAPT_W), h->R(T_ADAPT_B), C, cfg->feat_h, cfg->feat_w, cfg->max_views, cfg->pe_normalize != 0, sine,
                        h->pe_table, stream);
  if (rc != POEM_OK) { poem_destroy(h); return rc; }
  {
    if (!take_kit(h)) { poem_destroy(h); return POEM_E_LAUNCH; }
  }
  {   // block-0 anchor tables (see poem_handle_s::tables_cached): handle-owned, built here once
    const size_t tf = poem_vector_attention_table_floats(cfg->nquery, C);
    const size_t cx = align_up((size_t)cfg->nquery * 3, 64);
    const size_t qf = align_up((size_t)cfg->nquery * 2 * C, 64);
    if (hipMalloc((void**)&h->tab_mem, (cx + 4 * tf + qf) * sizeof(float)) != hipSuccess) {
      g_last_hip_error = (int)hipGetLastError(); poem_destroy(h); return POEM_E_LAUNCH;
    }
    h->c_canon_xyz = h->tab_mem;
    Plan tp{};
    tp.canon_xyz = h->c_canon_xyz;
    for (int k = 0; k < 2; ++k) {
      tp.tab_g[k] = h->c_tab_g[k] = h->tab_mem + cx + (size_t)(2 * k) * tf;
      tp.tab_p[k] = h->c_tab_p[k] = h->tab_mem + cx + (size_t)(2 * k + 1) * tf;
    }
    rc = build_anchor_tables(h, tp, s, true);
    if (rc != POEM_OK) { poem_destroy(h); return rc; }
    h->c_qeqp0 = h->tab_mem + cx + 4 * tf;
    if (poem_launch_gemm_split(h->R(T_QEMB), C, h->fused[0].w[1], h->fused[0].b[1], nullptr, 0, h->c_qeqp0, 2 * C, cfg->nquery, 2 * C, C,
                               POEM_ACT_NONE, 2 * C, POEM_ACT_NONE, s) != hipSuccess) {
      g_last_hip_error = (int)hipGetLastError(); poem_destroy(h); return POEM_E_LAUNCH;
    }
  }
  *out = h;
  return POEM_OK;
}

void poem_destroy(poem_handle_t h) {
  if (!h) return;
  POEM_TRACE("destroy h=%p graphs=%zu", (void*)h, h->graph_cache.size());
  for (auto e : h->prof_ev) (void)hipEventDestroy(e);
  if (h->tab_mem) (void)hipFree(h->tab_mem);
  if (h->split_mem) (void)hipFree(h->split_mem);
  if (h->gemm_split) (void)hipFree(h->gemm_split);
  if (h->native16) (void)hipFree(h->native16);
  if (h->gemm_scales) (void)hipFree(h->gemm_scales);
  park_execs(h);
  return_kit(h);
  delete h;
}

int poem_set_overlap(poem_handle_t h, int enable) {
  if (!h) return POEM_E_ARG;
  h->overlap = enable != 0;
  return POEM_OK;
}

int poem_set_chains(poem_handle_t h, int enable) {
  if (!h) return POEM_E_ARG;
  h->chains = enable != 0;
  return POEM_OK;
}

int poem_set_option(poem_handle_t h, const char* name, int value) {
  if (!h || !name) return POEM_E_ARG;
  const std::string k(name);
  if (k == "overlap") h->overlap = value != 0;
  else if (k == "anchor_tables") h->anchor_tables = value != 0;
  else if (k == "chains") h->chains = value != 0;
  else if (k == "knn_early") h->knn_early = value != 0;
  else if (k == "fused_sampling") h->fused_sampling = value != 0;
  else if (k == "chain_combine") h->chain_combine = value != 0;
  else if (k == "xattn_merge") { if (value < -1 || value > 1) return POEM_E_ARG; h->xattn_merge = value; }
  else if (k == "tables_first") h->tables_first = value != 0;
  else if (k == "tables_cached") h->tables_cached = value != 0;
  else if (k == "knn_fma") h->knn_fma = value != 0;
  else if (k == "knn_query") {
    if (value < 0 || value > 32) return POEM_E_ARG;
    if (h->precision != POEM_PRECISION_FP32 && value != 0 && value != 32) return POEM_E_UNSUPPORTED;
    h->knn_query = value;
  }
  else if (k == "graphs") h->graphs = value != 0;
  else if (k == "graph_eager") h->graph_eager = value != 0;
  else if (k == "gemm_xcd_map") { poem_gemm_xcd_map(value != 0); g_proc_switches.fetch_and(~1); g_proc_switches.fetch_or(value ? 1 : 0); }
  else if (k == "f1_split") h->f1_split = value != 0;
  else if (k == "xattn_half") { poem_cross_attention_half(value != 0); g_proc_switches.fetch_and(~4); g_proc_switches.fetch_or(value ? 4 : 0); }
  else if (k == "xattn_tail") { poem_cross_attention_tail_halves(value != 0); g_proc_switches.fetch_and(~8); g_proc_switches.fetch_or(value ? 8 : 0); }
  else if (k == "wait_merge") { if (value < -1 || value > 7) return POEM_E_ARG; h->wait_merge = value; }
  else if (k == "d2_first") { if (value < 0 || value > 1) return POEM_E_ARG; h->d2_first = value; }
  else if (k == "va_p1") { if (value < -1 || value > 2) return POEM_E_ARG; h->va_p1 = value; }
  else if (k == "gemm_kslab") { poem_gemm_kslab(value != 0); g_proc_switches.fetch_and(~2); g_proc_switches.fetch_or(value ? 2 : 0); }
  else if (k == "small_batch") h->small_batch = value;
  else if (k == "group_xcd") h->group_xcd = value != 0;
  else if (k == "group_min_views") { if (value < -1) return POEM_E_ARG; h->group_min_views = value; }
  else if (k == "bps_defer") { if (value < 0 || value > 3) return POEM_E_ARG; h->bps_defer = value; }
  else if (k == "chain_tile") { if (value < 0 || value > 3) return POEM_E_ARG; h->chain_tile = value; }
  else return POEM_E_ARG;
  return POEM_OK;
}

// [0] execs cached by this handle  [1] captures  [2] instantiations  [3] graph replays  [4] forwards on plain launches
// [5] view-layout uploads  [6] execs parked in the process  [7] parked execs re-used by update  [8] updates the runtime refused
// [9] (n >= 10) parked execs passed over because their last launch had not finished
int poem_graph_stats(poem_handle_t h, int64_t* out, int n) {
  if (!h || !out || n < 9) return POEM_E_ARG;
  out[0] = (int64_t)h->graph_cache.size();
  out[1] = h->graph_captures; out[2] = h->graph_instantiations; out[3] = h->graph_replays; out[4] = h->plain_forwards;
  out[5] = h->layout_uploads;
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  out[6] = (int64_t)g_parked_execs.size(); out[7] = g_exec_reuses; out[8] = g_exec_update_failures;
  if (n >= 10) out[9] = g_exec_busy_skips;
  return POEM_OK;
}

// The MANO layer of the parametric tail inside the forward (include/poem_hip.h).  The table is the caller's device memory.
int poem_attach_mano(poem_handle_t h, const void* table, int center_idx) {
  if (!h || center_idx < -1 || center_idx > 20) return POEM_E_ARG;
  if (table && !h->cfg.parametric) return POEM_E_UNSUPPORTED;
  h->mano_table = (const float*)table;
  h->mano_center = center_idx;
  return POEM_OK;
}

int poem_set_anchor_tables(poem_handle_t h, int enable) {
  if (!h) return POEM_E_ARG;
  h->anchor_tables = enable != 0;
  return POEM_OK;
}

int poem_set_precision(poem_handle_t h, int mode) {
  if (!h || mode < POEM_PRECISION_FP32 || mode > POEM_PRECISION_SPLIT_F16X3_ALL) return POEM_E_ARG;
  if (mode != POEM_PRECISION_FP32 && (!h->split_mem || !h->gemm_split)) return POEM_E_UNSUPPORTED;      // embed < 128
  if (mode != POEM_PRECISION_FP32 && (h->cfg.knn != 32 || (h->knn_query && h->knn_query != 32)))
    return POEM_E_UNSUPPORTED;      // the split-precision vector attention has no masked form (decoder.cpp checks again)
  h->precision = mode;
  return POEM_OK;
}

int poem_enable_taps(poem_handle_t h, int enable) {
  if (!h) return POEM_E_ARG;
  h->taps = enable != 0;
  return POEM_OK;
}

int64_t poem_tap(poem_handle_t h, const char* name, void* dst, int64_t dst_elems, void* stream) {
  if (!h || !name) return POEM_E_ARG;
  auto it = h->tapmap.find(name);
  if (it == h->tapmap.end()) return POEM_E_ARG;
  if (dst) {
    if (dst_elems < it->second.elems) return POEM_E_ARG;
    HIPCHK(hipMemcpyAsync(dst, it->second.p, (size_t)it->second.elems * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  }
  return it->second.elems;
}
int poem_profile_enable(poem_handle_t h, int max_launches) {
  if (!h || max_launches < 0) return POEM_E_ARG;
  for (auto e : h->prof_ev) (void)hipEventDestroy(e);
  h->prof_ev.clear();
  h->prof_used = 0;
  h->prof_on = max_launches > 0;
  h->prof_kind.assign((size_t)max_launches, 0);
  for (int i = 0; i < 2 * max_launches; ++i) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    h->prof_ev.push_back(e);
  }
  return POEM_OK;
}

static int profile_sum(poem_handle_t h, int kind, int* launches, float* total_ms) {
  float tot = 0.f;
  int n = 0;
  for (int i = 0; i < h->prof_used; ++i) {
    if (h->prof_kind[i] != kind) continue;
    HIPCHK(hipEventSynchronize(h->prof_ev[2 * i + 1]));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->prof_ev[2 * i], h->prof_ev[2 * i + 1]));
    tot += ms;
    ++n;
  }
  *launches = n;
  *total_ms = tot;
  return POEM_OK;
}

int poem_profile_read(poem_handle_t h, int* launches, float* total_ms, int reset) {
  if (!h || !launches || !total_ms) return POEM_E_ARG;
  const int rc = profile_sum(h, 0, launches, total_ms);
  if (rc == POEM_OK && reset) h->prof_used = 0;
  return rc;
}

int poem_profile_read_anchored(poem_handle_t h, int* launches, float* total_ms) {
  if (!h || !launches || !total_ms) return POEM_E_ARG;
  return profile_sum(h, 1, launches, total_ms);
}

int poem_profile_read_stage(poem_handle_t h, int stage, int* launches, float* total_ms) {
  if (!h || !launches || !total_ms || stage < 0 || stage > 7) return POEM_E_ARG;
  return profile_sum(h, stage, launches, total_ms);
}
}  // extern "C"
