// Host-side schedule of one MultiViewUNet.forward on one MI355X: the whole forward
// (camera encoder -> reference-image encoder UNet -> adapter K/V -> main UNet) is issued
// from C++ behind a single C-ABI call; Python only hands over device pointers.
//
// Follows /root/reference/src/models/mvd_unet.py:179-338 (orchestration),
// image_encoder.py:97-112 (encoder pass at t=0, 16 captured maps),
// attention.py:48-188 (adapter branch), camera_encoder.py:160-255 (embedding + FiLM) and
// the diffusers-0.32.2 UNet2DConditionModel layer order.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mvd_hip.h"
#include "kernels.h"

int mvd_launch_silu_to_bf16(const float* x, int64_t n, bf16_t* y, hipStream_t s);

// measurement / bisection switches (bench.py --debug-flags, tools/): bit 0 = no LayerNorm fold through the small-M kernels,
// bit 1 = small-M kernels never split K, bit 2 = small-M kernels off, bit 3 = no split-KV attention, bit 4 = no side stream,
// bit 7 (128) = X-stationary kernels off
static int g_debug_flags = 0;
extern "C" int mvd_debug_set_flags(int flags) { g_debug_flags = flags; return 0; }
int mvd_debug_flags() { return g_debug_flags; }

namespace {

struct Weight { const void* p; int64_t numel; int dtype; };

struct Arena {
  char* base = nullptr;
  size_t cap = 0, off = 0, high = 0;
  bool dry = false;
  void* alloc(size_t bytes) {
    off = (off + 255) & ~size_t(255);
    void* p = dry ? (void*)(uintptr_t)(0x1000 + off) : (void*)(base + off);
    off += bytes;
    if (off > high) high = off;
    return p;
  }
  bool overflow() const { return !dry && high > cap; }
};

struct Act {  // NHWC bf16 activation
  bf16_t* p = nullptr;
  int B = 0, H = 0, W = 0, C = 0;
  int hw() const { return H * W; }
  int rows() const { return B * H * W; }
};

struct FeatureInfo { std::string name; int level; int C; int heads; std::string key; /* weight-slot prefix of the block */ };
constexpr int MVD_REF_ONLY_INTERNAL = 1 << 30;   // set by mvd_engine_reference_encode only (rejected on the public entry)

}  // namespace

struct mvd_engine {
  mvd_config_t cfg;
  std::unordered_map<std::string, Weight> w[2];
  Arena tmp, act, persist;
  void* ws_ptr = nullptr; int64_t ws_bytes = 0;
  void* rc_ptr = nullptr; int64_t rc_bytes = 0;
  std::vector<FeatureInfo> feats;
  // reference cache state (valid after a forward with MVD_USE_IMAGE)
  std::vector<bf16_t*> refkv;       // per feature: [ref_batch*hw][4C]
  std::vector<bf16_t*> feat_keep;   // per feature: [ref_batch][hw][C] when kept
  int rc_batch = 0, rc_h = 0, rc_w = 0; bool rc_valid = false; bool rc_keep = false;
  // global Q2 statistics (mvd_engine_reference_encode / _finish): the encoder pass stops before the normalisation
  float* ref_stats_out = nullptr;   // set only while mvd_engine_reference_encode runs
  bool rc_pending = false;          // raw features are in feat_keep, waiting for the merged statistics
  int64_t ref_pixels(int h, int w) const { int64_t n = 0; for (auto& f : feats) n += (int64_t)(h >> f.level) * (w >> f.level); return n; }
  int64_t feat_pixel_off(int idx) const { int64_t n = 0; for (int i = 0; i < idx; ++i) n += (int64_t)(rc_h >> feats[i].level) * (rc_w >> feats[i].level); return n; }
  float* cam_emb = nullptr; int cam_batch = 0;
  // arrival counters of the in-kernel split-K combine (gemm_sm.hip): one zeroed block per entry call, carved from the
  // workspace, every split-K launch of the call takes its own slice (no counter is ever re-used inside a call)
  unsigned int* cnt_base = nullptr; int cnt_used = 0, cnt_cap = 0;
  // The reference-image encoder pass and the main pass are two chains of kernels of which the second needs the first only at
  // its adapter attentions (feature by feature) -- so the encoder pass is issued on a SIDE STREAM and the main pass waits per
  // feature on an event.  At batch 1 (infer.py) each chain alone leaves most of the chip idle.
  hipStream_t side = nullptr;
  std::vector<hipEvent_t> feat_ev;          // one per feature: its adapter K/V are complete
  hipEvent_t fork_ev = nullptr, join_ev = nullptr;
  bool dual_now = false;                    // this forward runs the encoder pass on the side stream
  int ensure_side_stream() {
    if (side) return 0;
    {
      // the side stream gets the HIGHEST priority: the encoder pass is the producer of every adapter attention's K/V, and with it
      // ahead the main pass fills the gaps (cfg4 cold, same box: default priority 63.06 / 62.83 ms, lowest 62.92 / 62.78, highest
      // 62.69 / 62.58; batch 1 indifferent; mvd_debug_set_flags bit 64 = default priority, for A/B)
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      const int pr = (g_debug_flags & 64) ? 0 : hi;
      if (hipStreamCreateWithPriority(&side, hipStreamNonBlocking, pr) != hipSuccess) { mvd_set_error("engine: hipStreamCreate failed"); return -3; }
    }
    feat_ev.resize(feats.size());
    for (auto& ev : feat_ev) if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { mvd_set_error("engine: hipEventCreate failed"); return -3; }
    if (hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&join_ev, hipEventDisableTiming) != hipSuccess) { mvd_set_error("engine: hipEventCreate failed"); return -3; }
    return 0;
  }
  bool share_encoder = false;       // N4: the encoder pass reads weight set 0 (base UNet == image-encoder UNet)
  // hipGraph replay of whole forwards (mvd_engine_set_graph): one instantiated graph per distinct (arguments, cache state)
  struct HostState {                  // what a forward leaves behind on the host side
    std::vector<bf16_t*> refkv, feat_keep;
    int rc_batch, rc_h, rc_w; bool rc_valid, rc_keep;
    float* cam_emb; int cam_batch;
  };
  HostState host_state() const { return {refkv, feat_keep, rc_batch, rc_h, rc_w, rc_valid, rc_keep, cam_emb, cam_batch}; }
  void set_host_state(const HostState& h) {
    refkv = h.refkv; feat_keep = h.feat_keep; rc_batch = h.rc_batch; rc_h = h.rc_h; rc_w = h.rc_w; rc_valid = h.rc_valid;
    rc_keep = h.rc_keep; cam_emb = h.cam_emb; cam_batch = h.cam_batch;
  }
  struct GraphEntry { std::string key; hipGraph_t g; hipGraphExec_t x; HostState after; };
  bool dual_late = false;                   // set by the main pass once it runs (mostly) alone again: single-stream launch policy
  int dual_late_from = MVD_ENV_INT("MVD_DUAL_LATE_FROM", 99);   // up-block index from which that holds (99: never)
  bool graph_on = false;
  std::vector<GraphEntry> graphs;
  std::vector<std::string> graph_seen;      // keys that ran once un-captured (first-use initialisation happens there)
  void drop_graphs() {
    for (auto& ge : graphs) { (void)hipGraphExecDestroy(ge.x); (void)hipGraphDestroy(ge.g); }
    graphs.clear(); graph_seen.clear();
  }
  ~mvd_engine() {
    drop_graphs();
    for (hipEvent_t ev : ev_pool) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : feat_ev) if (ev) (void)hipEventDestroy(ev);
    if (fork_ev) (void)hipEventDestroy(fork_ev);
    if (join_ev) (void)hipEventDestroy(join_ev);
    if (side) (void)hipStreamDestroy(side);
  }
  std::vector<int> temb_off;        // per resnet offset into the fused time_emb_proj output
  int temb_total = 0;
  std::vector<int> tkv_off;         // per transformer column offset into the fused text K/V projection
  int tkv_total = 0;
  // optional per-kernel-class profiling (HIP events on the launch stream)
  bool prof = false;
  bool prof_overlap = false;      // mvd_engine_set_profiling(e, 2): record launches WITHOUT giving up the two-stream schedule
  struct ProfRec { int cls; double flops; double bytes; hipEvent_t e0, e1; int M, N, K, tag; };
  int prof_M = 0, prof_N = 0, prof_K = 0, prof_tag = 0;   // shape of the launch being recorded (per-shape dump)
  std::vector<ProfRec> prof_recs;
  std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
  hipEvent_t get_event() {
    if (ev_used == ev_pool.size()) { hipEvent_t ev = nullptr; (void)hipEventCreate(&ev); ev_pool.push_back(ev); }
    return ev_pool[ev_used++];
  }
};

// profiling classes: 0..5 = gemm tile config, 8..11 = attention NW (1,2,4,8), 16 groupnorm, 17 layernorm, 18 other
// profile class of a GEMM launch = the KERNEL that runs it (each is a distinct rocprof kernel name): tile config for the
// lock-step kernels of gemm.hip; for the ping-pong kernels of gemm_pp.hip 7 = dense A operand, 13 = implicit-GEMM 3x3
// convolution (incl. the fused 1x1 shortcut / upsample forms), 14 = their split-K forms, 15 = dense with the LayerNorm fold,
// 6 = GEGLU (with or without the fold).  (8..11 are attention.)  30..33 = the X-stationary kernels of gemm_xs.hip (dense, residual,
// LayerNorm, GEGLU).
static inline int gemm_class(const MvdGemmArgs& g, int cfg, int splitk) {
  if (cfg == 8) return 12;
  if (cfg == 7) return splitk > 1 ? 14 : (g.seg[0].mode == MVD_A_CONV3 ? 13 : (g.ln_c1 ? 15 : 7));
  return cfg;
}
int mvd_attention_pick_nw(const MvdAttnArgs& a);

namespace {

#define CHECK(x) do { int _r = (x); if (_r) return _r; } while (0)

#ifdef MVD_PROBE
struct WsCheckRec { std::string slot; int set; bool side; const char* p[4]; size_t n[4]; };
static std::vector<WsCheckRec> g_ws_recs;
static unsigned long long* g_ws_cnt = nullptr;      // pinned: [launch][2] = {elements beyond tolerance, elements compared}
__global__ void ws_check_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, long n, unsigned long long* cnt_l, unsigned long long* cnt, unsigned long long launch) {
  unsigned long long bad = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float va = bflo((unsigned)a[i]), vb = bflo((unsigned)b[i]);
    if (!(fabsf(va - vb) <= 0.03f * fabsf(vb) + 0.05f)) {
      ++bad;
      const unsigned long long k = __hip_atomic_fetch_add(cnt + 8192 * 2 - 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (k < 4096) { cnt[8192 * 2 + 2 * k] = (unsigned long long)i | (launch << 40); cnt[8192 * 2 + 2 * k + 1] = ((unsigned long long)__float_as_uint(va) << 32) | __float_as_uint(vb); }
    }
  }
  if (bad) __hip_atomic_fetch_add(cnt_l, bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(cnt_l + 1, (unsigned long long)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
extern "C" void mvd_debug_ws_check_report(void);
static const char* g_ws_ptrs[4]; static size_t g_ws_lens[4];
static void ws_check_launch(const std::string& slot, int set, bool side, const bf16_t* a, const bf16_t* b, long n, hipStream_t s) {
  if (!g_ws_cnt) {
    (void)hipHostMalloc(&g_ws_cnt, (8192 * 2 + 8192) * sizeof(unsigned long long), hipHostMallocMapped); memset(g_ws_cnt, 0, (8192 * 2 + 8192) * sizeof(unsigned long long));
    atexit(mvd_debug_ws_check_report);
  }
  if (g_ws_recs.size() >= 4096) return;
  g_ws_recs.push_back({slot, set, side, {g_ws_ptrs[0], g_ws_ptrs[1], g_ws_ptrs[2], g_ws_ptrs[3]}, {g_ws_lens[0], g_ws_lens[1], g_ws_lens[2], g_ws_lens[3]}});
  hipLaunchKernelGGL(ws_check_kernel, dim3(64), dim3(256), 0, s, a, b, n, g_ws_cnt + 2 * (g_ws_recs.size() - 1), g_ws_cnt, (unsigned long long)(g_ws_recs.size() - 1));
}
extern "C" void mvd_debug_ws_check_report(void) {
  (void)hipDeviceSynchronize();
  int nbad = 0;
  for (size_t i = 0; i < g_ws_recs.size(); ++i)
    if (g_ws_cnt[2 * i]) { ++nbad; fprintf(stderr, "ws check: launch %zu %s (set %d, %s stream): %llu of %llu elements beyond tolerance\n", i, g_ws_recs[i].slot.c_str(), g_ws_recs[i].set, g_ws_recs[i].side ? "side" : "main", g_ws_cnt[2 * i], g_ws_cnt[2 * i + 1]); }
  {   // where the mismatches sit: (launch, row m, channel n) histogram by row block of 64 and channel tile of 16
    const unsigned long long nlog = g_ws_cnt[8192 * 2 - 1] < 4096 ? g_ws_cnt[8192 * 2 - 1] : 4096;
    std::unordered_map<unsigned long long, int> tiles;
    for (unsigned long long k = 0; k < nlog; ++k) {
      const unsigned long long v = g_ws_cnt[8192 * 2 + 2 * k], launch = v >> 40, idx = v & ((1ull << 40) - 1);
      const int N = 0; (void)N;
      tiles[(launch << 40) | idx] = 1;
    }
    // print the first 24 raw entries (index decoded by the reader: idx = m * N + n)
    for (unsigned long long k = 0; k < nlog && k < 24; ++k) {
      const unsigned long long v = g_ws_cnt[8192 * 2 + 2 * k], w = g_ws_cnt[8192 * 2 + 2 * k + 1];
      const unsigned ua = (unsigned)(w >> 32), ub = (unsigned)w; float fa, fb; memcpy(&fa, &ua, 4); memcpy(&fb, &ub, 4);
      fprintf(stderr, "ws check: mismatch launch %llu idx %llu: ws %g tiled %g\n", v >> 40, v & ((1ull << 40) - 1), fa, fb);
    }
  }
  // operand ranges (x, sc0, sc1, out) of launches of DIFFERENT passes of one forward must never overlap
  static const char* kind[4] = {"x", "sc0", "sc1", "out"};
  int nov = 0;
  for (size_t i = 0; i < g_ws_recs.size() && nov < 12; ++i)
    for (size_t j = i + 1; j < g_ws_recs.size() && j < i + 40 && nov < 12; ++j) {
      if (g_ws_recs[i].side == g_ws_recs[j].side) continue;
      for (int u = 0; u < 4; ++u) for (int v = 0; v < 4; ++v) {
        const char* a0 = g_ws_recs[i].p[u]; const char* b0 = g_ws_recs[j].p[v];
        if (!a0 || !b0 || (u != 3 && v != 3)) continue;               // (an overlap matters when one side writes)
        if (a0 < b0 + g_ws_recs[j].n[v] && b0 < a0 + g_ws_recs[i].n[u]) {
          ++nov; fprintf(stderr, "ws check: OVERLAP launch %zu %s (%s stream) %s [%p, +%zu) with launch %zu %s (%s stream) %s [%p, +%zu)\n", i, g_ws_recs[i].slot.c_str(), g_ws_recs[i].side ? "side" : "main", kind[u], (const void*)a0, g_ws_recs[i].n[u], j, g_ws_recs[j].slot.c_str(), g_ws_recs[j].side ? "side" : "main", kind[v], (const void*)b0, g_ws_recs[j].n[v]);
        }
      }
    }
  fprintf(stderr, "ws check: %d operand overlaps between the two passes\n", nov);
  fprintf(stderr, "ws check: %zu conv_ws launches compared with the tiled kernel, %d with elements beyond tolerance\n", g_ws_recs.size(), nbad);
}
#endif

struct Ctx {
  mvd_engine* e;
  hipStream_t s;
  int set;          // weight set of the current UNet pass
  bool dry;
  int err = 0;
  bool nowait = false;   // split-K combines must not wait for other workgroups (kernels of two streams share the chip)

  const Weight* W(const std::string& name, int dtype, int64_t numel, int set_override = -1) {
    if (dry) return nullptr;  // sizing run: weights need not be registered
    const int st = set_override >= 0 ? set_override : set;
    auto it = e->w[st].find(name);
    if (it == e->w[st].end()) { mvd_set_error("missing weight slot '%s' in set %d", name.c_str(), st); err = -10; return nullptr; }
    if (it->second.dtype != dtype || it->second.numel != numel) {
      mvd_set_error("weight slot '%s' (set %d): expected dtype %d numel %lld, got dtype %d numel %lld", name.c_str(), st, dtype,
                    (long long)numel, it->second.dtype, (long long)it->second.numel);
      err = -11; return nullptr;
    }
    return &it->second;
  }
  const bf16_t* WB(const std::string& n, int64_t numel, int so = -1) { auto* w = W(n, 1, numel, so); return w ? (const bf16_t*)w->p : nullptr; }
  const float* WF(const std::string& n, int64_t numel, int so = -1) { auto* w = W(n, 0, numel, so); return w ? (const float*)w->p : nullptr; }
  bool has(const std::string& n, int so = -1) { const int st = so >= 0 ? so : set; return e->w[st].count(n) != 0; }

  template <class T> T* talloc(size_t n) { return (T*)e->tmp.alloc(n * sizeof(T)); }
  template <class T> T* aalloc(size_t n) { return (T*)e->act.alloc(n * sizeof(T)); }
  Act new_act(int B, int H, int W_, int C, bool longlived) {
    Act a; a.B = B; a.H = H; a.W = W_; a.C = C;
    a.p = longlived ? aalloc<bf16_t>((size_t)B * H * W_ * C) : talloc<bf16_t>((size_t)B * H * W_ * C);
    return a;
  }

  // ------------------------------------------------------------------ op wrappers
  template <class F> int profiled(int cls, double flops, double bytes, F&& launch) {
    if (!e->prof) return launch();
    hipEvent_t e0 = e->get_event(), e1 = e->get_event();
    (void)hipEventRecord(e0, s);
    const int r = launch();
    (void)hipEventRecord(e1, s);
    e->prof_recs.push_back({cls, flops, bytes, e0, e1, e->prof_M, e->prof_N, e->prof_K, e->prof_tag});
    return r;
  }
  // whether the small-M kernels (gemm_sm.hip) are in use at all: ONE predicate for Ctx::gemm's route and for ln_linear's
  // "the fold is available through them" test (a kill switch must make the forward fall back, not fail)
  static bool sm_enabled() {
    static const bool use_sm = MVD_ENV_INT("MVD_GEMM_SM", 1) != 0;
    return use_sm && !(g_debug_flags & 4);
  }
  int gemm(MvdGemmArgs& g) {
    if (err) return err;
    // small problems (one image's feature maps): the latency-oriented kernels of gemm_sm.hip, split-K combined in the kernel
    int sm_tile = 0, sm_ns = 0, sm_S = 1;
    if (sm_enabled() && mvd_gemm_sm_plan(g, &sm_tile, &sm_ns, &sm_S)) {
      if (g_debug_flags & 2) sm_S = 1;
      const size_t mark = e->tmp.off;
      if (sm_S > 1) {
        g.splitk = sm_S; g.splitk_nowait = nowait ? 1 : 0; g.part = talloc<float>((size_t)sm_S * g.M * g.N);
        const int tiles = ((g.M + 63) / 64) * (g.N / 64);           // (an upper bound for every tile shape)
        g.tile_cnt = e->cnt_base + e->cnt_used;
        e->cnt_used += tiles;
        if (!dry && e->cnt_used > e->cnt_cap) { mvd_set_error("forward: split-K tile counters exhausted (%d > %d)", e->cnt_used, e->cnt_cap); return err = -15; }
      }
      int r = 0;
      if (!dry) {
        const double fl = 2.0 * g.M * (double)g.N * g.Ktot;
        e->prof_M = g.M; e->prof_N = g.N; e->prof_K = g.Ktot; e->prof_tag = g.seg[0].mode * 100 + g.geglu * 10 + (sm_S > 1 ? sm_S : 0);
        r = profiled(20 + sm_tile, fl, 0.0, [&] { return mvd_launch_gemm_sm(g, s, sm_tile, sm_ns); });
      }
      e->tmp.off = mark;
      return r;
    }
    // split-K for tile grids that cannot fill the chip: fp32 partials in scoped workspace + a reduce/epilogue pass
    int S = mvd_gemm_pick_splitk(g);
    // While the encoder pass runs beside the main pass (two streams, forward_impl) a launch need not fill the chip by itself --
    // the other stream's kernels take the idle CUs -- so the two devices that exist only to fill it are dropped: the 256x320
    // tile is not split along K at 128 tiles (16x16 level; no fp32 partials, no reduce pass), and 100+ tiles of it are preferred
    // to 512 of the 128x160 tile (measured on one box, cfg4 cold: 65.6 -> 64.6 -> 63.5 ms per step; halving or dropping the
    // four-way split of the 8x8 level instead: no gain / a loss; profiles/r03_probe_dual_stream_policies.log).
    int fc = -1;
    if (e->dual_now && !e->dual_late && !(g_debug_flags & 32) && !g.geglu && !g.ln_c1 && !g.out_f32 && g.N % 320 == 0 && mvd_gemm_pp_applicable(g)) {
      const int cfg = mvd_gemm_pick_config(g);
      const long t7 = (long)((g.M + 255) / 256) * (g.N / 320);
      if (t7 >= 100) { S = 1; if (cfg != 7) fc = 7; }      // (below 100 tiles -- the 8x8 level -- the heuristic's tile and split stand)
    }
    const size_t mark = e->tmp.off;
    if (S > 1) { g.splitk = S; g.part = talloc<float>((size_t)S * g.M * g.N); } else g.splitk = 1;   // (decided: 0 = undecided)
    static const bool trace = MVD_ENV_INT("MVD_TRACE_GEMM", 0) != 0;
    if (trace && !dry) fprintf(stderr, "gemm M=%d N=%d K=%d mode=%d nseg=%d geglu=%d cfg=%d splitk=%d\n", g.M, g.N, g.Ktot, g.seg[0].mode, g.nseg, g.geglu, mvd_gemm_pick_config(g), S);
    int r = 0;
    if (!dry) {
      const double fl = 2.0 * g.M * (double)g.N * g.Ktot;
      e->prof_M = g.M; e->prof_N = g.N; e->prof_K = g.Ktot; e->prof_tag = g.seg[0].mode * 100 + g.geglu * 10 + (S > 1 ? S : 0);
      r = profiled(e->prof ? gemm_class(g, mvd_gemm_pick_config(g), S) : 0, fl, 0.0, [&] { return mvd_launch_gemm(g, s, fc); });
      if (!r && S > 1) r = mvd_launch_splitk_reduce(g, s);
    }
    e->tmp.off = mark;
    return r;
  }

  // X-stationary form (gemm_xs.hip) of a K = 320 projection over many rows: taken when the packed twin `<slot>.wx`
  // (packing.pack_xs: weights in fragment order, bias -- or the LayerNorm-folded c2 -- inside) is registered.  Returns 1 when
  // it launched, 0 when the caller should go on with the general kernels, < 0 on error.  `n_full`: rows of the packed matrix
  // (a pass may use a prefix of them: the q/k/v rows without the adapter's q_ref).
  int try_xs(const bf16_t* x, int K, int M, const std::string& slot, int64_t n_full, int N, bool geglu, bool ln,
             const bf16_t* res, int ldres, void* out, int ldo, int set_override = -1) {
    if (err) return err;
    if (dry || (g_debug_flags & 128) || K != 320 || slot.empty() || !has(slot, set_override)) return 0;
    MvdXsArgs a; memset(&a, 0, sizeof(a));
    a.x = x; a.ldx = K; a.M = M; a.K = K; a.units = N / 32; a.geglu = geglu; a.ln = ln; a.ln_eps = 1e-5f;
    a.res = res; a.ldres = ldres; a.out = (bf16_t*)out; a.ldo = ldo;
    if (N % 64 || !mvd_gemm_xs_applicable(a)) return 0;
    // two workgroups per CU or nothing: with fewer work items than ~1.75 per CU (N = 320 cannot be split into column parts: five
    // store groups) the ping-pong / lock-step tiles fill the chip better
    if ((long)((M + 255) / 256) * mvd_gemm_xs_pick_csplit(a) < 448) return 0;
    a.w = WB(slot, (n_full / 32) * 21 * 512, set_override);
    if (!a.w) return err;
    e->prof_M = M; e->prof_N = N; e->prof_K = K; e->prof_tag = geglu * 10 + (ln ? 1 : 0);
    const int r = profiled(geglu ? 33 : (res ? 31 : (ln ? 32 : 30)), 2.0 * M * (double)N * K, 0.0, [&] { return mvd_launch_gemm_xs(a, s); });
    return r ? r : 1;
  }

  // dense linear: out[M][N] = alpha*(A.W^T + bias) + res       (xs: slot name of the X-stationary twin, if the site has one)
  int linear(const bf16_t* a, const bf16_t* a2, int k1, int k2, int M, const bf16_t* w, const float* bias, int N,
             const bf16_t* res, int ldres, void* out, int ldo, bool geglu = false, bool out_f32 = false, int ldw = 0,
             const std::string& xs = std::string(), int xs_set = -1) {
    if (!xs.empty() && !a2 && !k2 && !geglu && !out_f32) {
      const int r = try_xs(a, k1, M, xs, N, N, false, false, res, ldres, out, ldo, xs_set);
      if (r) return r < 0 ? r : 0;
    }
    MvdGemmArgs g; memset(&g, 0, sizeof(g));
    g.ldw = ldw ? ldw : k1 + k2;
    g.seg[0].p0 = a; g.seg[0].p1 = a2; g.seg[0].c0 = k1; g.seg[0].c1 = k2; g.seg[0].mode = MVD_A_DENSE; g.seg[0].ksize = k1 + k2;
    g.nseg = 1; g.W = w; g.M = M; g.N = N; g.Ktot = k1 + k2; g.rows_per_batch = M; g.outH = 1; g.outW = M;
    g.bias = bias; g.res = res; g.ldres = ldres; g.alpha = 1.f; g.geglu = geglu; g.out = out; g.ldo = ldo; g.out_f32 = out_f32;
    return gemm(g);
  }

  // out = LayerNorm(x; gamma, beta).W^T + b (optionally GEGLU).  With the folded slots <slot>.wf (W.diag(gamma), bf16) and
  // <slot>.cf ([2][n_full]: c1 | c2) registered and a shape the fused kernel takes, ONE launch (gemm_pp.hip, LNF) that reads
  // the un-normalised rows; otherwise ln_kernel into `ln_tmp` and the plain GEMM.
  int ln_linear(const bf16_t* x, int M, int C, const std::string& ln_key, const std::string& slot, int64_t n_full, int N,
                const float* bias, bf16_t* ln_tmp, void* out, int ldo, bool geglu) {
    MvdGemmArgs g; memset(&g, 0, sizeof(g));
    g.ldw = C; g.seg[0].p0 = x; g.seg[0].c0 = C; g.seg[0].mode = MVD_A_DENSE; g.seg[0].ksize = C; g.nseg = 1;
    g.M = M; g.N = N; g.Ktot = C; g.rows_per_batch = M; g.outH = 1; g.outW = M; g.alpha = 1.f; g.geglu = geglu; g.out = out; g.ldo = ldo;
    {
      const int r = try_xs(x, C, M, slot + ".wx", n_full, N, geglu, true, nullptr, 0, out, ldo);
      if (r) return r < 0 ? r : 0;
    }
    static const bool use_fold = MVD_ENV_INT("MVD_LN_FOLD", 1) != 0;
    auto sm_fold_ok = [&]() {      // the small-M kernels (batch 1) fold at every level
      MvdGemmArgs t = g;
      static const float dummy = 0.f;
      t.ln_c1 = &dummy; t.bias = &dummy; t.W = reinterpret_cast<const bf16_t*>(&dummy);
      int a_, b_, c_;
      return sm_enabled() && !(g_debug_flags & 1) && mvd_gemm_sm_plan(t, &a_, &b_, &c_);
    };
    if (use_fold && !dry && has(slot + ".wf") && has(slot + ".cf") && (mvd_gemm_ln_fold_ok(g) || sm_fold_ok())) {
      const float* cf = WF(slot + ".cf", 2 * n_full);
      g.W = WB(slot + ".wf", n_full * C);
      g.ln_c1 = cf; g.bias = cf ? cf + n_full : nullptr; g.ln_eps = 1e-5f;
      return gemm(g);
    }
    CHECK(layernorm(x, M, C, WF(ln_key + ".g", C), WF(ln_key + ".b", C), ln_tmp));
    g.seg[0].p0 = ln_tmp; g.W = WB(slot + ".w", n_full * C); g.bias = bias;
    return gemm(g);
  }

  // Weight-streaming form (conv_ws.hip) of a stride-1 resnet convolution on one image's 8- or 16-wide map, taken when the packed
  // twin `<slot>.ws` (packing.pack_ws) is registered: batch 1 at the 8x8 / 16x16 / 32x32 levels is a weight stream, which the
  // implicit-GEMM kernels cut along K with a rendezvous between workgroups (21-63 us against 15-48 us, profiles/r04_probe_conv_ws.log).
  // Returns 1 when it launched, 0 when the caller should go on, < 0 on error.
  int try_ws(const Act& x, int ups, const std::string& slot, const float* bias, const float* rowvec, int ld_rowvec, const bf16_t* res,
             const bf16_t* sc0, const bf16_t* sc1, int scc0, int scc1, Act& out) {
    if (err) return err;
    if (dry || (g_debug_flags & 256) || slot.empty() || !has(slot)) return 0;
    MvdWsArgs a; memset(&a, 0, sizeof(a));
    a.x = x.p; a.B = x.B; a.H = x.H; a.W = x.W; a.C = x.C; a.ups = ups; a.sc0 = sc0; a.sc1 = sc1; a.scc0 = scc0; a.scc1 = scc1;
    a.bias = bias; a.rowvec = rowvec; a.ld_rowvec = ld_rowvec; a.res = res; a.ldres = out.C; a.out = out.p; a.ldo = out.C;
    a.M = out.rows(); a.N = out.C;
    // (debug flags, A/B only: 512 = maps of at most 256 pixels, 1024 = 64-pixel blocks everywhere, 2048 = no fused shortcut)
    if ((g_debug_flags & 512) && a.M > 256) return 0;
    if (g_debug_flags & 1024) a.variant = 1;
    if ((g_debug_flags & 2048) && scc0) return 0;
    if ((g_debug_flags & 4096) && nowait) return 0;             // (4096: not in the encoder pass of a two-stream forward; 8192: only there)
    if ((g_debug_flags & 8192) && !nowait) return 0;
    if (out.H != (ups ? 2 : 1) * x.H || out.W != (ups ? 2 : 1) * x.W || out.B != x.B || a.M > 1024) return 0;
    // every 64-pixel row block streams its column tile's whole weight panel (from L2 at best): with 16 row blocks x 80 column tiles
    // (the 16 -> 32 upsampling convolution, N = 1280) that is a tie with the tiled kernel (62.9 vs 60.5 us) -- not taken
    if ((long)(a.M / 64) * (a.N / 16) > 1000) return 0;     // (one 32x32 map at most: beyond, the tiled kernels have the FLOPs to win)
    a.w = reinterpret_cast<const bf16_t*>(&a);                                     // (placeholder for the shape test)
    if (!mvd_conv_ws_applicable(a)) return 0;
    a.w = WB(slot, (int64_t)mvd_conv_ws_packed_elems(x.C, scc0 + scc1, out.C));
    if (!a.w) return err;
    e->prof_M = a.M; e->prof_N = a.N; e->prof_K = 9 * x.C + scc0 + scc1; e->prof_tag = 100 + (scc0 ? 1 : 0);
    const int r = profiled(34, 2.0 * a.M * (double)a.N * (9.0 * x.C + scc0 + scc1), 0.0, [&] { return mvd_launch_conv_ws(a, s); });
    if (!r && (g_debug_flags & 16384)) return 0;              // (16384, A/B only: the tiled kernel runs as well and overwrites the result)
#ifdef MVD_PROBE
    // probe builds, flag 32768: the tiled kernel computes the same launch into a scratch tensor and a compare kernel counts the
    // elements that differ by more than rounding into pinned host memory -- no synchronise; mvd_debug_ws_check_report() prints
    if (!r && (g_debug_flags & 32768)) {
      const size_t mark = e->tmp.off;
      Act chk = new_act(out.B, out.H, out.W, out.C, false);
      const bf16_t* wold = WB(slot.substr(0, slot.size() - 1), (int64_t)out.C * (9 * x.C + scc0 + scc1));   // "<..>.convN.ws" -> "<..>.convN.w"
      if (wold) {
        const int saved = g_debug_flags; g_debug_flags |= 256;
        conv3(x, 1, ups, wold, bias, rowvec, ld_rowvec, res, sc0, sc1, scc0, scc1, chk);
        g_debug_flags = saved;
        g_ws_ptrs[0] = (const char*)x.p; g_ws_lens[0] = (size_t)x.rows() * x.C * 2;   // (input rows)
        g_ws_ptrs[1] = (const char*)sc0; g_ws_lens[1] = (size_t)out.rows() * scc0 * 2;
        g_ws_ptrs[2] = (const char*)sc1; g_ws_lens[2] = (size_t)out.rows() * scc1 * 2;
        g_ws_ptrs[3] = (const char*)out.p; g_ws_lens[3] = (size_t)out.rows() * out.C * 2;
        ws_check_launch(slot, set, nowait, out.p, chk.p, (long)out.rows() * out.C, s);
      }
      e->tmp.off = mark;
    }
#endif
    return r ? r : 1;
  }

  int conv3(const Act& x, int stride, int ups, const bf16_t* w, const float* bias, const float* rowvec, int ld_rowvec,
            const bf16_t* res, const bf16_t* sc0, const bf16_t* sc1, int scc0, int scc1, Act& out, const std::string& ws = std::string()) {
    if (!ws.empty() && stride == 1) {
      const int r = try_ws(x, ups, ws, bias, rowvec, ld_rowvec, res, sc0, sc1, scc0, scc1, out);
      if (r) return r < 0 ? r : 0;
    }
    MvdGemmArgs g; memset(&g, 0, sizeof(g));
    g.seg[0].p0 = x.p; g.seg[0].c0 = x.C; g.seg[0].mode = MVD_A_CONV3; g.seg[0].ksize = 9 * x.C;
    g.seg[0].inH = x.H; g.seg[0].inW = x.W; g.seg[0].stride = stride; g.seg[0].ups = ups;
    g.nseg = 1; g.Ktot = 9 * x.C;
    if (sc0) {
      g.seg[1].p0 = sc0; g.seg[1].p1 = sc1; g.seg[1].c0 = scc0; g.seg[1].c1 = scc1; g.seg[1].mode = MVD_A_DENSE;
      g.seg[1].ksize = scc0 + scc1; g.nseg = 2; g.Ktot += scc0 + scc1;
    }
    g.W = w; g.ldw = g.Ktot; g.M = out.rows(); g.N = out.C; g.rows_per_batch = out.hw(); g.outH = out.H; g.outW = out.W;
    g.bias = bias; g.rowvec = rowvec; g.ld_rowvec = ld_rowvec; g.res = res; g.ldres = out.C; g.alpha = 1.f;
    g.out = out.p; g.ldo = out.C;
    return gemm(g);
  }

  int groupnorm(const bf16_t* x0, const bf16_t* x1, int c0, int c1, int B, int hw, float eps, const float* g, const float* b,
                int silu, bf16_t* y) {
    if (err) return err;
    const int groups = e->cfg.norm_num_groups;
    float* ws = talloc<float>((size_t)B * MVD_GN_MAXCHUNK * groups * 2);
    if (dry) return 0;
    const double by = 3.0 * B * (double)hw * (c0 + c1) * 2;   // 2 reads + 1 write of the activation
    e->prof_M = B * hw; e->prof_N = c0 + c1; e->prof_K = 0; e->prof_tag = 2000 + silu;
    return profiled(16, 0.0, by, [&] { return mvd_launch_groupnorm(x0, x1, c0, c1, B, hw, groups, eps, g, b, silu, y, ws, s); });
  }
  int layernorm(const bf16_t* x, int rows, int c, const float* g, const float* b, bf16_t* y) {
    if (err) return err; if (dry) return 0;
    e->prof_M = rows; e->prof_N = c; e->prof_K = 0; e->prof_tag = 3000;
    return profiled(17, 0.0, 2.0 * rows * (double)c * 2, [&] { return mvd_launch_layernorm(x, rows, c, 1e-5f, g, b, y, s); });
  }
  int attention(MvdAttnArgs& a) {
    if (err) return err;
    // batch 1: split the keys over several workgroups (partials + counters from the scoped workspace / the call's counter block)
    const size_t mark = e->tmp.off;
    const int ns = (g_debug_flags & 8) ? 1 : mvd_attention_pick_split(a);
    if (ns > 1) {
      a.nsplit = ns;
      a.split_ws = e->tmp.alloc(mvd_attention_split_ws_bytes(a, ns));
      a.split_cnt = e->cnt_base + e->cnt_used;
      e->cnt_used += mvd_attention_split_counters(a);
      if (!dry && e->cnt_used > e->cnt_cap) { mvd_set_error("forward: tile counters exhausted (%d > %d)", e->cnt_used, e->cnt_cap); return err = -15; }
    }
    e->tmp.off = mark;
    if (dry) return 0;
    double fl = 0;
    for (int i = 0; i < a.nprob; ++i) fl += 4.0 * a.batch * a.heads * (double)a.p[i].nq * a.p[i].nk * 64;
    e->prof_M = a.p[0].nq; e->prof_N = a.p[0].nk; e->prof_K = a.heads; e->prof_tag = 1000 + a.nprob;
    return profiled(e->prof ? 8 + mvd_attention_pick_nw(a) : 8, fl, 0.0, [&] { return mvd_launch_attention(a, s); });
  }
};

// ---------------------------------------------------------------------- structure helpers
struct ResnetDesc { std::string key; int cin0, cin1, cout; };

std::vector<ResnetDesc> enumerate_resnets(const mvd_config_t& c) {
  std::vector<ResnetDesc> v;
  const int n = c.num_levels, L = c.layers_per_block;
  int prev = c.block_out_channels[0];
  for (int i = 0; i < n; ++i) {
    const int co = c.block_out_channels[i];
    for (int j = 0; j < L; ++j) v.push_back({"down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), j == 0 ? prev : co, 0, co});
    prev = co;
  }
  const int cm = c.block_out_channels[n - 1];
  v.push_back({"mid_block.resnets.0", cm, 0, cm});
  v.push_back({"mid_block.resnets.1", cm, 0, cm});
  int prev_out = cm;
  for (int i = 0; i < n; ++i) {
    const int out_c = c.block_out_channels[n - 1 - i];
    const int in_c = c.block_out_channels[n - 1 - (i + 1 < n ? i + 1 : n - 1)];
    for (int j = 0; j <= L; ++j) {
      const int skip = (j == L) ? in_c : out_c;
      const int hid = (j == 0) ? prev_out : out_c;
      v.push_back({"up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), hid, skip, out_c});
    }
    prev_out = out_c;
  }
  return v;
}

std::vector<FeatureInfo> enumerate_features(const mvd_config_t& c) {
  std::vector<FeatureInfo> f;
  const int n = c.num_levels, L = c.layers_per_block;
  for (int i = 0; i + 1 < n; ++i)
    for (int j = 0; j < L; ++j) f.push_back({"down_block_" + std::to_string(i) + "_attn_" + std::to_string(j), i, c.block_out_channels[i], c.num_heads[i],
                   "down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j)});
  f.push_back({"mid_block_attn_0", n - 1, c.block_out_channels[n - 1], c.num_heads[n - 1], "mid_block.attentions.0"});
  for (int i = 1; i < n; ++i)
    for (int j = 0; j <= L; ++j) f.push_back({"up_block_" + std::to_string(i) + "_attn_" + std::to_string(j), n - 1 - i, c.block_out_channels[n - 1 - i], c.num_heads[n - 1 - i],
                   "up_blocks." + std::to_string(i) + ".attentions." + std::to_string(j)});
  return f;
}

struct PassOpts {
  bool adapter = false;       // add the cross-view attention branch (needs e->refkv)
  bool film = false;          // apply camera FiLM hooks
  bool capture = false;       // encoder pass: produce reference K/V (+ keep features)
  bool defer_norm = false;    // capture, global-statistics form: per-pixel local (mean, M2) go to `stats`, the features are
  float* stats = nullptr;     // kept raw and normalisation + K/V projection wait for mvd_engine_reference_finish
  int ref_batch = 0;          // batch of the reference features (Q4 re-chunking)
  const std::unordered_map<std::string, std::pair<float*, float*>>* film_ss = nullptr;  // name -> (scale, shift) [B][dim]
};

// ---------------------------------------------------------------------- one UNet pass
struct UNetPass {
  Ctx& c;
  const mvd_config_t& cfg;
  PassOpts o;
  int B, H0, W0, L;
  const bf16_t* text;      // [B*L][xdim] bf16
  const float* tproj;      // [B][temb_total] fp32 (all time_emb_proj outputs incl. bias)
  bf16_t* tkv = nullptr;   // [B*L][tkv_total] bf16: text K/V of every attn2 site (one GEMM per pass)
  int resnet_idx = 0;
  int feat_idx = 0;

  int resnet(const std::string& key, const Act& x0, const Act* x1, int cout, Act& out) {
    const int cin0 = x0.C, cin1 = x1 ? x1->C : 0, cin = cin0 + cin1;
    const int B_ = x0.B, hw = x0.hw();
    const size_t mark = c.e->tmp.off;
    Act t1 = c.new_act(B_, x0.H, x0.W, cin, false);
    CHECK(c.groupnorm(x0.p, x1 ? x1->p : nullptr, cin0, cin1, B_, hw, cfg.norm_eps, c.WF(key + ".norm1.g", cin), c.WF(key + ".norm1.b", cin), 1, t1.p));
    Act h1 = c.new_act(B_, x0.H, x0.W, cout, false);
    const int toff = c.e->temb_off[resnet_idx++];
    CHECK(c.conv3(t1, 1, 0, c.WB(key + ".conv1.w", (int64_t)cout * 9 * cin), c.WF(key + ".conv1.b", cout), tproj + toff, c.e->temb_total,
                  nullptr, nullptr, nullptr, 0, 0, h1, key + ".conv1.ws"));
    Act t2 = c.new_act(B_, x0.H, x0.W, cout, false);
    CHECK(c.groupnorm(h1.p, nullptr, cout, 0, B_, hw, cfg.norm_eps, c.WF(key + ".norm2.g", cout), c.WF(key + ".norm2.b", cout), 1, t2.p));
    if (cin != cout) {
      CHECK(c.conv3(t2, 1, 0, c.WB(key + ".conv2.w", (int64_t)cout * (9 * cout + cin)), c.WF(key + ".conv2.b", cout), nullptr, 0, nullptr,
                    x0.p, x1 ? x1->p : nullptr, cin0, cin1, out, key + ".conv2.ws"));
    } else {
      CHECK(c.conv3(t2, 1, 0, c.WB(key + ".conv2.w", (int64_t)cout * 9 * cout), c.WF(key + ".conv2.b", cout), nullptr, 0, x0.p, nullptr,
                    nullptr, 0, 0, out, key + ".conv2.ws"));
    }
    c.e->tmp.off = mark;
    return c.err;
  }

  int transformer(const std::string& key, const Act& x, int heads, Act& out) {
    const int C = x.C, M = x.rows(), hw = x.hw(), B_ = x.B;
    const FeatureInfo& fi = c.e->feats[feat_idx];
    const bool ad = o.adapter;
    // set 0 may carry the adapter rows/columns even when this pass does not use them (no image conditioning):
    // q/k/v rows are a prefix of the fused matrix, the out-projection is addressed with its packed row stride.
    const bool packed = c.set == 0 && (c.dry || c.has(key + ".ref_kv.w", 0));
    if (ad && !packed) { mvd_set_error("adapter requested but no adapter weights registered for %s", key.c_str()); return -14; }
    const int ld_out = packed ? 2 * C : C;
    const char* bias_slot = (packed && !ad) ? ".out.b0" : ".out.b";
    const size_t mark = c.e->tmp.off;
    bf16_t* n0 = c.talloc<bf16_t>((size_t)M * C);
    CHECK(c.groupnorm(x.p, nullptr, C, 0, B_, hw, 1e-6f, c.WF(key + ".norm.g", C), c.WF(key + ".norm.b", C), 0, n0));
    bf16_t* h = c.talloc<bf16_t>((size_t)M * C);
    CHECK(c.linear(n0, nullptr, C, 0, M, c.WB(key + ".proj_in.w", (int64_t)C * C), c.WF(key + ".proj_in.b", C), C, nullptr, 0, h, C, false, false, 0, key + ".proj_in.wx"));
    bf16_t* ln = n0;  // reuse
    const float scale = 0.125f;   // (folded, with log2 e, into the packed to_q / to_q_ref rows: a.prescaled)
    bf16_t* o_self = c.talloc<bf16_t>((size_t)M * C);
    bf16_t* o_ref = ad ? c.talloc<bf16_t>((size_t)M * C) : nullptr;
    const bf16_t* rkv = ad ? c.e->refkv[feat_idx] : nullptr;
    // Q4: K/V rows of the reference are re-chunked by the HIDDEN batch size
    const int ref_nk = ad ? (int)((int64_t)o.ref_batch * hw / B_) : 0;
    if (ad && (int64_t)ref_nk * B_ != (int64_t)o.ref_batch * hw) { mvd_set_error("adapter: ref tokens %d x %d not divisible by batch %d", o.ref_batch, hw, B_); return -12; }

    // (dual-stream forwards: this feature's adapter K/V come from the encoder pass on the side stream)
    if (ad && c.e->dual_now && !c.dry && !c.err && hipStreamWaitEvent(c.s, c.e->feat_ev[feat_idx], 0) != hipSuccess) { mvd_set_error("forward: hipStreamWaitEvent failed"); return -3; }
    // ---- attn1 (self) + adapter branch "<feature>_self"
    {
      const int nq = ad ? 4 * C : 3 * C;
      bf16_t* qkv = c.talloc<bf16_t>((size_t)M * nq);
      CHECK(c.ln_linear(h, M, C, key + ".ln1", key + ".attn1.qkv", (int64_t)(packed ? 4 : 3) * C, nq, nullptr, ln, qkv, nq, false));
      MvdAttnArgs a; memset(&a, 0, sizeof(a));
      a.batch = B_; a.heads = heads; a.scale = scale; a.prescaled = 1; a.nprob = ad ? 2 : 1;
      a.p[0] = {qkv, qkv + C, qkv + 2 * C, o_self, nq, nq, nq, C, (int64_t)hw * nq, (int64_t)hw * nq, (int64_t)hw * nq, (int64_t)hw * C, hw, hw};
      if (ad) a.p[1] = {qkv + 3 * C, rkv, rkv + C, o_ref, nq, 4 * C, 4 * C, C, (int64_t)hw * nq, (int64_t)ref_nk * 4 * C, (int64_t)ref_nk * 4 * C, (int64_t)hw * C, hw, ref_nk};
      CHECK(c.attention(a));
      const int kout = ad ? 2 * C : C;
      (void)kout;
      CHECK(c.linear(o_self, o_ref, C, ad ? C : 0, M, c.WB(key + ".attn1.out.w", (int64_t)C * ld_out), c.WF(key + ".attn1" + bias_slot, C), C, h, C, h, C, false, false, ld_out,
                     ad ? std::string() : key + ".attn1.out.wx"));
    }
    // ---- attn2 (text cross) + adapter branch "<feature>_cross"
    {
      const int nq = ad ? 2 * C : C;
      bf16_t* q2 = c.talloc<bf16_t>((size_t)M * nq);
      const bf16_t* kv2 = tkv + c.e->tkv_off[feat_idx];   // [B*L][2C] slice of the fused text K/V
      const int ldkv = c.e->tkv_total;
      CHECK(c.ln_linear(h, M, C, key + ".ln2", key + ".attn2.q", (int64_t)(packed ? 2 : 1) * C, nq, nullptr, ln, q2, nq, false));
      MvdAttnArgs a; memset(&a, 0, sizeof(a));
      a.batch = B_; a.heads = heads; a.scale = scale; a.prescaled = 1; a.nprob = ad ? 2 : 1;
      a.p[0] = {q2, kv2, kv2 + C, o_self, nq, ldkv, ldkv, C, (int64_t)hw * nq, (int64_t)L * ldkv, (int64_t)L * ldkv, (int64_t)hw * C, hw, L};
      if (ad) a.p[1] = {q2 + C, rkv + 2 * C, rkv + 3 * C, o_ref, nq, 4 * C, 4 * C, C, (int64_t)hw * nq, (int64_t)ref_nk * 4 * C, (int64_t)ref_nk * 4 * C, (int64_t)hw * C, hw, ref_nk};
      CHECK(c.attention(a));
      const int kout = ad ? 2 * C : C;
      (void)kout;
      CHECK(c.linear(o_self, o_ref, C, ad ? C : 0, M, c.WB(key + ".attn2.out.w", (int64_t)C * ld_out), c.WF(key + ".attn2" + bias_slot, C), C, h, C, h, C, false, false, ld_out,
                     ad ? std::string() : key + ".attn2.out.wx"));
    }
    // ---- GEGLU feed-forward
    {
      bf16_t* ff = c.talloc<bf16_t>((size_t)M * 4 * C);
      CHECK(c.ln_linear(h, M, C, key + ".ln3", key + ".ff1", (int64_t)8 * C, 8 * C, c.WF(key + ".ff1.b", 8 * C), ln, ff, 4 * C, true));
      CHECK(c.linear(ff, nullptr, 4 * C, 0, M, c.WB(key + ".ff2.w", (int64_t)C * 4 * C), c.WF(key + ".ff2.b", C), C, h, C, h, C));
    }
    CHECK(c.linear(h, nullptr, C, 0, M, c.WB(key + ".proj_out.w", (int64_t)C * C), c.WF(key + ".proj_out.b", C), C, x.p, C, out.p, C, false, false, 0, key + ".proj_out.wx"));
    c.e->tmp.off = mark;

    if (o.capture) {  // encoder pass: reference normalisation (Q2) + adapter K/V for both processors of this feature
      const size_t mk = c.e->tmp.off;
      bf16_t* rn = c.talloc<bf16_t>((size_t)M * C);        // (also sized in the deferred form: _finish needs the same scratch)
      if (o.defer_norm) {
        if (!c.dry && !c.err) CHECK(mvd_launch_refstats(out.p, B_, hw, C, o.stats + 2 * c.e->feat_pixel_off(feat_idx), c.s));
      } else {
        if (!c.dry && !c.err) CHECK(mvd_launch_refnorm(out.p, B_, hw, C, rn, c.s));
        CHECK(c.linear(rn, nullptr, C, 0, M, c.WB(key + ".ref_kv.w", (int64_t)4 * C * C, 0), nullptr, 4 * C, nullptr, 0, c.e->refkv[feat_idx], 4 * C, false, false, 0,
                       key + ".ref_kv.wx", 0));
        if (c.e->dual_now && !c.dry && !c.err && hipEventRecord(c.e->feat_ev[feat_idx], c.s) != hipSuccess) { mvd_set_error("forward: hipEventRecord failed"); return -3; }
      }
      if (c.e->rc_keep && !c.dry && !c.err)
        CHECK((int)hipMemcpyAsync(c.e->feat_keep[feat_idx], out.p, (size_t)M * C * sizeof(bf16_t), hipMemcpyDeviceToDevice, c.s));
      c.e->tmp.off = mk;
    }
    (void)fi;
    ++feat_idx;
    return c.err;
  }

  int film(const std::string& name, const Act& x, Act& out) {
    auto it = o.film_ss->find(name);
    if (it == o.film_ss->end()) { out = x; return 0; }   // unknown modulator ("mid_0") -> identity (Q3)
    if (c.dry || c.err) return c.err;
    return mvd_launch_film(x.p, x.B, x.hw(), x.C, it->second.first, it->second.second, x.C, out.p, c.s);
  }

  int run(const Act& x_in, float* out_nchw) {
    const int n = cfg.num_levels, Lb = cfg.layers_per_block;
    const int C0 = cfg.block_out_channels[0];
    std::vector<Act> skips;
    // text K/V for all attn2 sites of this pass: one [B*L][xdim] x [tkv_total][xdim]^T GEMM instead of 16 small ones
    tkv = c.aalloc<bf16_t>((size_t)B * L * c.e->tkv_total);
    CHECK(c.linear(text, nullptr, cfg.cross_attention_dim, 0, B * L, c.WB("text_kv.w", (int64_t)c.e->tkv_total * cfg.cross_attention_dim), nullptr,
                   c.e->tkv_total, nullptr, 0, tkv, c.e->tkv_total));
    Act h = c.new_act(B, H0, W0, C0, true);
    // conv_in: x_in holds im2col rows [B*H*W][64] (k = tap*Cin + ch, zero padded) -> one K=64 MFMA GEMM
    CHECK(c.linear(x_in.p, nullptr, 64, 0, B * H0 * W0, c.WB("conv_in.w", (int64_t)C0 * 64), c.WF("conv_in.b", C0), C0, nullptr, 0, h.p, C0));
    skips.push_back(h);
    for (int i = 0; i < n; ++i) {
      const int co = cfg.block_out_channels[i];
      const std::string bk = "down_blocks." + std::to_string(i);
      for (int j = 0; j < Lb; ++j) {
        Act r = c.new_act(B, h.H, h.W, co, true);
        CHECK(resnet(bk + ".resnets." + std::to_string(j), h, nullptr, co, r));
        h = r;
        if (i + 1 < n) {
          Act t = c.new_act(B, h.H, h.W, co, true);
          CHECK(transformer(bk + ".attentions." + std::to_string(j), h, cfg.num_heads[i], t));
          h = t;
        }
        skips.push_back(h);
      }
      if (i + 1 < n) {
        Act d = c.new_act(B, (h.H + 1) / 2, (h.W + 1) / 2, co, true);
        CHECK(c.conv3(h, 2, 0, c.WB(bk + ".down.w", (int64_t)co * 9 * co), c.WF(bk + ".down.b", co), nullptr, 0, nullptr, nullptr, nullptr, 0, 0, d));
        h = d;
        skips.push_back(h);
      }
      if (o.film) {  // hook modulates the block's returned hidden_states only; skips stay unmodulated (Q6)
        Act m = c.new_act(B, h.H, h.W, h.C, true);
        Act res = m;
        CHECK(film("down_" + std::to_string(i), h, res));
        h = res;
      }
    }
    {
      const int cm = cfg.block_out_channels[n - 1];
      Act r = c.new_act(B, h.H, h.W, cm, true);
      CHECK(resnet("mid_block.resnets.0", h, nullptr, cm, r));
      Act t = c.new_act(B, h.H, h.W, cm, true);
      CHECK(transformer("mid_block.attentions.0", r, cfg.num_heads[n - 1], t));
      Act r2 = c.new_act(B, h.H, h.W, cm, true);
      CHECK(resnet("mid_block.resnets.1", t, nullptr, cm, r2));
      h = r2;
      if (o.film) { Act res = h; CHECK(film("mid_0", h, res)); h = res; }
    }
    for (int i = 0; i < n; ++i) {
      const int co = cfg.block_out_channels[n - 1 - i];
      const std::string bk = "up_blocks." + std::to_string(i);
      // main pass of a two-stream forward: from up block `dual_late_from` on the encoder pass (41 % of the work, ahead by design
      // and on the higher-priority stream) has normally drained -- launches must fill the chip by themselves again
      if (!o.capture && i >= c.e->dual_late_from) c.e->dual_late = true;
      for (int j = 0; j <= Lb; ++j) {
        Act skip = skips.back(); skips.pop_back();
        if (skip.H != h.H || skip.W != h.W) { mvd_set_error("up block %d: skip %dx%d vs hidden %dx%d (odd latent size unsupported)", i, skip.H, skip.W, h.H, h.W); return -13; }
        Act r = c.new_act(B, h.H, h.W, co, true);
        CHECK(resnet(bk + ".resnets." + std::to_string(j), h, &skip, co, r));
        h = r;
        if (i > 0) {
          Act t = c.new_act(B, h.H, h.W, co, true);
          CHECK(transformer(bk + ".attentions." + std::to_string(j), h, cfg.num_heads[n - 1 - i], t));
          h = t;
        }
      }
      if (i + 1 < n) {
        Act u = c.new_act(B, h.H * 2, h.W * 2, co, true);
        CHECK(c.conv3(h, 1, 1, c.WB(bk + ".up.w", (int64_t)co * 9 * co), c.WF(bk + ".up.b", co), nullptr, 0, nullptr, nullptr, nullptr, 0, 0, u, bk + ".up.ws"));
        h = u;
      }
      if (o.film) { Act res = h; CHECK(film("up_" + std::to_string(i), h, res)); h = res; }
    }
    if (out_nchw) {
      Act t = c.new_act(B, h.H, h.W, C0, false);
      CHECK(c.groupnorm(h.p, nullptr, C0, 0, B, h.hw(), cfg.norm_eps, c.WF("conv_norm_out.g", C0), c.WF("conv_norm_out.b", C0), 1, t.p));
      const bf16_t* wo = c.WB("conv_out.w", (int64_t)cfg.out_channels * 9 * C0);
      const float* bo = c.WF("conv_out.b", cfg.out_channels);
      if (!c.dry && !c.err) CHECK(mvd_launch_conv_out(t.p, B, h.H, h.W, C0, wo, bo, cfg.out_channels, out_nchw, c.s));
    }
    return c.err;
  }
};

// time embedding -> fused time_emb_proj for every resnet: returns [B][temb_total] fp32
int time_path(Ctx& c, const float* timesteps, int B, const float** tproj_out) {
  const mvd_config_t& cfg = c.e->cfg;
  const int C0 = cfg.block_out_channels[0], TD = 4 * C0;
  float* sinus = c.talloc<float>((size_t)B * C0);
  float* t1 = c.talloc<float>((size_t)B * TD);
  float* emb = c.talloc<float>((size_t)B * TD);
  bf16_t* act = c.talloc<bf16_t>((size_t)B * TD);
  float* tproj = c.aalloc<float>((size_t)B * c.e->temb_total);
  const bf16_t* w1 = c.WB("time.l1.w", (int64_t)TD * C0);
  const float* b1 = c.WF("time.l1.b", TD);
  const bf16_t* w2 = c.WB("time.l2.w", (int64_t)TD * TD);
  const float* b2 = c.WF("time.l2.b", TD);
  const bf16_t* wp = c.WB("temb_proj.w", (int64_t)c.e->temb_total * TD);
  const float* bp = c.WF("temb_proj.b", c.e->temb_total);
  *tproj_out = tproj;
  if (c.err) return c.err;
  if (c.dry) return 0;
  CHECK(mvd_launch_timestep_embedding(timesteps, B, C0, sinus, c.s));
  CHECK(mvd_launch_skinny_linear(sinus, C0, B, C0, w1, 1, b1, TD, 0, t1, TD, c.s));
  CHECK(mvd_launch_skinny_linear(t1, TD, B, TD, w2, 1, b2, TD, 1, emb, TD, c.s));
  // SiLU(emb) -> bf16 operand of the fused time_emb_proj GEMM (resnet: time_emb_proj(nonlinearity(temb)))
  CHECK(mvd_launch_silu_to_bf16(emb, (int64_t)B * TD, act, c.s));
  CHECK(c.linear(act, nullptr, TD, 0, B, wp, bp, c.e->temb_total, nullptr, 0, tproj, c.e->temb_total, false, true));
  return 0;
}

// camera encoder (fp32, Q9): cameras -> embedding [B][D]   (camera_encoder.py:160-196)
int camera_embed(Ctx& c, const float* src, const float* tgt, int cam_rows, const float* proj, int B, float* emb) {
  const mvd_config_t& cfg = c.e->cfg;
  const int D = cfg.cam_output_dim, Hd = cfg.cam_hidden_dim;
  const int nfreq = (D / 2) / 3, encd = 6 * nfreq;
  const bool simple = cfg.simple_cam_encoder != 0;
  float* rflat = c.talloc<float>((size_t)B * 9);
  float* enc = c.talloc<float>((size_t)B * encd);
  float* encp = c.talloc<float>((size_t)B * D);
  float* cat = c.talloc<float>((size_t)B * 2 * D);
  float* bufa = c.talloc<float>((size_t)B * (D > Hd ? D : Hd));
  float* bufb = c.talloc<float>((size_t)B * (D > Hd ? D : Hd));
  auto lin = [&](const std::string& k, const float* x, int ldx, int kin, int nout, float* y, int ldy) -> int {
    const float* w = c.WF("cam." + k + ".weight", (int64_t)nout * kin, 0);
    const float* b = c.WF("cam." + k + ".bias", nout, 0);
    if (c.err) return c.err; if (c.dry) return 0;
    return mvd_launch_skinny_linear(x, ldx, B, kin, w, 0, b, nout, 0, y, ldy, c.s);
  };
  auto lnorm = [&](const std::string& k, const float* x, int n, int silu, float* y) -> int {
    const float* g = c.WF("cam." + k + ".weight", n, 0);
    const float* b = c.WF("cam." + k + ".bias", n, 0);
    if (c.err) return c.err; if (c.dry) return 0;
    return mvd_launch_layernorm_f32(x, B, n, 1e-5f, g, b, silu, y, c.s);
  };
  if (!c.dry && !c.err) {
    CHECK(mvd_launch_camera_features(src, tgt, B, cam_rows, nfreq, 10.0f, rflat, enc, c.s));
    // Q1: projection by the per-call random matrix (no bias)
    CHECK(mvd_launch_skinny_linear(enc, encd, B, encd, proj, 0, nullptr, D, 0, encp, D, c.s));
  }
  const char* encs[2] = {"rotation_encoder", "translation_encoder"};
  for (int t = 0; t < 2; ++t) {   // -> cat[:, :D] and cat[:, D:]
    const std::string p = encs[t];
    const float* x = t == 0 ? rflat : encp;
    const int kin = t == 0 ? 9 : D;
    CHECK(lin(p + ".0", x, kin, kin, Hd, bufa, Hd));
    CHECK(lnorm(p + ".1", bufa, Hd, 1, bufb));
    if (simple) {
      CHECK(lin(p + ".3", bufb, Hd, Hd, D, cat + t * D, 2 * D));
    } else {
      CHECK(lin(p + ".3", bufb, Hd, Hd, Hd, bufa, Hd));
      CHECK(lnorm(p + ".4", bufa, Hd, 1, bufb));
      CHECK(lin(p + ".6", bufb, Hd, Hd, D, cat + t * D, 2 * D));
    }
  }
  CHECK(lin("final_projection.0", cat, 2 * D, 2 * D, D, bufa, D));
  CHECK(lnorm("final_projection.1", bufa, D, 1, bufb));
  CHECK(lin("final_projection.3", bufb, D, D, D, bufa, D));
  CHECK(lnorm("final_projection.4", bufa, D, 0, bufb));
  CHECK(lnorm("output_norm", bufb, D, 0, emb));
  return c.err;
}

// one modulator MLP: emb [B][D] -> processed FiLM scale/shift [B][dim]   (camera_encoder.py:215-222)
// (out_rows >= B: the processed scale/shift rows are replicated cyclically up to the sample batch)
int modulator(Ctx& c, const std::string& name, int dim, const float* emb, int B, float* sc, float* sh, int out_rows = 0) {
  const mvd_config_t& cfg = c.e->cfg;
  const int D = cfg.cam_output_dim;
  float* mh = c.talloc<float>((size_t)B * (D / 2));
  float* mh2 = c.talloc<float>((size_t)B * (D / 2));
  float* raw = c.talloc<float>((size_t)B * 2 * dim);
  const std::string k = "cam.modulators." + name;
  const float* w0 = c.WF(k + ".0.weight", (int64_t)(D / 2) * D, 0);
  const float* b0 = c.WF(k + ".0.bias", D / 2, 0);
  const float* g1 = c.WF(k + ".1.weight", D / 2, 0);
  const float* b1 = c.WF(k + ".1.bias", D / 2, 0);
  const float* w3 = c.WF(k + ".3.weight", (int64_t)2 * dim * (D / 2), 0);
  const float* b3 = c.WF(k + ".3.bias", 2 * dim, 0);
  if (c.err) return c.err;
  if (c.dry) return 0;
  CHECK(mvd_launch_skinny_linear(emb, D, B, D, w0, 0, b0, D / 2, 0, mh, D / 2, c.s));
  CHECK(mvd_launch_layernorm_f32(mh, B, D / 2, 1e-5f, g1, b1, 1, mh2, c.s));
  CHECK(mvd_launch_skinny_linear(mh2, D / 2, B, D / 2, w3, 0, b3, 2 * dim, 0, raw, 2 * dim, c.s));
  return mvd_launch_film_params(raw, B, dim, cfg.cam_modulation_strength, sc, sh, out_rows > B ? out_rows : B, c.s);
}

// modulators addressed by the hooks: name -> dim (mvd_unet.py:63-80); "mid" exists as a parameter but
// is never addressed (the hook asks for "mid_0", Q3)
std::vector<std::pair<std::string, int>> modulator_dims(const mvd_config_t& cfg, bool include_mid) {
  std::vector<std::pair<std::string, int>> mods;
  for (int i = 0; i < cfg.num_levels; ++i) mods.push_back({"down_" + std::to_string(i), cfg.block_out_channels[i]});
  for (int i = 0; i < cfg.num_levels; ++i) mods.push_back({"up_" + std::to_string(i), cfg.block_out_channels[cfg.num_levels - 1 - i]});
  if (include_mid) mods.push_back({"mid", cfg.block_out_channels[cfg.num_levels - 1]});
  mods.push_back({"output", 4});
  return mods;
}

// The camera batch Bc may be smaller than the sample batch B (classifier-free guidance doubles the latents but not the
// cameras, pipeline.py:141-152): the (Bc, C, 1, 1) scale/shift then broadcasts over the sample rows like torch does for
// Bc == 1; for 1 < Bc < B (Bc | B) row b takes camera b % Bc, i.e. the [uncond | cond] halves share their cameras.
int camera_path(Ctx& c, const mvd_forward_args_t& a, std::unordered_map<std::string, std::pair<float*, float*>>& ss) {
  const mvd_config_t& cfg = c.e->cfg;
  const int B = a.batch, D = cfg.cam_output_dim;
  const int Bc = a.cam_batch > 0 ? a.cam_batch : B;
  float* emb = c.aalloc<float>((size_t)Bc * D);
  c.e->cam_emb = emb; c.e->cam_batch = Bc;
  CHECK(camera_embed(c, a.source_camera, a.target_camera, a.cam_rows, a.fourier_proj, Bc, emb));
  // all hooked modulators at once when the concatenated slots are registered (packing.pack_camera): 4 launches instead of 4 each
  const auto mods = modulator_dims(cfg, false);
  if ((c.dry || c.has("cam.modcat.w0", 0)) && mods.size() <= 16) {
    const int nm = (int)mods.size(), Hh = D / 2;
    MvdSegTable seg; memset(&seg, 0, sizeof(seg));
    seg.n = nm;
    int tot = 0;
    for (int i = 0; i < nm; ++i) { tot += 2 * mods[i].second; seg.end[i] = tot; }
    float* ssbuf = c.aalloc<float>((size_t)B * tot);                  // per modulator: scale [B][dim] | shift [B][dim]
    const size_t mk = c.e->tmp.off;
    float* h1 = c.talloc<float>((size_t)Bc * nm * Hh);
    float* h2 = c.talloc<float>((size_t)Bc * nm * Hh);
    float* raw = c.talloc<float>((size_t)Bc * tot);
    const float* w0 = c.WF("cam.modcat.w0", (int64_t)nm * Hh * D, 0);
    const float* b0 = c.WF("cam.modcat.b0", (int64_t)nm * Hh, 0);
    const float* g1 = c.WF("cam.modcat.g1", (int64_t)nm * Hh, 0);
    const float* be1 = c.WF("cam.modcat.be1", (int64_t)nm * Hh, 0);
    const float* w3 = c.WF("cam.modcat.w3", (int64_t)tot * Hh, 0);
    const float* b3 = c.WF("cam.modcat.b3", tot, 0);
    if (!c.dry && !c.err) {
      CHECK(mvd_launch_skinny_linear(emb, D, Bc, D, w0, 0, b0, nm * Hh, 0, h1, nm * Hh, c.s));
      CHECK(mvd_launch_layernorm_f32(h1, Bc * nm, Hh, 1e-5f, g1, be1, 1, h2, c.s, nm));
      CHECK(mvd_launch_skinny_linear_grouped(h2, nm * Hh, Hh, Bc, Hh, w3, b3, tot, seg, raw, tot, c.s));
      CHECK(mvd_launch_film_params_grouped(raw, Bc, seg, cfg.cam_modulation_strength, ssbuf, B, c.s));
    }
    c.e->tmp.off = mk;
    int c0 = 0;
    for (int i = 0; i < nm; ++i) {
      float* sc = ssbuf + (size_t)B * 2 * c0;
      ss[mods[i].first] = {sc, sc + (size_t)B * mods[i].second};
      c0 += mods[i].second;
    }
    return c.err;
  }
  for (auto& m : modulator_dims(cfg, false)) {
    float* sc = c.aalloc<float>((size_t)B * m.second);
    float* sh = c.aalloc<float>((size_t)B * m.second);
    const size_t mk = c.e->tmp.off;
    CHECK(modulator(c, m.first, m.second, emb, Bc, sc, sh, B));
    c.e->tmp.off = mk;
    ss[m.first] = {sc, sh};
  }
  return c.err;
}

// Split-K arrival counters for one entry call: a block of the workspace, zeroed once (one memset node in front of the
// call's first kernel); Ctx::gemm hands every split-K launch its own slice.
constexpr int MVD_TILE_COUNTERS = 32768;
int setup_tile_counters(mvd_engine* e, unsigned int* block, hipStream_t s, bool dry) {
  e->cnt_base = block; e->cnt_used = 0; e->cnt_cap = MVD_TILE_COUNTERS;
  if (dry) return 0;
  hipError_t he = hipMemsetAsync(block, 0, MVD_TILE_COUNTERS * sizeof(unsigned int), s);
  if (he != hipSuccess) { mvd_set_error("forward: hipMemsetAsync(tile counters): %s", hipGetErrorString(he)); return -3; }
  return 0;
}

int forward_body(mvd_engine* e, const mvd_forward_args_t& a, hipStream_t s, bool dry) {
  const mvd_config_t& cfg = e->cfg;
  if (a.batch <= 0 || a.height <= 0 || a.width <= 0 || a.text_len <= 0) { mvd_set_error("forward: bad shape"); return -1; }
  const int div = 1 << (cfg.num_levels - 1);
  if (a.height % div || a.width % div) { mvd_set_error("forward: latent %dx%d must be divisible by %d", a.height, a.width, div); return -1; }
  // ref_only: mvd_engine_reference_encode -- the reference pass alone, normalisation deferred (global Q2 statistics)
  const bool ref_only = a.flags & MVD_REF_ONLY_INTERNAL;
  const bool use_cam = (a.flags & MVD_USE_CAMERA) && !ref_only, use_img = a.flags & MVD_USE_IMAGE, reuse = a.flags & MVD_REUSE_REF;
  if (ref_only && (!use_img || reuse)) { mvd_set_error("reference_encode: needs MVD_USE_IMAGE without MVD_REUSE_REF"); return -1; }
  if (!dry) {
    if (!ref_only && (!a.sample || !a.timesteps || !a.text || !a.out)) { mvd_set_error("forward: null sample/timesteps/text/out"); return -1; }
    if (use_cam && (!a.source_camera || !a.target_camera || !a.fourier_proj || (a.cam_rows != 3 && a.cam_rows != 4))) { mvd_set_error("forward: camera inputs missing"); return -1; }
    if (use_img && !reuse && (!a.source_latents || !a.encoder_text || a.ref_batch <= 0)) { mvd_set_error("forward: image-conditioning inputs missing"); return -1; }
    if (!e->ws_ptr) { mvd_set_error("forward: workspace not bound"); return -1; }
  }
  if (use_img && a.ref_batch <= 0) { mvd_set_error("forward: ref_batch must be > 0 with MVD_USE_IMAGE"); return -1; }
  if (use_cam && a.cam_batch > 0 && (a.cam_batch > a.batch || a.batch % a.cam_batch)) { mvd_set_error("forward: camera batch %d must divide the sample batch %d", a.cam_batch, a.batch); return -1; }
  // the input FiLM uses the 4-wide "output" modulator (mvd_unet.py:74-80, 256-258): any other in_channels is a broadcast
  // error in the reference and would read past the [B][4] scale/shift rows here -- reject before any launch
  if (!dry && use_cam && cfg.in_channels != 4) { mvd_set_error("forward: camera conditioning needs in_channels == 4 (the 'output' modulator is 4 wide), got %d", cfg.in_channels); return -1; }

  e->tmp.dry = e->act.dry = dry;
  e->tmp.off = e->tmp.high = 0;
  e->act.off = e->act.high = 0;
  Ctx c{e, s, 0, dry};
  CHECK(setup_tile_counters(e, (unsigned int*)e->act.alloc(MVD_TILE_COUNTERS * sizeof(unsigned int)), s, dry));
  const int B = a.batch, H = a.height, Wd = a.width, L = a.text_len, xd = cfg.cross_attention_dim;

  // ---- persistent reference cache layout
  if (use_img) {
    const bool keep = (a.flags & MVD_KEEP_FEATURES) || ref_only;
    if (reuse) {
      if (!e->rc_valid || e->rc_batch != a.ref_batch || e->rc_h != H || e->rc_w != Wd) { mvd_set_error("forward: MVD_REUSE_REF without a matching cached reference"); return -1; }
    } else {
      e->persist.dry = dry; e->persist.off = e->persist.high = 0;
      e->refkv.assign(e->feats.size(), nullptr);
      e->feat_keep.assign(e->feats.size(), nullptr);
      for (size_t i = 0; i < e->feats.size(); ++i) {
        const int lv = e->feats[i].level;
        const size_t hw = (size_t)(H >> lv) * (Wd >> lv);
        e->refkv[i] = (bf16_t*)e->persist.alloc((size_t)a.ref_batch * hw * 4 * e->feats[i].C * sizeof(bf16_t));
        if (keep) e->feat_keep[i] = (bf16_t*)e->persist.alloc((size_t)a.ref_batch * hw * e->feats[i].C * sizeof(bf16_t));
      }
      if (e->persist.overflow()) { mvd_set_error("forward: reference cache too small (%zu > %zu bytes)", e->persist.high, e->persist.cap); return -4; }
      e->rc_keep = keep; e->rc_batch = a.ref_batch; e->rc_h = H; e->rc_w = Wd;
      if (!dry) { e->rc_valid = false; e->rc_pending = false; }
    }
  }

  // ---- the encoder pass goes to the side stream (see mvd_engine::side): at batch 1 the two passes overlap almost completely
  // (cfg3 cold 9.8 -> 7.0 ms), at 32 pairs the second stream still fills the tails and the under-filled launches of the deep
  // levels (cfg4 65.0 -> 63.5 ms, same box).  `dual` (a function of the call's flags only) decides the workspace layout -- in
  // the sizing runs too; whether the side stream is really used also needs: no per-launch profiling.
  const bool dual = use_img && !reuse && !ref_only;
  e->dual_late = false;
  if (g_debug_flags & (3 << 19)) e->dual_late_from = (g_debug_flags >> 19) & 3;      // (A/B: debug-flag bits 19-20 = 1..3)
  // (Round 5: under hipGraph capture too.  The fork event is recorded on the capturing stream and waited for by the side stream,
  //  which thereby joins the capture; the per-feature events and the join event become edges of the graph, so a replayed
  //  forward keeps the two-branch schedule instead of serialising the passes.  Debug flag 65536 restores the one-stream capture.)
  e->dual_now = dual && !dry && !(e->graph_on && (g_debug_flags & 65536)) && (!e->prof || e->prof_overlap) && !(g_debug_flags & 16);
  // Where the side stream forks: in front of everything (the encoder pass needs nothing of the camera path).  Round 5's two-stream
  // kernel stats showed what that does to the ~30 latency-bound launches in front of the main pass (camera MLPs, time MLPs): the
  // encoder's persistent 256-workgroup kernels fill every CU's register file, so a small kernel of the other stream waits for
  // a whole big kernel to retire -- 145 us per time-MLP launch, 390 us for the grouped modulator layer, ~1.6 ms before the main
  // pass (the step's critical path) can start.  The alternative `early` (debug flag 4194304: those launches of BOTH passes first,
  // on an otherwise idle chip, the fork behind them) was measured and LOSES: cfg4 60.19 -> 60.96 ms, cfg3 6.46 -> 6.67 ms (three /
  // two same-box alternations) -- an idle chip for 0.75 ms costs more than the slowed-down front matter.  What helped instead is
  // making those launches cheap (misc.hip skinny_mfma_kernel).  `early` is a function of the call's flags only, so the sizing run
  // allocates in the same order.
  const bool early = dual && (g_debug_flags & 4194304);
  auto fork = [&]() -> int {
    if (!e->dual_now) return 0;
    CHECK(e->ensure_side_stream());
    if (hipEventRecord(e->fork_ev, s) != hipSuccess || hipStreamWaitEvent(e->side, e->fork_ev, 0) != hipSuccess) { mvd_set_error("forward: stream fork failed"); return -3; }
    return 0;
  };
  if (!early) CHECK(fork());

  // ---- camera path (fp32) -> embedding + FiLM scale/shift per modulator
  std::unordered_map<std::string, std::pair<float*, float*>> film_ss;
  // (running this path on a side stream concurrently with the reference pass was measured: +0.1 %, within noise --
  //  the persistent GEMMs leave no free CU resources for it -- so it stays on the caller's stream)
  if (use_cam) CHECK(camera_path(c, a, film_ss));
  const float* tproj_main = nullptr;
  const float* tproj_enc = nullptr;
  if (early) {
    c.set = 0;
    CHECK(time_path(c, a.timesteps, B, &tproj_main));
    c.set = e->share_encoder ? 0 : 1;
    float* tz = c.aalloc<float>(a.ref_batch);
    if (!dry) CHECK((int)hipMemsetAsync(tz, 0, a.ref_batch * sizeof(float), s));
    CHECK(time_path(c, tz, a.ref_batch, &tproj_enc));
    c.set = 0;
    CHECK(fork());
  }

  // ---- reference image encoder pass (frozen UNet at t = 0, plain attention) -> adapter K/V
  // Small batches: on the side stream, concurrently with the main pass (see mvd_engine::side).  Not under graph capture /
  // per-launch profiling (one stream each), not for the deferred-statistics form.
  if (use_img && !reuse) {
    // (dual: the camera path's kernels on the caller's stream may still be using their scoped temporaries when the side
    //  stream starts -- the encoder pass gets a region of its own behind them)
    if (dual) e->tmp.off = e->tmp.high;
    const size_t tm = e->tmp.off, am = e->act.off;
    if (e->dual_now) { c.s = e->side; c.nowait = true; }
    c.set = e->share_encoder ? 0 : 1;
    const int Br = a.ref_batch;
    const float* tproj = tproj_enc;
    if (!tproj) {
      float* tz = c.talloc<float>(Br);
      if (!dry) CHECK((int)hipMemsetAsync(tz, 0, Br * sizeof(float), c.s));
      CHECK(time_path(c, tz, Br, &tproj));
    }
    bf16_t* tx = c.aalloc<bf16_t>((size_t)Br * L * xd);
    Act xin = c.new_act(Br, H, Wd, 64, true);   // im2col rows for conv_in
    if (!dry && !c.err) {
      CHECK(mvd_launch_f32_to_bf16(a.encoder_text, (int64_t)Br * L * xd, tx, c.s));
      CHECK(mvd_launch_im2col_in(a.source_latents, Br, cfg.in_channels, H, Wd, nullptr, nullptr, 0, xin.p, c.s));
    }
    PassOpts po; po.capture = true; po.ref_batch = Br;
    po.defer_norm = ref_only; po.stats = e->ref_stats_out;
    UNetPass pass{c, cfg, po, Br, H, Wd, L, tx, tproj};
    CHECK(pass.run(xin, nullptr));
    if (!dry) { e->rc_valid = !ref_only; e->rc_pending = ref_only; }
    if (dual) {
      // the encoder pass may still be RUNNING while the main pass is issued: its activations stay, the main pass's scoped
      // temporaries start behind the encoder's high-water mark.  (Main-pass kernels may rendezvous again: the side stream's
      // never wait, so every workgroup of a main-pass kernel becomes resident eventually.)
      e->tmp.off = e->tmp.high;
      if (e->dual_now && hipEventRecord(e->join_ev, e->side) != hipSuccess) { mvd_set_error("forward: hipEventRecord failed"); return -3; }
      c.s = s; c.nowait = false;
    } else {
      e->tmp.off = tm; e->act.off = am;   // encoder activations are dead; reuse their memory
    }
  }
  if (ref_only) {
    if (c.err) return c.err;
    if (!dry && (e->tmp.high + e->act.high > (size_t)e->ws_bytes)) { mvd_set_error("reference_encode: workspace too small"); return -4; }
    return 0;
  }

  // ---- main pass
  c.set = 0;
  const float* tproj = tproj_main;
  if (!tproj) CHECK(time_path(c, a.timesteps, B, &tproj));
  bf16_t* tx = c.aalloc<bf16_t>((size_t)B * L * xd);
  Act xin = c.new_act(B, H, Wd, 64, true);   // im2col rows for conv_in
  if (!dry && !c.err) {
    CHECK(mvd_launch_f32_to_bf16(a.text, (int64_t)B * L * xd, tx, s));
    const float* sc = nullptr; const float* sh = nullptr;
    if (use_cam) { sc = film_ss["output"].first; sh = film_ss["output"].second; }   // mvd_unet.py:256-258 (input FiLM)
    CHECK(mvd_launch_im2col_in(a.sample, B, cfg.in_channels, H, Wd, sc, sh, 4, xin.p, s));
  }
  PassOpts po; po.adapter = use_img; po.film = use_cam; po.ref_batch = a.ref_batch; po.film_ss = &film_ss;
  UNetPass pass{c, cfg, po, B, H, Wd, L, tx, tproj};
  CHECK(pass.run(xin, a.out));
  if (e->dual_now) {   // join: everything the side stream did (kept features included) precedes whatever follows on s
    if (hipStreamWaitEvent(s, e->join_ev, 0) != hipSuccess) { mvd_set_error("forward: stream join failed"); return -3; }
    e->dual_now = false;
  }
  if (c.err) return c.err;
  if (!dry && (e->tmp.high + e->act.high > (size_t)e->ws_bytes)) { mvd_set_error("forward: workspace too small"); return -4; }
  return 0;
}

// Every exit of a forward that forked the side stream joins it: an error return between fork and join (a failed check in the
// camera path or a pass, a workspace that is too small) must not leave the encoder pass writing the reference cache and the
// activation arena while the caller frees or reuses them.  The success path has joined with an event already.
int forward_impl(mvd_engine* e, const mvd_forward_args_t& a, hipStream_t s, bool dry) {
  const int r = forward_body(e, a, s, dry);
  if (e->dual_now) {
    if (e->side) (void)hipStreamSynchronize(e->side);      // error path only: the cheap join is not worth an event here
    e->dual_now = false;
  }
  return r;
}

// Second half of the global-statistics path: normalise the kept raw features with the merged per-pixel (mean, k) and
// project them to the adapter K/V -- the same two launches per feature the encoder pass makes in the local form.
int reference_finish_impl(mvd_engine* e, const float* mean_k, hipStream_t s, bool dry) {
  e->tmp.dry = dry;
  e->tmp.off = e->tmp.high = 0;
  Ctx c{e, s, 0, dry};
  CHECK(setup_tile_counters(e, (unsigned int*)e->tmp.alloc(MVD_TILE_COUNTERS * sizeof(unsigned int)), s, dry));
  c.set = 0;                                         // the adapter's ref_kv weights live with the base set
  for (size_t i = 0; i < e->feats.size(); ++i) {
    const int lv = e->feats[i].level, C = e->feats[i].C;
    const int hw = (e->rc_h >> lv) * (e->rc_w >> lv), M = e->rc_batch * hw;
    const std::string& key = e->feats[i].key;
    const size_t mk = e->tmp.off;
    bf16_t* rn = c.talloc<bf16_t>((size_t)M * C);
    if (!dry && !c.err) CHECK(mvd_launch_refapply(e->feat_keep[i], e->rc_batch, hw, C, mean_k + 2 * e->feat_pixel_off((int)i), rn, s));
    CHECK(c.linear(rn, nullptr, C, 0, M, c.WB(key + ".ref_kv.w", (int64_t)4 * C * C, 0), nullptr, 4 * C, nullptr, 0, e->refkv[i], 4 * C));
    e->tmp.off = mk;
  }
  return c.err;
}

}  // namespace

// silu -> bf16 helper kernel (time embedding activation)
__global__ void silu_to_bf16_kernel(const float* __restrict__ x, long n, bf16_t* __restrict__ y) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = f2bf(silu_f(x[i]));
}
int mvd_launch_silu_to_bf16(const float* x, int64_t n, bf16_t* y, hipStream_t s) {
  int grid = (int)((n + 255) / 256); if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(silu_to_bf16_kernel, dim3(grid), dim3(256), 0, s, x, (long)n, y);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mvd_set_error("silu_to_bf16 launch: %s", hipGetErrorString(e)); return -3; }
  return 0;
}

// =========================================================================== C ABI
extern "C" {

int mvd_engine_create(const mvd_config_t* cfg, mvd_engine_t** out) {
  if (!cfg || !out) { mvd_set_error("engine_create: null argument"); return -1; }
  if (cfg->num_levels < 2 || cfg->num_levels > MVD_MAX_LEVELS || cfg->layers_per_block < 1) { mvd_set_error("engine_create: unsupported topology"); return -1; }
  for (int i = 0; i < cfg->num_levels; ++i) {
    const int c = cfg->block_out_channels[i];
    if (c % 64 || c % cfg->norm_num_groups || cfg->num_heads[i] * 64 != c) { mvd_set_error("engine_create: level %d: channels %d must be a multiple of 64/groups with head_dim 64 (heads %d)", i, c, cfg->num_heads[i]); return -1; }
  }
  if (cfg->cross_attention_dim % 64) { mvd_set_error("engine_create: cross_attention_dim must be a multiple of 64"); return -1; }
  if (cfg->in_channels > 7 || cfg->out_channels > 8) { mvd_set_error("engine_create: in_channels must be <= 7 (9*Cin <= 64) and out_channels <= 8"); return -1; }
  mvd_engine* e = new mvd_engine();
  e->cfg = *cfg;
  e->feats = enumerate_features(*cfg);
  int off = 0;
  for (auto& r : enumerate_resnets(*cfg)) { e->temb_off.push_back(off); off += r.cout; }
  e->temb_total = off;
  off = 0;
  for (auto& f : e->feats) { e->tkv_off.push_back(off); off += 2 * f.C; }
  e->tkv_total = off;
  *out = e;
  return 0;
}

int mvd_engine_destroy(mvd_engine_t* e) { delete e; return 0; }

int mvd_engine_set_weight(mvd_engine_t* e, int set, const char* slot, const void* ptr, int64_t numel, int dtype) {
  if (e) e->drop_graphs();
  if (!e || set < 0 || set > 1 || !slot || !ptr || numel <= 0 || dtype < 0 || dtype > 1) { mvd_set_error("set_weight: bad argument"); return -1; }
  if ((uintptr_t)ptr & 15) { mvd_set_error("set_weight: '%s' must be 16-byte aligned", slot); return -1; }
  e->w[set][slot] = Weight{ptr, numel, dtype};
  return 0;
}
int mvd_engine_clear_weights(mvd_engine_t* e, int set) {
  if (e) e->drop_graphs();
  if (!e || set < 0 || set > 1) { mvd_set_error("clear_weights: bad argument"); return -1; }
  e->w[set].clear();
  return 0;
}

static void fill_dry_args(mvd_forward_args_t& a, int batch, int h, int w, int L, int ref_batch) {
  memset(&a, 0, sizeof(a));
  a.batch = batch; a.height = h; a.width = w; a.text_len = L; a.ref_batch = ref_batch; a.cam_rows = 4;
  a.flags = MVD_USE_CAMERA | (ref_batch > 0 ? MVD_USE_IMAGE : 0);
}

int64_t mvd_engine_workspace_bytes(mvd_engine_t* e, int batch, int height, int width, int text_len, int ref_batch) {
  if (!e) { mvd_set_error("workspace_bytes: null engine"); return -1; }
  mvd_forward_args_t a; fill_dry_args(a, batch, height, width, text_len, ref_batch);
  const bool valid = e->rc_valid;
  std::vector<bf16_t*> kv = e->refkv, fk = e->feat_keep;
  const bool keep = e->rc_keep; const int rb = e->rc_batch, rh = e->rc_h, rw = e->rc_w;
  int r = forward_impl(e, a, nullptr, true);
  e->rc_valid = valid; e->refkv = kv; e->feat_keep = fk; e->rc_keep = keep; e->rc_batch = rb; e->rc_h = rh; e->rc_w = rw;
  if (r) return r;
  // act and tmp arenas share one buffer: [act | tmp]
  return (int64_t)(((e->act.high + 255) & ~size_t(255)) + e->tmp.high + 4096);
}

int64_t mvd_engine_refcache_bytes(mvd_engine_t* e, int ref_batch, int height, int width, int keep_features) {
  if (!e || ref_batch <= 0) { mvd_set_error("refcache_bytes: bad argument"); return -1; }
  size_t total = 0;
  for (auto& f : e->feats) {
    const size_t hw = (size_t)(height >> f.level) * (width >> f.level);
    total += (((size_t)ref_batch * hw * 4 * f.C * 2) + 255) & ~size_t(255);
    if (keep_features) total += (((size_t)ref_batch * hw * f.C * 2) + 255) & ~size_t(255);
  }
  return (int64_t)total + 4096;
}

int mvd_engine_bind_workspace(mvd_engine_t* e, void* ws, int64_t ws_bytes, void* refcache, int64_t refcache_bytes) {
  if (e) e->drop_graphs();
  if (!e || !ws || ws_bytes <= 0) { mvd_set_error("bind_workspace: bad argument"); return -1; }
  if (((uintptr_t)ws & 255) || (refcache && ((uintptr_t)refcache & 255))) { mvd_set_error("bind_workspace: buffers must be 256-byte aligned"); return -1; }
  e->ws_ptr = ws; e->ws_bytes = ws_bytes;
  e->rc_ptr = refcache; e->rc_bytes = refcache_bytes;
  e->persist.base = (char*)refcache; e->persist.cap = (size_t)(refcache ? refcache_bytes : 0);
  e->rc_valid = false; e->rc_pending = false;
  return 0;
}

int64_t mvd_engine_reference_pixels(mvd_engine_t* e, int height, int width) {
  if (!e || height <= 0 || width <= 0) { mvd_set_error("reference_pixels: bad argument"); return -1; }
  return e->ref_pixels(height, width);
}

int mvd_engine_reference_encode(mvd_engine_t* e, const mvd_forward_args_t* args, float* local_stats, void* stream) {
  if (!e || !args || !local_stats) { mvd_set_error("reference_encode: null argument"); return -1; }
  e->drop_graphs();
  mvd_forward_args_t a = *args;
  a.flags = (a.flags & ~MVD_REUSE_REF) | MVD_USE_IMAGE | MVD_KEEP_FEATURES | MVD_REF_ONLY_INTERNAL;
  if (a.batch <= 0) a.batch = a.ref_batch;
  int r = forward_impl(e, a, nullptr, true);
  e->rc_valid = false; e->rc_pending = false;
  if (r) return r;
  const size_t act_bytes = (e->act.high + 255) & ~size_t(255);
  if (!e->ws_ptr || act_bytes + e->tmp.high > (size_t)e->ws_bytes) {
    mvd_set_error("reference_encode: workspace too small: need %zu bytes, bound %lld", act_bytes + e->tmp.high, (long long)e->ws_bytes);
    return -4;
  }
  e->act.base = (char*)e->ws_ptr; e->act.cap = act_bytes;
  e->tmp.base = (char*)e->ws_ptr + act_bytes; e->tmp.cap = (size_t)e->ws_bytes - act_bytes;
  e->ref_stats_out = local_stats;
  r = forward_impl(e, a, (hipStream_t)stream, false);
  e->ref_stats_out = nullptr;
  return r;
}

int mvd_engine_reference_finish(mvd_engine_t* e, const float* mean_k, void* stream) {
  if (!e || !mean_k) { mvd_set_error("reference_finish: null argument"); return -1; }
  if (!e->rc_pending) { mvd_set_error("reference_finish: no pending reference (call mvd_engine_reference_encode first)"); return -1; }
  e->drop_graphs();
  int r = reference_finish_impl(e, mean_k, nullptr, true);
  if (r) return r;
  if (!e->ws_ptr || e->tmp.high > (size_t)e->ws_bytes) { mvd_set_error("reference_finish: workspace too small (%zu > %lld bytes)", e->tmp.high, (long long)e->ws_bytes); return -4; }
  e->tmp.base = (char*)e->ws_ptr; e->tmp.cap = (size_t)e->ws_bytes;
  r = reference_finish_impl(e, mean_k, (hipStream_t)stream, false);
  if (r) return r;
  e->rc_pending = false; e->rc_valid = true;
  return 0;
}

int mvd_unet_forward(mvd_engine_t* e, const mvd_forward_args_t* args, void* stream) {
  if (!e || !args) { mvd_set_error("forward: null argument"); return -1; }
  if (args->flags & MVD_REF_ONLY_INTERNAL) { mvd_set_error("forward: unknown flag bits 0x%x", args->flags); return -1; }
  // size the two arenas for this shape with a dry run (pure host arithmetic), then run for real
  const bool valid = e->rc_valid;
  std::vector<bf16_t*> kv = e->refkv, fk = e->feat_keep;
  const bool keep = e->rc_keep; const int rb = e->rc_batch, rh = e->rc_h, rw = e->rc_w;
  int r = forward_impl(e, *args, nullptr, true);
  e->rc_valid = valid; e->refkv = kv; e->feat_keep = fk; e->rc_keep = keep; e->rc_batch = rb; e->rc_h = rh; e->rc_w = rw;
  if (r) return r;
  const size_t act_bytes = (e->act.high + 255) & ~size_t(255);
  if (!e->ws_ptr || act_bytes + e->tmp.high > (size_t)e->ws_bytes) {
    mvd_set_error("forward: workspace too small: need %zu bytes, bound %lld", act_bytes + e->tmp.high, (long long)e->ws_bytes);
    return -4;
  }
  e->act.base = (char*)e->ws_ptr; e->act.cap = act_bytes;
  e->tmp.base = (char*)e->ws_ptr + act_bytes; e->tmp.cap = (size_t)e->ws_bytes - act_bytes;
  hipStream_t st = (hipStream_t)stream;
  if (!e->graph_on || e->prof || !st) return forward_impl(e, *args, st, false);
  // ---- hipGraph replay.  A forward is a pure function of the argument block (pointers, shapes, flags), the bound buffers
  // and the reference-cache state it starts from; the host-side state it leaves behind is the same every time.  The first
  // call with a given key runs as usual (kernels' first-use set-up must not happen inside a capture), the second is
  // captured and instantiated, later ones are one hipGraphLaunch.
  std::string key((const char*)args, sizeof(*args));
  const int st8[8] = {e->rc_valid, e->rc_batch, e->rc_h, e->rc_w, e->rc_keep, e->share_encoder, 0, 0};
  key.append((const char*)st8, sizeof(st8));
  key.append((const char*)&e->ws_ptr, sizeof(void*));
  key.append((const char*)&e->rc_ptr, sizeof(void*));
  for (auto& ge : e->graphs)
    if (ge.key == key) {
      hipError_t he = hipGraphLaunch(ge.x, st);
      if (he != hipSuccess) { mvd_set_error("forward: hipGraphLaunch: %s", hipGetErrorString(he)); return -7; }
      e->set_host_state(ge.after);        // (same key => the captured run started from this state and ended in that one)
      return 0;
    }
  bool seen = false;
  for (auto& k : e->graph_seen) seen |= k == key;
  if (!seen) { e->graph_seen.push_back(key); return forward_impl(e, *args, st, false); }
  if (e->graphs.size() >= 8) e->drop_graphs();
  hipError_t he = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  if (he != hipSuccess) { mvd_set_error("forward: hipStreamBeginCapture: %s", hipGetErrorString(he)); return -7; }
  r = forward_impl(e, *args, st, false);
  hipGraph_t g = nullptr;
  he = hipStreamEndCapture(st, &g);
  if (r) { if (g) (void)hipGraphDestroy(g); return r; }
  if (he != hipSuccess || !g) { mvd_set_error("forward: hipStreamEndCapture: %s", hipGetErrorString(he)); return -7; }
  hipGraphExec_t x = nullptr;
  he = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
  if (he != hipSuccess) { (void)hipGraphDestroy(g); mvd_set_error("forward: hipGraphInstantiate: %s", hipGetErrorString(he)); return -7; }
  e->graphs.push_back({key, g, x, e->host_state()});
  he = hipGraphLaunch(x, st);
  if (he != hipSuccess) { mvd_set_error("forward: hipGraphLaunch: %s", hipGetErrorString(he)); return -7; }
  return 0;
}

int mvd_engine_set_graph(mvd_engine_t* e, int enable) {
  if (!e) { mvd_set_error("set_graph: null engine"); return -1; }
  e->graph_on = enable != 0;
  if (!e->graph_on) e->drop_graphs();
  return 0;
}

int mvd_engine_set_profiling(mvd_engine_t* e, int enable) {
  if (!e) { mvd_set_error("set_profiling: null engine"); return -1; }
  // 1: one stream, launches back to back (serial kernel times); 2: the forward's own schedule (the encoder pass on the side
  // stream): durations of launches that share the chip with the other pass's -- what rocprofv3 sees in the timed region
  e->prof = enable != 0; e->prof_overlap = enable == 2; e->prof_recs.clear(); e->ev_used = 0;
  return 0;
}

// Sums the recorded launches per kernel class (after synchronising the last event) and resets the records.
// Arrays hold `cap` entries; returns the number of classes written (<0 on error).
int mvd_engine_profile_summary(mvd_engine_t* e, int cap, int* cls, int* launches, double* ms, double* flops, double* bytes) {
  if (!e || cap <= 0 || !cls || !launches || !ms || !flops || !bytes) { mvd_set_error("profile_summary: bad argument"); return -1; }
  int n = 0;
  for (auto& r : e->prof_recs) {
    if (hipEventSynchronize(r.e1) != hipSuccess) { mvd_set_error("profile_summary: event sync failed"); return -2; }
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) { mvd_set_error("profile_summary: elapsed time failed"); return -2; }
    int j = 0;
    while (j < n && cls[j] != r.cls) ++j;
    if (j == n) { if (n == cap) continue; cls[n] = r.cls; launches[n] = 0; ms[n] = 0; flops[n] = 0; bytes[n] = 0; ++n; }
    launches[j] += 1; ms[j] += t; flops[j] += r.flops; bytes[j] += r.bytes;
  }
  e->prof_recs.clear(); e->ev_used = 0;
  return n;
}

int mvd_engine_profile_shapes(mvd_engine_t* e, char* buf, int cap) {
  if (!e || !buf || cap <= 0) { mvd_set_error("profile_shapes: bad argument"); return -1; }
  struct Acc { int cls, M, N, K, tag, n; double ms, fl; };
  std::vector<Acc> v;
  for (auto& r : e->prof_recs) {
    if (hipEventSynchronize(r.e1) != hipSuccess) { mvd_set_error("profile_shapes: event sync failed"); return -2; }
    float t = 0.f; (void)hipEventElapsedTime(&t, r.e0, r.e1);
    size_t j = 0;
    while (j < v.size() && !(v[j].cls == r.cls && v[j].M == r.M && v[j].N == r.N && v[j].K == r.K && v[j].tag == r.tag)) ++j;
    if (j == v.size()) v.push_back({r.cls, r.M, r.N, r.K, r.tag, 0, 0.0, 0.0});
    v[j].n++; v[j].ms += t; v[j].fl += r.flops;
  }
  int off = 0;
  for (auto& x : v) {
    const int w = snprintf(buf + off, (size_t)(cap - off), "cls=%d M=%d N=%d K=%d tag=%d launches=%d ms=%.3f tflops=%.0f\n", x.cls, x.M, x.N, x.K,
                           x.tag, x.n, x.ms, x.ms > 0 ? x.fl / x.ms / 1e9 : 0.0);
    if (w < 0 || off + w >= cap) break;
    off += w;
  }
  return off;
}

int mvd_engine_share_encoder_weights(mvd_engine_t* e, int enable) {
  if (e) e->drop_graphs();
  if (!e) { mvd_set_error("share_encoder_weights: null engine"); return -1; }
  e->share_encoder = enable != 0;
  return 0;
}

int mvd_engine_num_features(mvd_engine_t* e) { return e ? (int)e->feats.size() : -1; }
int mvd_engine_feature_shape(mvd_engine_t* e, int idx, int* channels, int* height, int* width) {
  if (!e || idx < 0 || idx >= (int)e->feats.size() || !e->rc_h) { mvd_set_error("feature_shape: bad index or no reference pass yet"); return -1; }
  *channels = e->feats[idx].C; *height = e->rc_h >> e->feats[idx].level; *width = e->rc_w >> e->feats[idx].level;
  return 0;
}
int mvd_engine_get_feature(mvd_engine_t* e, int idx, float* out_nchw, void* stream) {
  if (!e || idx < 0 || idx >= (int)e->feats.size() || !out_nchw) { mvd_set_error("get_feature: bad argument"); return -1; }
  if (!e->rc_valid || !e->rc_keep || !e->feat_keep[idx]) { mvd_set_error("get_feature: no kept features (run forward with MVD_KEEP_FEATURES)"); return -1; }
  const int lv = e->feats[idx].level;
  return mvd_launch_nhwc_to_nchw_f32(e->feat_keep[idx], e->rc_batch, (e->rc_h >> lv) * (e->rc_w >> lv), e->feats[idx].C, out_nchw, (hipStream_t)stream);
}
// Standalone CameraEncoder.encode_cameras (camera_encoder.py:160-176) through the engine's kernels.
int mvd_engine_encode_cameras(mvd_engine_t* e, const float* source_camera, const float* target_camera, int cam_rows,
                              int batch, const float* fourier_proj, float* out_emb, void* stream) {
  if (!e || !source_camera || !target_camera || !fourier_proj || !out_emb || batch <= 0 || (cam_rows != 3 && cam_rows != 4)) { mvd_set_error("encode_cameras: bad argument"); return -1; }
  if (!e->ws_ptr) { mvd_set_error("encode_cameras: workspace not bound"); return -1; }
  {   // size the scratch with a dry pass: nothing is launched into a workspace that is too small
    e->tmp.dry = true; e->tmp.off = e->tmp.high = 0;
    Ctx d{e, nullptr, 0, true};
    if (int r = camera_embed(d, source_camera, target_camera, cam_rows, fourier_proj, batch, out_emb)) return r;
    if (e->tmp.high > (size_t)e->ws_bytes) { mvd_set_error("encode_cameras: workspace too small (%zu > %lld bytes)", e->tmp.high, (long long)e->ws_bytes); return -4; }
  }
  e->tmp.dry = false; e->tmp.base = (char*)e->ws_ptr; e->tmp.cap = (size_t)e->ws_bytes; e->tmp.off = e->tmp.high = 0;
  Ctx c{e, (hipStream_t)stream, 0, false};
  return camera_embed(c, source_camera, target_camera, cam_rows, fourier_proj, batch, out_emb);
}

// Standalone CameraEncoder.apply_modulation_to_tensor (camera_encoder.py:207-255) on an NCHW fp32 tensor.
// Unknown modulator names are the identity (returns 1 and leaves `out` untouched), like the reference.
int mvd_engine_apply_modulation(mvd_engine_t* e, const char* name, const float* emb, int batch, const float* x_nchw,
                                int channels, int hw, float* out_nchw, void* stream) {
  if (!e || !name || !emb || !x_nchw || !out_nchw || batch <= 0 || channels <= 0 || hw <= 0) { mvd_set_error("apply_modulation: bad argument"); return -1; }
  if (!e->ws_ptr) { mvd_set_error("apply_modulation: workspace not bound"); return -1; }
  int dim = -1;
  for (auto& m : modulator_dims(e->cfg, true)) if (m.first == name) dim = m.second;
  if (dim < 0) return 1;
  if (dim != channels) { mvd_set_error("apply_modulation: modulator '%s' has %d channels, tensor has %d", name, dim, channels); return -1; }
  {   // dry sizing pass first (see encode_cameras)
    e->tmp.dry = true; e->tmp.off = e->tmp.high = 0;
    Ctx d{e, nullptr, 0, true};
    float* sc0 = d.talloc<float>((size_t)batch * dim);
    float* sh0 = d.talloc<float>((size_t)batch * dim);
    if (int r = modulator(d, name, dim, emb, batch, sc0, sh0)) return r;
    if (e->tmp.high > (size_t)e->ws_bytes) { mvd_set_error("apply_modulation: workspace too small (%zu > %lld bytes)", e->tmp.high, (long long)e->ws_bytes); return -4; }
  }
  e->tmp.dry = false; e->tmp.base = (char*)e->ws_ptr; e->tmp.cap = (size_t)e->ws_bytes; e->tmp.off = e->tmp.high = 0;
  Ctx c{e, (hipStream_t)stream, 0, false};
  float* sc = c.talloc<float>((size_t)batch * dim);
  float* sh = c.talloc<float>((size_t)batch * dim);
  if (int r = modulator(c, name, dim, emb, batch, sc, sh)) return r;
  return mvd_launch_film_nchw_f32(x_nchw, batch, channels, hw, sc, sh, out_nchw, (hipStream_t)stream);
}

int mvd_engine_get_camera_embedding(mvd_engine_t* e, float* out, void* stream) {
  if (!e || !out || !e->cam_emb) { mvd_set_error("get_camera_embedding: no camera pass yet"); return -1; }
  return (int)hipMemcpyAsync(out, e->cam_emb, (size_t)e->cam_batch * e->cfg.cam_output_dim * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream);
}

// ------------------------------------------------------------------ operator-level entry points
// small-M kernels with split-K (force_cfg >= 100): the tile counters live behind the partials in the caller's split-K
// workspace (MVD_OP_SPLITK_COUNTERS extra 4-byte words, zeroed here per call); the kernel combines the slices itself
static int op_sm_splitk(MvdGemmArgs& g, hipStream_t s, int force_cfg) {
  g.tile_cnt = reinterpret_cast<unsigned int*>(g.part + (size_t)g.splitk * g.M * g.N);
  hipError_t e = hipMemsetAsync(g.tile_cnt, 0, MVD_OP_SPLITK_COUNTERS * sizeof(unsigned int), s);
  if (e != hipSuccess) { mvd_set_error("op split-K: hipMemsetAsync: %s", hipGetErrorString(e)); return -3; }
  if ((long)((g.M + 63) / 64) * (g.N / 64) > MVD_OP_SPLITK_COUNTERS) { mvd_set_error("op split-K: more than %d output tiles", MVD_OP_SPLITK_COUNTERS); return -1; }
  return mvd_launch_gemm(g, s, force_cfg);
}
int mvd_op_linear(const void* a, const void* a2, int k1, int k2, const void* w, const float* bias, const float* rowvec,
                  int ld_rowvec, int rows_per_batch, const void* res, float alpha, int geglu, void* out, int out_f32, int m,
                  int n, int force_cfg, int splitk, float* splitk_ws, void* stream) {
  MvdGemmArgs g; memset(&g, 0, sizeof(g));
  g.seg[0].p0 = (const bf16_t*)a; g.seg[0].p1 = (const bf16_t*)a2; g.seg[0].c0 = k1; g.seg[0].c1 = k2;
  g.seg[0].mode = MVD_A_DENSE; g.seg[0].ksize = k1 + k2; g.nseg = 1;
  g.W = (const bf16_t*)w; g.M = m; g.N = n; g.Ktot = k1 + k2; g.ldw = k1 + k2;
  g.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : m; g.outH = 1; g.outW = g.rows_per_batch;
  g.bias = bias; g.rowvec = rowvec; g.ld_rowvec = ld_rowvec; g.res = (const bf16_t*)res;
  const int on = geglu ? n / 2 : n;
  g.ldres = on; g.alpha = alpha; g.geglu = geglu; g.out = out; g.ldo = on; g.out_f32 = out_f32;
  g.part = splitk_ws;      // (also the stamp buffer of probe builds)
  if (splitk > 1) {
    g.splitk = splitk; g.part = splitk_ws;
    if (force_cfg >= 100) return op_sm_splitk(g, (hipStream_t)stream, force_cfg);
    if (int r = mvd_launch_gemm(g, (hipStream_t)stream, force_cfg)) return r;
    return mvd_launch_splitk_reduce(g, (hipStream_t)stream);
  }
  g.splitk = 1;            // the caller DECIDED not to split (0 would mean "undecided": the tile heuristic may then assume a deep split)
  return mvd_launch_gemm(g, (hipStream_t)stream, force_cfg);
}

int mvd_op_ln_linear(const void* x, int k, const void* w_folded, const float* c1, const float* c2, float eps, int geglu,
                     void* out, int m, int n, void* stream) {
  MvdGemmArgs g; memset(&g, 0, sizeof(g));
  g.seg[0].p0 = (const bf16_t*)x; g.seg[0].c0 = k; g.seg[0].mode = MVD_A_DENSE; g.seg[0].ksize = k; g.nseg = 1;
  g.W = (const bf16_t*)w_folded; g.M = m; g.N = n; g.Ktot = k; g.ldw = k; g.rows_per_batch = m; g.outH = 1; g.outW = m;
  g.bias = c2; g.ln_c1 = c1; g.ln_eps = eps; g.alpha = 1.f; g.geglu = geglu; g.out = out; g.ldo = geglu ? n / 2 : n; g.ldres = g.ldo;
  if (!c1 || !c2) { mvd_set_error("mvd_op_ln_linear: c1 and c2 are required"); return -1; }
  if (!mvd_gemm_ln_fold_ok(g)) {
    int tile = 0, ns = 0, S = 1;           // small problems (batch 1): the fold of the small-M kernels
    if (mvd_gemm_sm_plan(g, &tile, &ns, &S) && S == 1) return mvd_launch_gemm_sm(g, (hipStream_t)stream, tile, ns);
    mvd_set_error("mvd_op_ln_linear: M=%d N=%d K=%d geglu=%d is not a shape of the fused LayerNorm GEMMs", m, n, k, geglu);
    return -1;
  }
  return mvd_launch_gemm(g, (hipStream_t)stream);
}

int mvd_op_conv3x3(const void* x, int batch, int in_h, int in_w, int cin, int stride, int upsample, int asym_pad, const void* w,
                   const float* bias, const float* rowvec, int ld_rowvec, const void* res, const void* sc, const void* sc2,
                   int sc_c1, int sc_c2, void* out, int cout, int force_cfg, int splitk, float* splitk_ws, void* stream) {
  MvdGemmArgs g; memset(&g, 0, sizeof(g));
  const int oh = upsample ? in_h * 2 : (stride == 2 ? (in_h + 1) / 2 : in_h);
  const int ow = upsample ? in_w * 2 : (stride == 2 ? (in_w + 1) / 2 : in_w);
  g.seg[0].p0 = (const bf16_t*)x; g.seg[0].c0 = cin; g.seg[0].mode = MVD_A_CONV3; g.seg[0].ksize = 9 * cin;
  g.seg[0].inH = in_h; g.seg[0].inW = in_w; g.seg[0].stride = stride; g.seg[0].ups = upsample; g.seg[0].asym = asym_pad;
  g.nseg = 1; g.Ktot = 9 * cin;
  if (sc) {
    g.seg[1].p0 = (const bf16_t*)sc; g.seg[1].p1 = (const bf16_t*)sc2; g.seg[1].c0 = sc_c1; g.seg[1].c1 = sc_c2;
    g.seg[1].mode = MVD_A_DENSE; g.seg[1].ksize = sc_c1 + sc_c2; g.nseg = 2; g.Ktot += sc_c1 + sc_c2;
  }
  g.W = (const bf16_t*)w; g.ldw = g.Ktot; g.M = batch * oh * ow; g.N = cout; g.rows_per_batch = oh * ow; g.outH = oh; g.outW = ow;
  g.bias = bias; g.rowvec = rowvec; g.ld_rowvec = ld_rowvec; g.res = (const bf16_t*)res; g.ldres = cout; g.alpha = 1.f;
  g.out = out; g.ldo = cout;
  g.part = splitk_ws;      // (also the stamp buffer of probe builds)
  if (splitk > 1) {
    g.splitk = splitk; g.part = splitk_ws;
    if (force_cfg >= 100) return op_sm_splitk(g, (hipStream_t)stream, force_cfg);
    if (int r = mvd_launch_gemm(g, (hipStream_t)stream, force_cfg)) return r;
    return mvd_launch_splitk_reduce(g, (hipStream_t)stream);
  }
  g.splitk = 1;            // the caller DECIDED not to split (0 would mean "undecided": the tile heuristic may then assume a deep split)
  return mvd_launch_gemm(g, (hipStream_t)stream, force_cfg);
}

int mvd_op_attention(const void* q, const void* k, const void* v, void* o, int batch, int heads, int nq, int nk, int ldq,
                     int ldk, int ldv, int ldo, float scale, void* stream) {
  MvdAttnArgs a; memset(&a, 0, sizeof(a));
  a.nprob = 1; a.batch = batch; a.heads = heads; a.scale = scale;
  a.prescaled = scale == 0.f;   // q already carries softmax_scale * log2(e)
  a.p[0] = {(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, ldq, ldk, ldv, ldo,
            (int64_t)nq * ldq, (int64_t)nk * ldk, (int64_t)nk * ldv, (int64_t)nq * ldo, nq, nk};
  return mvd_launch_attention(a, (hipStream_t)stream);
}

// split-KV form of mvd_op_attention (prescaled queries only: scale == 0): ws = mvd_op_attention_split_ws_bytes(...) bytes
// (partials + the arrival counters, which the call zeroes)
int64_t mvd_op_attention_split_ws_bytes(int batch, int heads, int nq, int nsplit) {
  MvdAttnArgs a; memset(&a, 0, sizeof(a));
  a.nprob = 1; a.batch = batch; a.heads = heads; a.p[0].nq = nq;
  return (int64_t)mvd_attention_split_ws_bytes(a, nsplit) + (int64_t)mvd_attention_split_counters(a) * 4 + 256;
}
int mvd_op_attention_split(const void* q, const void* k, const void* v, void* o, int batch, int heads, int nq, int nk, int ldq,
                           int ldk, int ldv, int ldo, int nsplit, void* ws, void* stream) {
  MvdAttnArgs a; memset(&a, 0, sizeof(a));
  a.nprob = 1; a.batch = batch; a.heads = heads; a.scale = 0.f; a.prescaled = 1;
  a.p[0] = {(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, ldq, ldk, ldv, ldo,
            (int64_t)nq * ldq, (int64_t)nk * ldk, (int64_t)nk * ldv, (int64_t)nq * ldo, nq, nk};
  if (nsplit > 1) {
    if (!ws) { mvd_set_error("op_attention_split: null workspace"); return -1; }
    const size_t pb = (mvd_attention_split_ws_bytes(a, nsplit) + 255) & ~size_t(255);
    a.nsplit = nsplit; a.split_ws = ws; a.split_cnt = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(ws) + pb);
    if (hipMemsetAsync(a.split_cnt, 0, (size_t)mvd_attention_split_counters(a) * 4, (hipStream_t)stream) != hipSuccess) { mvd_set_error("op_attention_split: hipMemsetAsync failed"); return -3; }
  }
  return mvd_launch_attention(a, (hipStream_t)stream);
}

int mvd_op_groupnorm(const void* x0, const void* x1, int c0, int c1, int batch, int hw, int groups, float eps,
                     const float* gamma, const float* beta, int silu, void* y, float* ws, void* stream) {
  return mvd_launch_groupnorm((const bf16_t*)x0, (const bf16_t*)x1, c0, c1, batch, hw, groups, eps, gamma, beta, silu, (bf16_t*)y, ws, (hipStream_t)stream);
}
int mvd_op_layernorm(const void* x, int rows, int c, float eps, const float* gamma, const float* beta, void* y, void* stream) {
  return mvd_launch_layernorm((const bf16_t*)x, rows, c, eps, gamma, beta, (bf16_t*)y, (hipStream_t)stream);
}
int mvd_op_refnorm(const void* x, int batch, int hw, int c, void* y, void* stream) {
  return mvd_launch_refnorm((const bf16_t*)x, batch, hw, c, (bf16_t*)y, (hipStream_t)stream);
}
int mvd_op_film(const void* x, int batch, int hw, int c, const float* scale, const float* shift, void* y, void* stream) {
  return mvd_launch_film((const bf16_t*)x, batch, hw, c, scale, shift, c, (bf16_t*)y, (hipStream_t)stream);
}
int mvd_op_conv_in(const void* x, int batch, int h, int w, int cin, const float* wt, const float* bias, int cout, void* y, void* stream) {
  return mvd_launch_conv_in((const bf16_t*)x, batch, h, w, cin, wt, bias, cout, (bf16_t*)y, (hipStream_t)stream);
}
int mvd_op_conv_out(const void* x, int batch, int h, int w, int c, const void* wt, const float* bias, int cout, float* y, void* stream) {
  return mvd_launch_conv_out((const bf16_t*)x, batch, h, w, c, (const bf16_t*)wt, bias, cout, y, (hipStream_t)stream);
}
int mvd_op_nchw_to_nhwc(const float* x, int batch, int c, int hw, const float* scale, const float* shift, void* y, void* stream) {
  return mvd_launch_nchw_to_nhwc(x, batch, c, hw, scale, shift, c, (bf16_t*)y, (hipStream_t)stream);
}
int mvd_op_nhwc_to_nchw(const void* x, int batch, int hw, int c, float* y, void* stream) {
  return mvd_launch_nhwc_to_nchw_f32((const bf16_t*)x, batch, hw, c, y, (hipStream_t)stream);
}
int mvd_op_f32_to_bf16(const float* x, int64_t n, void* y, void* stream) {
  return mvd_launch_f32_to_bf16(x, n, (bf16_t*)y, (hipStream_t)stream);
}
// the fp32 linear layer of the camera / time MLPs: y[b][o] = sum_k act(x[b][k]) W[o][k] + bias[o]  (W fp32, or bf16 with wbf16 = 1;
// act_in = 1: SiLU on the inputs)
int mvd_op_skinny_linear(const float* x, int ldx, int batch, int k, const void* w, int wbf16, const float* bias, int n, int act_in,
                         float* y, int ldy, void* stream) {
  return mvd_launch_skinny_linear(x, ldx, batch, k, w, wbf16, bias, n, act_in, y, ldy, (hipStream_t)stream);
}

}  // extern "C"
