// lcr_api.hip — C ABI (include/lcr.h) over the HIP kernels: context, batch binding, stage drivers
// and the small sequential host epilogues (dense-cluster sweep, candidate.rs:465-526).
// There is NO CPU fallback: every stage launches HIP kernels and fails with LCR_E_DEVICE otherwise.
#include <atomic>
#include <thread>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>

#include <chrono>
#include "lcr_dev.h"
#include "lcr_phase_host.h"

struct lcr_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;

  // bound batch
  bool loaded = false;
  BatchView bv{};
  int64_t n_cols = 0, n_bases = 0, n_cigar = 0;
  int32_t n_tiles = 0;
  std::vector<int64_t> h_start0, h_col_off;
  std::vector<int32_t> h_len, h_read_begin, h_region_first_tile;
  DevBuf in_[16];  // device copies of host inputs (LCR_MEM_HOST)
  // asynchronous input path (lcr_load_batch_async / lcr_bind_batch): two staging slots, filled on an upload stream
  struct UploadSlot { DevBuf buf[16]; hipEvent_t ev = nullptr; bool filled = false; lcr_reads rd{}; lcr_regions rg{}; } up[2];
  hipStream_t up_stream = nullptr;
  hipEvent_t ev_dl = nullptr, ev_cand_dl = nullptr;   // lcr_candidates: the candidate records' download on the phase stage's second queue
  bool cand_dl_other = false;
  hipStream_t fill_stream = nullptr;           // zero fill of the count planes beside K0 (lcr_pileup)
  hipEvent_t ev_fill0 = nullptr, ev_fill1 = nullptr;
  int bound_slot = -1;
  bool bound_host = false;   // the bound batch was copied into in_[] (LCR_MEM_HOST)
  DevBuf scan_tmp, read_region, read_bin, read_rend, tile_region, tile_col0, first_tile, k0_tile_fill, k0_items, tile_nbase, tile_order;
  DevBuf desc_tile, desc_val, chunks, chunk_off;   // K0's chunk descriptors, the same sorted by tile, their per-tile offsets
  DevBuf blk_first_read, read_scan, cig_compact, cig_off_new, cig_new_off32;   // K0 op blocks (k0_ops.hip)
  uint64_t cig0 = 0;      // index of the batch's first op in bv.cigar
  uint32_t n_ops = 0;     // ops of the batch (one flat op space)
  int64_t n_items = 0;

  // K1
  bool have_planes = false;
  DevBuf planes;
  DevParams dp{};
  float sor_thr = -1.f;
  HostBuf h_planes;
  HostBuf h_nnz;              // pinned: first entry of every region of the fragment matrix, [ng] = entry count (lcr_fragments -> frag_settle)
  DevBuf region_e_off, frag_tmp_col, frag_tmp_val;
  hipEvent_t ev_nnz = nullptr, ev_cand = nullptr, ev_ctl = nullptr, ev_sv = nullptr;
  int32_t sv_cap_guess = 0;   // lcr_candidates: survivors the buffers are sized for before their number is known (the last call's + a quarter; 0: wait first)
  bool nnz_pending = false, cand_pending = false;
  HostBuf h_order;      // pinned: k0_pack raises it when a region's reads are not sorted by position
  static constexpr int UP_LANES = 4;   // staging lanes of pageable host uploads (upload_bytes): two page-locked 8 MB buffers + events each
  HostBuf h_up[2 * UP_LANES]; hipEvent_t ev_up[2 * UP_LANES] = {}; bool up_busy[2 * UP_LANES] = {};
  HostBuf h_stage[4];   // pinned staging of lcr_candidates / lcr_fragments: survivor offsets, candidate records, keep flags, region rows

  // K2
  bool have_cand = false;
  DevBuf flags, tile_count, tile_off, total, survivors, sv_region_off, hist, cand_tmp, keep;
  DevBuf hit_cnt, hit_list, ovf_list;   // k2_hist's (read, survivor) hits for K3; the overflow counter sits behind the histograms
  bool hits_valid = false; int32_t hits_n_sv = 0;
  int dbg_bg_tiles = 0;     // lcr_debug_set("bg_tiles"): > 0 = the record-free tiles' stores by this many workgroups on a second queue beside the tally
  int dbg_prefill = 0;      // lcr_debug_set("plane_prefill"): 1 = the count planes are zeroed on a second queue while K0 runs -- measured: 0.69 instead of 0.63 ms for the stage; 0: k1_empty_tiles writes the record-free tiles
  int dbg_hist_tiles = 0;   // lcr_debug_set("hist_tiles"): 0 = by survivor density, 1 = the tile form whenever it applies, -1 = never
  int dbg_zf_fused = 0;     // lcr_debug_set("zonefix_fused", 1) (measurement switch): HiFi presets -- the poly-A pass and the record-free tiles' stores in one launch; measured slower (HISTORY.md Appendix C)
  int dbg_zf_overlap = 0;   // lcr_debug_set("zonefix_overlap", 1) (measurement switch): HiFi presets -- the record-free tiles' stores on a second queue beside the poly-A pass; measured slower with the asynchronous phase stage (HISTORY.md Appendix C)
  int dbg_spec_compact = 1; // lcr_debug_set("spec_compact"): 0 = lcr_candidates waits for the survivors' number before it queues their compaction
  int dbg_fuse_filter = 1;  // lcr_debug_set("fuse_filter"): 0 = pass 1 of the candidate filters always by k2_filter (its own pass over the planes)
  bool flt_fused = false;   // the last lcr_pileup left k2_filter's flags and per-tile counts (ONT presets: no poly-A pass behind the tally)
  DevParams flt_dp{};       // ... computed with these parameters
  int dbg_k3_hits = 1;      // lcr_debug_set("k3_hits"): 0 = K3's count pass walks every read's CIGAR itself (the path of batches without hit lists)
  std::vector<lcr_candidate> h_cand;
  std::vector<int32_t> h_cand_off;
  DevBuf d_cand, d_cand_off;

  // K3
  bool have_frag = false;
  uint32_t min_linkers = 1;
  int32_t n_rows = 0;
  int64_t nnz = 0;
  std::vector<int32_t> h_row_region_off;
  DevBuf region_rows, row_region_off, row_cnt, row_links, row_ptr, col, val;
  HostBuf h_row_ptr, h_row_read, h_col, h_val, h_row_fp, h_row_links;

  // K4 + post-phase
  bool have_phase = false;
  bool res_valid = false;   // lcr_collect_phase: the last lcr_phase's results (host + HBM) are intact -- they outlive lcr_load_batch / lcr_pileup of the next batch
  int32_t res_ng = 0;
  int phase_slot = -1;      // staging slot of the batch whose (asynchronous) phase stage may be in flight: -1 = the caller's own device arrays, -2 = in_[] (a host batch)
  PhaseHost phase;
  std::vector<int32_t> ld_off, ld_snps;   // lcr_get_ld_blocks

  // region discovery (N3)
  DevBuf rd_start, rd_end, rd_diff, rd_ex, rd_cnt, rd_off, rd_s, rd_e, rd_max;
  std::vector<int64_t> rl_start0;
  std::vector<int32_t> rl_len;
  std::vector<uint32_t> rl_max;

  // timing
  bool timing = false;
  uint32_t timing_mask = 0;   // lcr_debug_set("timing_mask"): bit k = LCR_K_* k is timed; 0 = all of them (every timer is two event records on the stream)
  hipEvent_t ev[LCR_NKERNELS][2] = {};
  bool ev_valid[LCR_NKERNELS] = {};
  int64_t pileup_bytes = 0, stage_bytes = 0;
};

// ---- block cache (lcr_dev.h): freed device / page-locked blocks per device, first fit in size order
#include <map>
#include <mutex>
namespace {
struct BlockCache {
  std::mutex mu;
  std::multimap<size_t, void*> blocks[2][16];   // [host][device]: capacity -> block
  size_t held[2][16] = {};
  size_t limit[2] = {(size_t)24 << 30, (size_t)2 << 30};   // bytes kept per device: HBM blocks, page-locked host blocks
};
BlockCache& block_cache() { static BlockCache* c = new BlockCache(); return *c; }   // (never destroyed: contexts may outlive static destructors)
}  // namespace
void* lcr_cache_take(int host, size_t want, size_t* cap) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  BlockCache& bc = block_cache();
  std::lock_guard<std::mutex> g(bc.mu);
  auto& m = bc.blocks[host ? 1 : 0][dev];
  auto it = m.lower_bound(want);
  if (it == m.end() || it->first > 2 * want + (1u << 20)) return nullptr;
  void* p = it->second;
  *cap = it->first;
  bc.held[host ? 1 : 0][dev] -= it->first;
  m.erase(it);
  return p;
}
bool lcr_cache_put(int host, void* p, size_t cap) {
  int dev = 0;
  if (!p || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
  BlockCache& bc = block_cache();
  std::lock_guard<std::mutex> g(bc.mu);
  const int h = host ? 1 : 0;
  if (bc.held[h][dev] + cap > bc.limit[h]) return false;
  bc.blocks[h][dev].emplace(cap, p);
  bc.held[h][dev] += cap;
  return true;
}

namespace {

struct Timer {  // records HIP events on the ctx stream around one kernel
  lcr_ctx* c; int k;
  bool on() const { return c->timing && (c->timing_mask == 0 || ((c->timing_mask >> k) & 1u)); }
  Timer(lcr_ctx* c_, int k_) : c(c_), k(k_) { if (on()) { (void)hipEventRecord(c->ev[k][0], c->stream); } }
  ~Timer() { if (on()) { (void)hipEventRecord(c->ev[k][1], c->stream); c->ev_valid[k] = true; } }
};

DevParams to_dev(const lcr_params* p, float sor_thr) {
  DevParams d{};
  d.ont = p->platform == LCR_PLATFORM_ONT;
  d.dist_to_end = (int32_t)p->dist_to_end;
  d.polya_len = (int32_t)p->polya_len;
  d.min_baseq = p->min_baseq; d.min_depth = p->min_depth; d.max_depth = p->max_depth; d.min_qual = p->min_qual;
  d.low_cnt_cut = p->low_cnt_cut; d.min_linkers = p->min_linkers; d.use_strand_bias = p->use_strand_bias;
  d.min_af = p->min_af; d.min_af_intron = p->min_af_intron; d.low_frac_cut = p->low_frac_cut;
  d.sor_threshold = sor_thr;
  return d;
}

// Host -> device copy of a caller's (pageable) array on the context's stream.  Page-locked sources (hipHostMalloc / hipHostRegister: what
// lcr_load_batch_async asks for) go straight to the DMA engines.  Pageable ones are staged through page-locked buffers of the context,
// 8 MB at a time: the runtime's own path for them pins the caller's pages chunk by chunk, and on this stack (ROCm 7, MI355X) a device memory
// fault inside that path (rocr VMFaultHandler under hsaCopyStagedOrPinned / addPinnedMem, the caller still inside hipMemcpyAsync) aborted one
// test-suite run in five -- in torch's own .to() as well as here.  Small copies (<= 64 KB) are staged by the runtime itself either way.
int upload_bytes(lcr_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t q = nullptr) {
  if (!q) q = c->stream;
  if (bytes <= 64 * 1024) { HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, q)); return LCR_OK; }
  hipPointerAttribute_t at{};
  if (hipPointerGetAttributes(&at, src) == hipSuccess && at.type == hipMemoryTypeHost) {
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, q));
    return LCR_OK;
  }
  (void)hipGetLastError();   // (an unregistered pointer is reported as an error by some runtimes)
  constexpr size_t CH = 8u << 20;
  // one staging lane = two buffers + their events; large uploads run UP_LANES lanes on threads of their own (one thread's memcpy is
  // ~10 GB/s against the link's 50: 33 ms instead of 21 per C3 batch with a single lane)
  const int lanes = bytes >= (64u << 20) ? lcr_ctx::UP_LANES : 1;
  for (int k = 0; k < 2 * lanes; k++) {
    HIPCHK(c, c->h_up[k].reserve(CH));
    if (!c->ev_up[k]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_up[k], hipEventDisableTiming));
  }
  const size_t n_ch = (bytes + CH - 1) / CH;
  std::atomic<int> bad{0};
  auto lane_fn = [&](int w) {
    (void)hipSetDevice(c->device);
    int use = 0;
    for (size_t i = (size_t)w; i < n_ch && !bad.load(std::memory_order_relaxed); i += (size_t)lanes, use ^= 1) {
      const int k = 2 * w + use;
      const size_t off = i * CH, n = std::min(CH, bytes - off);
      hipError_t e = c->up_busy[k] ? hipEventSynchronize(c->ev_up[k]) : hipSuccess;
      if (e == hipSuccess) {
        memcpy(c->h_up[k].p, (const uint8_t*)src + off, n);
        e = hipMemcpyAsync((uint8_t*)dst + off, c->h_up[k].p, n, hipMemcpyHostToDevice, q);
      }
      if (e == hipSuccess) e = hipEventRecord(c->ev_up[k], q);
      if (e != hipSuccess) { bad.store((int)e); return; }
      c->up_busy[k] = true;
    }
  };
  if (lanes == 1) lane_fn(0);
  else {
    std::vector<std::thread> th;
    try { for (int w = 1; w < lanes; w++) th.emplace_back(lane_fn, w); } catch (...) { }   // (no thread to be had: their chunks are taken below)
    const int started = (int)th.size() + 1;
    lane_fn(0);
    for (auto& t : th) t.join();
    for (int w = started; w < lanes; w++) lane_fn(w);
  }
  if (bad.load()) { c->err = std::string("host upload: ") + hipGetErrorString((hipError_t)bad.load()); return LCR_E_DEVICE; }
  return LCR_OK;
}

template <class T>
int upload(lcr_ctx* c, DevBuf& buf, const T* src, size_t n, const T** dst, int mem) {
  if (mem == LCR_MEM_DEVICE) { *dst = src; return LCR_OK; }
  HIPCHK(c, buf.reserve(std::max<size_t>(n, 1) * sizeof(T)));
  if (n) { const int rc = upload_bytes(c, buf.p, src, n * sizeof(T)); if (rc) return rc; }
  *dst = buf.as<T>();
  return LCR_OK;
}

}  // namespace

int g_lcr_own_fill = 1;
namespace {
__global__ void __launch_bounds__(256) lcr_fill_kernel(uint8_t* p, uint32_t word, size_t head, size_t n16, size_t tail) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
  if (t < head) p[t] = (uint8_t)word;
  uint4* const body = reinterpret_cast<uint4*>(p + head);
  const uint4 w = make_uint4(word, word, word, word);
  for (size_t i = t; i < n16; i += nt) body[i] = w;
  if (t < tail) p[head + 16 * n16 + t] = (uint8_t)word;
}
}
namespace {
struct FillRanges { uint8_t* p[4]; size_t head[4], n16[4], tail[4]; uint32_t word[4]; };
__global__ void __launch_bounds__(256) lcr_fill_multi_kernel(FillRanges r) {
  const int k = blockIdx.y;
  uint8_t* const p = r.p[k];
  if (!p) return;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
  const uint32_t word = r.word[k];
  if (t < r.head[k]) p[t] = (uint8_t)word;
  uint4* const body = reinterpret_cast<uint4*>(p + r.head[k]);
  const uint4 w = make_uint4(word, word, word, word);
  for (size_t i = t; i < r.n16[k]; i += nt) body[i] = w;
  if (t < r.tail[k]) p[r.head[k] + 16 * r.n16[k] + t] = (uint8_t)word;
}
}
// up to four fills in one launch (every launch of its own costs ~5 us of queue time on this platform)
hipError_t lcr_fill_multi_async(int n, void* const* ptrs, const int* bytes_val, const size_t* sizes, hipStream_t s) {
  if (!g_lcr_own_fill) { for (int k = 0; k < n; k++) if (sizes[k]) { const hipError_t e = hipMemsetAsync(ptrs[k], bytes_val[k], sizes[k], s); if (e != hipSuccess) return e; } return hipSuccess; }
  FillRanges r{};
  size_t most = 0;
  int m = 0;
  for (int k = 0; k < n && m < 4; k++) {
    if (!sizes[k]) continue;
    const size_t mis = (size_t)((uintptr_t)ptrs[k] & 15), head = std::min(sizes[k], mis ? 16 - mis : 0), n16 = (sizes[k] - head) / 16;
    const uint32_t b = (uint32_t)(bytes_val[k] & 255);
    r.p[m] = (uint8_t*)ptrs[k]; r.head[m] = head; r.n16[m] = n16; r.tail[m] = sizes[k] - head - 16 * n16; r.word[m] = b | (b << 8) | (b << 16) | (b << 24);
    most = std::max(most, n16); m++;
  }
  if (!m) return hipSuccess;
  const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>((most + 255) / 256, 1024));
  hipLaunchKernelGGL(lcr_fill_multi_kernel, dim3(blocks, (unsigned)m), dim3(256), 0, s, r);
  return hipGetLastError();
}

hipError_t lcr_fill_async(void* p, int byte, size_t bytes, hipStream_t s) {
  if (!bytes) return hipSuccess;
  if (!g_lcr_own_fill) return hipMemsetAsync(p, byte, bytes, s);
  const size_t mis = (size_t)((uintptr_t)p & 15), head = std::min(bytes, mis ? 16 - mis : 0), n16 = (bytes - head) / 16, tail = bytes - head - 16 * n16;
  const uint32_t b = (uint32_t)(byte & 255), word = b | (b << 8) | (b << 16) | (b << 24);
  const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>((n16 + 255) / 256, 2048));
  hipLaunchKernelGGL(lcr_fill_kernel, dim3(blocks), dim3(256), 0, s, (uint8_t*)p, word, head, n16, tail);
  return hipGetLastError();
}

int g_lcr_host_trace = 0;
namespace {
struct HtMark { const char* tag; std::chrono::steady_clock::time_point t; };
thread_local std::vector<HtMark> g_ht;
}
void lcr_host_trace_mark(const char* tag) { g_ht.push_back({tag, std::chrono::steady_clock::now()}); }
void lcr_host_trace_flush() {
  if (g_ht.empty()) return;
  std::string line = "[host]";
  char buf[96];
  for (const HtMark& m : g_ht) { snprintf(buf, sizeof buf, " %s %.1f", m.tag, std::chrono::duration<double, std::micro>(m.t - g_ht[0].t).count()); line += buf; }
  fprintf(stderr, "%s\n", line.c_str());
  g_ht.clear();
}

extern "C" {

const char* lcr_version(void) { return "liblcr 0.1 (gfx950)"; }

int lcr_release_cached_memory(void) {   // every block the cache holds goes back to the runtime (all devices)
  BlockCache& bc = block_cache();
  std::lock_guard<std::mutex> g(bc.mu);
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (int dev = 0; dev < 16; dev++) {
    if (bc.blocks[0][dev].empty() && bc.blocks[1][dev].empty()) continue;
    (void)hipSetDevice(dev);
    for (auto& kv : bc.blocks[0][dev]) (void)hipFree(kv.second);
    for (auto& kv : bc.blocks[1][dev]) (void)hipHostFree(kv.second);
    bc.blocks[0][dev].clear(); bc.blocks[1][dev].clear();
    bc.held[0][dev] = bc.held[1][dev] = 0;
  }
  (void)hipSetDevice(cur);
  return LCR_OK;
}
int lcr_set_cache_limits(int64_t device_bytes, int64_t host_bytes) {   // per device; 0 = keep nothing (every block is freed when it is given back)
  if (device_bytes < 0 || host_bytes < 0) return LCR_E_ARG;
  BlockCache& bc = block_cache();
  std::lock_guard<std::mutex> g(bc.mu);
  bc.limit[0] = (size_t)device_bytes; bc.limit[1] = (size_t)host_bytes;
  return LCR_OK;
}

int lcr_params_preset(int preset, lcr_params* o) {
  if (!o || preset < 0 || preset > 3) return LCR_E_ARG;
  memset(o, 0, sizeof *o);
  const bool ont = preset >= 2;  // main.rs:272-396
  o->platform = ont ? LCR_PLATFORM_ONT : LCR_PLATFORM_HIFI;
  o->min_baseq = 10; o->dist_to_end = ont ? 20 : 40; o->polya_len = 5;
  o->min_depth = ont ? 10 : 6; o->max_depth = 50000; o->min_qual = 2;
  o->dense_win = 100; o->min_dense_cnt = 5; o->low_cnt_cut = 10; o->min_linkers = 1; o->max_enum_snps = 10;
  o->ld_weight_threshold = 1;
  o->use_strand_bias = (preset == 0 || preset == 2) ? 1 : 0;
  o->min_af = ont ? 0.20f : 0.15f; o->min_af_intron = 0.0f; o->low_frac_cut = 0.05f;
  o->min_phase_score = ont ? 13.0f : 11.0f;
  o->read_assign_cutoff = 0.0; o->seed = 2025;
  return LCR_OK;
}

int lcr_ctx_create(int device, lcr_ctx** out) {
  if (!out) return LCR_E_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return LCR_E_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return LCR_E_DEVICE;
  lcr_ctx* c = new lcr_ctx();
  c->device = device;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return LCR_E_DEVICE; }
  c->own_stream = true;
  for (int k = 0; k < LCR_NKERNELS; k++)
    for (int j = 0; j < 2; j++)
      if (hipEventCreate(&c->ev[k][j]) != hipSuccess) { delete c; return LCR_E_DEVICE; }
  *out = c;
  return LCR_OK;
}

void lcr_ctx_destroy(lcr_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->cand_pending && c->cand_dl_other && c->ev_cand_dl) (void)hipEventSynchronize(c->ev_cand_dl);   // (the candidate download on the phase stage's queue writes h_stage[1..2], freed below)
  (void)c->phase.settle(nullptr);   // (an lcr_phase whose results nobody collected: its queues are drained before anything is freed)
  if (c->phase.main_q) (void)hipStreamSynchronize(c->phase.main_q);
  if (c->phase.side) (void)hipStreamSynchronize(c->phase.side);
  if (c->phase.aux) (void)hipStreamSynchronize(c->phase.aux);
  (void)hipStreamSynchronize(c->stream);
  for (auto& b : c->in_) b.release();
  if (c->up_stream) (void)hipStreamSynchronize(c->up_stream);
  for (auto& u : c->up) { for (auto& b : u.buf) b.release(); if (u.ev) (void)hipEventDestroy(u.ev); }
  if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
  if (c->ev_dl) (void)hipEventDestroy(c->ev_dl);
  if (c->ev_cand_dl) (void)hipEventDestroy(c->ev_cand_dl);
  if (c->fill_stream) { (void)hipStreamSynchronize(c->fill_stream); (void)hipStreamDestroy(c->fill_stream); }
  if (c->ev_fill0) (void)hipEventDestroy(c->ev_fill0);
  if (c->ev_fill1) (void)hipEventDestroy(c->ev_fill1);
  DevBuf* bufs[] = {&c->rd_start, &c->rd_end, &c->rd_diff, &c->rd_ex, &c->rd_cnt, &c->rd_off, &c->rd_s, &c->rd_e, &c->rd_max,
                    &c->scan_tmp, &c->read_region, &c->read_bin, &c->read_rend, &c->tile_region, &c->tile_col0, &c->first_tile, &c->desc_tile, &c->desc_val, &c->chunks, &c->chunk_off,
                    &c->k0_tile_fill, &c->k0_items, &c->region_e_off, &c->frag_tmp_col, &c->frag_tmp_val, &c->tile_nbase, &c->blk_first_read, &c->read_scan, &c->cig_compact, &c->cig_off_new, &c->cig_new_off32, &c->planes, &c->flags,
                    &c->tile_count, &c->tile_off, &c->total, &c->survivors, &c->sv_region_off, &c->hist, &c->cand_tmp,
                    &c->keep, &c->d_cand, &c->d_cand_off, &c->region_rows, &c->row_region_off, &c->row_cnt,
                    &c->row_links, &c->row_ptr, &c->col, &c->val, &c->tile_order, &c->hit_cnt, &c->hit_list, &c->ovf_list};
  for (auto* b : bufs) b->release();
  if (c->ev_nnz) (void)hipEventDestroy(c->ev_nnz);
  if (c->ev_cand) (void)hipEventDestroy(c->ev_cand);
  if (c->ev_sv) (void)hipEventDestroy(c->ev_sv);
  if (c->ev_ctl) (void)hipEventDestroy(c->ev_ctl);
  for (int k = 0; k < 2 * lcr_ctx::UP_LANES; k++) { if (c->ev_up[k]) (void)hipEventDestroy(c->ev_up[k]); c->h_up[k].release(); }
  HostBuf* hb[] = {&c->h_order, &c->h_nnz, &c->h_stage[0], &c->h_stage[1], &c->h_stage[2], &c->h_stage[3], &c->h_planes, &c->h_row_ptr, &c->h_row_read, &c->h_col, &c->h_val, &c->h_row_fp, &c->h_row_links};
  for (auto* b : hb) b->release();
  c->phase.release();
  for (int k = 0; k < LCR_NKERNELS; k++) for (int j = 0; j < 2; j++) if (c->ev[k][j]) (void)hipEventDestroy(c->ev[k][j]);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* lcr_last_error(const lcr_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

// lcr_phase leaves its kernels in flight on the phase stage's own queues (lcr_phase_host.h): whoever needs its results, or is about
// to overwrite what it reads / writes, collects them first
static int phase_settle(lcr_ctx* c) { return c->phase.settle(&c->err); }

int lcr_ctx_set_stream(lcr_ctx* c, void* s) {
  if (!c) return LCR_E_ARG;
  (void)hipSetDevice(c->device);
  { int rc = phase_settle(c); if (rc) return rc; }
  if (c->own_stream && c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
  if (s) { c->stream = (hipStream_t)s; c->own_stream = false; }
  else { HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
  return LCR_OK;
}

int lcr_ctx_sync(lcr_ctx* c) {
  if (!c) return LCR_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  { int rc = phase_settle(c); if (rc) return rc; }
  if (c->cand_pending && c->cand_dl_other) HIPCHK(c, hipEventSynchronize(c->ev_cand_dl));   // (the candidate records' download rides on the phase stage's second queue)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LCR_OK;
}

int lcr_enable_timing(lcr_ctx* c, int on) { if (!c) return LCR_E_ARG; c->timing = on != 0; return LCR_OK; }

int lcr_kernel_ms(lcr_ctx* c, int k, float* ms) {
  if (!c || !ms || k < 0 || k >= LCR_NKERNELS) return LCR_E_ARG;
  if (!c->ev_valid[k]) return LCR_E_STATE;
  HIPCHK(c, hipEventSynchronize(c->ev[k][1]));
  HIPCHK(c, hipEventElapsedTime(ms, c->ev[k][0], c->ev[k][1]));
  return LCR_OK;
}

int lcr_pileup_bytes(lcr_ctx* c, int64_t* bytes) {
  if (!c || !bytes) return LCR_E_ARG;
  *bytes = c->pileup_bytes;
  return LCR_OK;
}

int lcr_pileup_stage_bytes(lcr_ctx* c, int64_t* bytes) {
  if (!c || !bytes) return LCR_E_ARG;
  *bytes = c->stage_bytes;
  return LCR_OK;
}

int lcr_load_batch(lcr_ctx* c, const lcr_reads* rd, const lcr_regions* rg) {
  if (!c || !rd || !rg) return LCR_E_ARG;
  if (rd->n_reads < 0 || rg->n_regions < 0 || rd->mem != rg->mem) { c->err = "bad batch header"; return LCR_E_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  // a device-resident batch may be bound (and its pileup queued) while the previous batch's phase stage is still running: nothing
  // here or in lcr_pileup touches what that stage reads.  A host batch is copied into the context's staging buffers, which hold
  // the previous batch's region table: the stage has to be done first.
  if (rd->mem == LCR_MEM_HOST) { int rc = phase_settle(c); if (rc) return rc; }
  c->loaded = c->have_planes = c->have_cand = c->have_frag = c->have_phase = false;
  c->bound_slot = -1;
  c->bound_host = rd->mem == LCR_MEM_HOST;
  const int nr = rd->n_reads, ng = rg->n_regions, mem = rd->mem;
  // host copies of the small per-region arrays
  c->h_start0.assign(ng, 0); c->h_len.assign(ng, 0); c->h_col_off.assign(ng + 1, 0); c->h_read_begin.assign(ng + 1, 0);
  if (mem == LCR_MEM_HOST) {
    if (ng) { memcpy(c->h_start0.data(), rg->start0, ng * sizeof(int64_t)); memcpy(c->h_len.data(), rg->len, ng * sizeof(int32_t)); }
    memcpy(c->h_col_off.data(), rg->col_off, (ng + 1) * sizeof(int64_t));
    memcpy(c->h_read_begin.data(), rg->read_begin, (ng + 1) * sizeof(int32_t));
  } else {   // device-resident batch: one kernel writes the four small region arrays into pinned host memory, one wait
    const size_t o1 = (size_t)ng * 8, o2 = o1 + (size_t)(ng + 1) * 8, o3 = o2 + (size_t)ng * 4, tot = o3 + (size_t)(ng + 1) * 4;
    HIPCHK(c, c->h_stage[0].reserve(tot + 16));
    HIPCHK(c, c->first_tile.reserve((ng + 1) * 4));
    uint8_t* st = c->h_stage[0].as<uint8_t>();
    uint8_t* dst = nullptr;   // the pinned buffer as the device sees it
    HIPCHK(c, hipHostGetDevicePointer((void**)&dst, st, 0));
    // the same launch and the same wait bring the geometry of the flat op space: first op, end of the last read's ops, "CIGARs lie back to back"
    HIPCHK(c, c->h_order.reserve(64));
    memset(c->h_order.p, 0, 64);
    { int32_t* d_flag = nullptr; HIPCHK(c, hipHostGetDevicePointer((void**)&d_flag, c->h_order.p, 0));
      Timer t(c, LCR_K_BIND_TABLE);
      launch_k0_bind_a(rg->start0, rg->len, rg->col_off, rg->read_begin, ng, c->first_tile.as<int32_t>(), (int64_t*)dst,
                       (int32_t*)(dst + o2), (int64_t*)(dst + o1), (int32_t*)(dst + o3), rd->cig_off, rd->n_cig, nr, rd->n_cigar, d_flag, c->stream); }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    if (ng) { memcpy(c->h_start0.data(), st, ng * sizeof(int64_t)); memcpy(c->h_len.data(), st + o2, ng * sizeof(int32_t)); }
    memcpy(c->h_col_off.data(), st + o1, (ng + 1) * sizeof(int64_t));
    memcpy(c->h_read_begin.data(), st + o3, (ng + 1) * sizeof(int32_t));
  }
  if (c->h_read_begin[ng] != nr || c->h_col_off[0] != 0) { c->err = "read_begin/col_off inconsistent"; return LCR_E_ARG; }
  for (int g = 0; g < ng; g++)
    if (c->h_len[g] < 0 || c->h_col_off[g + 1] - c->h_col_off[g] != c->h_len[g] || c->h_read_begin[g + 1] < c->h_read_begin[g]) {
      c->err = "region table inconsistent"; return LCR_E_ARG;
    }
  c->n_cols = c->h_col_off[ng];
  // (tile column origins and the intron difference array are indexed with int32)
  if (c->n_cols + ng + 1 > (int64_t)INT32_MAX) { c->err = "batch too large: columns + regions must stay below 2^31; split it"; return LCR_E_ARG; }
  c->n_bases = rd->n_bases; c->n_cigar = rd->n_cigar;

  BatchView& b = c->bv;
  b.n_reads = nr; b.n_regions = ng; b.n_bases = rd->n_bases;
  int rc;
#define UP(i, field, T, n) if ((rc = upload<T>(c, c->in_[i], (const T*)rd->field, (size_t)(n), (const T**)&b.field, mem))) return rc
  UP(0, pos, int32_t, nr); UP(1, seq_len, int32_t, nr);
  if ((rc = upload<int32_t>(c, c->in_[2], rd->lead_clip, nr, &b.lead, mem))) return rc;
  if ((rc = upload<int32_t>(c, c->in_[3], rd->trail_clip, nr, &b.trail, mem))) return rc;
  UP(4, flags, uint8_t, nr); UP(5, seq_off, uint64_t, nr); UP(6, cig_off, uint64_t, nr); UP(7, n_cig, uint32_t, nr);
  UP(8, bases, uint8_t, rd->n_bases); UP(9, quals, uint8_t, rd->n_bases); UP(10, cigar, uint32_t, rd->n_cigar);
#undef UP
  if ((rc = upload<int64_t>(c, c->in_[11], rg->start0, ng, &b.start0, mem))) return rc;
  if ((rc = upload<int32_t>(c, c->in_[12], rg->len, ng, &b.len, mem))) return rc;
  if ((rc = upload<int64_t>(c, c->in_[13], rg->col_off, ng + 1, &b.col_off, mem))) return rc;
  if ((rc = upload<int32_t>(c, c->in_[14], rg->read_begin, ng + 1, &b.read_begin, mem))) return rc;
  if ((rc = upload<uint8_t>(c, c->in_[15], rg->ref, c->n_cols, &b.ref, mem))) return rc;

  // tile table: tiles never cross a region; the host only needs the tile count, the table is built on the device
  c->h_region_first_tile.assign(ng + 1, 0);
  for (int g = 0; g < ng; g++) c->h_region_first_tile[g + 1] = c->h_region_first_tile[g] + (c->h_len[g] + LCR_TILE - 1) / LCR_TILE;
  c->n_tiles = c->h_region_first_tile[ng];
  HIPCHK(c, c->tile_region.reserve(std::max<size_t>(c->n_tiles, 1) * 4));
  HIPCHK(c, c->tile_col0.reserve(std::max<size_t>(c->n_tiles, 1) * 4));
  HIPCHK(c, c->first_tile.reserve((ng + 1) * 4));
  if (mem == LCR_MEM_HOST)   // (a device-resident batch had its prefix sums computed with the region fetch above)
    launch_k0_region_setup(b.start0, b.len, b.col_off, b.read_begin, ng, c->first_tile.as<int32_t>(), nullptr, nullptr, nullptr, nullptr, c->stream);
  HIPCHK(c, c->read_rend.reserve(std::max<size_t>(nr, 1) * 4));
  b.read_rend = c->read_rend.as<int32_t>();
  HIPCHK(c, c->read_region.reserve(std::max(nr, 1) * 4));
  b.read_region = c->read_region.as<int32_t>();
  b.region_first_tile = c->first_tile.as<int32_t>(); b.error_flag = nullptr;   // set by lcr_pileup
  HIPCHK(c, c->read_bin.reserve(std::max<size_t>(nr, 1) * sizeof(ReadBin)));
  // ---- the flat op space of K0 (k0_ops.hip): ops [cig0, cig0 + n_ops) of bv.cigar, read after read
  bool contiguous = true, cig_oob = false;
  uint64_t cig0 = 0, cig_end = 0, cig_total = 0;
  if (mem == LCR_MEM_HOST) {
    HIPCHK(c, c->h_order.reserve(64));
    // (a device-resident batch bound before this one returned without a wait: its k0_pack may still be about to raise the flag)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memset(c->h_order.p, 0, 64);
    if (nr) { cig0 = rd->cig_off[0]; cig_end = rd->cig_off[nr - 1] + rd->n_cig[nr - 1]; }
    for (int r = 0; r + 1 < nr && contiguous; r++) contiguous = rd->cig_off[r + 1] == rd->cig_off[r] + rd->n_cig[r];
    for (int r = 0; r < nr; r++) {   // (any layout: every read's ops inside the caller's array; the total in 64 bits)
      cig_total += rd->n_cig[r];
      cig_oob |= rd->cig_off[r] > (uint64_t)rd->n_cigar || (uint64_t)rd->n_cig[r] > (uint64_t)rd->n_cigar - rd->cig_off[r];
    }
  } else {
    const uint64_t* g = reinterpret_cast<const uint64_t*>(c->h_order.as<uint8_t>() + 16);   // written by k0_cig_check, waited for above
    contiguous = c->h_order.as<int32_t>()[1] == 0;
    cig_oob = c->h_order.as<int32_t>()[2] != 0;
    cig0 = g[0]; cig_end = g[1];
  }
  if (cig_oob) { c->err = "cig_off / n_cig reach beyond n_cigar"; return LCR_E_ARG; }
  // (every read lies inside [0, n_cigar): a total beyond 2^31 needs n_cigar beyond it or overlapping reads -- the scan below is int32)
  if (!contiguous && (rd->n_cigar > 0x7FFFFFF0ll || cig_total > 0x7FFFFFF0ull)) { c->err = "batch too large: CIGAR ops must stay below 2^31; split it"; return LCR_E_ARG; }
  if (!contiguous) {   // the ABI allows any cig_off: copy the CIGARs back to back once (rare; every producer here is contiguous)
    HIPCHK(c, c->cig_new_off32.reserve(((size_t)nr + 2) * 4));
    HIPCHK(c, c->cig_off_new.reserve(std::max<size_t>(nr, 1) * 8));
    int32_t* total = c->cig_new_off32.as<int32_t>() + nr;
    launch_scan_i32(c->scan_tmp, (const int32_t*)b.n_cig, c->cig_new_off32.as<int32_t>(), nr, total, c->stream);
    int32_t h_total = 0;
    HIPCHK(c, hipMemcpyAsync(&h_total, total, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (h_total < 0) { c->err = "batch too large: CIGAR ops must stay below 2^31; split it"; return LCR_E_ARG; }
    HIPCHK(c, c->cig_compact.reserve(std::max<size_t>((size_t)h_total, 1) * 4));
    launch_k0_cig_compact(b.cigar, b.cig_off, b.n_cig, c->cig_new_off32.as<int32_t>(), nr, c->cig_compact.as<uint32_t>(),
                          c->cig_off_new.as<uint64_t>(), c->stream);
    b.cigar = c->cig_compact.as<uint32_t>(); b.cig_off = c->cig_off_new.as<uint64_t>();
    cig0 = 0; cig_end = (uint64_t)h_total;
  }
  if (cig_end < cig0 || cig_end - cig0 > 0xFFF00000ull || (contiguous && cig_end > (uint64_t)std::max<int64_t>(rd->n_cigar, 0))) {
    c->err = "cig_off / n_cig inconsistent with n_cigar, or more than 2^32 CIGAR ops in one batch"; return LCR_E_ARG;
  }
  c->cig0 = cig0; c->n_ops = (uint32_t)(cig_end - cig0);
  *c->h_order.as<int32_t>() = 0;   // (the previous batch's k0_pack finished long ago: every lcr_pileup waits behind it)
  // ONE launch: read -> region and tile tables, the packed read headers, the order check, the op blocks' first reads (k0_bind_b)
  { int32_t* d_flag = nullptr; HIPCHK(c, hipHostGetDevicePointer((void**)&d_flag, c->h_order.p, 0));
    const int opb = launch_k0_opb();
    const int32_t n_blocks = (int32_t)(((uint64_t)c->n_ops + opb - 1) / opb);
    HIPCHK(c, c->blk_first_read.reserve(((size_t)n_blocks + 2) * 4));
    Timer t(c, LCR_K_BIND);
    launch_k0_bind_b(b, c->read_bin.as<ReadBin>(), d_flag, c->read_region.as<int32_t>(), c->n_tiles, c->tile_region.as<int32_t>(), c->tile_col0.as<int32_t>(),
                     c->cig0, opb, n_blocks, c->blk_first_read.as<int32_t>(), c->stream); }
  // host batch: the caller's arrays are free again when this returns; device batch: no wait, the next stage queues
  // behind these kernels on the same stream (the arrays stay the caller's to keep alive, include/lcr.h)
  if (mem == LCR_MEM_HOST) HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  c->loaded = true;
  return LCR_OK;
}

int lcr_load_batch_async(lcr_ctx* c, const lcr_reads* rd, const lcr_regions* rg, int32_t slot) {
  if (!c || !rd || !rg) return LCR_E_ARG;
  if (slot < 0 || slot > 1) { c->err = "lcr_load_batch_async: slot must be 0 or 1"; return LCR_E_ARG; }
  if (rd->mem != LCR_MEM_HOST || rg->mem != LCR_MEM_HOST) { c->err = "lcr_load_batch_async takes LCR_MEM_HOST batches (a device-resident batch needs no upload)"; return LCR_E_ARG; }
  if (rd->n_reads < 0 || rg->n_regions < 0 || rd->n_bases < 0 || rd->n_cigar < 0) { c->err = "bad batch header"; return LCR_E_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  // a phase stage in flight reads the region table of ITS batch: it has to be done only if that batch lives in the slot rewritten here
  // (two slots alternate: batch k + 1 is uploaded while batch k's stage runs -- lcr_collect_phase fetches batch k's results afterwards)
  if (c->phase.pending && c->phase_slot == slot) { int rc = phase_settle(c); if (rc) return rc; }
  if (!c->up_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
  lcr_ctx::UploadSlot& u = c->up[slot];
  if (!u.ev) HIPCHK(c, hipEventCreateWithFlags(&u.ev, hipEventDisableTiming));
  if (c->bound_slot == slot) {   // the bound batch lives in this slot: its kernels must be done before it is overwritten
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->loaded = c->have_planes = c->have_cand = c->have_frag = c->have_phase = false;
    c->bound_slot = -1;
  }
  u.filled = false;
  const int nr = rd->n_reads, ng = rg->n_regions;
  int64_t n_cols = 0;
  if (ng) { if (!rg->col_off) { c->err = "col_off missing"; return LCR_E_ARG; } n_cols = rg->col_off[ng]; }
  if (n_cols < 0) { c->err = "region table inconsistent"; return LCR_E_ARG; }
  u.rd = lcr_reads{}; u.rg = lcr_regions{};
  u.rd.mem = LCR_MEM_DEVICE; u.rg.mem = LCR_MEM_DEVICE;
  u.rd.n_reads = nr; u.rd.n_bases = rd->n_bases; u.rd.n_cigar = rd->n_cigar; u.rg.n_regions = ng;
  auto put = [&](int i, const void* src, size_t bytes, const void** dst) -> int {
    HIPCHK(c, u.buf[i].reserve(std::max<size_t>(bytes, 1)));
    if (bytes) {
      if (!src) { c->err = "lcr_load_batch_async: null array"; return LCR_E_ARG; }
      { const int rc2 = upload_bytes(c, u.buf[i].p, src, bytes, c->up_stream); if (rc2) return rc2; }   // (page-locked arrays: asynchronous; pageable ones are staged)
    }
    *dst = u.buf[i].p;
    return LCR_OK;
  };
  int rc;
#define PUT(i, obj, field, T, n) if ((rc = put(i, (obj)->field, (size_t)(n) * sizeof(T), (const void**)&u.obj.field))) return rc
  PUT(0, rd, pos, int32_t, nr); PUT(1, rd, seq_len, int32_t, nr); PUT(2, rd, lead_clip, int32_t, nr); PUT(3, rd, trail_clip, int32_t, nr);
  PUT(4, rd, flags, uint8_t, nr); PUT(5, rd, seq_off, uint64_t, nr); PUT(6, rd, cig_off, uint64_t, nr); PUT(7, rd, n_cig, uint32_t, nr);
  PUT(8, rd, bases, uint8_t, rd->n_bases); PUT(9, rd, quals, uint8_t, rd->n_bases); PUT(10, rd, cigar, uint32_t, rd->n_cigar);
  PUT(11, rg, start0, int64_t, ng); PUT(12, rg, len, int32_t, ng); PUT(13, rg, col_off, int64_t, ng + 1); PUT(14, rg, read_begin, int32_t, ng + 1);
  PUT(15, rg, ref, uint8_t, n_cols);
#undef PUT
  HIPCHK(c, hipEventRecord(u.ev, c->up_stream));
  u.filled = true;
  return LCR_OK;
}

int lcr_bind_batch(lcr_ctx* c, int32_t slot) {
  if (!c) return LCR_E_ARG;
  if (slot < 0 || slot > 1) { c->err = "lcr_bind_batch: slot must be 0 or 1"; return LCR_E_ARG; }
  lcr_ctx::UploadSlot& u = c->up[slot];
  if (!u.filled) { c->err = "lcr_bind_batch before lcr_load_batch_async on this slot"; return LCR_E_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamWaitEvent(c->stream, u.ev, 0));   // the ctx stream continues behind the slot's upload
  const int rc = lcr_load_batch(c, &u.rd, &u.rg);      // (device-resident form: region tables fetched with one wait; the upload is complete when it returns)
  if (rc == LCR_OK) c->bound_slot = slot;
  return rc;
}

int lcr_host_alloc(size_t bytes, void** out) {
  if (!out) return LCR_E_ARG;
  *out = nullptr;
  return hipHostMalloc(out, std::max<size_t>(bytes, 1), hipHostMallocDefault) == hipSuccess ? LCR_OK : LCR_E_NOMEM;
}
void lcr_host_free(void* p) { if (p) (void)hipHostFree(p); }
int lcr_host_register(void* p, size_t bytes) {
  if (!p || !bytes) return LCR_E_ARG;
  return hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess ? LCR_OK : LCR_E_DEVICE;
}
int lcr_host_unregister(void* p) { return p && hipHostUnregister(p) == hipSuccess ? LCR_OK : LCR_E_ARG; }

int lcr_pileup(lcr_ctx* c, const lcr_params* p) {
  HT("pileup");
  if (!c || !p) return LCR_E_ARG;
  if (!c->loaded) { c->err = "lcr_pileup before lcr_load_batch"; return LCR_E_STATE; }
  if (p->polya_len == 0) { c->err = "polya_len must be >= 1"; return LCR_E_ARG; }
  // (the ends kernel of the poly-A mask takes dist_to_end <= 63 and polya_len in 2..16 -- every preset --, the per-offset kernel the rest;
  // the thread index of the latter runs over n_reads x 2 x dist_to_end)
  if ((uint64_t)c->bv.n_reads * 2ull * p->dist_to_end > 0x7FFFFF00ull * (uint64_t)LCR_BLOCK) { c->err = "dist_to_end x reads too large for one launch; split the batch"; return LCR_E_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  if (c->sor_thr < 0.f) c->sor_thr = lcr_device_sor_threshold(c->stream);  // candidate.rs:49-51, evaluated by the device's logf
  c->dp = to_dev(p, c->sor_thr);
  c->dp.dbg = 0;
  HIPCHK(c, c->planes.reserve(std::max<size_t>((size_t)c->n_cols * LCR_NPLANES, 1) * 4));
  bool gated = false;   // (async_phase: K0 waits for the restarts of a phase stage still in flight -- below, behind the fill in front of it)
  BatchView& b = c->bv;
  const int nt = c->n_tiles;
  // ---- K0: decode every CIGAR once into per-tile records (one op-parallel pass; a block's records lie back to back in the
  // pool, grouped by tile, each group announced by a chunk descriptor)
  // records <= M / D / I ops + their tile crossings (M: <= bases / tile) + at most two per intron.  Long D runs can exceed
  // the estimate: K0 then flags an overflow (writes are bounds-checked) and the stage is repeated with larger pools.
  const int opb = launch_k0_opb();
  const int32_t n_blocks = (int32_t)(((uint64_t)c->n_ops + opb - 1) / opb);
  // (a block's records of one tile take whole 16-slot units: + 15 slots per (block, tile) group at most)
  size_t desc_cap64 = (size_t)n_blocks * 128 + (size_t)b.n_reads / 4 + 1024;
  size_t pool_cap64 = (size_t)c->n_ops + (size_t)c->n_ops / 2 + (size_t)b.n_reads + (size_t)c->n_bases / LCR_TILE + 8 * desc_cap64 + 1024;
  // one cleared buffer: tile fill counters [0, nt) | control block at nt + 1 (pool top, items, records, error flag, descriptor
  // top) | tile-level intron difference array | chunks per tile | bin cursors | K0's accounting slots
  const size_t o_ndiff = (size_t)nt + 16, o_nch = o_ndiff + nt + 8, o_cur = o_nch + nt + 8, o_acct = o_cur + nt + 8;
  const size_t o_tmp = o_acct + launch_k0_acct_words();   // scratch of the tile passes (class counts, cursors, block sums)
  const size_t fill_words = (o_tmp + launch_k1_tiles_tmp_words(nt) + 63) & ~(size_t)63;   // (a multiple of 256 bytes: one fill kernel, not a body and a tail)
  HIPCHK(c, c->k0_tile_fill.reserve(fill_words * 4));
  int32_t* const fill = c->k0_tile_fill.as<int32_t>();
  b.error_flag = fill + nt + 4;
  HIPCHK(c, c->tile_order.reserve(std::max(nt, 1) * 4));
  HIPCHK(c, c->tile_nbase.reserve(std::max(nt, 1) * 4));
  HIPCHK(c, c->chunk_off.reserve(((size_t)nt + 2) * 4));
  HIPCHK(c, c->read_scan.reserve(std::max<size_t>(b.n_reads, 1) * 8));
  HIPCHK(c, c->h_stage[0].reserve(64));
  int32_t n_recs = 0, bad = 0, n_ops = 0;
  // Three quarters of a spliced batch's tiles hold no record: all their planes are zeros (the intron plane: a constant).  That
  // store stream (52 B per column) is the stage's largest HBM write, and K0 -- instruction-bound -- leaves the memory system idle:
  // ALL planes are zeroed on a second queue while K0 runs, K1 then writes the tiles with records and the intron constants.
  // (Under the tally the same stream hurts: the tiles' dependent loads queue behind it.  DESIGN.md K1.)
  const bool prefill = c->dbg_prefill != 0 && nt > 0;
  // (measurement switch) HiFi presets: the record-free tiles' stores in ONE launch with the poly-A pass
  const bool zf_fused = c->dbg_zf_fused != 0 && !c->dp.ont && c->dp.dist_to_end > 0 && c->dp.dist_to_end <= 63 && c->dp.polya_len >= 2 && c->dp.polya_len <= 16 &&
                        nt > 0 && !prefill && c->dbg_bg_tiles == 0 && c->dbg_zf_overlap == 0 && b.n_reads > 0;
  // (measurement switch) HiFi presets: the record-free tiles' stores beside the poly-A pass
  const bool zf_overlap = c->dbg_zf_overlap != 0 && !c->dp.ont && c->dp.dist_to_end > 0 && nt > 0 && !prefill && c->dbg_bg_tiles == 0;
  if ((prefill || c->dbg_bg_tiles > 0 || c->dbg_bg_tiles == -1 || zf_overlap) && !c->fill_stream) { HIPCHK(c, hipStreamCreateWithFlags(&c->fill_stream, hipStreamNonBlocking)); HIPCHK(c, hipEventCreateWithFlags(&c->ev_fill0, hipEventDisableTiming)); HIPCHK(c, hipEventCreateWithFlags(&c->ev_fill1, hipEventDisableTiming)); }
  if (prefill) {
    if (!c->fill_stream) { HIPCHK(c, hipStreamCreateWithFlags(&c->fill_stream, hipStreamNonBlocking)); HIPCHK(c, hipEventCreateWithFlags(&c->ev_fill0, hipEventDisableTiming)); HIPCHK(c, hipEventCreateWithFlags(&c->ev_fill1, hipEventDisableTiming)); }
    HIPCHK(c, hipEventRecord(c->ev_fill0, c->stream));            // (the planes' last readers of the previous batch are ahead in the ctx stream)
    HIPCHK(c, hipStreamWaitEvent(c->fill_stream, c->ev_fill0, 0));
    HIPCHK(c, hipMemsetAsync(c->planes.p, 0, (size_t)c->n_cols * LCR_NPLANES * 4, c->fill_stream));
    HIPCHK(c, hipEventRecord(c->ev_fill1, c->fill_stream));
  }
  // pass 1 of the candidate filters inside the tally's epilogue (k2_eval.h): presets whose planes are final when K1 stores them (ONT: the
  // HiFi presets subtract the poly-A mask afterwards, k1_zonefix); lcr_candidates uses the flags if it is called with the same filters
  const bool fuse = c->dbg_fuse_filter != 0 && c->dp.ont && nt > 0;
  c->flt_fused = false;
  if (fuse) { HIPCHK(c, c->flags.reserve(std::max<size_t>(c->n_cols, 1))); HIPCHK(c, c->tile_count.reserve(std::max(nt, 1) * 4)); }
  for (;;) {
    // the pool and the descriptor array are cut into launch_k0_acct_slots() shards (a block allocates from shard blockIdx % shards)
    const size_t nsh = (size_t)launch_k0_acct_slots();
    const size_t pool_sub64 = (pool_cap64 + nsh - 1) / nsh + 256, desc_sub64 = (desc_cap64 + nsh - 1) / nsh + 64;
    if (pool_sub64 * nsh > 0xFFFFFFF0ull || desc_sub64 * nsh > 0x7FFFFFF0ull) { c->err = "batch too large for the 32-bit record pool: split it"; return LCR_E_ARG; }
    const unsigned int pool_sub = (unsigned int)pool_sub64, desc_sub = (unsigned int)desc_sub64;
    HIPCHK(c, c->k0_items.reserve(pool_sub64 * nsh * 8));
    HIPCHK(c, c->desc_tile.reserve(desc_sub64 * nsh * 4));
    HIPCHK(c, c->desc_val.reserve(desc_sub64 * nsh * 8));
    // entries of 16 slots: a group of c records inside a block's tile window takes ceil(c / 16) entries AND ceil(c / 16) * 16
    // pool slots, a record outside the window one pool slot, one descriptor and one entry of its own -- so the entry list is
    // bounded by pool / 16 + descriptors, not by pool / 16 (thousands of reads across an intron of > 65 536 columns)
    HIPCHK(c, c->chunks.reserve((pool_sub64 * nsh / 16 + desc_sub64 * nsh + 16) * 8));
    HIPCHK(c, lcr_fill_async(fill, 0, fill_words * 4, c->stream));
    // (async_phase: behind the restarts of a phase stage still in flight, beside its tails -- a matter of speed, not of order: the fill runs early)
    if (!gated) { HIPCHK(c, c->phase.gate_stream(c->stream)); gated = true; }
    { Timer t(c, LCR_K_SPANS);
      launch_k0_ops(b, c->read_bin.as<ReadBin>(), c->blk_first_read.as<int32_t>(), c->cig0, c->n_ops, c->dp.ont, c->dp.dist_to_end, nt,
                    fill, fill + o_nch, fill + o_ndiff, fill + nt + 1, (unsigned int*)(fill + o_acct), pool_sub, c->k0_items.as<unsigned long long>(),
                    desc_sub, c->desc_tile.as<uint32_t>(), c->desc_val.p, c->read_scan.p, c->stream); }
    // tile order for K1, introns per whole tile, chunk offsets, K0's accounting (one workgroup); K0's verdict (CIGAR
    // validation, pool overflow) and counts then leave for the host before the rest is queued: the host waits for them while
    // K1 runs and returns without waiting for K1 -- later calls queue behind it
    int32_t* const ctl = c->h_stage[0].as<int32_t>();
    if (!c->ev_ctl) HIPCHK(c, hipEventCreateWithFlags(&c->ev_ctl, hipEventDisableTiming));
    { Timer t(c, LCR_K_PILEUP);   // (the tally kernel with its ordering and chunk-binning passes)
      if (nt > 0) {   // (k1_tiles_a writes K0's verdict and counts straight into the pinned block: no copy in the queue in front of k1_tiles_b)
        unsigned int* d_ctl = nullptr;
        HIPCHK(c, hipHostGetDevicePointer((void**)&d_ctl, ctl, 0));
        launch_k1_tiles_a(nt, fill, fill + o_ndiff, fill + o_nch, fill + o_tmp, (unsigned int*)(fill + o_acct), launch_k0_acct_slots(),
                          (unsigned int*)(fill + nt + 1), d_ctl, c->stream);
      } else HIPCHK(c, hipMemcpyAsync(ctl, fill + nt + 1, 32, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipEventRecord(c->ev_ctl, c->stream));
      if (nt > 0) launch_k1_tiles_b(nt, fill, fill + o_ndiff, fill + o_nch, fill + o_tmp, c->tile_nbase.as<int32_t>(), c->chunk_off.as<int32_t>(),
                                    c->tile_order.as<int32_t>(), c->stream);
      const bool early_empty = c->dbg_bg_tiles == -1 && nt > 0 && !prefill;
      if (early_empty)   // (measurement switch) the record-free tiles' store stream beside k0_desc_bin and the start of the tally
        launch_k1_empty_early(b, c->tile_region.as<int32_t>(), c->tile_col0.as<int32_t>(), nt, c->n_cols, c->tile_nbase.as<int32_t>(), c->planes.as<uint32_t>(),
                              c->tile_order.as<int32_t>(), fill + o_tmp, c->stream, c->fill_stream, c->ev_fill0, c->ev_fill1, fuse ? c->tile_count.as<int32_t>() : nullptr);
      if (nt > 0 && c->n_ops > 0)
        launch_k0_desc_bin(fill + nt + 1, (const unsigned int*)(fill + o_acct), desc_sub, c->desc_tile.as<uint32_t>(), c->desc_val.p, c->chunk_off.as<int32_t>(), fill + o_cur,
                           c->chunks.p, n_blocks / 8 + 1, c->stream);
      // ---- K1: per-tile tally from the records (leaves at once if K0 flagged an error); K1z: poly-A / homopolymer
      // mask of the HiFi presets
      if (prefill) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_fill1, 0));
      launch_k1_pileup(b, c->dp, c->tile_region.as<int32_t>(), c->tile_col0.as<int32_t>(), nt, c->n_cols, fill, c->chunk_off.as<int32_t>(),
                       c->chunks.p, c->k0_items.as<unsigned long long>(), c->tile_nbase.as<int32_t>(), c->planes.as<uint32_t>(),
                       c->tile_order.as<int32_t>(), fill + o_tmp, prefill ? 1 : 0, c->stream, (c->dbg_bg_tiles > 0 || c->dbg_bg_tiles == -1 || zf_overlap) ? c->fill_stream : nullptr, c->ev_fill0, c->ev_fill1, zf_fused ? -4 : zf_overlap ? -3 : c->dbg_bg_tiles,
                       fuse ? c->flags.as<uint8_t>() : nullptr, fuse ? c->tile_count.as<int32_t>() : nullptr);
      if (zf_fused)
        (void)launch_k1_zonefix_tiles(b, c->dp.dist_to_end, c->dp.polya_len, c->n_cols, c->planes.as<uint32_t>(), c->tile_region.as<int32_t>(), c->tile_col0.as<int32_t>(), nt,
                                      c->tile_nbase.as<int32_t>(), c->tile_order.as<int32_t>(), fill + o_tmp, fuse ? c->tile_count.as<int32_t>() : nullptr, c->stream);
      else if (!c->dp.ont && c->dp.dist_to_end > 0)
        launch_k1_zonefix(b, c->read_bin.as<ReadBin>(), c->dp.dist_to_end, c->dp.polya_len, c->n_cols, c->planes.as<uint32_t>(), c->stream);
      if (zf_overlap) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_fill1, 0)); }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventSynchronize(c->ev_ctl));
  HT("pile:ctl");
    n_ops = ctl[1]; n_recs = ctl[2]; bad = ctl[3];
    if (*c->h_order.as<int32_t>() != 0) { c->err = "the reads of a region must be sorted by position (lcr_reads.pos)"; return LCR_E_ARG; }
    if (bad == 1) { c->err = "unknown CIGAR operation (reference panics: util.rs:944)"; return LCR_E_CIGAR; }
    if (bad == 2) { c->err = "CIGAR inconsistent with l_seq / soft clips"; return LCR_E_CIGAR; }
    if (bad == 0) break;
    // overflow: K0 kept counting -- the true record and descriptor counts are known now
    // (ctl[0] / ctl[4]: what the fullest shard asked for)
    pool_cap64 = std::max<size_t>(pool_cap64 * 2, ((size_t)(uint32_t)ctl[0] + 1024) * nsh);
    desc_cap64 = std::max<size_t>(desc_cap64 * 2, ((size_t)(uint32_t)ctl[4] + 1024) * nsh);
  }
  c->n_items = n_recs;
  // bytes K1 itself has to move: read bases once + 8-byte records + reference byte per column, 13 u32
  // planes written per column
  // (8 bytes per M / D / I / N item: the extra records of items that cross a tile boundary are overhead, not algorithm)
  c->pileup_bytes = c->n_bases + 8 * (int64_t)n_ops + (4 * LCR_NPLANES + 1) * c->n_cols;
  c->stage_bytes = c->n_bases + 4 * c->n_cigar + 37 * (int64_t)b.n_reads + (4 * LCR_NPLANES + 1) * c->n_cols;
  c->have_planes = true;
  c->have_cand = c->have_frag = c->have_phase = false;
  c->flt_fused = fuse; c->flt_dp = c->dp;
  return LCR_OK;
}

int lcr_get_columns(lcr_ctx* c, lcr_columns* out) {
  if (!c || !out) return LCR_E_ARG;
  if (!c->have_planes) { c->err = "lcr_get_columns before lcr_pileup"; return LCR_E_STATE; }
  const size_t bytes = (size_t)c->n_cols * LCR_NPLANES * 4;
  HIPCHK(c, c->h_planes.reserve(std::max<size_t>(bytes, 1)));
  if (bytes) HIPCHK(c, hipMemcpyAsync(c->h_planes.p, c->planes.p, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  out->n_cols = c->n_cols;
  out->planes = c->h_planes.as<uint32_t>();
  return LCR_OK;
}

static int cand_settle(lcr_ctx* c);
static int read_records_fresh(lcr_ctx* c);

int lcr_candidates(lcr_ctx* c, const lcr_params* p) {
  HT("cand");
  if (!c || !p) return LCR_E_ARG;
  if (!c->have_planes) { c->err = "lcr_candidates before lcr_pileup"; return LCR_E_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  { int rc = phase_settle(c); if (rc) return rc; }   // (the previous batch's phase stage reads the candidate / fragment buffers rewritten from here on)
  HT("cand:settled");
  c->res_valid = false;   // (its results are rewritten from here on: lcr_collect_phase had to come before this call)
  c->dp = to_dev(p, c->dp.sor_threshold);
  const int ng = c->bv.n_regions, nt = c->n_tiles;
  if (c->cand_pending && c->cand_dl_other) HIPCHK(c, hipEventSynchronize(c->ev_cand_dl));   // (a previous call's download nobody picked up: its source is rewritten below)
  HIPCHK(c, c->flags.reserve(std::max<size_t>(c->n_cols, 1)));
  HIPCHK(c, c->tile_count.reserve(std::max(nt, 1) * 4));
  HIPCHK(c, c->tile_off.reserve((std::max(nt, 1) + 1) * 4));
  HIPCHK(c, c->total.reserve(16));
  // (the tally's epilogue has taken pass 1 already when lcr_pileup ran with the same filter parameters: ONT presets, k2_eval.h)
  const DevParams &fa = c->flt_dp, &fb = c->dp;
  const bool have_flt = c->flt_fused && c->dbg_fuse_filter != 0 && fa.ont == fb.ont && fa.min_depth == fb.min_depth && fa.max_depth == fb.max_depth && fa.low_cnt_cut == fb.low_cnt_cut &&
                        fa.use_strand_bias == fb.use_strand_bias && fa.min_af_intron == fb.min_af_intron && fa.low_frac_cut == fb.low_frac_cut && fa.sor_threshold == fb.sor_threshold;
  { Timer t(c, LCR_K_CAND_FILTER);
    if (!have_flt)
    launch_k2_filter(c->bv, c->dp, c->tile_region.as<int32_t>(), c->tile_col0.as<int32_t>(), nt, c->n_cols,
                     c->planes.as<uint32_t>(), c->k0_tile_fill.as<int32_t>(), c->flags.as<uint8_t>(), c->tile_count.as<int32_t>(), c->stream);
  }
  // survivors per region = tile offsets at the regions' first tiles (gathered on the device, pinned D2H)
  HIPCHK(c, c->sv_region_off.reserve((ng + 1) * 4));
  HIPCHK(c, c->h_stage[0].reserve((ng + 2) * 4));
  int32_t* const sv_off = c->h_stage[0].as<int32_t>();
  { int32_t* d_sv = nullptr;   // (the gather writes the offsets into the pinned block as well: the wait needs no copy behind it)
    HIPCHK(c, hipHostGetDevicePointer((void**)&d_sv, sv_off, 0));
    launch_scan_i32(c->scan_tmp, c->tile_count.as<int32_t>(), c->tile_off.as<int32_t>(), nt, c->total.as<int32_t>(), c->stream);
    launch_gather_i32(c->tile_off.as<int32_t>(), c->first_tile.as<int32_t>(), ng + 1, nt, c->total.as<int32_t>(), c->sv_region_off.as<int32_t>(), c->stream, d_sv); }
  // Round 6: the survivors' compaction (and the fill of their histograms) is queued BEFORE the host knows how many there are, into buffers sized
  // by the last call's count + a quarter, and the host waits for an event in front of it: the round trip (28 us on C3) runs under that kernel
  // instead of in front of it.  More survivors than the guess (or no guess yet): the kernel dropped the rest, and runs again below.
  const int32_t cap_guess = (c->dbg_spec_compact && nt > 0) ? c->sv_cap_guess : 0;
  if (cap_guess > 0) {
    if (!c->ev_sv) HIPCHK(c, hipEventCreateWithFlags(&c->ev_sv, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ev_sv, c->stream));
    HIPCHK(c, c->survivors.reserve((size_t)cap_guess * sizeof(Survivor)));
    HIPCHK(c, c->hist.reserve((size_t)cap_guess * 124 * 4 + 64));
    HIPCHK(c, lcr_fill_async(c->hist.p, 0, (size_t)cap_guess * 124 * 4 + 64, c->stream));
    launch_k2_compact(c->bv, c->dp, c->tile_region.as<int32_t>(), c->tile_col0.as<int32_t>(), nt, c->n_cols,
                      c->planes.as<uint32_t>(), c->flags.as<uint8_t>(), c->tile_count.as<int32_t>(), c->tile_off.as<int32_t>(),
                      c->survivors.as<Survivor>(), cap_guess, c->stream);
    HIPCHK(c, hipEventSynchronize(c->ev_sv));
  } else HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  const int32_t n_sv = sv_off[ng];
  const bool compacted = cap_guess > 0 && n_sv <= cap_guess;   // (survivors and a cleared hist are in place, or on their way)
  c->sv_cap_guess = n_sv > 0 ? n_sv + n_sv / 4 + 64 : 0;
  HT("cand:n_sv");
  if (c->phase.dbg.prof) fprintf(stderr, "[cand] %d survivors of the count filters in %lld columns, %d reads\n", n_sv, (long long)c->n_cols, c->bv.n_reads);
  HIPCHK(c, c->survivors.reserve(std::max(n_sv, 1) * sizeof(Survivor)));
  HIPCHK(c, c->hist.reserve(std::max<size_t>(n_sv, 1) * 124 * 4 + 64));   // (+ the hit lists' overflow counter: cleared with the histograms)
  HIPCHK(c, c->cand_tmp.reserve(std::max<size_t>(n_sv, 1) * sizeof(lcr_candidate)));
  HIPCHK(c, c->keep.reserve(((size_t)std::max(n_sv, 1) * 3 + 2) * 4));   // keep | pos (+1) | het/hom index scratch
  HIPCHK(c, c->d_cand.reserve(std::max<size_t>(n_sv, 1) * sizeof(lcr_candidate)));   // (capacity: every survivor kept)
  HIPCHK(c, c->d_cand_off.reserve((ng + 1) * 4));
  int32_t* const d_keep = c->keep.as<int32_t>();
  int32_t* const d_pos = d_keep + std::max(n_sv, 1);
  int32_t* const d_idx = d_pos + std::max(n_sv, 1) + 1;
  c->h_cand.clear();
  c->h_cand_off.assign(ng + 1, 0);
  c->hits_valid = false;
  if (n_sv) {
    // quality histograms of the survivors: from K0's per-tile records when the survivors are dense (>= 1 per 8 columns: a second
    // pileup -- C5), else by walking the reads that cover them.  The tile form needs the ONT presets (end trim already cut out of
    // the records) and u16 counters (a survivor's depth is <= max_depth).
    const bool tiles_ok = c->dp.ont && p->max_depth <= 65535u;
    const bool hist_tiles = tiles_ok && c->dbg_hist_tiles >= 0 && (c->dbg_hist_tiles > 0 || (int64_t)n_sv * 8 >= c->n_cols);
    if (!compacted) HIPCHK(c, lcr_fill_async(c->hist.p, 0, (size_t)n_sv * 124 * 4 + 64, c->stream));
    c->hits_valid = !hist_tiles && c->dbg_k3_hits != 0;   // (the walk below leaves K3 its hits; the tile form does not walk reads)
    c->hits_n_sv = n_sv;
    if (c->hits_valid) {
      HIPCHK(c, c->hit_cnt.reserve(std::max<size_t>(c->bv.n_reads, 1) * 4));
      HIPCHK(c, c->hit_list.reserve(std::max<size_t>(c->bv.n_reads, 1) * LCR_HITS * 8));
      HIPCHK(c, c->ovf_list.reserve(std::max<size_t>(c->bv.n_reads, 1) * 4));
    }
    { Timer t(c, LCR_K_CAND_HIST);
      if (!compacted)
      launch_k2_compact(c->bv, c->dp, c->tile_region.as<int32_t>(), c->tile_col0.as<int32_t>(), nt, c->n_cols,
                        c->planes.as<uint32_t>(), c->flags.as<uint8_t>(), c->tile_count.as<int32_t>(), c->tile_off.as<int32_t>(),
                        c->survivors.as<Survivor>(), n_sv, c->stream);
      if (hist_tiles)
        launch_k2_hist_tiles(c->bv, c->tile_col0.as<int32_t>(), nt, c->tile_count.as<int32_t>(), c->tile_off.as<int32_t>(), c->survivors.as<Survivor>(),
                             c->chunk_off.as<int32_t>(), c->chunks.p, c->k0_items.as<unsigned long long>(), c->hist.as<uint32_t>(), c->stream);
      else
        launch_k2_hist(c->bv, c->dp, c->read_bin.as<ReadBin>(), c->survivors.as<Survivor>(), c->tile_off.as<int32_t>(), nt, n_sv, c->hist.as<uint32_t>(),
                       c->hits_valid ? c->hit_cnt.as<int32_t>() : nullptr, c->hit_list.p, (int32_t*)(c->hist.as<uint32_t>() + (size_t)n_sv * 124), c->ovf_list.as<int32_t>(),
                       c->stream); }
    { Timer t(c, LCR_K_CAND_GT);
      launch_k2_gt(c->dp, c->survivors.as<Survivor>(), n_sv, c->hist.as<uint32_t>(), c->bv.start0,
                   c->cand_tmp.as<lcr_candidate>(), d_keep, c->stream); }
  }
  // ordered compaction of the kept candidates + dense-cluster sweep (candidate.rs:465-526) on the device; the
  // host copy (getters, chain-region host steps) arrives with the same round trip as the offsets
  // (the kept records and their offsets leave for pinned host memory inside the last kernel, which knows the count: a copy of the records'
  // capacity on a second queue -- 3 MB on C3 -- held up the fragment stage's first kernel for 30 us)
  HIPCHK(c, c->h_stage[1].reserve(std::max<size_t>(n_sv, 1) * sizeof(lcr_candidate)));
  HIPCHK(c, c->h_stage[2].reserve((size_t)(ng + 1) * 4));
  { lcr_candidate* hp = nullptr; int32_t* ho = nullptr;
    HIPCHK(c, hipHostGetDevicePointer((void**)&hp, c->h_stage[1].p, 0));
    HIPCHK(c, hipHostGetDevicePointer((void**)&ho, c->h_stage[2].p, 0));
    if (ng == 0) c->h_stage[2].as<int32_t>()[0] = 0;
    launch_k2_finish(c->scan_tmp, c->cand_tmp.as<lcr_candidate>(), d_keep, n_sv, c->sv_region_off.as<int32_t>(), ng, d_pos, d_idx,
                     c->d_cand.as<lcr_candidate>(), c->d_cand_off.as<int32_t>(), p->dense_win, p->min_dense_cnt, c->stream, hp, ho); }
  HT("cand:finish_q");
  c->cand_dl_other = false;
  // rows of the fragment matrix per region (fragment.rs:51-54) depend on the candidates only: computed here so
  // that lcr_fragments starts without a round trip
  HIPCHK(c, c->region_rows.reserve(std::max(ng, 1) * 4));
  HIPCHK(c, c->h_stage[3].reserve(std::max(ng, 1) * 4));
  HIPCHK(c, c->row_region_off.reserve((ng + 1) * 4));
  { int32_t* d_rr = nullptr;   // (the rows per region also go straight into the pinned block: no copy in the queue)
    HIPCHK(c, hipHostGetDevicePointer((void**)&d_rr, c->h_stage[3].p, 0));
    launch_k3_rows_offsets(c->bv, c->d_cand.as<lcr_candidate>(), c->d_cand_off.as<int32_t>(), c->region_rows.as<int32_t>(), c->row_region_off.as<int32_t>(), c->stream, d_rr); }
  // no wait here: the host copies are picked up by whoever needs them first (cand_settle) -- lcr_fragments queues its
  // count pass before it does, so the GPU does not idle across the call boundary
  if (!c->ev_cand) HIPCHK(c, hipEventCreateWithFlags(&c->ev_cand, hipEventDisableTiming));
  HIPCHK(c, hipEventRecord(c->ev_cand, c->stream));
  HIPCHK(c, hipGetLastError());
  c->cand_pending = true;
  c->have_cand = true;
  HT("cand:ret");
  c->have_frag = c->have_phase = false;
  return LCR_OK;
}

int lcr_get_candidates(lcr_ctx* c, lcr_candidate_list* out) {
  if (!c || !out) return LCR_E_ARG;
  if (!c->have_cand) { c->err = "lcr_get_candidates before lcr_candidates"; return LCR_E_STATE; }
  { int rc = cand_settle(c); if (rc) return rc; }
  { int rc = phase_settle(c); if (rc) return rc; }   // (after lcr_phase the records carry its results)
  out->n_cand = (int32_t)c->h_cand.size();
  out->n_regions = c->bv.n_regions;
  out->cand = c->h_cand.data();
  out->region_off = c->h_cand_off.data();
  return LCR_OK;
}

// lcr_candidates leaves its last copies in flight: candidate records, per-region offsets, rows per region
static int cand_settle(lcr_ctx* c) {
  if (!c->cand_pending) return LCR_OK;
  HIPCHK(c, hipEventSynchronize(c->ev_cand));
  if (c->cand_dl_other) HIPCHK(c, hipEventSynchronize(c->ev_cand_dl));
  const int ng = c->bv.n_regions;
  memcpy(c->h_cand_off.data(), c->h_stage[2].p, (size_t)(ng + 1) * 4);
  c->h_cand.assign(c->h_stage[1].as<lcr_candidate>(), c->h_stage[1].as<lcr_candidate>() + c->h_cand_off[ng]);
  c->cand_pending = false;
  return LCR_OK;
}

// lcr_fragments leaves the fill pass running; the entry count arrives on the host before that pass ends
static int frag_settle(lcr_ctx* c) {
  if (!c->nnz_pending) return LCR_OK;
  HIPCHK(c, hipEventSynchronize(c->ev_nnz));
  c->nnz = c->h_nnz.as<int64_t>()[c->bv.n_regions];
  c->nnz_pending = false;
  return LCR_OK;
}

int lcr_fragments(lcr_ctx* c, const lcr_params* p) {
  HT("frag");
  if (!c || !p) return LCR_E_ARG;
  if (!c->have_cand) { c->err = "lcr_fragments before lcr_candidates"; return LCR_E_STATE; }
  if (p->min_linkers == 0) { c->err = "min_linkers must be > 0 (fragment.rs:252)"; return LCR_E_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  const int ng = c->bv.n_regions;
  c->min_linkers = p->min_linkers;
  // The count pass is queued before the host knows the row count: buffers are sized for one row per read (rows are
  // a prefix of every region's reads), counts of the unused tail stay 0, so the scan puts the entry total at
  // row_ptr[n_rows] as well as at its end.
  const int nr_cap = c->bv.n_reads;
  HIPCHK(c, c->row_cnt.reserve(std::max(nr_cap, 1) * 4));   // (row_region_off is on the device since lcr_candidates)
  HIPCHK(c, c->frag_tmp_col.reserve((size_t)std::max(nr_cap, 1) * launch_k3_inline() * 4));   // provisional entries of the count pass
  HIPCHK(c, c->frag_tmp_val.reserve((size_t)std::max(nr_cap, 1) * launch_k3_inline()));
  HIPCHK(c, c->row_links.reserve(std::max(nr_cap, 1) * 4));
  HIPCHK(c, c->row_ptr.reserve((std::max(nr_cap, 1) + 1) * 8));
  if (nr_cap) HIPCHK(c, lcr_fill_async(c->row_cnt.p, 0, (size_t)nr_cap * 4, c->stream));
  HT("frag:memset_q");
  // the count pass takes the (read, survivor) hits lcr_candidates' walk left (candidates are a subset of the survivors): no second
  // CIGAR walk; without them (dense survivors: the tile histograms) it walks the reads itself
  K3Hits hits{};
  if (c->hits_valid) {
    const int32_t* d_keep = c->keep.as<int32_t>();
    hits = K3Hits{c->hit_cnt.as<int32_t>(), c->hit_list.p, (const int32_t*)(c->hist.as<uint32_t>() + (size_t)c->hits_n_sv * 124), c->ovf_list.as<int32_t>(),
                  d_keep, d_keep + std::max(c->hits_n_sv, 1)};
  }
  { Timer t(c, LCR_K_FRAG_COUNT);
    launch_k3_count(c->bv, c->read_bin.as<ReadBin>(), c->d_cand.as<lcr_candidate>(), c->d_cand_off.as<int32_t>(), c->row_region_off.as<int32_t>(), nr_cap,
                    c->row_cnt.as<int32_t>(), c->row_links.as<uint32_t>(), c->frag_tmp_col.as<int32_t>(), c->frag_tmp_val.as<uint8_t>(), hits, c->stream);
    launch_scan_i32_to_i64(c->scan_tmp, c->row_cnt.as<int32_t>(), c->row_ptr.as<int64_t>(), nr_cap, c->stream); }
  // the regions' first entries ([ng] = all entries) follow the count pass to the host: the phase stage sizes its
  // work from them without a round trip of its own
  HIPCHK(c, c->h_nnz.reserve((size_t)(ng + 1) * 8));
  HIPCHK(c, c->region_e_off.reserve((size_t)(ng + 1) * 8));
  if (!c->ev_nnz) HIPCHK(c, hipEventCreateWithFlags(&c->ev_nnz, hipEventDisableTiming));
  { int64_t* d_nnz = nullptr;   // (straight into the pinned block: no copy in the queue in front of the fill pass)
    HIPCHK(c, hipHostGetDevicePointer((void**)&d_nnz, c->h_nnz.p, 0));
    launch_k3_region_entries(c->row_ptr.as<int64_t>(), c->row_region_off.as<int32_t>(), ng, c->region_e_off.as<int64_t>(), c->stream, d_nnz); }
  HIPCHK(c, hipEventRecord(c->ev_nnz, c->stream));
  c->nnz_pending = true;
  HT("frag:count_q");
  // now the candidates' host copies (long since there): rows per region, candidates per region
  { int rc = cand_settle(c); if (rc) return rc; }
  if (c->phase.dbg.prof && c->hits_valid) {
  HT("frag:cand_settled");
    int32_t n_ovf = 0;
    HIPCHK(c, hipMemcpy(&n_ovf, c->hist.as<uint32_t>() + (size_t)c->hits_n_sv * 124, 4, hipMemcpyDeviceToHost));
    fprintf(stderr, "[frag] %d reads with more than %d survivor hits (walked again)\n", n_ovf, LCR_HITS);
  }
  const int32_t* rr = c->h_stage[3].as<int32_t>();   // rows per region, from lcr_candidates
  c->h_row_region_off.assign(ng + 1, 0);
  for (int g = 0; g < ng; g++) c->h_row_region_off[g + 1] = c->h_row_region_off[g] + rr[g];
  c->n_rows = c->h_row_region_off[ng];
  const int nrow = c->n_rows;
  // entries: at most rows x candidates per region.  When that bound is affordable the fill pass is queued right
  // behind the count pass and the true count is picked up later (frag_settle); otherwise wait for it first.
  int64_t bound = 0;
  for (int g = 0; g < ng; g++) bound += (int64_t)rr[g] * (c->h_cand_off[g + 1] - c->h_cand_off[g]);
  int64_t cap = bound;
  if (bound > ((int64_t)1 << 28)) {
    int rc = frag_settle(c);
    if (rc) return rc;
    cap = c->nnz;
  }
  HIPCHK(c, c->col.reserve(std::max<int64_t>(cap, 1) * 4));
  HIPCHK(c, c->val.reserve(std::max<int64_t>(cap, 1)));
  { Timer t(c, LCR_K_FRAG_FILL);
    launch_k3_fill(c->bv, c->read_bin.as<ReadBin>(), c->d_cand.as<lcr_candidate>(), c->d_cand_off.as<int32_t>(), c->row_region_off.as<int32_t>(), nrow,
                   c->row_cnt.as<int32_t>(), c->row_ptr.as<int64_t>(), c->frag_tmp_col.as<int32_t>(), c->frag_tmp_val.as<uint8_t>(),
                   c->col.as<int32_t>(), c->val.as<uint8_t>(), hits, c->stream); }
  HIPCHK(c, hipGetLastError());
  c->have_frag = true;
  HT("frag:ret");
  c->have_phase = false;
  return LCR_OK;
}

int lcr_get_candidates_device(lcr_ctx* c, const lcr_candidate** dev_cand, int32_t* n_cand) {
  if (!c || !dev_cand || !n_cand) return LCR_E_ARG;
  if (!c->have_cand) { c->err = "lcr_get_candidates_device before lcr_candidates"; return LCR_E_STATE; }
  { int rc = cand_settle(c); if (rc) return rc; }
  { int rc = phase_settle(c); if (rc) return rc; }
  *dev_cand = c->d_cand.as<lcr_candidate>();
  *n_cand = (int32_t)c->h_cand.size();
  return LCR_OK;
}

int lcr_get_fragmat(lcr_ctx* c, lcr_fragmat* out) {
  if (!c || !out) return LCR_E_ARG;
  if (!c->have_frag) { c->err = "lcr_get_fragmat before lcr_fragments"; return LCR_E_STATE; }
  { int rc = frag_settle(c); if (rc) return rc; }
  const int nrow = c->n_rows, ng = c->bv.n_regions;
  const int64_t nnz = c->nnz;
  HIPCHK(c, c->h_row_ptr.reserve((nrow + 1) * 8));
  HIPCHK(c, c->h_row_read.reserve(std::max(nrow, 1) * 4));
  HIPCHK(c, c->h_col.reserve(std::max<int64_t>(nnz, 1) * 4));
  HIPCHK(c, c->h_val.reserve(std::max<int64_t>(nnz, 1)));
  HIPCHK(c, c->h_row_fp.reserve(std::max(nrow, 1)));
  HIPCHK(c, c->h_row_links.reserve(std::max(nrow, 1) * 4));
  HIPCHK(c, hipMemcpyAsync(c->h_row_ptr.p, c->row_ptr.p, (size_t)(nrow + 1) * 8, hipMemcpyDeviceToHost, c->stream));
  if (nrow) HIPCHK(c, hipMemcpyAsync(c->h_row_links.p, c->row_links.p, (size_t)nrow * 4, hipMemcpyDeviceToHost, c->stream));
  if (nnz) {
    HIPCHK(c, hipMemcpyAsync(c->h_col.p, c->col.p, (size_t)nnz * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_val.p, c->val.p, (size_t)nnz, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int32_t* rread = c->h_row_read.as<int32_t>();
  uint8_t* fp = c->h_row_fp.as<uint8_t>();
  const uint32_t* links = c->h_row_links.as<uint32_t>();
  for (int g = 0; g < ng; g++)
    for (int r = c->h_row_region_off[g]; r < c->h_row_region_off[g + 1]; r++) rread[r] = c->h_read_begin[g] + (r - c->h_row_region_off[g]);
  for (int r = 0; r < nrow; r++) fp[r] = links[r] >= c->min_linkers ? 1 : 0;
  out->n_rows = nrow; out->nnz = nnz; out->n_regions = ng;
  out->row_region_off = c->h_row_region_off.data();
  out->row_ptr = c->h_row_ptr.as<int64_t>(); out->row_read = rread; out->col = c->h_col.as<int32_t>();
  out->val = c->h_val.as<uint8_t>(); out->row_for_phasing = fp; out->row_links = links;
  return LCR_OK;
}

int lcr_phase(lcr_ctx* c, const lcr_params* p) {
  if (!c || !p) return LCR_E_ARG;
  if (!c->have_frag) { c->err = "lcr_phase before lcr_fragments"; return LCR_E_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  { int rc = frag_settle(c); if (rc) return rc; }
  PhaseInputs in;
  in.n_regions = c->bv.n_regions; in.n_rows = c->n_rows; in.nnz = c->nnz;
  in.row_region_off = c->h_row_region_off.data(); in.cand_region_off = c->h_cand_off.data();
  in.region_start0 = c->h_start0.data();
  in.region_e_off = c->h_nnz.as<int64_t>();
  in.d_row_ptr = c->row_ptr.as<int64_t>(); in.d_col = c->col.as<int32_t>(); in.d_val = c->val.as<uint8_t>();
  in.d_row_links = c->row_links.as<uint32_t>();
  in.cand = &c->h_cand;
  in.d_cand = c->d_cand.as<lcr_candidate>(); in.d_cand_off = c->d_cand_off.as<int32_t>();
  in.d_row_region_off = c->row_region_off.as<int32_t>(); in.d_start0 = c->bv.start0;
  Timer t(c, LCR_K_PHASE);
  int rc = c->phase.run(in, *p, c->stream, &c->err);
  if (rc) {   // (an error part of the way through an asynchronous stage leaves kernels queued that read the candidate / fragment buffers)
    if (c->phase.q_first) (void)hipStreamSynchronize(c->phase.q_first);
    if (c->phase.side) (void)hipStreamSynchronize(c->phase.side);
    if (c->phase.aux) (void)hipStreamSynchronize(c->phase.aux);
    return rc;
  }
  c->have_phase = true;
  c->res_valid = true; c->res_ng = c->bv.n_regions;
  c->phase_slot = c->bound_slot >= 0 ? c->bound_slot : (c->bound_host ? -2 : -1);
  return LCR_OK;
}

int lcr_discover_regions(lcr_ctx* c, int32_t mem, int32_t n_reads, const int32_t* ref_start, const int32_t* ref_end,
                         int64_t contig_len, lcr_region_list* out) {
  if (!c || !out || n_reads < 0 || contig_len < 0 || contig_len > 0x7FFFFFF0ll) return LCR_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  c->rl_start0.clear(); c->rl_len.clear(); c->rl_max.clear();
  out->n_regions = 0; out->start0 = nullptr; out->len = nullptr; out->max_cov = nullptr;
  if (n_reads == 0 || contig_len == 0) return LCR_OK;
  const int32_t *d_s = ref_start, *d_e = ref_end;
  if (mem == LCR_MEM_HOST) {
    HIPCHK(c, c->rd_start.reserve((size_t)n_reads * 4)); HIPCHK(c, c->rd_end.reserve((size_t)n_reads * 4));
    { int rc2 = upload_bytes(c, c->rd_start.p, ref_start, (size_t)n_reads * 4); if (rc2) return rc2; rc2 = upload_bytes(c, c->rd_end.p, ref_end, (size_t)n_reads * 4); if (rc2) return rc2; }
    d_s = c->rd_start.as<int32_t>(); d_e = c->rd_end.as<int32_t>();
  }
  // the window of the contig that reads cover at all (host spans: one loop here; device spans: one small reduction)
  int64_t w_lo = contig_len, w_hi = 0;
  if (mem == LCR_MEM_HOST) {
    for (int32_t r = 0; r < n_reads; r++) {
      const int64_t s = ref_start[r], e = std::min<int64_t>(ref_end[r], contig_len);
      if (s >= 0 && s < e) { w_lo = std::min(w_lo, s); w_hi = std::max(w_hi, e); }
    }
  } else {
    HIPCHK(c, c->rd_cnt.reserve(16));
    int32_t init[2] = {INT_MAX, 0}, got[2] = {INT_MAX, 0};
    HIPCHK(c, hipMemcpyAsync(c->rd_cnt.p, init, 8, hipMemcpyHostToDevice, c->stream));
    launch_k5_span_window(d_s, d_e, n_reads, contig_len, c->rd_cnt.as<int32_t>(), c->stream);
    HIPCHK(c, hipMemcpyAsync(got, c->rd_cnt.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (got[1] > 0) { w_lo = got[0]; w_hi = got[1]; }
  }
  if (w_hi <= w_lo) return LCR_OK;   // no valid span
  const int64_t full_len = contig_len;
  contig_len = w_hi - w_lo;          // from here on positions are relative to w_lo
  const size_t nd = (size_t)contig_len + 2;
  HIPCHK(c, c->rd_diff.reserve(nd * 4)); HIPCHK(c, c->rd_ex.reserve((nd + 1) * 4));
  HIPCHK(c, hipMemsetAsync(c->rd_diff.p, 0, nd * 4, c->stream));
  launch_k5_span_diff(d_s, d_e, n_reads, full_len, w_lo, c->rd_diff.as<uint32_t>(), c->stream);
  launch_scan_i32(c->scan_tmp, (const int32_t*)c->rd_diff.p, c->rd_ex.as<int32_t>(), (int32_t)nd, nullptr, c->stream);
  const int32_t nb = (int32_t)((contig_len + 1023) / 1024);
  HIPCHK(c, c->rd_cnt.reserve(((size_t)nb + 1) * 4)); HIPCHK(c, c->rd_off.reserve(((size_t)nb + 2) * 4));
  launch_k5_bounds(false, c->rd_ex.as<int32_t>(), contig_len, nb, c->rd_cnt.as<int32_t>(), nullptr, nullptr, nullptr, c->stream);
  launch_scan_i32(c->scan_tmp, c->rd_cnt.as<int32_t>(), c->rd_off.as<int32_t>(), nb, c->rd_off.as<int32_t>() + nb, c->stream);
  int32_t n_isl = 0;
  HIPCHK(c, hipMemcpyAsync(&n_isl, c->rd_off.as<int32_t>() + nb, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  if (n_isl == 0) return LCR_OK;
  HIPCHK(c, c->rd_s.reserve((size_t)n_isl * 4)); HIPCHK(c, c->rd_e.reserve((size_t)n_isl * 4)); HIPCHK(c, c->rd_max.reserve((size_t)n_isl * 4));
  launch_k5_bounds(true, c->rd_ex.as<int32_t>(), contig_len, nb, nullptr, c->rd_off.as<int32_t>(), c->rd_s.as<int32_t>(), c->rd_e.as<int32_t>(), c->stream);
  launch_k5_island_max(c->rd_ex.as<int32_t>(), c->rd_s.as<int32_t>(), c->rd_e.as<int32_t>(), n_isl, c->rd_max.as<uint32_t>(), c->stream);
  std::vector<int32_t> hs(n_isl), he(n_isl);
  std::vector<uint32_t> hm(n_isl);
  HIPCHK(c, hipMemcpyAsync(hs.data(), c->rd_s.p, (size_t)n_isl * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(he.data(), c->rd_e.p, (size_t)n_isl * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(hm.data(), c->rd_max.p, (size_t)n_isl * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  // util.rs:287-330: cursors and max_coverage are reset only when a region is emitted, and a region is
  // emitted only if region_end > region_start: a single-column island stays pending and becomes the
  // start of the region that ends with the next island (the gap between them included).
  uint32_t running = 0;
  int64_t pend = -1;
  for (int i = 0; i < n_isl; i++) {
    hs[i] += (int32_t)w_lo; he[i] += (int32_t)w_lo;   // back to contig positions
    running = std::max(running, hm[i]);
    if (pend < 0) pend = hs[i];
    if ((int64_t)he[i] > pend) {
      c->rl_start0.push_back(pend); c->rl_len.push_back((int32_t)(he[i] - pend + 1)); c->rl_max.push_back(running);
      pend = -1; running = 0;
    }
  }
  out->n_regions = (int32_t)c->rl_start0.size();
  out->start0 = c->rl_start0.data(); out->len = c->rl_len.data(); out->max_cov = c->rl_max.data();
  return LCR_OK;
}

int lcr_get_phase_result(lcr_ctx* c, lcr_phase_result* out) {
  if (!c || !out) return LCR_E_ARG;
  if (!c->have_phase) { c->err = "lcr_get_phase_result before lcr_phase"; return LCR_E_STATE; }
  { int rc = phase_settle(c); if (rc) return rc; }
  out->n_rows = c->n_rows; out->n_regions = c->bv.n_regions;
  out->haplotag = c->phase.r_haplotag; out->assignment = c->phase.r_assignment;
  out->phase_set = c->phase.r_phase_set; out->objective = c->phase.objective.data();
  return LCR_OK;
}

int lcr_get_read_records_device(lcr_ctx* c, const lcr_read_record** dev_rec, int32_t* n_rows) {
  if (!c || !dev_rec || !n_rows) return LCR_E_ARG;
  if (!c->have_phase) { c->err = "lcr_get_read_records_device before lcr_phase"; return LCR_E_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  { int rc = phase_settle(c); if (rc) return rc; }
  static_assert(sizeof(lcr_read_record) == 12, "lcr_read_record is 12 bytes");
  { int rc = read_records_fresh(c); if (rc) return rc; }
  *dev_rec = c->phase.d_read_rec.as<lcr_read_record>();
  *n_rows = c->n_rows;
  return LCR_OK;
}

static int read_records_fresh(lcr_ctx* c) {   // regions that took the host epilogue (debug hook / fallback): the HBM records are rebuilt from the host arrays
  if (!c->phase.read_rec_stale) return LCR_OK;
  std::vector<lcr_read_record> h((size_t)std::max(c->n_rows, 0));
  for (int r = 0; r < c->n_rows; r++) h[r] = lcr_read_record{r, c->phase.r_haplotag[r], c->phase.r_assignment[r], 0, c->phase.r_phase_set[r]};
  if (c->n_rows) { const int rc2 = upload_bytes(c, c->phase.d_read_rec.p, h.data(), h.size() * sizeof(lcr_read_record)); if (rc2) return rc2; HIPCHK(c, hipStreamSynchronize(c->stream)); }
  c->phase.read_rec_stale = false;
  return LCR_OK;
}

// The pipelined consumer's getter: everything the last lcr_phase produced, valid although the NEXT batch has been bound
// (lcr_load_batch / lcr_bind_batch) and its pileup queued -- none of those touch the buffers named here -- until the next lcr_candidates.
int lcr_collect_phase(lcr_ctx* c, lcr_phase_collected* out) {
  if (!c || !out) return LCR_E_ARG;
  if (!c->res_valid) { c->err = "lcr_collect_phase: no phase results (call it after lcr_phase and before the next lcr_candidates)"; return LCR_E_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  { int rc = phase_settle(c); if (rc) return rc; }
  { int rc = read_records_fresh(c); if (rc) return rc; }
  out->n_regions = c->res_ng; out->n_rows = c->n_rows; out->n_cand = (int32_t)c->h_cand.size(); out->pad_ = 0;
  out->cand = c->h_cand.data(); out->cand_region_off = c->h_cand_off.data(); out->row_region_off = c->h_row_region_off.data();
  out->haplotag = c->phase.r_haplotag; out->assignment = c->phase.r_assignment; out->phase_set = c->phase.r_phase_set;
  out->objective = c->phase.objective.data();
  out->dev_cand = c->d_cand.as<lcr_candidate>(); out->dev_read_rec = c->phase.d_read_rec.as<lcr_read_record>();
  return LCR_OK;
}

int lcr_ctx_set_async_phase(lcr_ctx* c, int on) {
  if (!c) return LCR_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  { int rc = phase_settle(c); if (rc) return rc; }
  c->phase.dbg.async_phase = on != 0;
  if (on) {   // the stage's four queues need hardware queues of their own: the runtime's default of 4 per process maps two of them onto one
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    if (!q || atoi(q) < 8) {
      c->err = "asynchronous phase stage: GPU_MAX_HW_QUEUES is unset or below 8 in this process's environment -- the HIP runtime then shares hardware queues "
               "between the stage's streams (results are the same, the overlap is lost); export GPU_MAX_HW_QUEUES=8 before the process starts";
      return LCR_W_HW_QUEUES;
    }
  }
  return LCR_OK;
}

int lcr_ctx_set_lock_dir(lcr_ctx* c, const char* dir) {
  if (!c || !dir) return LCR_E_ARG;
  c->phase.lock_dir = dir;
  return LCR_OK;
}

int lcr_debug_set(lcr_ctx* c, const char* key, int64_t value) {
  if (!c || !key) return LCR_E_ARG;
  PhaseDebug& d = c->phase.dbg;
  const std::string k(key);
  if (k == "phase_prof") d.prof = (int)value;
  else if (k == "post_host") d.post_host = (int)value;
  else if (k == "grid_min_entries") d.grid_min = value;
  else if (k == "grid_generic") d.grid_generic = (int)value;
  else if (k == "post_half") d.post_half = (int)value;
  else if (k == "enum_force_big") d.enum_force_big = (int)value;
  else if (k == "enum_force_stream") d.enum_force_stream = (int)value;
  else if (k == "host_threads") d.host_threads = (int)value;
  else if (k == "async_phase") d.async_phase = value != 0;
  else if (k == "host_trace") g_lcr_host_trace = value != 0;
  else if (k == "own_fill") g_lcr_own_fill = value != 0;
  else if (k == "fill_selftest") {   // lcr_fill_async / lcr_fill_multi_async against the host on every alignment of both ends (a test hook: returns LCR_E_DEVICE on a difference)
    HIPCHK(c, hipSetDevice(c->device));
    const size_t N = 4096;
    DevBuf buf; HIPCHK(c, buf.reserve(4 * N));
    std::vector<uint8_t> host(4 * N), want(4 * N);
    for (size_t off = 0; off < 18; off++)
      for (size_t len : {(size_t)0, (size_t)1, (size_t)15, (size_t)16, (size_t)17, (size_t)255, (size_t)1000 + off, (size_t)2049}) {
        HIPCHK(c, hipMemsetAsync(buf.p, 0xAB, 4 * N, c->stream));
        std::fill(want.begin(), want.end(), (uint8_t)0xAB);
        uint8_t* base = buf.as<uint8_t>();
        HIPCHK(c, lcr_fill_async(base + off, 0x5C, len, c->stream));
        std::fill(want.begin() + off, want.begin() + off + len, (uint8_t)0x5C);
        void* const ptrs[4] = {base + N + off, base + 2 * N + 3, nullptr, base + 3 * N + off};
        const int vals[4] = {0x80, 0, 7, 0xFF};
        const size_t sizes[4] = {len, 33, 0, len / 2};
        HIPCHK(c, lcr_fill_multi_async(4, ptrs, vals, sizes, c->stream));
        std::fill(want.begin() + N + off, want.begin() + N + off + len, (uint8_t)0x80);
        std::fill(want.begin() + 2 * N + 3, want.begin() + 2 * N + 3 + 33, (uint8_t)0);
        std::fill(want.begin() + 3 * N + off, want.begin() + 3 * N + off + len / 2, (uint8_t)0xFF);
        HIPCHK(c, hipMemcpyAsync(host.data(), buf.p, 4 * N, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (host != want) { c->err = "fill_selftest: a fill kernel wrote other bytes than asked (offset " + std::to_string(off) + ", length " + std::to_string(len) + ")"; return LCR_E_DEVICE; }
      }
  }
  else if (k == "chain_ties") d.chain_ties = value != 0;
  else if (k == "phase_prio") d.phase_prio = value != 0;
  else if (k == "no_gate") d.no_gate = value != 0;
  else if (k == "redo_lds") d.redo_lds = (int)std::max<int64_t>(0, std::min<int64_t>(value, 128 * 1024));
  else if (k == "tie_arith") d.tie_arith = (int)std::max<int64_t>(0, std::min<int64_t>(value, 3));   // (3 = the default: all four classes in the enumeration branch)
  else if (k == "timing_mask") c->timing_mask = (uint32_t)value;
  else if (k == "plane_prefill") c->dbg_prefill = value != 0;
  else if (k == "bg_tiles") c->dbg_bg_tiles = (int)std::max<int64_t>(-2, std::min<int64_t>(value, 4096));
  else if (k == "hist_tiles") c->dbg_hist_tiles = value > 0 ? 1 : value < 0 ? -1 : 0;
  else if (k == "k3_hits") c->dbg_k3_hits = value != 0;
  else if (k == "fuse_filter") c->dbg_fuse_filter = value != 0;
  else if (k == "spec_compact") c->dbg_spec_compact = value != 0;
  else if (k == "zonefix_overlap") c->dbg_zf_overlap = value != 0;
  else if (k == "zonefix_fused") c->dbg_zf_fused = value != 0;
  else if (k == "grid_spec_batch") d.spec_batch = (int)value;
  else if (k == "enum_bits") d.enum_bits = (int)value;
  else if (k == "grid_spec_lanes") d.spec_lanes = (int)std::max<int64_t>(1, std::min<int64_t>(value, 16));
  else { c->err = "lcr_debug_set: unknown key " + k; return LCR_E_ARG; }
  return LCR_OK;
}

int lcr_get_ld_blocks(lcr_ctx* c, int32_t region, int32_t* n_blocks, const int32_t** block_off, const int32_t** snp_idx) {
  if (!c || !n_blocks || !block_off || !snp_idx) return LCR_E_ARG;
  if (!c->have_phase) { c->err = "lcr_get_ld_blocks before lcr_phase"; return LCR_E_STATE; }
  if (region < 0 || region >= c->bv.n_regions) { c->err = "lcr_get_ld_blocks: no such region"; return LCR_E_ARG; }
  { int rc = phase_settle(c); if (rc) return rc; }
  PhaseInputs in;
  in.n_regions = c->bv.n_regions;
  in.cand_region_off = c->h_cand_off.data();
  const int rc = c->phase.ld_blocks(in, region, &c->ld_off, &c->ld_snps, c->stream, &c->err);
  if (rc != LCR_OK) return rc;
  *n_blocks = (int32_t)c->ld_off.size() - 1; *block_off = c->ld_off.data(); *snp_idx = c->ld_snps.data();
  return LCR_OK;
}

int lcr_get_tie_census(lcr_ctx* c, uint64_t out[8]) {
  if (!c || !out) return LCR_E_ARG;
  if (!c->have_phase) { c->err = "lcr_get_tie_census before lcr_phase"; return LCR_E_STATE; }
  { int rc = phase_settle(c); if (rc) return rc; }
  for (int i = 0; i < 8; i++) out[i] = i < TIE_NCTR ? (uint64_t)c->phase.tie_census[i] : 0;
  return LCR_OK;
}

}  // extern "C"
