// lcr_bam.cpp — SURVEY §8(f) N1: BGZF / BAM decode -> lcr_reads on the host (no GPU code in this file).
//
// Replaces the rust-htslib IndexedReader use of the reference (src/util.rs:636-691, src/fragment.rs:19-59;
// region discovery's pass over the file, util.rs:256-287): the reference inflates every region's blocks
// twice (once for the pileup, once for the fragments) and decodes records one at a time; here a contig's blocks are
// inflated in parallel, its records indexed once, and the batches handed to lcr_load_batch are cut out of that
// index (one decode shared by K1 and K3).
//
// Memory: the compressed file is mapped, not read; lcr_bam_open keeps only the BGZF block table and, per contig, the
// range of the inflated stream its records occupy (one bounded-window pass over the file: CRC check, record chain).
// The inflated bytes and the record index of ONE contig at a time are resident (loaded on demand by lcr_bam_spans /
// lcr_bam_batch / lcr_bam_write_phased, replaced when another contig is asked for); the phased-BAM writer deflates and
// writes in chunks of 16 MB.  lcr_bam_resident reports the current and the peak size of those buffers.
//
// What is kept from the reference, by line:
//   * read filter: mapq < min_mapq | l_seq < min_read_length | unmapped | secondary | supplementary, then
//     `de:f` >= divergence only if the tag exists with type f (util.rs:652-668, fragment.rs:32-49);
//   * record.pos(), cigar().leading_softclips() / trailing_softclips() (look past one hard clip),
//     seq() decoded to upper-case =ACMGRSVTWYHKDBN, qual() raw phred, the `ts:A` aux tag (util.rs:674-691);
//   * fetch((chr, start, end)) takes the 1-based region numbers as a 0-based half-open interval
//     (util.rs:637): a record is returned iff pos < end && bam_endpos > beg, with
//     bam_endpos = pos + max(reference length of the CIGAR, 1);
//   * reference_start() / reference_end() for region discovery (util.rs:281-285).
// The input must be coordinate-sorted (the reference needs a .bai, i.e. a sorted file, too); batches list a
// region's reads in file order.
#include "../../include/lcr.h"

#include <zlib.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

struct Rec {
  uint64_t off;        // first byte after block_size in the inflated stream
  uint32_t size;       // block_size
  int32_t ref_id, pos, l_seq, ref_len, lead, trail;
  uint32_t n_cig, l_rn;   // n_cig / cig_at: the real CIGAR (the CG:B,I tag's for a long-CIGAR record), n_cig_core: the core field
  uint32_t n_cig_core;
  uint64_t cig_at;        // offset of the real CIGAR's first op in the inflated stream
  uint64_t cg_tag_at;     // offset of the CG:B,I tag's first op (0: none), with cg_tag_n ops
  uint32_t cg_tag_n;
  uint16_t flag;
  uint8_t mapq, ts, has_de;
  float de;
};

inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline int32_t rdi32(const uint8_t* p) { return (int32_t)rd32(p); }

// Persistent host workers: a parallel_for per 64 MB window (open) or per output chunk (writers) would otherwise create and join
// up to hardware_concurrency threads each time (~10 us per thread).  One pool per process, grown on demand; one parallel loop
// runs on it at a time -- a second caller (another host thread with its own lcr_bam), or a forked child, falls back to threads
// of its own.
class Pool {
 public:
  // runs job() on the caller and on n_workers pool threads; false: the pool is busy (the caller spawns its own threads)
  bool run(int n_workers, const std::function<void()>& job) {
    if (getpid() != pid_) return false;   // a forked child has the object but none of its threads
    std::unique_lock<std::mutex> own(run_m_, std::try_to_lock);
    if (!own.owns_lock()) return false;
    try { while ((int)th_.size() < n_workers) th_.emplace_back([this] { worker(); }); } catch (...) { n_workers = (int)th_.size(); }
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = &job; want_ = n_workers; started_ = 0; finished_ = 0; gen_++;
    }
    cv_.notify_all();
    job();
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return finished_ == want_; });
    job_ = nullptr;
    return true;
  }

 private:
  void worker() {
    uint64_t seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(m_);
      cv_.wait(lk, [&] { return stop_ || (gen_ != seen && started_ < want_); });
      if (stop_) return;
      seen = gen_; started_++;
      const std::function<void()>* j = job_;
      lk.unlock();
      (*j)();
      lk.lock();
      if (++finished_ == want_) done_.notify_all();
    }
  }
  std::mutex run_m_, m_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> th_;
  const std::function<void()>* job_ = nullptr;
  uint64_t gen_ = 0;
  int want_ = 0, started_ = 0, finished_ = 0;
  bool stop_ = false;
  const pid_t pid_ = getpid();
};
// (never destroyed: the workers sleep on the condition variable until the process exits)
Pool& pool() { static Pool* p = new Pool; return *p; }

// n items over up to nt threads, chunked through an atomic counter; fn(i) must not throw
void parallel_for(int64_t n, int nt, int64_t chunk, const std::function<void(int64_t)>& fn) {
  if (n <= 0) return;
  nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, (n + chunk - 1) / chunk));
  if (nt == 1) { for (int64_t i = 0; i < n; i++) fn(i); return; }
  std::atomic<int64_t> next{0};
  const std::function<void()> work = [&]() {
    for (;;) {
      const int64_t b = next.fetch_add(chunk);
      if (b >= n) return;
      const int64_t e = std::min(n, b + chunk);
      for (int64_t i = b; i < e; i++) fn(i);
    }
  };
  if (pool().run(nt - 1, work)) return;
  std::vector<std::thread> th;
  for (int t = 1; t < nt; t++) th.emplace_back(work);
  work();
  for (auto& t : th) t.join();
}

// uninitialised, growable byte storage for the large arrays (the inflated contig, a batch's bases / qualities / CIGARs / names):
// std::vector::resize would zero-fill -- and first-touch -- hundreds of MB on the calling thread before the workers overwrite
// them.  2 MB-aligned with a transparent-huge-page hint from 4 MB on (fewer page faults when the workers fill it in parallel).
template <class T>
struct RawBuf {
  T* p = nullptr;
  size_t n = 0, cap = 0;
  RawBuf() = default;
  RawBuf(const RawBuf&) = delete;
  RawBuf& operator=(const RawBuf&) = delete;
  ~RawBuf() { free(p); }
  void reset() { free(p); p = nullptr; n = cap = 0; }
  bool resize(size_t want) {   // contents are NOT kept
    if (want > cap) {
      free(p); p = nullptr; cap = n = 0;
      const size_t bytes = std::max<size_t>(want * sizeof(T), 64);
      void* q = nullptr;
      if (bytes >= (4u << 20)) {
        const size_t rounded = (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
        if (posix_memalign(&q, 2u << 20, rounded) != 0) q = nullptr;
        if (q) (void)madvise(q, rounded, MADV_HUGEPAGE);
      } else q = malloc(bytes);
      if (!q) return false;
      p = static_cast<T*>(q); cap = want;
    }
    n = want;
    return true;
  }
  bool resize_keep(size_t want) {   // the first min(n, want) elements are kept
    if (want <= cap) { n = want; return true; }
    RawBuf<T> nb;
    if (!nb.resize(want + want / 4)) return false;
    if (n) memcpy(nb.p, p, n * sizeof(T));
    free(p); p = nb.p; cap = nb.cap; nb.p = nullptr;
    n = want;
    return true;
  }
  T* data() { return p; }
  const T* data() const { return p; }
  T* get() { return p; }
  const T* get() const { return p; }
  size_t size() const { return n; }
  size_t capacity() const { return cap; }
  explicit operator bool() const { return p != nullptr; }
};

}  // namespace

struct Blk { uint64_t coff; uint32_t clen, crc, isize; uint64_t uoff; };   // one BGZF block: compressed payload, its offset in the inflated stream

struct lcr_bam {
  std::string err;
  int n_threads = 1;
  // the compressed file, mapped read-only (pages are the kernel's to drop), and its block table
  const uint8_t* mm = nullptr;
  size_t mm_size = 0;
  std::vector<Blk> blks;
  uint64_t total_inflated = 0;
  std::vector<uint8_t> header;   // inflated bytes before the first record (magic, text, reference table)
  size_t header_size = 0;
  std::vector<std::string> ref_names;
  std::vector<const char*> ref_name_ptrs;
  std::vector<int64_t> ref_len;
  int64_t n_records = 0;
  // per contig (index ref_id + 1; 0 = unmapped tail): range of the inflated stream that holds its records, record count
  std::vector<uint64_t> ctg_u0, ctg_u1;
  std::vector<int64_t> ctg_n;
  // block_size of every record in stream order (4 B per record) and, per contig, its first record: a contig whose records lie
  // back to back (a sorted file) is indexed from this table without walking the block_size chain through the inflated bytes again
  std::vector<uint32_t> rec_bs;
  std::vector<int64_t> ctg_first;
  std::vector<uint8_t> ctg_scattered;
  // the ONE resident contig: inflated bytes of its blocks and its record index (Rec::off is relative to `data`)
  int32_t cur_ref = INT32_MIN;
  RawBuf<uint8_t> data;
  size_t data_size = 0;
  // a file whose inflated stream fits keep_bytes stays inflated from the open pass on (`full`): a contig is then a view into it and
  // is never inflated a second time
  uint64_t keep_bytes = 0;
  RawBuf<uint8_t> full;
  const uint8_t* view = nullptr;   // the resident contig's bytes: data.get(), or a window of `full`
  std::vector<Rec> recs;
  int64_t resident_now = 0, resident_peak = 0;
  // results of the last lcr_bam_spans / lcr_bam_batch call
  std::vector<int32_t> sp_start, sp_end;
  std::vector<int32_t> b_pos, b_seq_len, b_lead, b_trail, b_read_begin;
  std::vector<uint8_t> b_flags;
  RawBuf<uint8_t> b_bases, b_quals;
  std::vector<uint64_t> b_seq_off, b_cig_off, b_name_off;
  std::vector<uint32_t> b_n_cig;
  RawBuf<uint32_t> b_cigar;
  RawBuf<char> b_names;
  ~lcr_bam() { if (mm && mm_size) munmap(const_cast<uint8_t*>(mm), mm_size); }
};

namespace {

bool passes(const Rec& r, const lcr_read_filter& f) {   // util.rs:652-668
  if (r.mapq < f.min_mapq || r.l_seq < f.min_read_length) return false;
  if (r.flag & (0x4 | 0x100 | 0x800)) return false;
  if (r.has_de && r.de >= f.divergence) return false;
  return true;
}
inline int32_t end_pos(const Rec& r) { return r.pos + (r.ref_len > 0 ? r.ref_len : 1); }   // htslib bam_endpos

// aux block: the `de` tag of type f and the `ts` tag of type A; every other tag is skipped by its type
bool aux_scan(const uint8_t* p, const uint8_t* end, Rec& r, const uint8_t* base) {
  r.has_de = 0; r.de = 0.f; r.ts = 0; r.cg_tag_at = 0; r.cg_tag_n = 0;
  while (p + 3 <= end) {
    const uint8_t t0 = p[0], t1 = p[1], typ = p[2];
    p += 3;
    switch (typ) {
      case 'A': case 'c': case 'C':
        if (p + 1 > end) return false;
        if (typ == 'A' && t0 == 't' && t1 == 's') r.ts = p[0] == '+' ? 1 : (p[0] == '-' ? 2 : 0);
        p += 1; break;
      case 's': case 'S': p += 2; break;
      case 'i': case 'I': p += 4; break;
      case 'f':
        if (p + 4 > end) return false;
        if (t0 == 'd' && t1 == 'e') { uint32_t u = rd32(p); memcpy(&r.de, &u, 4); r.has_de = 1; }
        p += 4; break;
      case 'Z': case 'H':
        while (p < end && *p) p++;
        if (p >= end) return false;
        p++; break;
      case 'B': {
        if (p + 5 > end) return false;
        const uint8_t sub = p[0]; const uint32_t cnt = rd32(p + 1);
        size_t w = 0;
        switch (sub) { case 'c': case 'C': w = 1; break; case 's': case 'S': w = 2; break; case 'i': case 'I': case 'f': w = 4; break; default: return false; }
        if (t0 == 'C' && t1 == 'G' && (sub == 'I' || sub == 'i') && p + 5 + (size_t)cnt * 4 <= end) { r.cg_tag_at = (uint64_t)(p + 5 - base); r.cg_tag_n = cnt; }
        p += 5 + (size_t)cnt * w; break;
      }
      default: return false;
    }
    if (p > end) return false;
  }
  return true;
}

// true iff the aux block [p, end) holds a field with this two-letter tag (bam_aux_get)
bool aux_has(const uint8_t* p, const uint8_t* end, char a, char c) {
  while (p + 3 <= end) {
    if (p[0] == (uint8_t)a && p[1] == (uint8_t)c) return true;
    const uint8_t typ = p[2];
    p += 3;
    switch (typ) {
      case 'A': case 'c': case 'C': p += 1; break;
      case 's': case 'S': p += 2; break;
      case 'i': case 'I': case 'f': p += 4; break;
      case 'Z': case 'H': while (p < end && *p) p++; p++; break;
      case 'B': {
        if (p + 5 > end) return false;
        const uint8_t sub = p[0]; const uint32_t cnt = rd32(p + 1);
        const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
        p += 5 + (size_t)cnt * w; break;
      }
      default: return false;
    }
  }
  return false;
}

int fail(lcr_bam* b, int code, const std::string& msg) { b->err = msg; return code; }

#ifdef LCR_BAM_PROF   // measurement builds only (tools/build_variant.sh ... -DLCR_BAM_PROF): phase times on stderr
struct ProfT {
  const char* what; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(); double acc[8] = {0}; const char* nm[8] = {nullptr};
  explicit ProfT(const char* w) : what(w) {}
  void lap(int k, const char* name) { const auto t = std::chrono::steady_clock::now(); acc[k] += std::chrono::duration<double>(t - t0).count(); nm[k] = name; t0 = t; }
  ~ProfT() { fprintf(stderr, "[lcr_bam %s]", what); for (int k = 0; k < 8; k++) if (nm[k]) fprintf(stderr, " %s %.1f ms", nm[k], acc[k] * 1e3); fprintf(stderr, "\n"); }
};
#define PROF(name) ProfT prof_(name)
#define LAP(k, name) prof_.lap(k, name)
#else
#define PROF(name)
#define LAP(k, name)
#endif

}  // namespace

extern "C" {

}  // extern "C"

namespace {

// inflate the blocks [b0, b1) into dst (dst + (uoff - uoff of b0)); CRC-checked; returns the first bad block or -1
int64_t inflate_blocks(const lcr_bam* b, size_t b0, size_t b1, uint8_t* dst, int n_threads) {
  std::atomic<int64_t> bad{-1};
  const uint64_t base = b0 < b->blks.size() ? b->blks[b0].uoff : 0;
  parallel_for((int64_t)(b1 - b0), n_threads, 4, [&](int64_t i) {
    const Blk& k = b->blks[b0 + (size_t)i];
    if (k.isize == 0) return;   // EOF marker and other empty blocks
    // one inflate state per worker thread, reset per block (inflateInit2 allocates ~40 KB: 10^4 blocks on 256 threads would
    // spend their time in the allocator)
    struct Inflater {
      z_stream zs; bool ok;
      Inflater() { memset(&zs, 0, sizeof(zs)); ok = inflateInit2(&zs, -15) == Z_OK; }
      ~Inflater() { if (ok) inflateEnd(&zs); }
    };
    static thread_local Inflater inf;
    if (!inf.ok || inflateReset2(&inf.zs, -15) != Z_OK) { bad.store((int64_t)(b0 + i)); return; }
    z_stream& zs = inf.zs;
    zs.next_in = const_cast<Bytef*>(b->mm + k.coff); zs.avail_in = (uInt)k.clen;
    zs.next_out = dst + (k.uoff - base); zs.avail_out = k.isize;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == k.isize;
    if (!ok || crc32(crc32(0L, Z_NULL, 0), dst + (k.uoff - base), k.isize) != k.crc) bad.store((int64_t)(b0 + i));
  });
  return bad.load();
}

void note_resident(lcr_bam* b, int64_t extra) {
  b->resident_now = (int64_t)b->data_size + (int64_t)b->full.size() + (int64_t)(b->recs.capacity() * sizeof(Rec)) + (int64_t)(b->rec_bs.capacity() * 4) + extra;
  b->resident_peak = std::max(b->resident_peak, b->resident_now);
}

// decode the fixed fields of record r (r.off / r.size set) from the buffer d; false = malformed, *long_bad = placeholder without CG
bool index_record(Rec& r, const uint8_t* d, bool* long_bad) {
  const uint8_t* q = &d[r.off];
  r.ref_id = rdi32(q); r.pos = rdi32(q + 4); r.l_rn = q[8]; r.mapq = q[9];
  r.n_cig = r.n_cig_core = rd16(q + 12); r.flag = rd16(q + 14); r.l_seq = rdi32(q + 16);
  const uint64_t need = 32ull + r.l_rn + 4ull * r.n_cig + (uint64_t)((r.l_seq + 1) / 2) + (uint64_t)r.l_seq;
  if (r.l_seq < 0 || r.l_rn == 0 || need > r.size) return false;
  if (!aux_scan(q + need, q + r.size, r, d)) return false;
  const uint8_t* cg = q + 32 + r.l_rn;
  r.cig_at = r.off + 32 + r.l_rn;
  // long CIGAR (> 65535 ops; SAM spec 4.2.2, applied by htslib when it reads a record): the core field holds the
  // placeholder <l_seq>S<ref_len>N and the real CIGAR travels in the CG:B,I tag
  // htslib's bam_tag2cigar: mapped record (tid >= 0, pos >= 0) whose first op is <l_seq>S and that carries CG:B,I (or B,i);
  // a placeholder without the tag is left as it is (one soft clip + one intron: it covers nothing)
  if (r.n_cig >= 1 && r.ref_id >= 0 && r.pos >= 0 && (rd32(cg) & 15) == 4 && (int64_t)(rd32(cg) >> 4) == (int64_t)r.l_seq && r.cg_tag_n) {
    cg = d + r.cg_tag_at; r.cig_at = r.cg_tag_at; r.n_cig = r.cg_tag_n;
  }
  (void)long_bad;
  int64_t rl = 0;
  for (uint32_t k = 0; k < r.n_cig; k++) {
    const uint32_t w = rd32(cg + 4 * k), op = w & 15;
    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += w >> 4;   // M D N = X consume the reference
  }
  r.ref_len = (int32_t)rl;
  r.lead = r.trail = 0;
  if (r.n_cig) {   // leading / trailing soft clips, looking past one hard clip
    const uint32_t w0 = rd32(cg), wl = rd32(cg + 4 * (r.n_cig - 1));
    if ((w0 & 15) == 4) r.lead = (int32_t)(w0 >> 4);
    else if ((w0 & 15) == 5 && r.n_cig > 1 && (rd32(cg + 4) & 15) == 4) r.lead = (int32_t)(rd32(cg + 4) >> 4);
    if ((wl & 15) == 4) r.trail = (int32_t)(wl >> 4);
    else if ((wl & 15) == 5 && r.n_cig > 1 && (rd32(cg + 4 * (r.n_cig - 2)) & 15) == 4) r.trail = (int32_t)(rd32(cg + 4 * (r.n_cig - 2)) >> 4);
  }
  return true;
}

// make contig ref_id (-1: the unmapped tail) the resident one: inflate its blocks, index its records
int load_contig(lcr_bam* b, int32_t ref_id) {
  if (b->cur_ref == ref_id) return LCR_OK;
  b->cur_ref = INT32_MIN;
  b->recs.clear(); b->data.reset(); b->data_size = 0; b->view = nullptr;
  const size_t ci = (size_t)((int64_t)ref_id + 1);
  if (ref_id < -1 || ci >= b->ctg_n.size() || b->ctg_n[ci] == 0) { b->cur_ref = ref_id; note_resident(b, 0); return LCR_OK; }   // no records
  const uint64_t u0 = b->ctg_u0[ci], u1 = b->ctg_u1[ci];
  // blocks that cover [u0, u1)
  size_t b0 = (size_t)(std::upper_bound(b->blks.begin(), b->blks.end(), u0, [](uint64_t u, const Blk& k) { return u < k.uoff + k.isize; }) - b->blks.begin());
  size_t b1 = (size_t)(std::lower_bound(b->blks.begin(), b->blks.end(), u1, [](const Blk& k, uint64_t u) { return k.uoff < u; }) - b->blks.begin());
  if (b0 >= b1) return fail(b, LCR_E_ARG, "inconsistent contig range");
  const uint64_t base = b->blks[b0].uoff, bytes = b->blks[b1 - 1].uoff + b->blks[b1 - 1].isize - base;
  PROF("load_contig");
  if (b->full) b->view = b->full.get() + base;   // (inflated and CRC-checked by the open pass)
  else {
    if (!b->data.resize((size_t)bytes + 1)) return fail(b, LCR_E_NOMEM, "out of memory for the inflated contig");
    LAP(0, "alloc");
    b->data_size = (size_t)bytes;
    if (inflate_blocks(b, b0, b1, b->data.get(), b->n_threads) >= 0) return fail(b, LCR_E_ARG, "BGZF block does not inflate / CRC mismatch");
    b->view = b->data.get();
  }
  LAP(1, "inflate");
  const uint8_t* d = b->view;
  try { b->recs.reserve((size_t)b->ctg_n[ci]); } catch (...) { return fail(b, LCR_E_NOMEM, "out of memory for the record index"); }
  if (!b->ctg_scattered[ci]) {   // sorted file: the contig's records lie back to back, their sizes are known from the open pass
    size_t p = (size_t)(u0 - base);
    const uint32_t* bsz = b->rec_bs.data() + b->ctg_first[ci];
    b->recs.resize((size_t)b->ctg_n[ci]);
    for (int64_t k = 0; k < b->ctg_n[ci]; k++) { Rec r{}; r.off = p + 4; r.size = bsz[k]; b->recs[(size_t)k] = r; p += 4 + (size_t)bsz[k]; }
    if (p != (size_t)(u1 - base)) return fail(b, LCR_E_ARG, "inconsistent record table");
  } else {
    for (size_t p = (size_t)(u0 - base), e = (size_t)(u1 - base); p < e;) {
      const uint32_t bs = rd32(&d[p]);
      if (bs < 32 || p + 4 + (size_t)bs > e) return fail(b, LCR_E_ARG, "truncated record");
      if (rdi32(&d[p + 4]) == ref_id) { Rec r{}; r.off = p + 4; r.size = bs; b->recs.push_back(r); }   // (an unsorted file interleaves contigs)
      p += 4 + (size_t)bs;
    }
  }
  LAP(2, "chain");
  std::atomic<int64_t> bad_rec{-1}, long_cigar_bad{-1};
  parallel_for((int64_t)b->recs.size(), b->n_threads, 1024, [&](int64_t i) {
    bool lb = false;
    if (!index_record(b->recs[(size_t)i], d, &lb)) { if (lb) long_cigar_bad.store(i); else bad_rec.store(i); }
  });
  if (bad_rec.load() >= 0) return fail(b, LCR_E_ARG, "malformed record " + std::to_string(bad_rec.load()) + " of contig " + std::to_string(ref_id));
  if (long_cigar_bad.load() >= 0)
    return fail(b, LCR_E_ARG, "record " + std::to_string(long_cigar_bad.load()) + " has the long-CIGAR placeholder (<l_seq>S<n>N) but no CG:B,I tag");
  LAP(3, "index");
  b->cur_ref = ref_id;
  note_resident(b, 0);
  return LCR_OK;
}

}  // namespace

extern "C" {

int lcr_bam_open(const char* path, int32_t n_threads, lcr_bam** out) { return lcr_bam_open_keep(path, n_threads, (int64_t)4 << 30, out); }

int lcr_bam_open_keep(const char* path, int32_t n_threads, int64_t keep_bytes, lcr_bam** out) {
  if (!path || !out) return LCR_E_ARG;
  *out = nullptr;
  lcr_bam* b = new (std::nothrow) lcr_bam();
  if (!b) return LCR_E_NOMEM;
  *out = b;   // returned even on failure so that lcr_bam_last_error can explain; the caller closes it
  if (n_threads < 1) n_threads = (int32_t)std::max(1u, std::thread::hardware_concurrency());
  b->n_threads = n_threads;
  b->keep_bytes = keep_bytes > 0 ? (uint64_t)keep_bytes : 0;
  // ---- the compressed file: mapped, not read
  {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(b, LCR_E_ARG, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 0) { close(fd); return fail(b, LCR_E_ARG, "stat failed"); }
    b->mm_size = (size_t)st.st_size;
    if (b->mm_size) {
      void* m = mmap(nullptr, b->mm_size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m == MAP_FAILED) { close(fd); b->mm_size = 0; return fail(b, LCR_E_ARG, std::string("cannot map ") + path); }
      b->mm = static_cast<const uint8_t*>(m);
    }
    close(fd);
  }
  const uint8_t* raw = b->mm;
  const size_t raw_size = b->mm_size;
  // ---- BGZF block table (gzip member with the BC extra subfield; SAM spec 4.1)
  std::vector<Blk>& blks = b->blks;
  uint64_t total = 0;
  for (size_t off = 0; off < raw_size;) {
    if (off + 18 > raw_size || raw[off] != 0x1f || raw[off + 1] != 0x8b || raw[off + 2] != 8 || !(raw[off + 3] & 4))
      return fail(b, LCR_E_ARG, "not a BGZF block at offset " + std::to_string(off));
    const size_t xlen = rd16(&raw[off + 10]);
    if (off + 12 + xlen > raw_size) return fail(b, LCR_E_ARG, "truncated BGZF header");
    int64_t bsize = -1;
    for (size_t p = off + 12; p + 4 <= off + 12 + xlen;) {
      const size_t slen = rd16(&raw[p + 2]);
      if (raw[p] == 66 && raw[p + 1] == 67 && slen == 2 && p + 6 <= off + 12 + xlen) bsize = rd16(&raw[p + 4]);
      p += 4 + slen;
    }
    if (bsize < 0) return fail(b, LCR_E_ARG, "BGZF block without BC subfield at offset " + std::to_string(off));
    const size_t blen = (size_t)bsize + 1;
    if (blen < 12 + xlen + 8 || off + blen > raw_size) return fail(b, LCR_E_ARG, "truncated BGZF block at offset " + std::to_string(off));
    Blk k;
    k.coff = off + 12 + xlen; k.clen = (uint32_t)(blen - (12 + xlen) - 8);
    k.crc = rd32(&raw[off + blen - 8]); k.isize = rd32(&raw[off + blen - 4]);
    if (k.isize > 65536) return fail(b, LCR_E_ARG, "BGZF block larger than 64 KiB");
    k.uoff = total; total += k.isize;
    blks.push_back(k);
    off += blen;
  }
  b->total_inflated = total;
  // ---- one pass over the inflated stream in windows of <= 1024 blocks (64 MiB): CRC check of every block, BAM header,
  //      the block_size chain of the records -> per contig the range of the stream it occupies and its record count
  PROF("open");
  const bool keep_all = total > 0 && total <= b->keep_bytes;   // one window = the whole stream, kept
  const size_t WIN = keep_all ? std::max<size_t>(blks.size(), 1) : 1024;
  RawBuf<uint8_t> buf;            // [carry of the previous window | this window]
  size_t carry = 0;               // bytes at the front of buf that belong to an unfinished item
  std::vector<std::pair<uint64_t, uint32_t>> win_recs;   // (offset past block_size, block_size) of the complete records of the window
  uint64_t buf_u0 = 0;            // inflated offset of buf[0]
  bool have_header = false;
  size_t hp = 0;                  // parse position inside buf
  // parse_header: the BAM header (SAM spec 4.2) from d[0 .. n) -- may span windows, is parsed when it is complete (*done); p = first record
  auto parse_header = [&](const uint8_t* d, size_t n, size_t& p, bool* done) -> int {
    *done = false;
    if (n < 12) return LCR_OK;
    if (memcmp(d, "BAM\1", 4) != 0) return fail(b, LCR_E_ARG, "not a BAM file");
    size_t q = 8 + (size_t)rd32(&d[4]);
    if (q + 4 > n) return LCR_OK;
    const int32_t n_ref = rdi32(&d[q]);
    q += 4;
    if (n_ref < 0) return fail(b, LCR_E_ARG, "bad reference count");
    std::vector<std::string> names; std::vector<int64_t> lens;
    for (int32_t i = 0; i < n_ref; i++) {
      if (q + 4 > n) return LCR_OK;
      const uint32_t l_name = rd32(&d[q]);
      if (l_name == 0) return fail(b, LCR_E_ARG, "truncated reference table");
      if (q + 4 + (size_t)l_name + 4 > n) return LCR_OK;
      names.emplace_back(reinterpret_cast<const char*>(&d[q + 4]), l_name - 1);
      lens.push_back(rdi32(&d[q + 4 + l_name]));
      q += 8 + l_name;
    }
    b->ref_names = names; b->ref_len = lens;
    b->header_size = q; b->header.assign(d, d + q);
    p = q;
    for (auto& s : b->ref_names) b->ref_name_ptrs.push_back(s.c_str());
    b->ctg_u0.assign(b->ref_names.size() + 1, UINT64_MAX); b->ctg_u1.assign(b->ref_names.size() + 1, 0); b->ctg_n.assign(b->ref_names.size() + 1, 0);
    b->ctg_first.assign(b->ref_names.size() + 1, -1); b->ctg_scattered.assign(b->ref_names.size() + 1, 0);
    *done = true;
    return LCR_OK;
  };
  // walk: the block_size chain of the complete records in d[p .. n) (d[0] = inflated offset u0): contig ranges, record sizes
  auto walk = [&](const uint8_t* d, size_t n, bool last, uint64_t u0, size_t& p, std::vector<std::pair<uint64_t, uint32_t>>& out) -> int {
    while (p < n) {
      if (p + 4 > n) { if (last) return fail(b, LCR_E_ARG, "truncated record header"); break; }
      const uint32_t bs = rd32(&d[p]);
      if (bs < 32) return fail(b, LCR_E_ARG, "truncated record at inflated offset " + std::to_string(u0 + p));
      if (p + 4 + (size_t)bs > n) { if (last) return fail(b, LCR_E_ARG, "truncated record at inflated offset " + std::to_string(u0 + p)); break; }
      const int32_t rid = rdi32(&d[p + 4]);
      const int64_t ci = (int64_t)rid + 1;
      if (ci < 0 || (size_t)ci >= b->ctg_n.size()) return fail(b, LCR_E_ARG, "malformed record " + std::to_string(b->n_records) + ": reference id out of range");
      b->ctg_u0[(size_t)ci] = std::min(b->ctg_u0[(size_t)ci], u0 + p);
      b->ctg_u1[(size_t)ci] = std::max(b->ctg_u1[(size_t)ci], u0 + p + 4 + bs);
      if (b->ctg_n[(size_t)ci] == 0) b->ctg_first[(size_t)ci] = b->n_records;
      else if (b->ctg_first[(size_t)ci] + b->ctg_n[(size_t)ci] != b->n_records) b->ctg_scattered[(size_t)ci] = 1;
      try { b->rec_bs.push_back(bs); out.push_back(std::make_pair((uint64_t)(p + 4), bs)); } catch (...) { return fail(b, LCR_E_NOMEM, "out of memory for the record table"); }
      b->ctg_n[(size_t)ci]++; b->n_records++;
      p += 4 + (size_t)bs;
    }
    return LCR_OK;
  };
  // the fixed fields / aux block of the records in `recs` (offsets into d) validated in parallel, so that a malformed file is
  // refused at open without a second inflate; first_index = file index of recs[0]
  auto validate = [&](const uint8_t* d, const std::vector<std::pair<uint64_t, uint32_t>>& recs, int64_t first_index) -> int {
    std::atomic<int64_t> bad_rec{-1};
    parallel_for((int64_t)recs.size(), n_threads, 1024, [&](int64_t i) {
      Rec r{}; r.off = recs[(size_t)i].first; r.size = recs[(size_t)i].second;
      bool lb = false;
      if (!index_record(r, d, &lb)) bad_rec.store(i);
    });
    if (bad_rec.load() >= 0) return fail(b, LCR_E_ARG, "malformed record " + std::to_string(first_index + bad_rec.load()) + " (fixed fields / aux block)");
    return LCR_OK;
  };
  if (keep_all) {
    // The whole stream stays: windows of 1 024 blocks are inflated to their final place, and while the workers inflate window
    // i + 1 ONE thread walks the record chain of window i (serial pointer chasing through freshly written memory: ~190 ns per
    // record, as long as the inflate itself); the records are validated in one parallel pass at the end.
    if (!buf.resize((size_t)total + 1)) return fail(b, LCR_E_NOMEM, "out of memory for the inflated stream");
    LAP(1, "resize");
    b->resident_peak = std::max(b->resident_peak, (int64_t)buf.size());
    const uint8_t* d = buf.data();
    size_t p = 0;
    int chain_rc = LCR_OK;
    std::thread chain;
    const size_t KWIN = 1024;
    for (size_t w0 = 0; w0 < blks.size(); w0 += KWIN) {
      const size_t w1 = std::min(blks.size(), w0 + KWIN);
      const int64_t badb = inflate_blocks(b, w0, w1, buf.data() + blks[w0].uoff, n_threads);
      if (chain.joinable()) chain.join();
      if (badb >= 0) return fail(b, LCR_E_ARG, "BGZF block " + std::to_string(badb) + " does not inflate / CRC mismatch");
      if (chain_rc != LCR_OK) return chain_rc;
      const size_t n = (size_t)(blks[w1 - 1].uoff + blks[w1 - 1].isize);
      const bool last = w1 >= blks.size();
      auto chain_step = [&, n, last]() {
        //  this runs on a thread of its own: an exception here (bad_alloc / length_error from a hostile n_ref or l_name
        // in the header, or from the record index) must become an error code, not std::terminate
        try {
          if (!have_header) {
            bool done = false;
            chain_rc = parse_header(d, n, p, &done);
            if (chain_rc != LCR_OK) return;
            if (!done) { if (last) chain_rc = fail(b, LCR_E_ARG, n < 12 || memcmp(d, "BAM\1", 4) != 0 ? "not a BAM file" : "truncated BAM header"); return; }
            have_header = true;
          }
          chain_rc = walk(d, n, last, 0, p, win_recs);
        } catch (const std::bad_alloc&) { chain_rc = fail(b, LCR_E_NOMEM, "out of memory while indexing the BAM records");
        } catch (const std::exception& e) { chain_rc = fail(b, LCR_E_ARG, std::string("malformed BAM: ") + e.what()); }
      };
      try { chain = std::thread(chain_step); } catch (...) { chain_step(); }   // (no thread to be had: walk here)
    }
    if (chain.joinable()) chain.join();
    if (chain_rc != LCR_OK) return chain_rc;
    LAP(2, "inflate + chain");
    if (!have_header) return fail(b, LCR_E_ARG, "not a BAM file");
    { const int rc = validate(d, win_recs, 0); if (rc != LCR_OK) return rc; }
    LAP(4, "validate");
    carry = total - p; buf_u0 = p;
  } else
  for (size_t w0 = 0; w0 < blks.size() || !have_header; w0 += WIN) {
    const size_t w1 = std::min(blks.size(), w0 + WIN);
    const size_t wbytes = w0 < blks.size() ? (size_t)(blks[w1 - 1].uoff + blks[w1 - 1].isize - blks[w0].uoff) : 0;
    LAP(0, "other");
    if (!buf.resize_keep(carry + wbytes + 1)) return fail(b, LCR_E_NOMEM, "out of memory for the scan window");
    LAP(1, "resize");
    b->resident_peak = std::max(b->resident_peak, (int64_t)buf.size());
    if (wbytes) {
      const int64_t badb = inflate_blocks(b, w0, w1, buf.data() + carry, n_threads);
      if (badb >= 0) return fail(b, LCR_E_ARG, "BGZF block " + std::to_string(badb) + " does not inflate / CRC mismatch");
    }
    LAP(2, "inflate");
    const size_t n = carry + wbytes;
    const uint8_t* d = buf.data();
    const bool last = w1 >= blks.size();
    size_t p = hp;
    if (!have_header) {
      bool done = false;
      { const int rc = parse_header(d, n, p, &done); if (rc != LCR_OK) return rc; }
      if (!done) {
        if (last) return fail(b, LCR_E_ARG, n < 12 || memcmp(d, "BAM\1", 4) != 0 ? "not a BAM file" : "truncated BAM header");
        carry = n; hp = 0;   // keep everything, read on
        continue;
      }
      have_header = true;
    }
    // ---- records of this window
    win_recs.clear();
    const int64_t first_index = b->n_records;
    { const int rc = walk(d, n, last, buf_u0, p, win_recs); if (rc != LCR_OK) return rc; }
    LAP(3, "chain");
    { const int rc = validate(d, win_recs, first_index); if (rc != LCR_OK) return rc; }
    LAP(4, "validate");
    // the unfinished tail moves to the front of the next window
    carry = n - p;
    if (carry) memmove(buf.data(), buf.data() + p, carry);
    buf_u0 += p; hp = 0;
    if (last) break;
  }
  // (contigs are indexed on demand, load_contig; a small file stays inflated, of a large one nothing is resident after open)
  if (keep_all) {
    // the single window held the whole stream from offset 0 (the last iteration moved nothing: carry is the unparsed tail, 0)
    if (buf_u0 + carry != total) return fail(b, LCR_E_ARG, "inconsistent stream length");
    // buf_u0 == parse position of the end: the buffer still holds the stream from its first byte
    b->full.p = buf.p; b->full.n = (size_t)total; b->full.cap = buf.cap; buf.p = nullptr; buf.n = buf.cap = 0;
    note_resident(b, 0);
  }
  return LCR_OK;
}

void lcr_bam_close(lcr_bam* b) { delete b; }

const char* lcr_bam_last_error(const lcr_bam* b) { return b ? b->err.c_str() : "null handle"; }

int lcr_bam_refs(lcr_bam* b, int32_t* n_ref, const char* const** names, const int64_t** lengths) {
  if (!b || !n_ref || !names || !lengths) return LCR_E_ARG;
  *n_ref = (int32_t)b->ref_names.size();
  *names = b->ref_name_ptrs.data();
  *lengths = b->ref_len.data();
  return LCR_OK;
}

int lcr_bam_n_records(lcr_bam* b, int64_t* n) {
  if (!b || !n) return LCR_E_ARG;
  *n = b->n_records;
  return LCR_OK;
}

int lcr_bam_resident(lcr_bam* b, int64_t* now, int64_t* peak) {
  if (!b || !now || !peak) return LCR_E_ARG;
  *now = b->resident_now; *peak = b->resident_peak;
  return LCR_OK;
}

int lcr_bam_spans(lcr_bam* b, int32_t ref_id, const lcr_read_filter* f, int32_t* n, const int32_t** ref_start, const int32_t** ref_end) {
  if (!b || !f || !n || !ref_start || !ref_end) return LCR_E_ARG;
  { const int rc = load_contig(b, ref_id); if (rc != LCR_OK) return rc; }
  b->sp_start.clear(); b->sp_end.clear();
  for (const Rec& r : b->recs)
    if (r.ref_id == ref_id && passes(r, *f)) { b->sp_start.push_back(r.pos); b->sp_end.push_back(end_pos(r)); }
  *n = (int32_t)b->sp_start.size();
  *ref_start = b->sp_start.data(); *ref_end = b->sp_end.data();
  return LCR_OK;
}

int lcr_bam_batch(lcr_bam* b, int32_t ref_id, const lcr_read_filter* f, int32_t n_regions, const int64_t* start0, const int32_t* len,
                  lcr_reads* reads, const int32_t** read_begin, const uint64_t** name_off, const char** names) {
  if (!b || !f || !reads || !read_begin || n_regions < 0 || (n_regions && (!start0 || !len))) return LCR_E_ARG;
  { const int rc = load_contig(b, ref_id); if (rc != LCR_OK) return rc; }
  // passing records of the contig, by position (file order for a sorted file), with a running maximum of their ends
  std::vector<uint32_t> idx;
  for (size_t i = 0; i < b->recs.size(); i++)
    if (b->recs[i].ref_id == ref_id && passes(b->recs[i], *f)) idx.push_back((uint32_t)i);
  for (size_t i = 1; i < idx.size(); i++)
    if (b->recs[idx[i]].pos < b->recs[idx[i - 1]].pos) return fail(b, LCR_E_ARG, "BAM is not coordinate-sorted");
  std::vector<int32_t> ipos(idx.size()), iend_max(idx.size());
  int32_t run = INT32_MIN;
  for (size_t i = 0; i < idx.size(); i++) {
    ipos[i] = b->recs[idx[i]].pos;
    run = std::max(run, end_pos(b->recs[idx[i]]));
    iend_max[i] = run;
  }
  // fetch rule per region: pos < end && end_pos > beg over [beg, end) = [start0 + 1, start0 + len + 1)  (util.rs:637)
  std::vector<uint32_t> take;
  b->b_read_begin.assign((size_t)n_regions + 1, 0);
  for (int32_t g = 0; g < n_regions; g++) {
    if (len[g] < 0) return fail(b, LCR_E_ARG, "negative region length");
    const int64_t beg = start0[g] + 1, end = start0[g] + len[g] + 1;
    const size_t hi = (size_t)(std::lower_bound(ipos.begin(), ipos.end(), end, [](int32_t v, int64_t e) { return (int64_t)v < e; }) - ipos.begin());
    size_t lo = (size_t)(std::upper_bound(iend_max.begin(), iend_max.end(), beg, [](int64_t bg, int32_t v) { return bg < (int64_t)v; }) - iend_max.begin());
    for (; lo < hi; lo++)
      if ((int64_t)end_pos(b->recs[idx[lo]]) > beg) take.push_back(idx[lo]);
    if (take.size() > (size_t)INT32_MAX) return fail(b, LCR_E_ARG, "more than 2^31 reads in one batch");
    b->b_read_begin[(size_t)g + 1] = (int32_t)take.size();
  }
  const size_t nr = take.size();
  b->b_pos.resize(nr); b->b_seq_len.resize(nr); b->b_lead.resize(nr); b->b_trail.resize(nr); b->b_flags.resize(nr);
  b->b_seq_off.resize(nr); b->b_cig_off.resize(nr); b->b_n_cig.resize(nr); b->b_name_off.resize(nr + 1);
  uint64_t so = 0, co = 0, no = 0;
  for (size_t k = 0; k < nr; k++) {
    const Rec& r = b->recs[take[k]];
    b->b_seq_off[k] = so; b->b_cig_off[k] = co; b->b_name_off[k] = no;
    so += (uint64_t)r.l_seq; co += r.n_cig; no += r.l_rn;   // names keep their NUL
  }
  b->b_name_off[nr] = no;
  if (!b->b_bases.resize(so) || !b->b_quals.resize(so) || !b->b_cigar.resize(co) || !b->b_names.resize(no)) return fail(b, LCR_E_NOMEM, "out of memory for the batch");
  static const char NT16[] = "=ACMGRSVTWYHKDBN";
  const uint8_t* d = b->view;
  parallel_for((int64_t)nr, b->n_threads, 256, [&](int64_t k) {
    const Rec& r = b->recs[take[(size_t)k]];
    const uint8_t* q = d + r.off;
    b->b_pos[k] = r.pos; b->b_seq_len[k] = r.l_seq; b->b_lead[k] = r.lead; b->b_trail[k] = r.trail;
    b->b_flags[k] = (uint8_t)(((r.flag & 0x10) ? 1 : 0) | (r.ts << 1));
    b->b_n_cig[k] = r.n_cig;
    memcpy(b->b_names.data() + b->b_name_off[k], q + 32, r.l_rn);
    const uint8_t* cg = d + r.cig_at;
    uint32_t* co_ = b->b_cigar.data() + b->b_cig_off[k];
    for (uint32_t c = 0; c < r.n_cig; c++) co_[c] = rd32(cg + 4 * c);
    const uint8_t* sq = q + 32 + r.l_rn + 4 * (size_t)r.n_cig_core;
    uint8_t* bo = b->b_bases.data() + b->b_seq_off[k];
    for (int32_t i = 0; i + 1 < r.l_seq; i += 2) { const uint8_t v = sq[i >> 1]; bo[i] = (uint8_t)NT16[v >> 4]; bo[i + 1] = (uint8_t)NT16[v & 15]; }
    if (r.l_seq & 1) bo[r.l_seq - 1] = (uint8_t)NT16[sq[r.l_seq >> 1] >> 4];
    memcpy(b->b_quals.data() + b->b_seq_off[k], sq + (r.l_seq + 1) / 2, (size_t)r.l_seq);
  });
  memset(reads, 0, sizeof(*reads));
  reads->mem = LCR_MEM_HOST;
  reads->n_reads = (int32_t)nr; reads->n_bases = (int64_t)so; reads->n_cigar = (int64_t)co;
  reads->pos = b->b_pos.data(); reads->seq_len = b->b_seq_len.data(); reads->lead_clip = b->b_lead.data(); reads->trail_clip = b->b_trail.data();
  reads->flags = b->b_flags.data(); reads->seq_off = b->b_seq_off.data(); reads->cig_off = b->b_cig_off.data(); reads->n_cig = b->b_n_cig.data();
  reads->bases = b->b_bases.data(); reads->quals = b->b_quals.data(); reads->cigar = b->b_cigar.data();
  *read_begin = b->b_read_begin.data();
  if (name_off) *name_off = b->b_name_off.data();
  if (names) *names = b->b_names.data();
  return LCR_OK;
}


// ---- phased BAM (thread.rs:307-361) ---------------------------------------------------------
int lcr_bam_write_phased(lcr_bam* b, const char* out_path, int32_t n_regions, const int32_t* region_ref, const int64_t* start0,
                         const int32_t* len, int64_t n_tagged, const uint64_t* name_off, const char* names, const int32_t* hp,
                         const uint32_t* ps, int32_t level, int32_t n_threads) {
  if (!b || !out_path || n_regions < 0 || n_tagged < 0 || (n_regions && (!region_ref || !start0 || !len)) ||
      (n_tagged && (!name_off || !names || !hp || !ps)) || level < -1 || level > 9)
    return LCR_E_ARG;
  if (n_threads < 1) n_threads = b->n_threads;
  // qname -> HP / PS, first entry wins (thread.rs:308-325: the queues are drained in order; a read of two regions
  // keeps what its first region said).  hp < 0: no assignment entry; ps == 0: no phase-set entry.
  std::unordered_map<std::string, int32_t> m_hp;
  std::unordered_map<std::string, uint32_t> m_ps;
  for (int64_t i = 0; i < n_tagged; i++) {
    const std::string nm(names + name_off[i]);
    if (hp[i] >= 0) m_hp.emplace(nm, hp[i]);
    if (ps[i] != 0) m_ps.emplace(nm, ps[i]);
  }
  // ---- output: BGZF blocks of 0xff00 inflated bytes (htslib's BGZF_BLOCK_SIZE), deflated in parallel and written in
  //      chunks of 256 blocks; the payload is the header as in the input, then the records with HP:i (int32) / PS:I
  //      (uint32) appended to their aux blocks
  FILE* f = fopen(out_path, "wb");
  if (!f) return fail(b, LCR_E_ARG, std::string("cannot create ") + out_path);
  const uint64_t BLK = 0xff00;
  const size_t CHUNK_BLOCKS = 256;
  std::vector<uint8_t> pend;   // inflated bytes not written yet
  pend.reserve(std::max<size_t>(b->header.size(), CHUNK_BLOCKS * (size_t)BLK) + (1u << 20));
  pend.assign(b->header.begin(), b->header.end());
  std::vector<std::vector<uint8_t>> comp;
  bool io_ok = true;
  auto flush = [&](bool final_) -> bool {   // writes the complete blocks of `pend` (all of it, and the EOF block, at the end)
    const size_t nfull = pend.size() / BLK;
    const size_t nblk = final_ ? (pend.size() + BLK - 1) / BLK : nfull;
    const size_t n_out = nblk + (final_ ? 1 : 0);
    if (n_out == 0) return true;
    comp.assign(n_out, {});
    std::atomic<int> bad{0};
    parallel_for((int64_t)n_out, n_threads, 4, [&](int64_t i) {
      const uint64_t off = (uint64_t)i * BLK;
      const uint32_t n = (size_t)i >= nblk ? 0u : (uint32_t)std::min<uint64_t>(BLK, pend.size() - off);
      std::vector<uint8_t>& c = comp[(size_t)i];
      c.resize(18 + compressBound(n) + 8);
      z_stream zs;
      memset(&zs, 0, sizeof(zs));
      if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad.store(1); return; }
      zs.next_in = n ? pend.data() + off : nullptr; zs.avail_in = n;
      zs.next_out = c.data() + 18; zs.avail_out = (uInt)(c.size() - 18 - 8);
      const int rc = deflate(&zs, Z_FINISH);
      const size_t clen = zs.total_out;
      deflateEnd(&zs);
      if (rc != Z_STREAM_END || 18 + clen + 8 > 65536) { bad.store(1); return; }
      static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
      memcpy(c.data(), head, 16);
      const uint32_t bsize = (uint32_t)(18 + clen + 8 - 1);
      c[16] = (uint8_t)bsize; c[17] = (uint8_t)(bsize >> 8);
      const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), n ? pend.data() + off : nullptr, n);
      uint8_t* t = c.data() + 18 + clen;
      t[0] = (uint8_t)crc; t[1] = (uint8_t)(crc >> 8); t[2] = (uint8_t)(crc >> 16); t[3] = (uint8_t)(crc >> 24);
      t[4] = (uint8_t)n; t[5] = (uint8_t)(n >> 8); t[6] = (uint8_t)(n >> 16); t[7] = (uint8_t)(n >> 24);
      c.resize(18 + clen + 8);
    });
    if (bad.load()) return false;
    for (auto& c : comp) io_ok = io_ok && fwrite(c.data(), 1, c.size(), f) == c.size();
    const size_t done = std::min(pend.size(), nblk * (size_t)BLK);
    pend.erase(pend.begin(), pend.begin() + (ptrdiff_t)done);
    return true;
  };
  // records per contig by position with a running maximum of their ends (all records: the writer's fetch has no
  // mapq / length filter, thread.rs:337-340); the contig's inflated bytes are loaded when a region needs them
  std::vector<int32_t> ipos, iend_max;
  int32_t cur_ref = INT32_MIN;
  int rc_all = LCR_OK;
  for (int32_t g = 0; g < n_regions && rc_all == LCR_OK; g++) {
    if (len[g] < 0) { rc_all = fail(b, LCR_E_ARG, "negative region length"); break; }
    if (region_ref[g] != cur_ref || b->cur_ref != region_ref[g]) {
      cur_ref = region_ref[g];
      if ((rc_all = load_contig(b, cur_ref)) != LCR_OK) break;
      ipos.clear(); iend_max.clear();
      int32_t run = INT32_MIN;
      for (const Rec& r : b->recs) {
        if (!ipos.empty() && r.pos < ipos.back()) { rc_all = fail(b, LCR_E_ARG, "BAM is not coordinate-sorted"); break; }
        ipos.push_back(r.pos);
        run = std::max(run, end_pos(r)); iend_max.push_back(run);
      }
      if (rc_all != LCR_OK) break;
    }
    const uint8_t* d = b->view;
    const int64_t beg = start0[g] + 1, end = start0[g] + len[g] + 1;   // fetch((chr, start, end)), thread.rs:332-334
    const size_t hi = (size_t)(std::lower_bound(ipos.begin(), ipos.end(), end, [](int32_t v, int64_t e) { return (int64_t)v < e; }) - ipos.begin());
    size_t lo = (size_t)(std::upper_bound(iend_max.begin(), iend_max.end(), beg, [](int64_t bg, int32_t v) { return bg < (int64_t)v; }) - iend_max.begin());
    for (; lo < hi; lo++) {
      const Rec& r = b->recs[lo];
      if ((int64_t)end_pos(r) <= beg) continue;
      if (r.flag & (0x4 | 0x100 | 0x800)) continue;                                      // thread.rs:337-339
      // reference_start + 1 < region.start || reference_end + 1 > region.end -> skipped (thread.rs:340-345);
      // reference_end is htslib's bam_endpos
      if ((int64_t)r.pos + 1 < beg || (int64_t)end_pos(r) + 1 > end) continue;
      const uint8_t* q = d + r.off;
      const std::string nm(reinterpret_cast<const char*>(q + 32));
      const uint64_t fixed = 32ull + r.l_rn + 4ull * r.n_cig_core + (uint64_t)((r.l_seq + 1) / 2) + (uint64_t)r.l_seq;
      int32_t add_hp = 0, hpv = 0; uint32_t add_ps = 0, psv = 0;
      auto fh = m_hp.find(nm);
      if (fh != m_hp.end() && fh->second != 0 && !aux_has(q + fixed, q + r.size, 'H', 'P')) { add_hp = 1; hpv = fh->second; }   // thread.rs:347-352
      auto fp = m_ps.find(nm);
      if (fp != m_ps.end() && !aux_has(q + fixed, q + r.size, 'P', 'S')) { add_ps = 1; psv = fp->second; }                         // thread.rs:353-356
      const uint32_t bs = r.size + 7u * (uint32_t)add_hp + 7u * add_ps;
      const size_t at = pend.size();
      pend.resize(at + 4 + bs);
      uint8_t* w = pend.data() + at;
      w[0] = (uint8_t)bs; w[1] = (uint8_t)(bs >> 8); w[2] = (uint8_t)(bs >> 16); w[3] = (uint8_t)(bs >> 24);
      memcpy(w + 4, q, r.size);
      w += 4 + r.size;
      if (add_hp) { w[0] = 'H'; w[1] = 'P'; w[2] = 'i'; const uint32_t x = (uint32_t)hpv; w[3] = (uint8_t)x; w[4] = (uint8_t)(x >> 8); w[5] = (uint8_t)(x >> 16); w[6] = (uint8_t)(x >> 24); w += 7; }
      if (add_ps) { w[0] = 'P'; w[1] = 'S'; w[2] = 'I'; const uint32_t x = psv; w[3] = (uint8_t)x; w[4] = (uint8_t)(x >> 8); w[5] = (uint8_t)(x >> 16); w[6] = (uint8_t)(x >> 24); }
      if (pend.size() >= CHUNK_BLOCKS * (size_t)BLK) {
        note_resident(b, (int64_t)pend.capacity());
        if (!flush(false)) { rc_all = fail(b, LCR_E_ARG, "deflate failed"); break; }
      }
    }
  }
  if (rc_all == LCR_OK) {
    note_resident(b, (int64_t)pend.capacity());
    if (!flush(true)) rc_all = fail(b, LCR_E_ARG, "deflate failed");
  }
  io_ok = (fclose(f) == 0) && io_ok;
  if (rc_all != LCR_OK) return rc_all;
  if (!io_ok) return fail(b, LCR_E_ARG, std::string("write failed: ") + out_path);
  return LCR_OK;
}

// ---- decoded reads -> BAM (one contig): the inverse of lcr_bam_batch, for synthetic data sets and round-trip tests.
// Records: name "r<index>", mapq 60, flag 0 / 16 (lcr_reads.flags bit 0), CIGAR as given, bases packed 4-bit, qualities,
// `ts:A:+/-` when flags bits 1-2 say so.  Reads must be sorted by position (they are written in the order given).
int lcr_bam_write_reads(const char* out_path, const char* contig, int64_t contig_len, const lcr_reads* rd, int32_t level, int32_t n_threads) {
  if (!out_path || !contig || !rd || rd->mem != LCR_MEM_HOST || rd->n_reads < 0 || contig_len < 0 || level < -1 || level > 9) return LCR_E_ARG;
  if (n_threads < 1) n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
  const int64_t nr = rd->n_reads;
  // header: magic, l_text, text, n_ref, (l_name, name, l_ref)
  std::string text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:" + std::string(contig) + "\tLN:" + std::to_string(contig_len) + "\n";
  std::vector<uint8_t> head;
  auto put32 = [](std::vector<uint8_t>& v, uint32_t x) { for (int i = 0; i < 4; i++) v.push_back((uint8_t)(x >> (8 * i))); };
  head.insert(head.end(), {'B', 'A', 'M', 1});
  put32(head, (uint32_t)text.size()); head.insert(head.end(), text.begin(), text.end());
  put32(head, 1); put32(head, (uint32_t)strlen(contig) + 1);
  head.insert(head.end(), contig, contig + strlen(contig) + 1);
  put32(head, (uint32_t)contig_len);
  // record offsets
  std::vector<uint64_t> off((size_t)nr + 1);
  off[0] = head.size();
  for (int64_t r = 0; r < nr; r++) {
    char nm[24];
    const int ln = snprintf(nm, sizeof nm, "r%lld", (long long)r) + 1;
    const int ts = (rd->flags[r] >> 1) & 3;
    const uint64_t l = (uint64_t)rd->seq_len[r];
    off[(size_t)r + 1] = off[(size_t)r] + 4 + 32 + (uint64_t)ln + 4ull * rd->n_cig[r] + (l + 1) / 2 + l + (ts ? 4 : 0);
  }
  std::vector<uint8_t> pay(off[(size_t)nr]);
  memcpy(pay.data(), head.data(), head.size());
  static const uint8_t code[256] = {};   // (filled below: ASCII -> 4-bit BAM code)
  uint8_t enc[256];
  memset(enc, 15, sizeof enc);
  { const char* a = "=ACMGRSVTWYHKDBN"; for (int i = 0; i < 16; i++) enc[(uint8_t)a[i]] = (uint8_t)i; }
  (void)code;
  parallel_for(nr, n_threads, 256, [&](int64_t r) {
    uint8_t* w = pay.data() + off[(size_t)r];
    auto w32 = [&](uint32_t x) { w[0] = (uint8_t)x; w[1] = (uint8_t)(x >> 8); w[2] = (uint8_t)(x >> 16); w[3] = (uint8_t)(x >> 24); w += 4; };
    char nm[24];
    const int ln = snprintf(nm, sizeof nm, "r%lld", (long long)r) + 1;
    const uint32_t l = (uint32_t)rd->seq_len[r], nc = rd->n_cig[r];
    const int ts = (rd->flags[r] >> 1) & 3;
    w32((uint32_t)(off[(size_t)r + 1] - off[(size_t)r] - 4));
    w32(0);                                            // refID
    w32((uint32_t)rd->pos[r]);
    w32((uint32_t)ln | (60u << 8) | (4680u << 16));     // l_read_name, mapq, bin (unused by sequential readers)
    w32(nc | ((rd->flags[r] & 1 ? 16u : 0u) << 16));    // n_cigar_op, flag
    w32(l); w32(0xFFFFFFFFu); w32(0xFFFFFFFFu); w32(0);
    memcpy(w, nm, (size_t)ln); w += ln;
    memcpy(w, rd->cigar + rd->cig_off[r], 4ull * nc); w += 4ull * nc;
    const uint8_t* bs = rd->bases + rd->seq_off[r];
    for (uint32_t i = 0; i + 1 < l; i += 2) *w++ = (uint8_t)((enc[bs[i]] << 4) | enc[bs[i + 1]]);
    if (l & 1) *w++ = (uint8_t)(enc[bs[l - 1]] << 4);
    memcpy(w, rd->quals + rd->seq_off[r], l); w += l;
    if (ts) { w[0] = 't'; w[1] = 's'; w[2] = 'A'; w[3] = ts == 1 ? '+' : '-'; }
  });
  // BGZF
  FILE* f = fopen(out_path, "wb");
  if (!f) return LCR_E_ARG;
  const uint64_t BLK = 0xff00;
  const size_t nblk = (pay.size() + BLK - 1) / BLK, n_out = nblk + 1;
  std::vector<std::vector<uint8_t>> comp(n_out);
  std::atomic<int> bad{0};
  parallel_for((int64_t)n_out, n_threads, 4, [&](int64_t i) {
    const uint64_t o = (uint64_t)i * BLK;
    const uint32_t n = (size_t)i >= nblk ? 0u : (uint32_t)std::min<uint64_t>(BLK, pay.size() - o);
    std::vector<uint8_t>& c = comp[(size_t)i];
    c.resize(18 + compressBound(n) + 8);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad.store(1); return; }
    zs.next_in = n ? pay.data() + o : nullptr; zs.avail_in = n;
    zs.next_out = c.data() + 18; zs.avail_out = (uInt)(c.size() - 18 - 8);
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END || 18 + clen + 8 > 65536) { bad.store(1); return; }
    static const uint8_t hd[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(c.data(), hd, 16);
    const uint32_t bsize = (uint32_t)(18 + clen + 8 - 1);
    c[16] = (uint8_t)bsize; c[17] = (uint8_t)(bsize >> 8);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), n ? pay.data() + o : nullptr, n);
    uint8_t* t = c.data() + 18 + clen;
    for (int k = 0; k < 4; k++) { t[k] = (uint8_t)(crc >> (8 * k)); t[4 + k] = (uint8_t)(n >> (8 * k)); }
    c.resize(18 + clen + 8);
  });
  bool io_ok = !bad.load();
  for (auto& c : comp) io_ok = io_ok && fwrite(c.data(), 1, c.size(), f) == c.size();
  io_ok = (fclose(f) == 0) && io_ok;
  return io_ok ? LCR_OK : LCR_E_ARG;
}

}  // extern "C"
