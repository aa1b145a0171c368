// lcr_bam.cpp — SURVEY §8(f) N1: BGZF / BAM decode -> lcr_reads on the host (no GPU code in this file).
//
// Replaces the rust-htslib IndexedReader use of the reference (src/util.rs:636-691, src/fragment.rs:19-59;
// region discovery's pass over the file, util.rs:256-287): the reference inflates every region's blocks
// twice (once for the pileup, once for the fragments) and decodes records one at a time; here the file is
// inflated once, all BGZF blocks in parallel, every record is indexed once, and the batches handed to
// lcr_load_batch are cut out of that index (one decode shared by K1 and K3).
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

#include <algorithm>
#include <atomic>
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

// n items over up to nt threads, chunked through an atomic counter; fn(i) must not throw
void parallel_for(int64_t n, int nt, int64_t chunk, const std::function<void(int64_t)>& fn) {
  if (n <= 0) return;
  nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, (n + chunk - 1) / chunk));
  if (nt == 1) { for (int64_t i = 0; i < n; i++) fn(i); return; }
  std::atomic<int64_t> next{0};
  auto work = [&]() {
    for (;;) {
      const int64_t b = next.fetch_add(chunk);
      if (b >= n) return;
      const int64_t e = std::min(n, b + chunk);
      for (int64_t i = b; i < e; i++) fn(i);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nt; t++) th.emplace_back(work);
  work();
  for (auto& t : th) t.join();
}

}  // namespace

struct lcr_bam {
  std::string err;
  int n_threads = 1;
  std::unique_ptr<uint8_t[]> data;   // inflated stream (not value-initialised: every byte is written by inflate)
  size_t data_size = 0;
  std::vector<std::string> ref_names;
  std::vector<const char*> ref_name_ptrs;
  std::vector<int64_t> ref_len;
  std::vector<Rec> recs;
  size_t header_size = 0;   // inflated bytes before the first record (magic, text, reference table)
  // results of the last lcr_bam_spans / lcr_bam_batch call
  std::vector<int32_t> sp_start, sp_end;
  std::vector<int32_t> b_pos, b_seq_len, b_lead, b_trail, b_read_begin;
  std::vector<uint8_t> b_flags, b_bases, b_quals;
  std::vector<uint64_t> b_seq_off, b_cig_off, b_name_off;
  std::vector<uint32_t> b_n_cig, b_cigar;
  std::vector<char> b_names;
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
        if (t0 == 'C' && t1 == 'G' && sub == 'I' && p + 5 + (size_t)cnt * 4 <= end) { r.cg_tag_at = (uint64_t)(p + 5 - base); r.cg_tag_n = cnt; }
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

}  // namespace

extern "C" {

int lcr_bam_open(const char* path, int32_t n_threads, lcr_bam** out) {
  if (!path || !out) return LCR_E_ARG;
  *out = nullptr;
  lcr_bam* b = new (std::nothrow) lcr_bam();
  if (!b) return LCR_E_NOMEM;
  *out = b;   // returned even on failure so that lcr_bam_last_error can explain; the caller closes it
  if (n_threads < 1) n_threads = (int32_t)std::max(1u, std::thread::hardware_concurrency());
  b->n_threads = n_threads;
  // ---- the compressed file
  std::vector<uint8_t> raw;
  {
    FILE* f = fopen(path, "rb");
    if (!f) return fail(b, LCR_E_ARG, std::string("cannot open ") + path);
    if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return fail(b, LCR_E_ARG, "seek failed"); }
    const long sz = ftell(f);
    if (sz < 0) { fclose(f); return fail(b, LCR_E_ARG, "tell failed"); }
    rewind(f);
    try { raw.resize((size_t)sz); } catch (...) { fclose(f); return fail(b, LCR_E_NOMEM, "out of memory reading the file"); }
    const size_t got = sz ? fread(raw.data(), 1, (size_t)sz, f) : 0;
    fclose(f);
    if (got != (size_t)sz) return fail(b, LCR_E_ARG, "short read");
  }
  // ---- BGZF block table (gzip member with the BC extra subfield; SAM spec 4.1)
  struct Blk { size_t coff, clen; uint32_t crc, isize; uint64_t uoff; };
  std::vector<Blk> blks;
  uint64_t total = 0;
  for (size_t off = 0; off < raw.size();) {
    if (off + 18 > raw.size() || raw[off] != 0x1f || raw[off + 1] != 0x8b || raw[off + 2] != 8 || !(raw[off + 3] & 4))
      return fail(b, LCR_E_ARG, "not a BGZF block at offset " + std::to_string(off));
    const size_t xlen = rd16(&raw[off + 10]);
    if (off + 12 + xlen > raw.size()) return fail(b, LCR_E_ARG, "truncated BGZF header");
    int64_t bsize = -1;
    for (size_t p = off + 12; p + 4 <= off + 12 + xlen;) {
      const size_t slen = rd16(&raw[p + 2]);
      if (raw[p] == 66 && raw[p + 1] == 67 && slen == 2 && p + 6 <= off + 12 + xlen) bsize = rd16(&raw[p + 4]);
      p += 4 + slen;
    }
    if (bsize < 0) return fail(b, LCR_E_ARG, "BGZF block without BC subfield at offset " + std::to_string(off));
    const size_t blen = (size_t)bsize + 1;
    if (blen < 12 + xlen + 8 || off + blen > raw.size()) return fail(b, LCR_E_ARG, "truncated BGZF block at offset " + std::to_string(off));
    Blk k;
    k.coff = off + 12 + xlen; k.clen = blen - (12 + xlen) - 8;
    k.crc = rd32(&raw[off + blen - 8]); k.isize = rd32(&raw[off + blen - 4]);
    if (k.isize > 65536) return fail(b, LCR_E_ARG, "BGZF block larger than 64 KiB");
    k.uoff = total; total += k.isize;
    blks.push_back(k);
    off += blen;
  }
  b->data.reset(new (std::nothrow) uint8_t[(size_t)total + 1]);
  if (!b->data) return fail(b, LCR_E_NOMEM, "out of memory for the inflated stream");
  b->data_size = (size_t)total;
  // ---- inflate: blocks are independent
  std::atomic<int> bad{-1};
  parallel_for((int64_t)blks.size(), n_threads, 16, [&](int64_t i) {
    const Blk& k = blks[(size_t)i];
    if (k.isize == 0) return;   // EOF marker and other empty blocks
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) { bad.store((int)i); return; }
    zs.next_in = const_cast<Bytef*>(raw.data() + k.coff); zs.avail_in = (uInt)k.clen;
    zs.next_out = b->data.get() + k.uoff; zs.avail_out = k.isize;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == k.isize;
    inflateEnd(&zs);
    if (!ok || crc32(crc32(0L, Z_NULL, 0), b->data.get() + k.uoff, k.isize) != k.crc) bad.store((int)i);
  });
  if (bad.load() >= 0) return fail(b, LCR_E_ARG, "BGZF block " + std::to_string(bad.load()) + " does not inflate / CRC mismatch");
  raw.clear(); raw.shrink_to_fit();
  // ---- BAM header (SAM spec 4.2)
  const uint8_t* const d = b->data.get();
  const size_t n = b->data_size;
  if (n < 12 || memcmp(d, "BAM\1", 4) != 0) return fail(b, LCR_E_ARG, "not a BAM file");
  size_t p = 8 + (size_t)rd32(&d[4]);
  if (p + 4 > n) return fail(b, LCR_E_ARG, "truncated BAM header");
  const int32_t n_ref = rdi32(&d[p]);
  p += 4;
  if (n_ref < 0) return fail(b, LCR_E_ARG, "bad reference count");
  for (int32_t i = 0; i < n_ref; i++) {
    if (p + 4 > n) return fail(b, LCR_E_ARG, "truncated reference table");
    const uint32_t l_name = rd32(&d[p]);
    if (l_name == 0 || p + 4 + (size_t)l_name + 4 > n) return fail(b, LCR_E_ARG, "truncated reference table");
    b->ref_names.emplace_back(reinterpret_cast<const char*>(&d[p + 4]), l_name - 1);
    b->ref_len.push_back(rdi32(&d[p + 4 + l_name]));
    p += 8 + l_name;
  }
  for (auto& s : b->ref_names) b->ref_name_ptrs.push_back(s.c_str());
  b->header_size = p;
  // ---- record index: the block_size chain is sequential, the records' fields are not
  while (p < n) {
    if (p + 4 > n) return fail(b, LCR_E_ARG, "truncated record header");
    const uint32_t bs = rd32(&d[p]);
    if (bs < 32 || p + 4 + (size_t)bs > n) return fail(b, LCR_E_ARG, "truncated record at inflated offset " + std::to_string(p));
    Rec r{};
    r.off = p + 4; r.size = bs;
    b->recs.push_back(r);
    p += 4 + (size_t)bs;
  }
  std::atomic<int64_t> bad_rec{-1}, long_cigar_bad{-1};
  parallel_for((int64_t)b->recs.size(), n_threads, 1024, [&](int64_t i) {
    Rec& r = b->recs[(size_t)i];
    const uint8_t* q = &d[r.off];
    r.ref_id = rdi32(q); r.pos = rdi32(q + 4); r.l_rn = q[8]; r.mapq = q[9];
    r.n_cig = r.n_cig_core = rd16(q + 12); r.flag = rd16(q + 14); r.l_seq = rdi32(q + 16);
    const uint64_t need = 32ull + r.l_rn + 4ull * r.n_cig + (uint64_t)((r.l_seq + 1) / 2) + (uint64_t)r.l_seq;
    if (r.l_seq < 0 || r.l_rn == 0 || need > r.size) { bad_rec.store(i); return; }
    if (!aux_scan(q + need, q + r.size, r, d)) { bad_rec.store(i); return; }
    const uint8_t* cg = q + 32 + r.l_rn;
    r.cig_at = r.off + 32 + r.l_rn;
    // long CIGAR (> 65535 ops; SAM spec 4.2.2, applied by htslib when it reads a record): the core field holds the
    // placeholder <l_seq>S<ref_len>N and the real CIGAR travels in the CG:B,I tag
    if (r.n_cig == 2 && (rd32(cg) & 15) == 4 && (int64_t)(rd32(cg) >> 4) == (int64_t)r.l_seq && (rd32(cg + 4) & 15) == 3) {
      if (!r.cg_tag_n) { long_cigar_bad.store(i); return; }
      cg = d + r.cg_tag_at; r.cig_at = r.cg_tag_at; r.n_cig = r.cg_tag_n;
    }
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
  });
  if (bad_rec.load() >= 0) return fail(b, LCR_E_ARG, "malformed record " + std::to_string(bad_rec.load()));
  if (long_cigar_bad.load() >= 0)
    return fail(b, LCR_E_ARG, "record " + std::to_string(long_cigar_bad.load()) + " has the long-CIGAR placeholder (<l_seq>S<n>N) but no CG:B,I tag");
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
  *n = (int64_t)b->recs.size();
  return LCR_OK;
}

int lcr_bam_spans(lcr_bam* b, int32_t ref_id, const lcr_read_filter* f, int32_t* n, const int32_t** ref_start, const int32_t** ref_end) {
  if (!b || !f || !n || !ref_start || !ref_end) return LCR_E_ARG;
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
  b->b_bases.resize(so); b->b_quals.resize(so); b->b_cigar.resize(co); b->b_names.resize(no);
  static const char NT16[] = "=ACMGRSVTWYHKDBN";
  const uint8_t* d = b->data.get();
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
  // records per contig by position with a running maximum of their ends (all records: the writer's fetch has no
  // mapq / length filter, thread.rs:337-340)
  const uint8_t* d = b->data.get();
  struct Out { uint32_t rec; int32_t hp; uint32_t ps; uint8_t add_hp, add_ps; uint64_t at; };
  std::vector<Out> outs;
  std::vector<uint32_t> idx;
  std::vector<int32_t> ipos, iend_max;
  int32_t cur_ref = INT32_MIN;
  for (int32_t g = 0; g < n_regions; g++) {
    if (len[g] < 0) return fail(b, LCR_E_ARG, "negative region length");
    if (region_ref[g] != cur_ref) {
      cur_ref = region_ref[g];
      idx.clear(); ipos.clear(); iend_max.clear();
      int32_t run = INT32_MIN;
      for (size_t i = 0; i < b->recs.size(); i++) {
        const Rec& r = b->recs[i];
        if (r.ref_id != cur_ref) continue;
        if (!ipos.empty() && r.pos < ipos.back()) return fail(b, LCR_E_ARG, "BAM is not coordinate-sorted");
        idx.push_back((uint32_t)i); ipos.push_back(r.pos);
        run = std::max(run, end_pos(r)); iend_max.push_back(run);
      }
    }
    const int64_t beg = start0[g] + 1, end = start0[g] + len[g] + 1;   // fetch((chr, start, end)), thread.rs:332-334
    const size_t hi = (size_t)(std::lower_bound(ipos.begin(), ipos.end(), end, [](int32_t v, int64_t e) { return (int64_t)v < e; }) - ipos.begin());
    size_t lo = (size_t)(std::upper_bound(iend_max.begin(), iend_max.end(), beg, [](int64_t bg, int32_t v) { return bg < (int64_t)v; }) - iend_max.begin());
    for (; lo < hi; lo++) {
      const Rec& r = b->recs[idx[lo]];
      if ((int64_t)end_pos(r) <= beg) continue;
      if (r.flag & (0x4 | 0x100 | 0x800)) continue;                                      // thread.rs:337-339
      // reference_start + 1 < region.start || reference_end + 1 > region.end -> skipped (thread.rs:340-345);
      // reference_end is htslib's bam_endpos
      if ((int64_t)r.pos + 1 < beg || (int64_t)end_pos(r) + 1 > end) continue;
      Out o{idx[lo], 0, 0, 0, 0, 0};
      const uint8_t* q = d + r.off;
      const std::string nm(reinterpret_cast<const char*>(q + 32));
      const uint64_t fixed = 32ull + r.l_rn + 4ull * r.n_cig_core + (uint64_t)((r.l_seq + 1) / 2) + (uint64_t)r.l_seq;
      auto fh = m_hp.find(nm);
      if (fh != m_hp.end() && fh->second != 0 && !aux_has(q + fixed, q + r.size, 'H', 'P')) { o.add_hp = 1; o.hp = fh->second; }   // thread.rs:347-352
      auto fp = m_ps.find(nm);
      if (fp != m_ps.end() && !aux_has(q + fixed, q + r.size, 'P', 'S')) { o.add_ps = 1; o.ps = fp->second; }                         // thread.rs:353-356
      outs.push_back(o);
    }
  }
  // ---- payload: header as in the input, then the records with HP:i (int32) / PS:I (uint32) appended to the aux block
  uint64_t total = b->header_size;
  for (Out& o : outs) { o.at = total; total += 4ull + b->recs[o.rec].size + 7ull * o.add_hp + 7ull * o.add_ps; }
  std::unique_ptr<uint8_t[]> pay(new (std::nothrow) uint8_t[(size_t)total + 1]);
  if (!pay) return fail(b, LCR_E_NOMEM, "out of memory for the output stream");
  memcpy(pay.get(), d, b->header_size);
  parallel_for((int64_t)outs.size(), n_threads, 512, [&](int64_t i) {
    const Out& o = outs[(size_t)i];
    const Rec& r = b->recs[o.rec];
    uint8_t* w = pay.get() + o.at;
    const uint32_t bs = r.size + 7u * o.add_hp + 7u * o.add_ps;
    w[0] = (uint8_t)bs; w[1] = (uint8_t)(bs >> 8); w[2] = (uint8_t)(bs >> 16); w[3] = (uint8_t)(bs >> 24);
    memcpy(w + 4, d + r.off, r.size);
    w += 4 + r.size;
    if (o.add_hp) { w[0] = 'H'; w[1] = 'P'; w[2] = 'i'; const uint32_t v = (uint32_t)o.hp; w[3] = (uint8_t)v; w[4] = (uint8_t)(v >> 8); w[5] = (uint8_t)(v >> 16); w[6] = (uint8_t)(v >> 24); w += 7; }
    if (o.add_ps) { w[0] = 'P'; w[1] = 'S'; w[2] = 'I'; const uint32_t v = o.ps; w[3] = (uint8_t)v; w[4] = (uint8_t)(v >> 8); w[5] = (uint8_t)(v >> 16); w[6] = (uint8_t)(v >> 24); }
  });
  // ---- BGZF: 0xff00-byte blocks (htslib's BGZF_BLOCK_SIZE), deflated in parallel, then the empty EOF block
  const uint64_t BLK = 0xff00;
  const size_t nblk = (size_t)((total + BLK - 1) / BLK);
  std::vector<std::vector<uint8_t>> comp(nblk + 1);
  std::atomic<int> bad{0};
  parallel_for((int64_t)nblk + 1, n_threads, 4, [&](int64_t i) {
    const uint64_t off = (uint64_t)i * BLK;
    const uint32_t n = (size_t)i == nblk ? 0u : (uint32_t)std::min<uint64_t>(BLK, total - off);
    std::vector<uint8_t>& c = comp[(size_t)i];
    c.resize(18 + compressBound(n) + 8);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad.store(1); return; }
    zs.next_in = pay.get() + off; zs.avail_in = n;
    zs.next_out = c.data() + 18; zs.avail_out = (uInt)(c.size() - 18 - 8);
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END || 18 + clen + 8 > 65536) { bad.store(1); return; }
    static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(c.data(), head, 16);
    const uint32_t bsize = (uint32_t)(18 + clen + 8 - 1);
    c[16] = (uint8_t)bsize; c[17] = (uint8_t)(bsize >> 8);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), pay.get() + off, n);
    uint8_t* t = c.data() + 18 + clen;
    t[0] = (uint8_t)crc; t[1] = (uint8_t)(crc >> 8); t[2] = (uint8_t)(crc >> 16); t[3] = (uint8_t)(crc >> 24);
    t[4] = (uint8_t)n; t[5] = (uint8_t)(n >> 8); t[6] = (uint8_t)(n >> 16); t[7] = (uint8_t)(n >> 24);
    c.resize(18 + clen + 8);
  });
  if (bad.load()) return fail(b, LCR_E_ARG, "deflate failed");
  FILE* f = fopen(out_path, "wb");
  if (!f) return fail(b, LCR_E_ARG, std::string("cannot create ") + out_path);
  bool ok = true;
  for (auto& c : comp) ok = ok && fwrite(c.data(), 1, c.size(), f) == c.size();
  ok = (fclose(f) == 0) && ok;
  if (!ok) return fail(b, LCR_E_ARG, std::string("write failed: ") + out_path);
  return LCR_OK;
}

}  // extern "C"
