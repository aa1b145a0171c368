// k4_grid.hip — the chain regions (S > max_enum_snps, phase.rs:1123-1233) entirely on the device.
//
// One code path, two scopes.  Every step of the chain is written against a "scope" that supplies the thread
// numbering and the barrier: WgScope = one workgroup per region (barrier = __syncthreads; the usual case, a few
// hundred regions of a batch side by side), GridScope = ALL workgroups of a persistent launch work on ONE region
// (barrier = a grid-wide barrier on an atomic counter, sigma / delta / eta in HBM, per-SNP sums by 64-bit global
// atomics) for regions whose matrix is far beyond one CU (BASELINE config C5: one 1 Mb island, 4 10^5 reads x
// 5 10^3 SNPs, 2 503 cross_optimize calls).  Both scopes compute the same integers in the same fixed-point
// arithmetic and the same ordered f64 sums, so their results are bit-identical (tested against each other and
// against the oracle).
//
// Steps (reference file:line under /root/reference/src):
//   ordered_columns   per SNP the phase entries in row order = snp_cover_fragments order (fragment.rs:293-306)
//   ld_pair_table     allele-pair counts of every SNP pair a read links (fragment.rs:208-240), banded S x W table
//   ld_graph          perfect-LD pairs (snp.rs:158-188, candidate.rs:626-692) as sorted adjacency lists
//   ld_components     kosaraju_scc of petgraph on that graph (candidate.rs:733): blocks and their node order
//   ld_seed           init_haplotypes_LD2 (phase.rs:609-671): random delta, BFS sign propagation inside a block
//   cross_optimize    phase.rs:810-976 (WgScope: k4_dev.h; GridScope: cross_optimize_grid below)
//   block_flip        cross_optimize_by_block (phase.rs:1298-1394): f64 sums of ratios in the reference's order
//   perturbation      phase.rs:1197-1233
// petgraph facts used (0.6.4; restated, the crate is not in /root/reference): GraphMap keeps nodes in insertion
// order and adjacency lists in edge-insertion order.  The edges are inserted in (i asc, j asc) order, so every
// adjacency list is ascending, the first-inserted node of a component is its smallest index, kosaraju_scc emits
// the components in DESCENDING order of that node, and inside a component the nodes come out in the preorder of
// a DFS from that node which takes the LARGEST unvisited neighbour first (Dfs pops a stack it pushed in ascending
// order).  Bfs marks nodes when it pushes them; the first visited node that has a perfect-LD pair with `nx`
// (phase.rs:628-657) is therefore nx's BFS parent.  With ld_weight_threshold = 1 (thread.rs:166 hard-codes it) no
// edge is ever removed; other values are rejected by lcr_phase.
#include <climits>
#include "k4_dev.h"
#include <type_traits>
#include "k4_grid.h"
#include "k4_post.h"

namespace {

constexpr int CH_THREADS = 1024;
constexpr int CH_WAVES = CH_THREADS / 64;

// region-relative views the chain steps work on
struct ChainView {
  int g, S, R, nrow, c0, r0, W, n_parts;
  int64_t e_base;                 // first entry of the region in K3's matrix
  MatView mv;                     // phase matrix (global memory)
  int8_t *sg, *dl, *et;           // working state
  int8_t *bsg, *bdl, *bet;        // best state
  uint32_t* tbl; int32_t* adj; int32_t* pcnt;
  int32_t *adj_ptr, *blk_ptr, *blk_of, *blk_pos, *blk_nodes, *stack, *queue, *blk_info, *flipcol, *erow, *cent;
  uint8_t *seen, *ok, *cons; int8_t* new_hap;
  double *qs, *qfs;
  const int8_t* vt; const uint8_t* fp;
  const int32_t* prow_src;
};

__device__ ChainView make_view(const ChainDev& C, const ChainDesc& d, const RegionDev& rd) {
  ChainView v;
  v.g = d.slot; v.S = rd.S; v.R = rd.R; v.W = d.W; v.n_parts = d.n_parts;
  v.r0 = C.row_region_off[v.g]; v.nrow = C.row_region_off[v.g + 1] - v.r0;
  v.c0 = C.cand_off[v.g];
  v.e_base = C.row_ptr[v.r0];
  v.mv = global_view(C.P, rd);
  v.bsg = C.P.st_sigma + rd.sig_off; v.bdl = C.P.st_delta + rd.snp_off; v.bet = C.P.st_eta + rd.snp_off;
  v.sg = C.w_sigma + rd.sig_off; v.dl = C.w_delta + rd.snp_off; v.et = C.w_eta + rd.snp_off;
  v.tbl = C.ld_tbl + 2 * d.tbl_off; v.adj = C.ld_adj + d.adj_off; v.pcnt = C.part_cnt + d.part_off;
  v.adj_ptr = C.adj_ptr + rd.cp_off; v.blk_ptr = C.blk_ptr + rd.cp_off;
  v.blk_of = C.blk_of + rd.snp_off; v.blk_pos = C.blk_pos + rd.snp_off; v.blk_nodes = C.blk_nodes + rd.snp_off;
  v.stack = C.stack + 2 * (int64_t)rd.snp_off; v.queue = C.queue + rd.snp_off; v.blk_info = C.blk_info + 2 * v.g;
  v.flipcol = C.flipcol + rd.sig_off; v.erow = C.erow + rd.e_off; v.cent = C.cent + rd.e_off;
  v.seen = C.seen + rd.snp_off; v.ok = C.ld_ok + rd.snp_off; v.cons = const_cast<uint8_t*>(C.P.snp_cons) + rd.snp_off;
  v.new_hap = C.new_hap + rd.snp_off;
  v.qs = C.qs + rd.snp_off; v.qfs = C.qfs + rd.snp_off;
  v.vt = C.P.snp_vt + rd.snp_off; v.fp = C.P.snp_fp + rd.snp_off;
  v.prow_src = C.prow_src + rd.sig_off;
  return v;
}

// ------------------------------------------------------------------------------------------------------------
// ordered column index: cent[cp[i] .. cp[i+1]) = the phase entries (indices into the CSR) of SNP i in row order,
// erow[e] = phasing row of entry e.  Stable counting sort by column: the rows are cut into n_parts runs, entries
// are counted per (part, SNP), and each part is filled by one wave, 64 entries at a time in CSR order.
// ------------------------------------------------------------------------------------------------------------
template <class SC>
__device__ void ordered_index(SC& sc, int R, int S, const int32_t* rp, const int32_t* pc, const int32_t* cp /* or nullptr */, int32_t* cp_out,
                              int32_t* erow, int32_t* cent, int32_t* pcnt, int np, int (*sm)[16]) {
  const int rq = max(1, (R + np - 1) / np);
  for (int64_t i = sc.tid(); i < (int64_t)np * S; i += sc.nt()) pcnt[i] = 0;
  sc.sync();
  for (int row = sc.tid(); row < R; row += sc.nt()) {
    int32_t* cnt = pcnt + (int64_t)(row / rq) * S;
    for (int e = rp[row]; e < rp[row + 1]; e++) { erow[e] = row; atomicAdd(&cnt[pc[e]], 1); }
  }
  sc.sync();
  if (!cp) {   // column offsets from the counts
    for (int i = sc.tid(); i < S; i += sc.nt()) { int t = 0; for (int q = 0; q < np; q++) t += pcnt[(int64_t)q * S + i]; cp_out[i] = t; }
    sc.sync();
    if (sc.blk() == 0) {
      int carry = 0;
      for (int base = 0; base < S; base += (int)blockDim.x) {
        const int i = base + threadIdx.x;
        const int d = i < S ? cp_out[i] : 0;
        int ex, d0, tot, d1;
        block_scan2_rt(d, 0, ex, d0, tot, d1, sm);
        if (i < S) cp_out[i] = carry + ex;
        carry += tot;
      }
      if (threadIdx.x == 0) cp_out[S] = carry;
    }
    sc.sync();
    cp = cp_out;
  }
  for (int i = sc.tid(); i < S; i += sc.nt()) {
    int at = cp[i];
    for (int q = 0; q < np; q++) { int32_t* p = pcnt + (int64_t)q * S + i; const int n = *p; *p = at; at += n; }
  }
  sc.sync();
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int q = sc.wave(); q < np; q += sc.nwaves()) {
    int32_t* cur = pcnt + (int64_t)q * S;
    const int e_lo = rp[(int)min((int64_t)q * rq, (int64_t)R)], e_hi = rp[(int)min((int64_t)(q + 1) * rq, (int64_t)R)];
    for (int base = e_lo; base < e_hi; base += 64) {
      const int e = base + lane;
      const bool valid = e < e_hi;
      const int c = valid ? pc[e] : -1;
      const int at = valid ? cur[c] : 0;          // all cursor reads of the chunk precede its cursor writes
      unsigned long long rem = __ballot(valid), mine = 0;
      while (rem) {                               // lanes with equal columns, in lane (= row) order
        const int cc = __shfl(c, __ffsll((long long)rem) - 1, 64);
        const unsigned long long m = __ballot(c == cc);
        if (c == cc) mine = m;
        rem &= ~m;
      }
      if (valid) {
        const int rank = __popcll(mine & below);
        cent[at + rank] = e;
        if (rank == 0) cur[c] = at + __popcll(mine);
      }
      wave_mem_sync();
    }
  }
  sc.sync();
}

// ------------------------------------------------------------------------------------------------------------
// LD blocks: pair table -> graph -> components (kosaraju order) -> LD-seeded start haplotypes
// ------------------------------------------------------------------------------------------------------------
template <class SC>
__device__ void ld_pair_table(SC& sc, const ChainDev& C, const ChainView& v) {
  const int S = v.S, W = v.W;
  const int64_t e_lo = C.row_ptr[v.r0], n_ent = C.row_ptr[v.r0 + v.nrow] - e_lo;
  // one workgroup with spare LDS: entries (SNP | allele bit, row remainder) and the table itself live in LDS, the pairs
  // are spread over the threads by their first entry (a thread per row waits for its longest row: 231 global atomics in
  // a row of 22 entries), the table goes to HBM once at the end
  uint32_t lds_bytes = 0;
  uint8_t* const lds = sc.lds_scratch(&lds_bytes);
  const uint64_t tbl_bytes = 8ull * (uint64_t)S * (uint64_t)W;
  const bool in_lds = lds && S < 32768 && n_ent < 65536 && tbl_bytes + 4ull * (uint64_t)n_ent + 16 <= (uint64_t)lds_bytes;
  if (!in_lds) for (int64_t i = sc.tid(); i < 2ll * S * W; i += sc.nt()) v.tbl[i] = 0;
  for (int i = sc.tid(); i < S; i += sc.nt()) {
    const lcr_candidate& c = C.cand[v.c0 + i];
    // for_phasing, exactly one of the two major alleles is the reference (candidate.rs:637-660), both fractions non-zero
    v.ok[i] = (v.fp[i] && ((c.allele1 == c.ref_base) != (c.allele2 == c.ref_base)) && c.af1 != 0.0f && c.af2 != 0.0f) ? 1 : 0;
    v.blk_of[i] = -1; v.seen[i] = 0; v.cons[i] = 0;
  }
  if (in_lds) {
    uint32_t* const ltbl = reinterpret_cast<uint32_t*>(lds);
    uint16_t* const eidx = reinterpret_cast<uint16_t*>(lds + ((tbl_bytes + 15) & ~15ull));   // SNP | allele << 15; 0xFFFF = not eligible
    uint16_t* const erem = eidx + n_ent;                                                     // entries behind this one in its row
    for (int i = sc.tid(); i < 2 * S * W; i += sc.nt()) ltbl[i] = 0;
    sc.sync();
    for (int e = sc.tid(); e < (int)n_ent; e += sc.nt()) {
      const int i = C.col[e_lo + e] - v.c0;
      eidx[e] = v.ok[i] ? (uint16_t)(i | ((C.val[e_lo + e] & 32) ? 0x8000 : 0)) : (uint16_t)0xFFFF;
    }
    for (int r = sc.tid(); r < v.nrow; r += sc.nt()) {
      const int eb = (int)(C.row_ptr[v.r0 + r] - e_lo), ee = (int)(C.row_ptr[v.r0 + r + 1] - e_lo);
      for (int e = eb; e < ee; e++) erem[e] = (uint16_t)(ee - e - 1);
    }
    sc.sync();
    for (int x = sc.tid(); x < (int)n_ent; x += sc.nt()) {
      const uint32_t a = eidx[x];
      if (a == 0xFFFFu) continue;
      const int i = (int)(a & 0x7FFFu), y1 = x + 1 + erem[x];
      uint32_t* const row = ltbl + 2 * (i * W - i - 1);
      for (int y = x + 1; y < y1; y++) {
        const uint32_t bb = eidx[y];
        if (bb != 0xFFFFu) atomicAdd(&row[2 * (int)(bb & 0x7FFFu) + (((bb ^ a) >> 15) & 1u)], 1u);
      }
    }
    sc.sync();
    for (int i = sc.tid(); i < 2 * S * W; i += sc.nt()) v.tbl[i] = ltbl[i];
    sc.sync();
    return;
  }
  sc.sync();
  // every fragment row, every pair of its entries (fragment.rs:208-240); only pairs of eligible SNPs are ever looked up
  for (int r = sc.tid(); r < v.nrow; r += sc.nt()) {
    const int64_t eb = C.row_ptr[v.r0 + r], ee = C.row_ptr[v.r0 + r + 1];
    for (int64_t x = eb; x < ee; x++) {
      const int i = C.col[x] - v.c0;
      if (!v.ok[i]) continue;
      const int pi = C.val[x] & 32;
      for (int64_t y = x + 1; y < ee; y++) {
        const int j = C.col[y] - v.c0;
        if (!v.ok[j]) continue;
        atomicAdd(&v.tbl[2 * ((int64_t)i * W + (j - i - 1)) + (((C.val[y] & 32) != pi) ? 1 : 0)], 1u);
      }
    }
  }
  sc.sync();
}

// perfect LD (snp.rs:158-188: score = c1 / c2 == 0 with c2 > 0): reads support cis or trans, never both
__device__ __forceinline__ int ld_pass(const ChainView& v, int a, int b) {   // a < b <= a + W; 0: no edge, 1: cis (+), 2: trans (-)
  if (!v.ok[a] || !v.ok[b]) return 0;
  const uint32_t* t = v.tbl + 2 * ((int64_t)a * v.W + (b - a - 1));
  const uint32_t cis = t[0], trans = t[1];
  if ((cis > 0) == (trans > 0)) return 0;
  return cis > 0 ? 1 : 2;
}

template <class SC>
__device__ void ld_graph(SC& sc, const ChainView& v, int (*sm)[16]) {
  const int S = v.S, W = v.W;
  for (int i = sc.tid(); i < S; i += sc.nt()) {
    int deg = 0;
    if (v.ok[i]) {
      for (int a = max(0, i - W); a < i; a++) deg += ld_pass(v, a, i) ? 1 : 0;
      for (int b = i + 1; b <= min(S - 1, i + W); b++) deg += ld_pass(v, i, b) ? 1 : 0;
    }
    v.queue[i] = deg;
  }
  sc.sync();
  if (sc.blk() == 0) {   // exclusive scan of the degrees (one workgroup)
    int carry = 0;
    for (int base = 0; base < S; base += (int)blockDim.x) {
      const int i = base + threadIdx.x;
      const int d = i < S ? v.queue[i] : 0;
      int ex, d0, tot, d1;
      block_scan2_rt(d, 0, ex, d0, tot, d1, sm);
      if (i < S) v.adj_ptr[i] = carry + ex;
      carry += tot;
    }
    if (threadIdx.x == 0) v.adj_ptr[S] = carry;
  }
  sc.sync();
  for (int i = sc.tid(); i < S; i += sc.nt()) {
    if (!v.ok[i]) continue;
    int at = v.adj_ptr[i];
    for (int a = max(0, i - W); a < i; a++) { const int p = ld_pass(v, a, i); if (p) v.adj[at++] = a | (p == 2 ? (int)0x80000000 : 0); }
    for (int b = i + 1; b <= min(S - 1, i + W); b++) { const int p = ld_pass(v, i, b); if (p) v.adj[at++] = b | (p == 2 ? (int)0x80000000 : 0); }
  }
  sc.sync();
}

// kosaraju_scc's output for this graph (see the header): blocks numbered in ASCENDING order of their smallest node
// (block 0 is the one the reference visits LAST), nodes of a block in the reference's order.  One wave.
__device__ void ld_components_wave(const ChainView& v) {
  const int S = v.S, lane = threadIdx.x & 63;
  int nb = 0, out = 0;
  for (int r = 0; r < S; r++) {
    if (v.adj_ptr[r + 1] == v.adj_ptr[r] || v.seen[r]) continue;   // (uniform)
    const int start = out;
    int sp = 0;
    auto visit = [&](int x) {
      if (lane == 0) { v.seen[x] = 1; v.blk_nodes[out] = x; v.blk_of[x] = nb; v.blk_pos[x] = out - start; }
      out++;
    };
    visit(r);
    int x = r, cur = v.adj_ptr[r + 1] - v.adj_ptr[r];   // top of the stack in registers
    wave_mem_sync();
    for (;;) {
      // the largest unvisited neighbour of x below position cur
      int found = -1;
      const int base = v.adj_ptr[x];
      while (cur > 0 && found < 0) {
        const int lo = max(0, cur - 64), idx = lo + lane;
        const bool valid = idx < cur;
        const int y = valid ? (v.adj[base + idx] & 0x7fffffff) : 0;
        const bool uns = valid && !v.seen[y];
        const unsigned long long m = __ballot(uns);
        if (m) { const int hi = 63 - __clzll((long long)m); found = __shfl(y, hi, 64); cur = lo + hi; }
        else cur = lo;
      }
      if (found < 0) {
        if (sp == 0) break;
        sp--;
        int px = 0, pc_ = 0;
        if (lane == 0) { px = v.stack[2 * sp]; pc_ = v.stack[2 * sp + 1]; }
        x = __shfl(px, 0, 64); cur = __shfl(pc_, 0, 64);
      } else {
        if (lane == 0) { v.stack[2 * sp] = x; v.stack[2 * sp + 1] = cur; }
        sp++;
        visit(found);
        x = found; cur = v.adj_ptr[found + 1] - v.adj_ptr[found];
        wave_mem_sync();
      }
    }
    nb++;
    if (lane == 0) v.blk_ptr[nb] = out;
  }
  if (lane == 0) { v.blk_ptr[0] = 0; v.blk_info[0] = nb; v.blk_info[1] = 0; }
  wave_mem_sync();
}

// init_haplotypes_LD2 (phase.rs:609-671) for the blocks: first node = hap 1, every other node takes its BFS parent's
// haplotype times the sign of their pair's weight; block members are "conserved".  One wave; delta holds the random draws.
__device__ void ld_seed_wave(const ChainView& v) {
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  const int nb = v.blk_info[0];
  for (int b = 0; b < nb; b++) {
    const int n0 = v.blk_ptr[b], n1 = v.blk_ptr[b + 1];
    const int start = v.blk_nodes[n0];
    int qh = 0, qt = 1;
    if (lane == 0) { v.queue[0] = start; v.seen[start] = 2; v.dl[start] = 1; }
    wave_mem_sync();
    while (qh < qt) {
      const int nx = v.queue[qh++];
      const int hnx = v.dl[nx];
      const int a0 = v.adj_ptr[nx], a1 = v.adj_ptr[nx + 1];
      for (int base = a0; base < a1; base += 64) {
        const int idx = base + lane;
        const bool valid = idx < a1;
        const int raw = valid ? v.adj[idx] : 0;
        const int y = raw & 0x7fffffff;
        const bool und = valid && v.seen[y] != 2;
        const unsigned long long m = __ballot(und);
        if (und) { v.seen[y] = 2; v.dl[y] = (int8_t)(raw < 0 ? -hnx : hnx); v.queue[qt + __popcll(m & below)] = y; }
        qt += __popcll(m);
        wave_mem_sync();
      }
    }
    for (int k = n0 + lane; k < n1; k += 64) v.cons[v.blk_nodes[k]] = 1;
  }
  wave_mem_sync();
}

// ------------------------------------------------------------------------------------------------------------
// cross_optimize at grid scope (phase.rs:810-976): the arithmetic of k4_dev.h's cross_optimize with the state in
// HBM; the per-SNP sums are entry-balanced (a thread owns a fixed run of CSC entries for the whole launch).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ long long cross_optimize_scope(GridScope& sc, const PhaseDev& P, const RegionDev& rd, const ChainView& v, bool keep_conserved,
                                          bool with_genotype, const long long* wl, unsigned long long* macc, int e0, int e1, int i_first) {
  const int32_t* rp = v.mv.rp; const int32_t* pc = v.mv.pc; const uint8_t* pv = v.mv.pv;
  const int32_t* cp = v.mv.cp; const int32_t* cr = v.mv.cr; const uint8_t* cv = v.mv.cv;
  const uint8_t* fp = v.mv.fp; const uint8_t* cons = v.mv.cons;
  int8_t* sg = v.sg; int8_t* dl = v.dl; int8_t* et = v.et;
  const long long* scn = P.snp_const + 4ll * rd.snp_off;
  const int tid = sc.tid(), nt = sc.nt();
  bool hg_inc = true, h_inc = true;
  int iters = 0;
  while (hg_inc | h_inc) {
    int any = 0;
    for (int row = tid; row < rd.R; row += nt) {
      const int s = sg[row];
      long long diff = 0;
      for (int e = rp[row]; e < rp[row + 1]; e++) {
        const int i = pc[e];
        const uint8_t x = pv[e];
        if (et[i] == 0) { const long long w = wl[x & 31]; diff += (((x & 32) ? 1 : -1) == s * dl[i]) ? w : -w; }
      }
      if (diff < 0) { sg[row] = (int8_t)(-s); any = 1; }
      else if (diff == 0 && rp[row + 1] > rp[row] && tie_row_decide(P, rp, pc, pv, row, dl, et, s, P.lut64->le, P.lut64->l1e)) sg[row] = (int8_t)(-s);
    }
    any = sc.sync_or(any);
    if (!any) h_inc = false; else { h_inc = true; hg_inc = true; }
    if (e0 < e1) {
      int i = i_first;
      int d = dl[i]; int cend = cp[i + 1];
      long long M = 0;
      for (int e = e0; e < e1; e++) {
        if (e >= cend) {
          if (M) atomicAdd(&macc[i], (unsigned long long)M);
          M = 0;
          do { i++; cend = cp[i + 1]; } while (e >= cend);
          d = dl[i];
        }
        const uint8_t x = cv[e];
        if (((x & 32) ? 1 : -1) == sg[cr[e]] * d) M += wl[x & 31];
      }
      if (M) atomicAdd(&macc[i], (unsigned long long)M);
    }
    sc.sync();
    any = 0;
    for (int i = tid; i < rd.S; i += nt) {
      const long long M = (long long)macc[i];
      macc[i] = 0;
      const int ncol = cp[i + 1] - cp[i];
      if (!fp[i] || (keep_conserved && cons[i]) || ncol == 0) continue;
      const int d = dl[i], h = et[i];
      const long long het = P.lut.f_het0 - (long long)ncol * P.lut.f_log2;
      const long long F = scn[4 * i], Wt = scn[4 * i + 1];
      const long long N[4] = {F + M + het, F + Wt - M + het, scn[4 * i + 2] + P.lut.f_homref, scn[4 * i + 3] + P.lut.f_homvar};
      int ch;
      if (with_genotype) { ch = 0; for (int t = 1; t < 4; t++) if (N[t] > N[ch]) ch = t; }
      else if (h == 0) ch = N[1] > N[0] ? 1 : 0;
      else ch = N[3] > N[2] ? 3 : 2;
      const int cur = h == 0 ? 0 : (h == 1 ? 2 : 3);
      if (N[ch] > N[cur]) any = 1;
      dl[i] = (int8_t)(ch == 1 ? -d : d);
      et[i] = (int8_t)(ch <= 1 ? 0 : (ch == 2 ? 1 : -1));
    }
    any = sc.sync_or(any);
    if (!any) hg_inc = false; else { hg_inc = true; h_inc = true; }
    if (++iters > 20) break;
  }
  long long acc = 0;
  for (int row = tid; row < rd.R; row += nt) {
    const int s = sg[row];
    for (int e = rp[row]; e < rp[row + 1]; e++) {
      const int i = pc[e];
      const uint8_t x = pv[e];
      const int xx = et[i] == 0 ? s * dl[i] : et[i];
      if (((x & 32) ? 1 : -1) == xx) acc += wl[x & 31];
    }
  }
  return rd.f_total + sc.sync_sum(acc);
}

// exact objective of the working state (phase.rs:257-276)
template <class SC>
__device__ long long objective_scope(SC& sc, const RegionDev& rd, const ChainView& v, const long long* wl) {
  long long acc = 0;
  for (int row = sc.tid(); row < rd.R; row += sc.nt()) {
    const int s = v.sg[row];
    for (int e = v.mv.rp[row]; e < v.mv.rp[row + 1]; e++) {
      const int i = v.mv.pc[e];
      const uint8_t x = v.mv.pv[e];
      const int xx = v.et[i] == 0 ? s * v.dl[i] : v.et[i];
      if (((x & 32) ? 1 : -1) == xx) acc += wl[x & 31];
    }
  }
  return rd.f_total + sc.sync_sum(acc);
}

// ------------------------------------------------------------------------------------------------------------
// cross_optimize_by_block (phase.rs:1298-1394).  For every block: q = sum over its SNPs (block order) of
// cal_delta_eta_sigma_log on the current state, q_flip = the same with delta negated and the haplotag negated for the
// observations whose read has only block members up to and including that SNP (`flip_read`, phase.rs:1331-1349: an
// entry BEFORE the SNP that is not in the block vetoes, entries after it are not looked at yet).  q < q_flip flips the
// block's SNPs; every block rewrites the haplotag of every read (phase.rs:1364-1378), so only the version of the block
// the reference visits last (block 0 here) survives.  f64 sums in the reference's observation order.
// ------------------------------------------------------------------------------------------------------------
struct FlipLut { double le[32], l1e[32]; double p_homref, p_homvar, log_theta, log2; };
__device__ __forceinline__ double lgf(const FlipLut& L, int sigma, int delta, int eta, uint8_t x) {   // log10(aki(...)), phase.rs:32-49
  const int pp = (x & 32) ? 1 : -1, xx = eta == 0 ? sigma * delta : eta;
  return pp == xx ? L.l1e[x & 31] : L.le[x & 31];
}
constexpr int SSTR = 65;   // stage row stride in doubles

// ordered sums over the phase entries of column i (row order): lane a < 4 accumulates term a of every entry in order
template <class Term>
__device__ __forceinline__ void col_sums4(double* stg, const ChainView& v, int i, Term term, double* acc) {
  const int lane = threadIdx.x & 63;
  double mine = 0.0;
  const int kb = v.mv.cp[i], ke = v.mv.cp[i + 1];
  for (int k0 = kb; k0 < ke; k0 += 64) {
    const int k = k0 + lane;
    double t[4] = {0.0, 0.0, 0.0, 0.0};
    if (k < ke) { const int e = v.cent[k]; term(v.erow[e], v.mv.pv[e], t); }
#pragma unroll
    for (int a = 0; a < 4; a++) stg[a * SSTR + lane] = t[a];
    wave_lds_sync();
    const int nk = min(64, ke - k0);
    if (lane < 4) {
      const double* src = stg + lane * SSTR;
      int j = 0;
      for (; j + 8 <= nk; j += 8) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = src[j + u];
#pragma unroll
        for (int u = 0; u < 8; u++) mine += x[u];
      }
      for (; j < nk; j++) mine += src[j];
    }
    wave_lds_sync();
  }
#pragma unroll
  for (int a = 0; a < 4; a++)
    acc[a] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mine), a), __builtin_amdgcn_readlane(__double2loint(mine), a));
}

template <class SC>
__device__ void block_flip(SC& sc, const ChainDev& C, const ChainView& v, const FlipLut& L, double* stage /* per wave 4*SSTR */) {
  const int lane = threadIdx.x & 63;
  const int nb = __hip_atomic_load(&v.blk_info[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (nb == 0) return;   // (uniform) no block: the pass changes nothing and the objective equals launch A's
  // flip_read of (row, SNP idx) <=> idx <= flipcol[row]: the last column of the row's leading run of entries that all
  // belong to the block of its first entry
  for (int k = sc.tid(); k < v.R; k += sc.nt()) {
    const int r = v.prow_src[k];
    const int64_t eb = C.row_ptr[v.r0 + r], ee = C.row_ptr[v.r0 + r + 1];
    int last = -1;
    if (eb < ee) {
      const int b0 = v.blk_of[C.col[eb] - v.c0];
      if (b0 >= 0)
        for (int64_t e = eb; e < ee; e++) { const int ci = C.col[e] - v.c0; if (v.blk_of[ci] != b0) break; last = ci; }
    }
    v.flipcol[k] = last;
  }
  sc.sync();
  // per block SNP: the two ratio scores (ColScores of the host path = cal_delta_eta_sigma_log, phase.rs:128-176)
  double* stg = stage + (threadIdx.x >> 6) * (4 * SSTR);
  const int n_nodes = v.blk_ptr[nb];
  for (int t = sc.wave(); t < n_nodes; t += sc.nwaves()) {
    const int idx = v.blk_nodes[t];
    const int d = v.dl[idx], h = v.et[idx];
    const int n = v.mv.cp[idx + 1] - v.mv.cp[idx];
    const double p_het = n == 0 ? L.log_theta : L.log_theta - (double)(uint32_t)n * L.log2;
    auto score = [&](const double* s) -> double {   // s = {het_d, het_nd, homref, homvar}; score(+1, h)
      double q1 = h == 0 ? s[0] : (h == 1 ? s[2] : s[3]);
      q1 += h == 0 ? p_het : (h == 1 ? L.p_homref : L.p_homvar);
      const double q2 = s[3] + L.p_homvar, q3 = s[0] + p_het, q4 = s[2] + L.p_homref, q5 = s[1] + p_het;
      return 1.0 - q1 / (q2 + q3 + q4 + q5);
    };
    double a[4], b[4];
    col_sums4(stg, v, idx, [&](int k, uint8_t x, double* tt) {
      const int s = v.sg[k];
      tt[0] = lgf(L, s, d, 0, x); tt[1] = lgf(L, s, -d, 0, x); tt[2] = lgf(L, s, d, 1, x); tt[3] = lgf(L, s, d, -1, x);
    }, a);
    col_sums4(stg, v, idx, [&](int k, uint8_t x, double* tt) {
      const int s = idx <= v.flipcol[k] ? -v.sg[k] : v.sg[k];
      tt[0] = lgf(L, s, -d, 0, x); tt[1] = lgf(L, s, d, 0, x); tt[2] = lgf(L, s, -d, 1, x); tt[3] = lgf(L, s, -d, -1, x);
    }, b);
    if (lane == 0) { v.qs[idx] = score(a); v.qfs[idx] = score(b); }
  }
  sc.sync();
  // per block: sums in block order, verdict, new haplotypes (a wave per block; lane 0 adds in order)
  for (int bk = sc.wave(); bk < nb; bk += sc.nwaves()) {
    const int n0 = v.blk_ptr[bk], n1 = v.blk_ptr[bk + 1];
    double q = 0.0, qf = 0.0;
    for (int k0 = n0; k0 < n1; k0 += 64) {
      const int k = k0 + lane;
      double x = 0.0, y = 0.0;
      if (k < n1) { const int idx = v.blk_nodes[k]; x = v.qs[idx]; y = v.qfs[idx]; }
      const int nk = min(64, n1 - k0);
      for (int j = 0; j < nk; j++) {
        q += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), j), __builtin_amdgcn_readlane(__double2loint(x), j));
        qf += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(y), j), __builtin_amdgcn_readlane(__double2loint(y), j));
      }
    }
    const bool flip = q < qf;
    for (int k = n0 + lane; k < n1; k += 64) { const int idx = v.blk_nodes[k]; v.new_hap[idx] = (int8_t)(flip ? -v.dl[idx] : v.dl[idx]); }
    if (bk == 0 && lane == 0) v.blk_info[1] = flip ? 1 : 0;
  }
  sc.sync();
  // haplotags: the flipped version of block 0 if it flips -- per read the value written for the last SNP (block order)
  // it covers (phase.rs:1343-1349 inserts in SNP order, later inserts overwrite)
  const int flip0 = __hip_atomic_load(&v.blk_info[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (flip0)
    for (int k = sc.tid(); k < v.R; k += sc.nt()) {
      int best_pos = -1, best_ci = -1;
      for (int e = v.mv.rp[k]; e < v.mv.rp[k + 1]; e++) {
        const int ci = v.mv.pc[e];
        if (v.blk_of[ci] == 0 && v.blk_pos[ci] > best_pos) { best_pos = v.blk_pos[ci]; best_ci = ci; }
      }
      if (best_ci >= 0 && best_ci <= v.flipcol[k]) v.sg[k] = (int8_t)(-v.sg[k]);
    }
  for (int t = sc.tid(); t < n_nodes; t += sc.nt()) { const int idx = v.blk_nodes[t]; v.dl[idx] = v.new_hap[idx]; }
  sc.sync();
}

// ------------------------------------------------------------------------------------------------------------
// the chain, written once for both scopes
// ------------------------------------------------------------------------------------------------------------
// tie8(): called (by every thread) when a configuration's fixed-point objective EQUALS the best one's -- class 8, `prob > largest_prob`
// (phase.rs:1140-1144 ...) between configurations of equal objective; 1: the working state becomes the best one, 0: the best one stays, 2: it
// stays and the working state is known to equal it (no load_best_configuration needed -- the common case: a round falls back into the optimum).
template <class SC, class Cross, class FastRounds, class Tie8>
__device__ __forceinline__ void chain_run(SC& sc, const ChainDev& C, const RegionDev& rd, const ChainView& v, const long long* wl, const FlipLut& L,
                          double* stage, int (*sm)[16], Cross cross, FastRounds fast_rounds, Tie8 tie8, int slot) {
  const int S = rd.S, R = rd.R;
  int n_mark = 0;
  auto mark = [&]() { if (C.dbg && sc.tid() == 0 && (sc.nblk() > 1 || blockIdx.x == 0)) C.dbg[n_mark] = (long long)wall_clock64(); n_mark++; };
  mark();
  ordered_index(sc, v.R, v.S, v.mv.rp, v.mv.pc, v.mv.cp, nullptr, v.erow, v.cent, v.pcnt, v.n_parts, sm);
  mark();
  ld_pair_table(sc, C, v);
  mark();
  ld_graph(sc, v, sm);
  mark();
  // start state (phase.rs:1124-1131): random delta (draws S+F ..), genotype from the variant type, random sigma
  const uint64_t SF = (uint64_t)S + (uint64_t)R;
  for (int i = sc.tid(); i < S; i += sc.nt()) { v.dl[i] = u01(rd.seed, SF + i) < 0.5 ? 1 : -1; v.et[i] = init_genotype(v.vt[i]); }
  for (int row = sc.tid(); row < R; row += sc.nt()) v.sg[row] = u01(rd.seed, SF + S + row) < 0.5 ? -1 : 1;
  sc.sync();
  if (sc.blk() == 0 && (threadIdx.x >> 6) == 0) { ld_components_wave(v); ld_seed_wave(v); }
  sc.sync();
  mark();
  long long best = cross(true, false);
  mark();
  auto save = [&]() {
    for (int i = sc.tid(); i < S; i += sc.nt()) { v.bdl[i] = v.dl[i]; v.bet[i] = v.et[i]; }
    for (int row = sc.tid(); row < R; row += sc.nt()) v.bsg[row] = v.sg[row];
  };
  auto load = [&]() {
    for (int i = sc.tid(); i < S; i += sc.nt()) { v.dl[i] = v.bdl[i]; v.et[i] = v.bet[i]; }
    for (int row = sc.tid(); row < R; row += sc.nt()) v.sg[row] = v.bsg[row];
  };
  // `prob > largest_prob` + load_best_configuration behind a cross_optimize (phase.rs:1140-1144, 1210-1231): afterwards the working state is the best one
  auto settle = [&](long long obj) {
    if (obj > best) { best = obj; save(); return; }   // (working == best now)
    if (obj == best) { const int t8 = tie8(); if (t8 == 1) { save(); return; } if (t8 == 2) return; }
    load();
  };
  save();   // (every thread saves / loads / perturbs the same elements: no barrier between those steps)
  block_flip(sc, C, v, L, stage);
  {
    const long long obj = objective_scope(sc, rd, v, wl);
    settle(obj);   // `prob > largest_prob` (phase.rs:1140-1144), load_best_configuration
  }
  mark();
  if (fast_rounds(best)) { mark(); return; }   // (grid scope: the rounds with device-coherent state, see below)
  for (int tidx = 0; tidx <= S / 4; tidx++) {   // phase.rs:1198-1233
    const uint64_t ctr_t = 2 * SF + (uint64_t)tidx * SF;
    const bool flip = (tidx & 1) == 1;
    for (int i = sc.tid(); i < S; i += sc.nt()) {
      const double rg = u01(rd.seed, ctr_t + i);
      if (rg < 0.1) v.dl[i] = flip ? 1 : -1;
      else if (rg >= 0.9) v.dl[i] = flip ? -1 : 1;
    }
    sc.sync();
    settle(cross(false, false));
    for (int row = sc.tid(); row < R; row += sc.nt())
      if (u01(rd.seed, ctr_t + S + row) < 0.1) v.sg[row] = (int8_t)(-v.sg[row]);
    sc.sync();
    settle(cross(false, false));
  }
  mark();
  if (sc.tid() == 0) C.P.st_obj[slot] = best;
}

// ------------------------------------------------------------------------------------------------------------
// The perturbation rounds (phase.rs:1198-1233) at grid scope with DEVICE-COHERENT state.  A round is two
// cross_optimize calls of ~6 half-step pairs each, and C5 has 1 251 rounds: the barrier count decides the run time.
// Here the matrix (immutable) is read with plain cached loads and stays in the L2s, while everything that changes --
// sigma as one 64-bit word per 64 rows, delta / eta as bytes -- is read and written with agent-scope relaxed atomics
// only, so the barriers need no L2 write-back / invalidate (GridScope::sync_light).  Ownership is fixed: wave w owns
// the 64-row groups j = w (mod #waves) and the SNPs i = w (mod #waves); save / load / perturb are owner-local and need
// no barrier.  A half step first copies the other half's state into LDS (delta / eta bytes for the sigma step, the
// whole sigma bit vector for the delta step).  The delta step is one wave per SNP -- sum, decision and the SNP's term
// of the objective in one go -- so an iteration costs two barriers and the objective none.
// Same integers as cross_optimize_scope / k4_dev.h's cross_optimize (bit-identical results, tested).
// ------------------------------------------------------------------------------------------------------------
template <class T> __device__ __forceinline__ T cload(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void cstore(T* p, T x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool chain_rounds_fast(GridScope& sc, const ChainDev& C, const RegionDev& rd, const ChainView& v, const long long* wl,
                                  uint8_t* dyn, long long best, int slot, const FlipLut& FL) {
  const int S = rd.S, R = rd.R, lane = threadIdx.x & 63;
  const int ng = (R + 63) >> 6;                         // 64-row groups
  unsigned long long* wsw0 = C.sig_words + 2 * ((int64_t)(rd.sig_off >> 6) + slot);   // working sigma words (sequential form)
  unsigned long long* bsw = wsw0 + ng;                  // best sigma words
  // SPECULATIVE HALF-ROUNDS.  Every half-round of phase.rs:1198-1233 starts from the BEST state (load_best_configuration
  // after every cross_optimize) and draws from a fixed place of the random stream, and very few of them raise the best
  // objective (3 of 240 / 11 of 426 on the 100 - 200 kb islands of the tests): so `lanes` consecutive half-rounds run at the
  // same time, each on a sub-grid of every lanes-th workgroup (with 8 lanes: the workgroups of ONE XCD) with a working state
  // of its own, and are committed in order -- the first one that raises the best objective becomes the new best and the
  // half-rounds behind it are run again from there.  Same integers, same order of commits: bit-identical to the sequential
  // form (lanes = 1; tests compare the two and the oracle).
  const int lanes = (C.spec_lanes > 1 && gridDim.x % C.spec_lanes == 0 && (int)gridDim.x >= 2 * C.spec_lanes && ng <= C.spec_ng && S <= C.spec_s8) ? C.spec_lanes : 1;
  const bool spec = lanes > 1;
  const int my = spec ? (int)blockIdx.x % lanes : 0;
  GridScope lane_sc = sc;                               // the scope a half-round runs in: the whole launch, or this workgroup's lane
  if (spec) { lane_sc.c = C.spec_ctl + my; lane_sc.gen = 0; lane_sc.stride = lanes; lane_sc.first = my; }
  GridScope& sub = spec ? lane_sc : sc;
  unsigned long long* wsw = spec ? C.spec_sig + (int64_t)my * C.spec_ng : wsw0;
  int8_t* dlw = spec ? C.spec_de + (int64_t)my * 2 * C.spec_s8 : v.dl;
  int8_t* etw = spec ? dlw + C.spec_s8 : v.et;
  uint32_t* s_sig = (uint32_t*)dyn;                     // LDS: sigma bits of every row
  int8_t* s_dl = (int8_t*)(s_sig + 2 * ng); int8_t* s_et = s_dl + S;
  int8_t* s_de = s_et + S;                              // delta of the het SNPs, 0 for the others (sigma step: one byte per entry)
  const int32_t* rp = v.mv.rp; const int32_t* cp = v.mv.cp;
  // the entries as one dword each (value byte << 24 | row-in-unit << 18 | SNP in row order, value byte << 24 | row in column
  // order): a lane reads four with one 16-byte load.  (A dword + a byte load per entry kept the texture addressers as busy as the VALUs, 34 % each on C5, with the
  // matrix streaming from beyond L2 in every half step.)
  uint32_t* const pkr = C.pk_csr; uint32_t* const pkc = C.pk_csc;
  {
    const int32_t* pc = v.mv.pc; const uint8_t* pv = v.mv.pv; const int32_t* cr = v.mv.cr; const uint8_t* cv = v.mv.cv;
    const int E = cp[S];
    for (int e = sc.tid(); e < E + 8; e += sc.nt()) pkc[e] = e < E ? ((uint32_t)cv[e] << 24) | (uint32_t)cr[e] : 0u;
    // row order: value byte << 24 | row inside its 32-row unit << 18 | SNP (S < 2^18, checked by the caller)
    for (int row = sc.tid(); row < R; row += sc.nt())
      for (int e = rp[row]; e < rp[row + 1]; e++) pkr[e] = ((uint32_t)pv[e] << 24) | ((uint32_t)(row & 31) << 18) | (uint32_t)pc[e];
    for (int e = E + sc.tid(); e < E + 8; e += sc.nt()) pkr[e] = 0u;
  }
  const uint8_t* fp = v.mv.fp;
  const long long* scn = C.P.snp_const + 4ll * rd.snp_off;
  const PhaseLutDev& lut = C.P.lut;
  const int fw0 = sc.wave(), fnw = sc.nwaves();                          // ownership over the whole launch (set-up, commits)
  const int fwj0 = (int)(threadIdx.x >> 6) * sc.nblk() + sc.blk();
  const int w0 = sub.wave(), nw = sub.nwaves();                          // ownership inside a half-round's scope
  // row groups / SNPs of one workgroup lie far apart (neighbouring rows and columns are equally long: a workgroup that
  // owned a run of them would be the slowest or the fastest of every half step)
  const int wj0 = (int)(threadIdx.x >> 6) * sub.nblk() + sub.blk();
  __shared__ unsigned long long t_sum[4][8];   // delta step: partial sums / arrivals of the four-wave teams
  __shared__ long long rsum_all[CH_THREADS / 64][32];   // sigma step: row sums of the unit a wave is working on
  long long* const rsum = rsum_all[threadIdx.x >> 6];
  __shared__ unsigned t_cnt[4][8];
  if (threadIdx.x < 32) { t_sum[threadIdx.x >> 3][threadIdx.x & 7] = 0; t_cnt[threadIdx.x >> 3][threadIdx.x & 7] = 0; }
  // ---- delta step order: SNPs by column length, dealt to the teams in serpentine order (longest to team 0, 1, .. T-1,
  // the next T to team T-1 .. 0, ...): a team's share of the entries is then within a few % of the mean.  In index
  // order the slowest workgroup took 24.5 us per iteration against a median of 12 (C5: columns of 300 .. 5 000 entries,
  // 4.6 SNPs per team), and every workgroup waits for it at the barrier.
  int32_t* const ord = v.queue;   // (free after the LD seeding)
  for (int i = sc.tid(); i < S; i += sc.nt()) {
    const int len = cp[i + 1] - cp[i];
    int rank = 0;
    for (int j = 0; j < S; j++) { const int lj = cp[j + 1] - cp[j]; rank += (lj > len || (lj == len && j < i)) ? 1 : 0; }
    ord[rank] = i;
  }
  // ---- the byte state of the generic steps (best == working) into words
  sc.sync();   // (fenced: the byte arrays were written with plain stores by other workgroups)
  for (int j = fwj0; j < ng; j += fnw) {
    const int row = 64 * j + lane;
    const unsigned long long word = __ballot(row < R && v.bsg[row] == 1);
    if (lane == 0) { cstore(&wsw0[j], word); cstore(&bsw[j], word); }
  }
  sc.sync();   // (fenced: the byte arrays were written with plain stores)
  auto cross = [&]() -> long long {   // cross_optimize(keep_conserved = false, with_genotype = false)
    bool hg_inc = true, h_inc = true;
    int iters = 0;
    long long obj = 0;
    while (hg_inc | h_inc) {
      // ---- sigma step: delta / eta of every SNP from LDS
      long long tk0 = 0;
      const bool tk = C.dbg && sub.tid() == 0 && my == 0;
      auto tick = [&](int slot_) { if (tk) { const long long t = (long long)wall_clock64(); C.dbg[slot_] += t - tk0; tk0 = t; } };
      if (tk) tk0 = (long long)wall_clock64();
      for (int i = threadIdx.x; i < S; i += blockDim.x) { const int8_t dv = cload(&dlw[i]), ev = cload(&etw[i]); s_dl[i] = dv; s_et[i] = ev; s_de[i] = ev == 0 ? (dv == 0 ? (int8_t)2 : dv) : (int8_t)0; }   // (2: het with delta 0 -- never a hit)
      __syncthreads();
      tick(8);
      const long long wg_t0 = C.dbg ? (long long)wall_clock64() : 0;   // (LCR_PHASE_PROF: every workgroup's own time in the half steps)
      int any = 0;
      // a wave's unit is HALF a 64-row group (32 rows): 10 400 units on C5.  The unit's entries are one dense run of the packed row
      // array: the lanes read it four entries per load whatever the rows' lengths (every lane busy, every load coalesced), an
      // entry names its row inside the unit, a lane sums its run of entries per row and adds the run to the row's accumulator
      // in LDS (integer, order-free); lanes 0-31 then take the rows' decisions.  The units of a wave are fixed, so where each
      // begins is fetched once per step for all of them: no load of a unit waits for another.  (Before: a group of lanes per
      // row -- row pointers, then entries: two dependent trips per unit, 78 % of the lanes with an entry, DPP row sums.)
      {
        const int n_units = 2 * ng;
        for (int k0 = 0; wj0 + k0 * nw < n_units; k0 += 64) {
          const int uk = wj0 + (k0 + lane) * nw;                        // lane <-> one of the wave's next 64 units
          const int ub_l = uk < n_units ? rp[min(32 * uk, R)] : 0, ue_l = uk < n_units ? rp[min(32 * uk + 32, R)] : 0;
          // (sigma of the unit's rows as well: the unit's half of its word is written by this wave only)
          const unsigned long long wd_l = uk < n_units ? cload(&wsw[uk >> 1]) : 0ull;
          const uint32_t sb_l = (uint32_t)(wd_l >> (32 * (uk & 1)));
          for (int k = 0; k < 64 && wj0 + (k0 + k) * nw < n_units; k++) {
            const int u = wj0 + (k0 + k) * nw, j = u >> 1;
            const int ub = __shfl(ub_l, k, 64), ue = __shfl(ue_l, k, 64);
            const uint32_t sbits = (uint32_t)__shfl((int)sb_l, k, 64);   // sigma of the unit's rows
            if (lane < 32) rsum[lane] = 0;
            wave_lds_sync();
            uint32_t hm = 0;
            auto run4 = [&](const uint4& t, int e) {
              const uint32_t en[4] = {t.x, t.y, t.z, t.w};
              long long acc = 0; uint32_t cur = 32u;
#pragma unroll
              for (int x4 = 0; x4 < 4; x4++) {
                if (e + x4 < ue) {
                  const uint32_t i = en[x4] & 0x3FFFFu, rid = (en[x4] >> 18) & 31u, x = en[x4] >> 24;
                  if (rid != cur) { if (cur < 32u && acc) atomicAdd(reinterpret_cast<unsigned long long*>(&rsum[cur]), (unsigned long long)acc); cur = rid; acc = 0; }
                  // +w when the entry's allele equals sigma * delta, -w when not, nothing at a hom site: the three signs as bits
                  const int t = s_de[i];
                  hm |= (uint32_t)(t != 0) << rid;
                  const long long w = wl[x & 31];
                  const uint32_t neg = ((x >> 5) ^ (sbits >> rid) ^ ((uint32_t)t >> 31)) & 1u;
                  acc += t == 0 ? 0ll : ((neg | (uint32_t)(t == 2)) ? -w : w);
                }
              }
              if (cur < 32u && acc) atomicAdd(reinterpret_cast<unsigned long long*>(&rsum[cur]), (unsigned long long)acc);
            };
            {   // the first 1 024 entries of the unit are requested before any is used
              uint4 t4[4];
#pragma unroll
              for (int q = 0; q < 4; q++) { const int e = ub + 4 * lane + 256 * q; t4[q] = e < ue ? *reinterpret_cast<const uint4*>(pkr + e) : make_uint4(0, 0, 0, 0); }
#pragma unroll
              for (int q = 0; q < 4; q++) { const int e = ub + 4 * lane + 256 * q; if (e < ue) run4(t4[q], e); }
              for (int e = ub + 4 * lane + 1024; e < ue; e += 256) run4(*reinterpret_cast<const uint4*>(pkr + e), e);
            }
            {   // which rows of the unit have an entry at a het site: OR over the lanes (DPP row shifts / broadcasts, the total in lane 63)
              int h = (int)hm;
              h |= __builtin_amdgcn_update_dpp(0, h, 0x111, 0xf, 0xf, false);
              h |= __builtin_amdgcn_update_dpp(0, h, 0x112, 0xf, 0xf, false);
              h |= __builtin_amdgcn_update_dpp(0, h, 0x114, 0xf, 0xf, false);
              h |= __builtin_amdgcn_update_dpp(0, h, 0x118, 0xf, 0xf, false);
              h |= __builtin_amdgcn_update_dpp(0, h, 0x142, 0xa, 0xf, false);
              h |= __builtin_amdgcn_update_dpp(0, h, 0x143, 0xc, 0xf, false);
              hm = (uint32_t)__builtin_amdgcn_readlane(h, 63);
            }
            wave_lds_sync();
            const long long diff = lane < 32 ? rsum[lane] : 0;
            unsigned long long fb = __ballot(lane < 32 && diff < 0);
            if (fb) any = 1;
            // rows whose sums tie exactly, with an entry at a het site: the reference-order f64 scores decide (a lane per row;
            // a few dozen rows per step on C5)
            const uint32_t tmask = (uint32_t)__ballot(lane < 32 && diff == 0 && 32 * u + lane < R) & hm;
            if (tmask) {
              bool tf = false;
              if (lane < 32 && ((tmask >> lane) & 1u))
                tf = tie_row_decide(C.P, rp, v.mv.pc, v.mv.pv, 32 * u + lane, s_dl, s_et, ((sbits >> lane) & 1u) ? 1 : -1, FL.le, FL.l1e);
              fb |= __ballot(tf);
            }
            wave_lds_sync();
            if (fb && lane == 0) cstore(reinterpret_cast<uint32_t*>(&wsw[j]) + (u & 1), sbits ^ (uint32_t)fb);   // (this unit's half of the word)
          }
        }
      }
      __syncthreads();
      if (C.dbg && threadIdx.x == 0 && my == 0 && sub.blk() < 1024) C.dbg[16 + sub.blk()] += (long long)wall_clock64() - wg_t0;
      tick(9);
      any = sub.sync_or_light(any);
      tick(10);
      if (!any) h_inc = false; else { h_inc = true; hg_inc = true; }
      // ---- delta / eta step: one wave per SNP, sigma bits of every row from LDS
      for (int j = threadIdx.x; j < ng; j += blockDim.x) { const unsigned long long word = cload(&wsw[j]); s_sig[2 * j] = (uint32_t)word; s_sig[2 * j + 1] = (uint32_t)(word >> 32); }
      __syncthreads();
      tick(11);
      const long long wg_t1 = C.dbg ? (long long)wall_clock64() : 0;
      any = 0;
      long long acc = 0;
      // a team of four waves per SNP (a wave per SNP leaves the largest column as the critical path); the waves' partial
      // sums meet in LDS and the last one to arrive decides -- no barrier inside the loop
      {
        const int team = threadIdx.x >> 8, wt = (threadIdx.x >> 6) & 3, nteams = sub.nblk() * 4, gteam = team * sub.nblk() + sub.blk();
        // the team's SNPs are fixed: which they are and where their columns lie is fetched once for 64 of them (a lane each), not
        // SNP by SNP in front of the column's entries (order -> column pointers -> entries were three dependent trips to memory
        // for each of a team's ~37 SNPs: the whole step's length)
        int it = 0, i_l = -1, c0_l = 0, c1_l = 0, fp_l = 0;
        long long F_l = 0, Wt_l = 0, D2_l = 0, D3_l = 0;   // the SNP's constants travel with its pointers (the deciding lane used to fetch them last)
        auto shfl_ll = [&](long long x, int k) -> long long {
          return ((long long)__shfl((int)(x >> 32), k, 64) << 32) | (unsigned int)__shfl((int)x, k, 64);
        };
        for (int base = 0; base < S; base += nteams, it++) {
          if ((it & 63) == 0) {
            const int itl = it + lane, pl = itl * nteams + ((itl & 1) ? nteams - 1 - gteam : gteam);
            i_l = ((int64_t)itl * nteams < S && pl < S) ? ord[pl] : -1;
            c0_l = i_l >= 0 ? cp[i_l] : 0; c1_l = i_l >= 0 ? cp[i_l + 1] : 0;
            if (i_l >= 0) { F_l = scn[4 * i_l]; Wt_l = scn[4 * i_l + 1]; D2_l = scn[4 * i_l + 2]; D3_l = scn[4 * i_l + 3]; fp_l = fp[i_l]; }
          }
          if ((it & 7) == 0 && it) __syncthreads();   // the ring of 8 slots per team wraps (uniform trip count)
          const int k = it & 63;
          const int i = __shfl(i_l, k, 64);
          const int c0 = __shfl(c0_l, k, 64), c1 = __shfl(c1_l, k, 64);
          if (i < 0) continue;
          if (c0 == c1) continue;
          const int d = s_dl[i], h = s_et[i];
          const uint32_t dneg = d < 0 ? 1u : 0u;
          const bool dzero = d == 0;   // (never a hit)
          long long M = 0;   // sum of w over the entries with p == sigma * d
          auto run4 = [&](const uint4& t, int e) {
            const uint32_t en[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const uint32_t row = en[u] & 0xFFFFFFu, x = en[u] >> 24;
              const uint32_t miss = ((x >> 5) ^ (s_sig[row >> 5] >> (row & 31)) ^ dneg) & 1u;   // allele != sigma * d
              if (e + u < c1 && !miss && !dzero) M += wl[x & 31];
            }
          };
          // (requesting the NEXT column's entries under this column's sums was measured: 47 instead of 8 spilled VGPRs, 277 vs 274 ms)
          {   // four consecutive entries per lane and load; the column's first 2 048 entries (C5: all of them) leave together
            const int ea = c0 + 4 * (64 * wt + lane), eb2 = ea + 1024;
            const uint4 ta = ea < c1 ? *reinterpret_cast<const uint4*>(pkc + ea) : make_uint4(0, 0, 0, 0);
            const uint4 tb = eb2 < c1 ? *reinterpret_cast<const uint4*>(pkc + eb2) : make_uint4(0, 0, 0, 0);
            if (ea < c1) run4(ta, ea);
            if (eb2 < c1) run4(tb, eb2);
            for (int e = ea + 2048; e < c1; e += 1024) run4(*reinterpret_cast<const uint4*>(pkc + e), e);
          }
          M = wave_sum_ll_dpp(M);
          const long long F = shfl_ll(F_l, k), Wt = shfl_ll(Wt_l, k), D2 = shfl_ll(D2_l, k), D3 = shfl_ll(D3_l, k);
          const int fpi = __shfl(fp_l, k, 64);
          if (lane == 0) {
            unsigned long long* ts = &t_sum[team][it & 7];
            unsigned* tc = &t_cnt[team][it & 7];
            __hip_atomic_fetch_add(ts, (unsigned long long)M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned arrived = __hip_atomic_fetch_add(tc, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (arrived == 3) {
              M = (long long)__hip_atomic_load(ts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              __hip_atomic_store(ts, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              __hip_atomic_store(tc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              // data terms of (d,0) (-d,0) (d,+1) (d,-1); with_genotype is false: a het site stays het (it may flip), a hom
              // site may change between homref and homvar.  The priors are equal inside a class, so the data terms decide.
              const long long D0 = F + M, D1 = F + Wt - M;
              long long chosen = h == 0 ? D0 : (h == 1 ? D2 : D3);
              if (fpi) {
                if (h == 0) { if (D1 > D0) { chosen = D1; any = 1; cstore(&dlw[i], (int8_t)(-d)); } }
                else {
                  const long long n2 = D2 + lut.f_homref, n3 = D3 + lut.f_homvar;
                  const bool to3 = n3 > n2;                       // first maximum wins: homref on a tie
                  const long long ncur = h == 1 ? n2 : n3, nch = to3 ? n3 : n2;
                  if (nch > ncur) any = 1;
                  if (to3 != (h == -1)) cstore(&etw[i], (int8_t)(to3 ? -1 : 1));
                  chosen = to3 ? D3 : D2;
                }
              }
              acc += chosen;
            }
          }
        }
      }
      __syncthreads();
      if (C.dbg && threadIdx.x == 0 && my == 0 && sub.blk() < 1024) C.dbg[16 + 1024 + sub.blk()] += (long long)wall_clock64() - wg_t1;
      tick(12);
      // (summing the objective once after the last iteration instead -- the sum costs this barrier two more device
      // round trips, 17.9 vs 11.7 us -- is a wash: 3.1 iterations per call save what the closing barrier costs)
      any = sub.sync_or_sum_light(any, acc, &obj);
      tick(13);
      if (!any) hg_inc = false; else { hg_inc = true; h_inc = true; }
      if (++iters > 20) break;
    }
    if (C.dbg && sub.tid() == 0) C.dbg[15] += iters;
    return obj;   // = f_total + sum of w over the hits: every phase entry lies in exactly one column
  };
  const uint64_t SF = (uint64_t)S + (uint64_t)R;
  // the state of half-round h (0 .. 2 T - 1: even = delta perturbation, odd = sigma flips of round h / 2) applied to the working
  // copy of this scope, which holds the best state; owner-local, no barrier inside
  auto perturb = [&](int h) {
    const int tidx = h >> 1;
    const uint64_t ctr_t = 2 * SF + (uint64_t)tidx * SF;
    const bool flip = (tidx & 1) == 1;
    if ((h & 1) == 0) {
      for (int i = w0; i < S; i += nw)
        if (lane == 0) {
          const double rg = u01(rd.seed, ctr_t + i);
          if (rg < 0.1) cstore(&dlw[i], (int8_t)(flip ? 1 : -1));
          else if (rg >= 0.9) cstore(&dlw[i], (int8_t)(flip ? -1 : 1));
        }
    } else {
      for (int j = wj0; j < ng; j += nw) {
        const int row = 64 * j + lane;
        const unsigned long long word = cload(&wsw[j]);
        const unsigned long long fm = __ballot(row < R && u01(rd.seed, ctr_t + S + row) < 0.1);
        if (lane == 0 && fm) cstore(&wsw[j], word ^ fm);
      }
    }
  };
  auto load = [&]() {   // best -> this scope's working copy (owner-local)
    for (int j = wj0; j < ng; j += nw) if (lane == 0) cstore(&wsw[j], cload(&bsw[j]));
    for (int i = w0; i < S; i += nw) if (lane == 0) { cstore(&dlw[i], cload(&v.bdl[i])); cstore(&etw[i], cload(&v.bet[i])); }
  };
  const int H = 2 * (S / 4 + 1);
  if (!spec) {
    auto save = [&]() {
      for (int j = wj0; j < ng; j += nw) if (lane == 0) cstore(&bsw[j], cload(&wsw[j]));
      for (int i = w0; i < S; i += nw) if (lane == 0) { cstore(&v.bdl[i], cload(&dlw[i])); cstore(&v.bet[i], cload(&etw[i])); }
    };
    for (int h = 0; h < H; h++) {   // (the working copy holds the best state here: set-up, or the load below)
      perturb(h);
      sub.sync_light();
      const long long obj = cross();
      if (obj > best) { best = obj; save(); }
      load();
    }
  } else {
    for (int h = 0; h < H;) {
      const int mine = h + my;
      if (mine < H) {
        load();                 // (the best state was settled before the barrier that ended the previous batch)
        perturb(mine);          // (owner-local on top of the owner's own loads: no barrier between them)
        sub.sync_light();
        const long long obj = cross();
        if (sub.tid() == 0) cstore(&C.spec_res[my], obj);
      } else if (sub.tid() == 0) cstore(&C.spec_res[my], (long long)LLONG_MIN);
      sc.sync_light();
      // commit in order: the first half-round of the batch that raises the best objective
      int win = -1;
      long long wobj = best;
      for (int l = 0; l < lanes; l++) { const long long o = cload(&C.spec_res[l]); if (win < 0 && o > best) { win = l; wobj = o; } }
      if (win >= 0) {
        best = wobj;
        const unsigned long long* ssrc = C.spec_sig + (int64_t)win * C.spec_ng;
        const int8_t* dsrc = C.spec_de + (int64_t)win * 2 * C.spec_s8; const int8_t* esrc = dsrc + C.spec_s8;
        for (int j = fwj0; j < ng; j += fnw) if (lane == 0) cstore(&bsw[j], cload(&ssrc[j]));
        for (int i = fw0; i < S; i += fnw) if (lane == 0) { cstore(&v.bdl[i], cload(&dsrc[i])); cstore(&v.bet[i], cload(&esrc[i])); }
        h += win + 1;
      } else h += lanes;
      sc.sync_light();          // the new best is in place (and spec_res may be written again)
    }
  }
  // ---- result: best sigma words back to bytes (delta / eta best arrays are up to date)
  for (int j = fwj0; j < ng; j += fnw) {
    const int row = 64 * j + lane;
    const unsigned long long word = cload(&bsw[j]);
    if (row < R) v.bsg[row] = ((word >> lane) & 1ull) ? 1 : -1;
  }
  if (sc.tid() == 0) C.P.st_obj[slot] = best;
  return true;
}

#include "k4_grid_batch.h"

__device__ __forceinline__ void load_flip_lut(const ChainDev& C, FlipLut* L) {
  if (threadIdx.x < 32) { L->le[threadIdx.x] = threadIdx.x < 31 ? C.le[threadIdx.x] : 0.0; L->l1e[threadIdx.x] = threadIdx.x < 31 ? C.l1e[threadIdx.x] : 0.0; }
  if (threadIdx.x == 0) { L->p_homref = C.p_homref; L->p_homvar = C.p_homvar; L->log_theta = C.log_theta; L->log2 = C.log2; }
}

// The f64 objectives of two configurations (phase.rs:257-276): ONE running sum each over the entries' terms in row order.  The terms of both --
// the table value picked by "does the entry's allele match" -- are written to global scratch by a thread per row, then wave 0 adds them up in
// order, 64 entries per coalesced 16-byte load, the two sums as two independent chains of v_add_f64 fed by v_readlane (padding terms are
// + 0.0).  (Only at class-8 ties whose match bits differ; a first form walked the rows on wave 0 alone -- four dependent loads per row -- and
// made the phase stage of C3 0.92 -> 2.3 ms.)
__device__ __forceinline__ bool objective_f64_greater_wg(const PhaseDev& P, const MatView& mv, int R, const int8_t* sg, const int8_t* dl, const int8_t* et,
                                                         const int8_t* bsg, const int8_t* bdl, const int8_t* bet, double* T, int* s_out) {
  const int lane = threadIdx.x & 63;
  const double* const le64 = P.lut64->le; const double* const l1e64 = P.lut64->l1e;
  for (int row = threadIdx.x; row < R; row += blockDim.x) {
    const int sc_ = sg[row], sb_ = bsg[row];
    for (int e = mv.rp[row]; e < mv.rp[row + 1]; e++) {
      const int i = mv.pc[e];
      const uint8_t v = mv.pv[e];
      const int p = (v & 32) ? 1 : -1, q = v & 31, hc = et[i], hb = bet[i];
      const double2 t = make_double2(p == (hc == 0 ? sc_ * dl[i] : hc) ? l1e64[q] : le64[q], p == (hb == 0 ? sb_ * bdl[i] : hb) ? l1e64[q] : le64[q]);
      reinterpret_cast<double2*>(T)[e] = t;
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int E = mv.rp[R];
    const double2* T2 = reinterpret_cast<const double2*>(T);
    double a = 0.0, b = 0.0;
    double2 nxt = lane < E ? T2[lane] : make_double2(0.0, 0.0);
    for (int base = 0; base < E; base += 64) {
      const double2 cur = nxt;
      nxt = base + 64 + lane < E ? T2[base + 64 + lane] : make_double2(0.0, 0.0);
      const int xl = (int)__double2loint(cur.x), xh = (int)__double2hiint(cur.x), yl = (int)__double2loint(cur.y), yh = (int)__double2hiint(cur.y);
#pragma unroll
      for (int k = 0; k < 64; k++) {
        a += __hiloint2double(__builtin_amdgcn_readlane(xh, k), __builtin_amdgcn_readlane(xl, k));
        b += __hiloint2double(__builtin_amdgcn_readlane(yh, k), __builtin_amdgcn_readlane(yl, k));
      }
    }
    if (lane == 0) *s_out = a > b ? 1 : 0;
  }
  __syncthreads();
  const int r = *s_out;
  __syncthreads();
  return r != 0;
}

// one workgroup of sixteen waves per chain region; the working state (and the matrix, when it fits) lives in dynamic
// LDS.  (Eight-wave workgroups, two regions per CU, were measured for batches with more chain regions than CUs -- 368 on
// the ONT-dRNA C3-shaped batch: every serial step of a region gets longer, phase stage 1.89 ms instead of 1.62 ms.)
// Ties (round 6).  The first pass over a region -- the fast form of cross_optimize -- decides sigma ties by the reference-order f64 scores
// (class 1) and configurations of equal objective by their f64 sums (class 8, tie8 below), and COUNTS the class-2 ties (a delta / eta choice
// with two equal maxima) and class-4 steps (only tie changes) it meets, which it leaves at "first maximum" / "no improvement".  A region that
// met one is run again by the same workgroup with the plain form of cross_optimize and its complete tie contract (k4_dev.h; f64 scores
// through global scratch, C.tie_qrow / tie_qsnp / tie_ch): the whole chain, the draws are counters.  A region that met none took no decision
// the complete contract takes differently, so its result stands.  Without C.tie_flag (debug key "chain_ties" = 0) the counts go to the census
// as unresolved.  (The second pass as a launch of its own behind the first, workgroups of unflagged regions leaving at once: 100 us on C3
// although nothing ran -- sixteen-wave workgroups with 64 KB of LDS wait for a whole free CU beside the next batch's K0.)
template <int NT>
__global__ void __launch_bounds__(NT) k4_chain_wg(ChainDev C, int32_t first, int32_t n) {
  constexpr int NW = NT / 64;
  constexpr int MACC = CROSS_MACC;
  __shared__ long long red[NW];
  __shared__ unsigned long long macc[MACC];
  __shared__ long long wl[32];
  __shared__ FlipLut L;
  __shared__ int sm[2][16];
  __shared__ double stage[NW * 4 * SSTR];
  extern __shared__ __attribute__((aligned(16))) int8_t dyn_state[];
  if ((int)blockIdx.x >= n) return;
  const ChainDesc d = C.desc[first + blockIdx.x];
  __shared__ unsigned long long s_tie[3];   // class-2 ties, class-4 steps, class-8 compares the fast instantiation met
  __shared__ int s_f64;
  if (threadIdx.x < 3) s_tie[threadIdx.x] = 0;
  for (int i = threadIdx.x; i < MACC; i += blockDim.x) macc[i] = 0;
  load_flip_lut(C, &L);
  load_w(C.P, wl);
  const RegionDev rd = C.P.reg[d.slot];
  const uint32_t E = (uint32_t)C.P.prow_ptr[rd.rp_off + rd.R];
  __shared__ int wg_bc;
  constexpr int SCN = 256;   // SNPs whose constants (four words each, read by every delta step) are kept in LDS
  __shared__ long long scn[4 * SCN];
  if (rd.S <= SCN) for (int i = threadIdx.x; i < 4 * rd.S; i += blockDim.x) scn[i] = C.P.snp_const[4ll * rd.snp_off + i];
  // Two instantiations of the same body: with state and matrix in LDS every pointer of the sweeps has ONE provenance,
  // so the compiler emits ds_read / ds_write for them; a pointer that is "LDS or HBM" at run time makes them flat_load /
  // flat_store, which take the long way round even when they hit LDS (the sweeps of cross_optimize were 3x slower).
  auto body = [&](auto in_lds, auto complete_pass) {
    constexpr bool COMPLETE = decltype(complete_pass)::value;
    ChainView v = make_view(C, d, rd);
    MatView mvl = v.mv;
    if constexpr (decltype(in_lds)::value) {
      v.sg = dyn_state; v.dl = v.sg + rd.R; v.et = v.dl + rd.S;
      mvl = stage_view(C.P, rd, (uint8_t*)dyn_state + C.P.scratch_stride, E);   // the chain sweeps it dozens of times
      v.mv = mvl;
      v.cons = const_cast<uint8_t*>(mvl.cons);   // (written by the LD seeding, read by cross_optimize: the LDS copy is the live one)
    } else if (C.P.lds_state) { v.sg = dyn_state; v.dl = v.sg + rd.R; v.et = v.dl + rd.S; }
    WgScope sc{red, &wg_bc};
    sc.scratch = reinterpret_cast<uint8_t*>(stage); sc.scratch_bytes = (uint32_t)sizeof(stage);   // (block_flip's stage rows: free before it)
    auto cross = [&](bool keep_conserved, bool with_genotype) -> long long {
      const bool timed = C.dbg && blockIdx.x == 0 && threadIdx.x == 0;   // (LCR_PHASE_PROF)
      int iters = 0;
      long long obj;
      if constexpr (COMPLETE)
        obj = cross_optimize(C.P, rd, mvl, v.sg, v.dl, v.et, keep_conserved, with_genotype, red, wl, macc, MACC, nullptr, 0,
                             rd.S <= SCN ? scn : nullptr, &iters, nullptr, nullptr, C.tie_qrow + 2ll * rd.sig_off, nullptr,
                             C.tie_qsnp + 2ll * rd.snp_off, C.tie_ch + 2ll * rd.snp_off);
      else
        obj = cross_optimize(C.P, rd, mvl, v.sg, v.dl, v.et, keep_conserved, with_genotype, red, wl, macc, MACC,
                             reinterpret_cast<unsigned long long*>(stage), NW * 4 * SSTR,   // (free outside block_flip)
#if defined(CHAIN_ABL) && (CHAIN_ABL & 4)
                             rd.S <= SCN ? scn : nullptr, &iters, timed ? C.dbg + 8 : nullptr, &sm[0][0], nullptr, nullptr);   // (measurement build: ties to the census)
#else
                             rd.S <= SCN ? scn : nullptr, &iters, timed ? C.dbg + 8 : nullptr, &sm[0][0], nullptr, s_tie);
#endif
      if (timed) { C.dbg[14] += 1; C.dbg[15] += iters; }
      return obj;
    };
    // Class 8: a configuration whose fixed-point objective equals the best one's.  The reference compares the f64 sums of the two, and those
    // are one running sum each over the entries' terms, a term = a table value picked by "does the entry's allele match": two states with
    // the same match bits (the same state; the mirrored one; states that differ in the sigma of rows without a het entry -- nearly every
    // equal compare on gene batches) have the same sum.  So: (1) the states, element by element (every thread compares what it saved
    // itself); (2) the match bits, a thread per row; (3) only if those differ the two sums, wave 0 in the reference's order -- decided in
    // place by both instantiations (without C.tie_flag: counted as unresolved).
    auto tie8 = [&]() -> int {
#if defined(CHAIN_ABL) && (CHAIN_ABL & 1)
      return 0;   // (measurement build: no class-8 handling)
#endif
      // (the best state lives in global memory: step 1 leaves a copy of its delta / eta in the stage rows -- free between cross_optimize
      // calls -- so that step 2's two reads per entry stay in LDS; read per entry from L2 they made the kernel 575 -> 660 us on C3)
      int8_t* const lb = reinterpret_cast<int8_t*>(stage);
      const bool copy = 2 * (size_t)rd.S <= sizeof(stage);
      int diff = 0;
      for (int i = threadIdx.x; i < rd.S; i += blockDim.x) {
        const int8_t bd = v.bdl[i], be = v.bet[i];
        if (copy) { lb[i] = bd; lb[rd.S + i] = be; }
        diff |= (v.dl[i] != bd) | (v.et[i] != be);
      }
      for (int row = threadIdx.x; row < rd.R; row += blockDim.x) diff |= v.sg[row] != v.bsg[row];
      if (!__syncthreads_or(diff)) return 2;
      auto match_differs = [&](const int8_t* bdl, const int8_t* bet) {
        int df = 0;
        for (int row = threadIdx.x; row < rd.R; row += blockDim.x) {
          const int sc_ = v.sg[row], sb_ = v.bsg[row];
          for (int e = mvl.rp[row]; e < mvl.rp[row + 1]; e++) {
            const int i = mvl.pc[e], p = (mvl.pv[e] & 32) ? 1 : -1;
            const int hc = v.et[i], hb = bet[i];
            df |= (p == (hc == 0 ? sc_ * v.dl[i] : hc)) != (p == (hb == 0 ? sb_ * bdl[i] : hb));
          }
        }
        return df;
      };
      diff = copy ? match_differs(lb, lb + rd.S) : match_differs(v.bdl, v.bet);
      if (!__syncthreads_or(diff)) return 0;
      if (!C.tie_flag) { if (threadIdx.x == 0) s_tie[2]++; return 0; }
      if (threadIdx.x == 0) TIE_COUNT(C.P.tie_ctr, TIE_BEST_F64, 1ull);
      return objective_f64_greater_wg(C.P, mvl, rd.R, v.sg, v.dl, v.et, v.bsg, v.bdl, v.bet, C.tie_terms + 2ll * d.term_off, &s_f64) ? 1 : 0;
    };
    chain_run(sc, C, rd, v, wl, L, stage, sm, cross, [](long long) { return false; }, tie8, d.slot);
  };
  const bool mat_in_lds = C.P.lds_state && matview_bytes(rd.R, rd.S, E) <= (uint32_t)C.P.lds_mat;
  if (mat_in_lds) body(std::true_type{}, std::false_type{});
  else body(std::false_type{}, std::false_type{});
  __syncthreads();
  const bool met = (s_tie[0] | s_tie[1]) != 0;
  if (!C.tie_flag) {
    if (threadIdx.x == 0) {
      if (s_tie[0]) TIE_COUNT(C.P.tie_ctr, TIE_DELTA_UNRES, s_tie[0]);
      if (s_tie[1]) TIE_COUNT(C.P.tie_ctr, TIE_STEP_UNRES, s_tie[1]);
      if (s_tie[2]) TIE_COUNT(C.P.tie_ctr, TIE_BEST_UNRES, s_tie[2]);
    }
    return;
  }
  if (!met) return;
#if defined(CHAIN_ABL) && (CHAIN_ABL & 2)
  return;   // (measurement build: no second pass in the kernel)
#else
  if (mat_in_lds) body(std::true_type{}, std::true_type{});
  else body(std::false_type{}, std::true_type{});
#endif
}

// all workgroups of the launch on one region (desc[which])
__global__ void __launch_bounds__(CH_THREADS) k4_chain_grid(ChainDev C, int32_t which) {
  __shared__ long long red[CH_WAVES];
  __shared__ unsigned long long bc[2];
  __shared__ long long wl[32];
  __shared__ FlipLut L;
  __shared__ int sm[2][16];
  __shared__ double stage[CH_WAVES * 4 * SSTR];
  load_flip_lut(C, &L);
  load_w(C.P, wl);
  const ChainDesc d = C.desc[which];
  const RegionDev rd = C.P.reg[d.slot];
  const ChainView v = make_view(C, d, rd);
  GridScope sc{C.ctl, red, bc, 0u};
  unsigned long long* macc = C.macc + rd.snp_off;
  for (int i = sc.tid(); i < rd.S; i += sc.nt()) macc[i] = 0;
  // this thread's fixed run of CSC entries and the column its first entry lies in
  const int E = v.mv.cp[rd.S];
  const int c = (int)(((int64_t)E + sc.nt() - 1) / sc.nt());
  const int e0 = (int)min((int64_t)E, (int64_t)sc.tid() * c), e1 = min(E, e0 + c);
  int i_first = 0;
  if (e0 < e1) { int lo = 0, hi = rd.S; while (lo < hi) { const int mid = (lo + hi) >> 1; if (v.mv.cp[mid + 1] <= e0) lo = mid + 1; else hi = mid; } i_first = lo; }
  sc.sync();
  auto cross = [&](bool keep_conserved, bool with_genotype) -> long long {
    return cross_optimize_scope(sc, C.P, rd, v, keep_conserved, with_genotype, wl, macc, e0, e1, i_first);
  };
  extern __shared__ __attribute__((aligned(16))) uint8_t dyn_fast[];
  auto fast_rounds = [&](long long best) -> bool {
    if (!C.pk_csr || (int64_t)v.mv.cp[rd.S] > C.pk_cap || rd.S >= (1 << 18) - 1 || rd.R >= (1 << 24)) return false;
    if (d.batch_lds && C.spec_batch && C.bt_pk4 && (int)gridDim.x <= K4_GRID_BATCH_MAX_WG)
      return chain_rounds_batch(sc, C, rd, v, wl, dyn_fast, best, d.slot, L);
    if (!d.fast_lds) return false;
    return chain_rounds_fast(sc, C, rd, v, wl, dyn_fast, best, d.slot, L);
  };
  chain_run(sc, C, rd, v, wl, L, stage, sm, cross, fast_rounds, []() { return 0; }, d.slot);
}


// ------------------------------------------------------------------------------------------------------------
// k4_stage_grid: k4_stage (k4_phase.hip) for ONE large region with all CUs: phasing rows x phase sites as CSR + CSC,
// per-SNP constants, region descriptor.  Workgroup b owns a contiguous slab of fragment rows; slab totals give every
// slab its offsets, so the CSR comes out in row order; the CSC is filled through per-column cursors (any order inside
// a column: its consumers only sum, and the chain kernel builds its own row-ordered index).
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(CH_THREADS) k4_stage_grid(StageIn in, StageOut out, PhaseLutDev lut, int32_t g, GridCtl* ctl, int32_t* blk_tot) {
  __shared__ long long red[CH_WAVES];
  __shared__ unsigned long long bc[2];
  __shared__ int sm[2][16];
  __shared__ int s_sum[5];
  __shared__ long long s_fe[32], s_f1e[32];
  GridScope sc{ctl, red, bc, 0u};
  const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.x, nb = gridDim.x;
  const int r0 = in.row_region_off[g], nrow = in.row_region_off[g + 1] - r0;
  const int c0 = in.cand_off[g], S = in.cand_off[g + 1] - c0;
  const int64_t e_base = in.row_ptr[r0];
  const int64_t E_all = in.row_ptr[r0 + nrow] - e_base;
  RegionDev rd{};
  rd.S = S; rd.rp_off = r0 + g; rd.cp_off = c0 + g; rd.e_off = e_base; rd.sig_off = r0; rd.snp_off = c0;
  rd.seed = region_seed(in.seed, in.start0[g]);
  if (tid < 32) { s_fe[tid] = tid < 31 ? lut.fe[tid] : 0; s_f1e[tid] = tid < 31 ? lut.f1e[tid] : 0; }
  int32_t* wmax = blk_tot + 2 * nb;
  for (int i = sc.tid(); i < S; i += sc.nt()) {
    const lcr_candidate& c = in.cand[c0 + i];
    out.snp_fp[c0 + i] = (c.flags & LCR_F_FOR_PHASING) ? 1 : 0;
    out.snp_vt[c0 + i] = (int8_t)c.variant_type;
    out.snp_cons[c0 + i] = 0;
    out.cursor[c0 + i] = 0;
  }
  if (sc.tid() == 0) *wmax = 0;
  sc.sync();
  const uint8_t* fp = out.snp_fp + c0;
  const int rs = (((nrow + nb - 1) / nb) + 63) & ~63;         // rows per slab
  const int s0 = min(nrow, b * rs), s1 = min(nrow, s0 + rs);
  auto row_info = [&](int r, int& isp, int& cnt, int& span) {
    isp = in.links[r0 + r] >= in.min_linkers ? 1 : 0;
    cnt = 0; span = 0;
    int first = -1, last = -1;
    for (int64_t e = in.row_ptr[r0 + r]; e < in.row_ptr[r0 + r + 1]; e++) { const int ci = in.col[e] - c0; if (fp[ci]) { cnt++; if (first < 0) first = ci; last = ci; } }
    if (last > first) span = last - first;
    if (!isp) cnt = 0;
  };
  // ---- slab totals
  if (tid < 5) s_sum[tid] = 0;
  __syncthreads();
  {
    int rows = 0, ents = 0, w = 0;
    for (int r = s0 + tid; r < s1; r += CH_THREADS) { int isp, cnt, span; row_info(r, isp, cnt, span); rows += isp; ents += cnt; w = max(w, span); }
    atomicAdd(&s_sum[0], rows); atomicAdd(&s_sum[1], ents); atomicMax(&s_sum[2], w);
  }
  __syncthreads();
  if (tid == 0) { blk_tot[2 * b] = s_sum[0]; blk_tot[2 * b + 1] = s_sum[1]; if (s_sum[2]) atomicMax(wmax, s_sum[2]); }
  sc.sync();
  // ---- offsets of this slab, totals of the region
  if (tid < 5) s_sum[tid] = 0;
  __syncthreads();
  for (int k = tid; k < nb; k += CH_THREADS) {
    const int rr = blk_tot[2 * k], ee = blk_tot[2 * k + 1];
    if (k < b) { atomicAdd(&s_sum[0], rr); atomicAdd(&s_sum[1], ee); }
    atomicAdd(&s_sum[3], rr); atomicAdd(&s_sum[4], ee);
  }
  __syncthreads();
  int R = s_sum[0], E = s_sum[1];
  const int R_tot = s_sum[3], E_tot = s_sum[4];
  int32_t* prp = out.prow_ptr + rd.rp_off;
  int32_t* pcp = out.ccol_ptr + rd.cp_off;
  // ---- CSR of the slab (row order), column counts
  for (int base = s0; base < s1; base += CH_THREADS) {
    const int r = base + tid;
    int isp = 0, cnt = 0, span = 0;
    if (r < s1) row_info(r, isp, cnt, span);
    int k, eo, tk, te;
    block_scan2n<CH_WAVES, 16>(isp, cnt, k, eo, tk, te, sm);
    if (isp) {
      k += R; eo += E;
      prp[k] = eo;
      out.prow_src[r0 + k] = r;
      for (int64_t e = in.row_ptr[r0 + r]; e < in.row_ptr[r0 + r + 1]; e++) {
        const int ci = in.col[e] - c0;
        if (!fp[ci]) continue;
        out.pcol[e_base + eo] = ci; out.pval[e_base + eo] = in.val[e] & 63;
        atomicAdd(&out.cursor[c0 + ci], 1);
        eo++;
      }
    }
    R += tk; E += te;
  }
  if (b == nb - 1 && tid == 0) prp[R_tot] = E_tot;
  sc.sync();
  // ---- column offsets (one workgroup)
  if (b == 0) {
    int carry = 0;
    for (int base = 0; base < S; base += CH_THREADS) {
      const int i = base + tid;
      const int x = i < S ? out.cursor[c0 + i] : 0;
      int ex, d0, tot, d1;
      block_scan2n<CH_WAVES, 16>(x, 0, ex, d0, tot, d1, sm);
      if (i < S) { pcp[i] = carry + ex; out.cursor[c0 + i] = carry + ex; }
      carry += tot;
    }
    if (tid == 0) pcp[S] = carry;
  }
  sc.sync();
  // ---- CSC mirror (phasing-row index, value)
  for (int k = sc.tid(); k < R_tot; k += sc.nt())
    for (int e = prp[k]; e < prp[k + 1]; e++) {
      const int pos = atomicAdd(&out.cursor[c0 + out.pcol[e_base + e]], 1);
      out.crow[e_base + pos] = k; out.cval[e_base + pos] = out.pval[e_base + e];
    }
  sc.sync();
  // ---- per-SNP constants: F = sum fe, W = sum w, Cref = sum (p==+1 ? f1e : fe), Cvar = sum (p==-1 ? f1e : fe)
  long long ft = 0;
  for (int i = sc.wave(); i < S; i += sc.nwaves()) {
    long long F = 0, W = 0, Cr = 0, Cv = 0;
    for (int e = pcp[i] + lane; e < pcp[i + 1]; e += 64) {
      const uint8_t x = out.cval[e_base + e];
      const long long fe = s_fe[x & 31], f1 = s_f1e[x & 31];
      F += fe; W += f1 - fe;
      Cr += (x & 32) ? f1 : fe; Cv += (x & 32) ? fe : f1;
    }
    F = wave_sum_ll_dpp(F); W = wave_sum_ll_dpp(W); Cr = wave_sum_ll_dpp(Cr); Cv = wave_sum_ll_dpp(Cv);
    if (lane == 0) { long long* scn = out.snp_const + 4ll * (c0 + i); scn[0] = F; scn[1] = W; scn[2] = Cr; scn[3] = Cv; ft += F; }
  }
  const long long ftot = sc.sync_sum(ft);
  if (sc.tid() == 0) {
    rd.R = R_tot; rd.f_total = ftot;
    out.reg[g] = rd;
    out.stat[g] = StageStat{R_tot, E_tot, INT_MAX, INT_MAX, (int)std::min<int64_t>(E_all, INT_MAX), *wmax};
  }
}

// ------------------------------------------------------------------------------------------------------------
// k4_gpost: the post-phase sequence (k4_post.h) with all CUs on ONE region, the region image in HBM
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(CH_THREADS) k4_gpost(PostIn in, PostScratch ps, int32_t g, PostLut lut) {
  __shared__ long long red[CH_WAVES];
  __shared__ unsigned long long bc[2];
  __shared__ int sm[2][16];
  __shared__ double s_lut[64];
  __shared__ double stage[CH_WAVES * 4 * POST_SSTR];
  GridScope sc{ps.ctl, red, bc, 0u};
  const int r0 = in.row_region_off[g], nrow = in.row_region_off[g + 1] - r0;
  const int c0 = in.cand_off[g], S = in.cand_off[g + 1] - c0;
  const int64_t e_base = in.row_ptr[r0];
  const int E = (int)(in.row_ptr[r0 + nrow] - e_base);
  PostView<int32_t> v;
  v.g = g; v.S = S; v.nrow = nrow; v.E = E; v.F = in.reg[g].R; v.r0 = r0; v.c0 = c0;
  v.le = s_lut; v.l1e = s_lut + 32;
  v.sps = ps.sps; v.rpa = ps.rpa; v.rpb = ps.rpb; v.sflags = ps.sflags; v.soflags = ps.soflags; v.parent = ps.parent;
  v.rptr = ps.rptr; v.ecol = ps.ecol; v.erow = ps.erow; v.cent = ps.cent; v.ccptr = ps.ccptr; v.ev = ps.ev;
  v.tag = ps.tag; v.asg = ps.asg; v.fp = ps.fp; v.lok = ps.lok; v.dirty = ps.dirty;
  v.shap = ps.shap; v.sgt = ps.sgt; v.svt = ps.svt; v.rcode = ps.rcode;
  v.fdirt = ps.fdirt; v.minf = ps.minf; v.ndraw = ps.ndraw; v.gwords = ps.gwords;
  v.cand = in.cand + c0;
  v.stage = stage;
  int n_mark = 0;
  auto mark = [&]() { if (in.dbg_clk && sc.tid() == 0) in.dbg_clk[(size_t)g * 16 + n_mark] = (long long)wall_clock64(); n_mark++; };
  mark();
  if (threadIdx.x < 31) { v.le[threadIdx.x] = lut.le[threadIdx.x]; v.l1e[threadIdx.x] = lut.l1e[threadIdx.x]; }
  for (int i = sc.tid(); i < S; i += sc.nt()) {
    v.sflags[i] = v.soflags[i] = v.cand[i].flags;
    v.shap[i] = in.st_delta[c0 + i]; v.sgt[i] = in.st_eta[c0 + i]; v.svt[i] = (int8_t)v.cand[i].variant_type;
    v.sps[i] = v.cand[i].phase_score;
    v.parent[i] = 0;
  }
  for (int r = sc.tid(); r < nrow; r += sc.nt()) {
    const int isp = in.links[r0 + r] >= in.min_linkers ? 1 : 0;
    v.rptr[r] = (int32_t)(in.row_ptr[r0 + r] - e_base);
    v.lok[r] = (uint8_t)isp; v.fp[r] = (uint8_t)isp; v.asg[r] = 0; v.tag[r] = 0;
  }
  if (sc.tid() == 0) v.rptr[nrow] = E;
  for (int e = sc.tid(); e < E; e += sc.nt()) { v.ecol[e] = in.col[e_base + e] - c0; v.ev[e] = in.val[e_base + e]; }
  sc.sync();
  for (int k = sc.tid(); k < v.F; k += sc.nt()) v.tag[in.prow_src[r0 + k]] = in.st_sigma[r0 + k];
  mark();
  ordered_index(sc, nrow, S, v.rptr, v.ecol, nullptr, v.ccptr, v.erow, v.cent, ps.pcnt, ps.n_parts, sm);
  mark();
  post_run(sc, in, lut, v, mark);
}
}  // namespace

// hipFuncSetAttribute is per device and costs tens of microseconds of host time: once per (kernel, device)
hipError_t k4_set_dyn_lds_once(const void* fn, int bytes, int slot) {
  static unsigned char done[8][64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (done[slot & 7][dev]) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done[slot & 7][dev] = 1;
  return e;
}

hipError_t k4_chain_launch_wg(const ChainDev& C, int first, int n, size_t dyn_lds, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipError_t e = k4_set_dyn_lds_once(reinterpret_cast<const void*>(&k4_chain_wg<CH_THREADS>), 64 * 1024, 1);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k4_chain_wg<CH_THREADS>, dim3((unsigned)n), dim3(CH_THREADS), dyn_lds, s, C, (int32_t)first, (int32_t)n);
  return hipGetLastError();
}

int k4_grid_blocks() {
  static int blocks = 0;
  if (blocks) return blocks;
  int dev = 0, ncu = 0, per = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, reinterpret_cast<const void*>(&k4_chain_grid), CH_THREADS, 0) != hipSuccess) return 0;
  if (per < 1 || ncu < 1) return 0;
  blocks = ncu;   // one workgroup per CU: fewer arrivals per barrier, and co-resident with room to spare
  return blocks;
}

hipError_t k4_chain_launch_grid(const ChainDev& C, int which, size_t dyn_lds, hipStream_t s) {
  const int nb = k4_grid_blocks();
  if (nb <= 0) return hipErrorInvalidDevice;
  hipError_t e = hipMemsetAsync(C.ctl, 0, sizeof(GridCtl), s);
  if (e != hipSuccess) return e;
  if (C.spec_lanes > 1) { e = hipMemsetAsync(C.spec_ctl, 0, sizeof(GridCtl) * (size_t)C.spec_lanes, s); if (e != hipSuccess) return e; }
  if (C.bt_ctl) { e = hipMemsetAsync(C.bt_ctl, 0, K4_GRID_BATCH_CTL_BYTES, s); if (e != hipSuccess) return e; }
  e = k4_set_dyn_lds_once(reinterpret_cast<const void*>(&k4_chain_grid), K4_GRID_FAST_LDS_MAX, 2);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k4_chain_grid, dim3((unsigned)nb), dim3(CH_THREADS), dyn_lds, s, C, (int32_t)which);
  return hipGetLastError();
}

hipError_t k4_stage_launch_grid(const StageIn& in, const StageOut& out, const PhaseLutDev& lut, int g, GridCtl* ctl, int32_t* blk_tot, hipStream_t s) {
  const int nb = k4_grid_blocks();
  if (nb <= 0) return hipErrorInvalidDevice;
  hipError_t e = hipMemsetAsync(ctl, 0, sizeof(GridCtl), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k4_stage_grid, dim3((unsigned)nb), dim3(CH_THREADS), 0, s, in, out, lut, (int32_t)g, ctl, blk_tot);
  return hipGetLastError();
}

hipError_t k4_post_launch_grid(const PostIn& post_in, const PostScratch& ps, int g, const PostLut& lut, hipStream_t s) {
  const int nb = k4_grid_blocks();
  if (nb <= 0) return hipErrorInvalidDevice;
  hipError_t e = hipMemsetAsync(ps.ctl, 0, sizeof(GridCtl), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k4_gpost, dim3((unsigned)nb), dim3(CH_THREADS), 0, s, post_in, ps, (int32_t)g, lut);
  return hipGetLastError();
}
