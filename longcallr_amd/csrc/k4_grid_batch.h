// k4_grid_batch.h — the perturbation rounds (phase.rs:1198-1233) at grid scope, EIGHT speculative half-rounds per pass over the
// matrix.  Included by k4_grid.hip (inside its anonymous namespace, after chain_rounds_fast, whose commit order it keeps).
//
// chain_rounds_fast runs its speculative half-rounds side by side, each on an eighth of the workgroups with a working state
// of its own: every one of them streams the whole packed matrix (row order for the sigma step, column order for the delta
// step) from beyond L2 in every half step, and the VALUs issue a third of the time.  Here ALL workgroups work on ALL eight
// states at once: an entry is loaded and decoded once and serves eight states, whose sigma / delta / eta are BITS of one
// word per row / per SNP:
//   sig8[pos]  bit s = [sigma_s == +1]                                  (row = perm[pos])
//   m32[snp]   het | dneg << 8 | dzero << 16 | homvar << 24, bit s each (written by whoever decides / loads the SNP)
// sigma step   a LANE per row: the rows are sorted by their number of entries (counting sort, any order inside a bin -- integer sums do
//              not depend on it) and cut into blocks of 64; a block's entries are 16-byte groups of FOUR entries of one row, stored
//              round-major (group j of the block's 64 rows side by side: one coalesced 1 KB load per round), rows padded to the
//              block's longest with entries of a null SNP -- sorted rows make that padding a few %.  A lane keeps its row's eight sums
//              in registers and takes the eight decisions itself: no atomics, no LDS accumulators, no cross-lane step.  Per
//              (entry, state) the work is three instructions: the masks are kept "spread" (state s at bit 2 s), so that
//              het | minus << 1 holds the 2-bit signed factor (-1, 0, +1) of every state, extracted by one v_bfe_i32 and applied to the
//              two limbs of w (low 23 bits, signed rest: w < 0 for q < 4) by two v_mad_i32_i24.  Blocks are dealt to the workgroups
//              in serpentine order of their length and taken by a workgroup's waves longest first from an LDS counter.
//              (Measured before: a wave per 32-row unit with lanes over the unit's groups and LDS atomics per group -- 28 us per
//              step on C5 against 12 here: 290 instructions per group, a third of them the flush, and units of 2.1 passes.)
//              Everything per row -- sigma bits, the column-order entries' row numbers -- is kept by sorted POSITION.
// delta step   teams of BATCH_TW waves per SNP (chain_rounds_fast: four); sigma comes from sig8 by a byte gather (333 KB on C5:
//              L2 resident), which is why the barrier in front of the step is followed by an L2 / L1 INVALIDATE (acquire
//              fence at agent scope, one wave per workgroup) -- the only non-coherent read of mutable data in the rounds.
//              Lanes 0-7 of the last wave of a team to arrive take the decisions of the eight states.
// het only     with_genotype is false in the rounds: a het site stays het and a hom site stays hom (it may move between hom-ref and
//              hom-var by its constants alone), in every state.  Entries at hom sites therefore never contribute to a row sum, and a
//              hom site's decision needs no column sum: the row-order array holds the het-site entries only, and the delta step
//              sweeps the het columns only (C5: 2 178 of 4 687 sites).
// barriers     two per iteration (as before), the second one carrying eight objective sums and eight "changed" bits.
// A batch ends when all its states have settled (or made 21 iterations); states that settle earlier are frozen.  Commits
// are owner-local: no barrier between a batch's last iteration and the next batch's perturbation.
// Same integers, same order of commits as the one-at-a-time form: bit-identical results (tests compare them).
#pragma once

struct BatchCtl {                      // device words of the batched rounds (zeroed by the launcher)
  unsigned alloc4, groups; unsigned pad_[30];  // 16-byte groups of the row-order entry array handed out so far; groups of the region
  unsigned hist[256], cursor[256];     // counting sort of the rows by their number of groups (255: that many or more)
  // The rounds' barrier: ARRIVAL = one plain coherent store of (stamp << 32 | flags) into the workgroup's own 128-byte line, nothing to wait
  // for; wave 0 of workgroup 0 polls the lines, ORs the flags, and publishes (stamp << 32 | OR) in res[0], which everybody polls.  A
  // delta-step barrier also carries the workgroup's shares of the eight objective sums in words 1-8 of its line (stored and acknowledged
  // before the arrival word); the polling wave sums the states that have just settled into res[1 + k] before it opens the barrier.
  // (A counter barrier costs two more device round trips, one for the returning arrival atomic and one for the last arriver's loads;
  // 2 304 atomics on nine words or every workgroup summing all the shares itself: 20 us per barrier, the readers queue on a few lines.)
  unsigned long long res[16];
  unsigned long long arr[K4_GRID_BATCH_MAX_WG][16];
};
static_assert(sizeof(BatchCtl) <= K4_GRID_BATCH_CTL_BYTES, "BatchCtl");

struct BatchCol { int i, c0, c1, fp; long long F, Wt, D2, D3; };   // a team's column of one round of the delta step (LDS, set once)
constexpr int BATCH_ROUNDS = 16;
constexpr int BATCH_TW = 2;                          // waves of a team (one column at a time); teams of a workgroup:
constexpr int BATCH_TEAMS = 16 / BATCH_TW;

__device__ __forceinline__ uint32_t spread8(uint32_t x) {   // bit s -> bit 2 s
  x = (x | (x << 4)) & 0x0F0Fu; x = (x | (x << 2)) & 0x3333u; x = (x | (x << 1)) & 0x5555u;
  return x;
}

// Eight per-lane sums (limbs: lo < 2^28, |hi| < 2^24 per lane) over the wave in ONE reduce-scatter + all-reduce: three halving
// exchanges over lane bits 0, 1, 2 (a lane keeps half of its values and receives the partner's for them), the survivor as a
// 64-bit value over lane bits 3, 4, 5.  Every lane ends with the wave's total of state 4 b0 + 2 b1 + b2 (b = bits of its index):
// ~70 instructions, against 8 x 26 for eight wave-wide sums.
__device__ __forceinline__ long long wave_reduce8(const uint32_t (&lo)[8], const int (&hi)[8], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
  uint32_t l4[4]; int h4[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t kl = b0 ? lo[k + 4] : lo[k], sl = b0 ? lo[k] : lo[k + 4];
    const int kh = b0 ? hi[k + 4] : hi[k], sh = b0 ? hi[k] : hi[k + 4];
    l4[k] = kl + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sl, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    h4[k] = kh + __builtin_amdgcn_update_dpp(0, sh, 0xB1, 0xf, 0xf, false);
  }
  uint32_t l2[2]; int h2[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const uint32_t kl = b1 ? l4[k + 2] : l4[k], sl = b1 ? l4[k] : l4[k + 2];
    const int kh = b1 ? h4[k + 2] : h4[k], sh = b1 ? h4[k] : h4[k + 2];
    l2[k] = kl + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sl, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    h2[k] = kh + __builtin_amdgcn_update_dpp(0, sh, 0x4E, 0xf, 0xf, false);
  }
  const uint32_t kl = b2 ? l2[1] : l2[0], sl = b2 ? l2[0] : l2[1];
  const int kh = b2 ? h2[1] : h2[0], sh = b2 ? h2[0] : h2[1];
  // lane ^ 4: the even banks of a row read four lanes up (row_shl:4), the odd banks four lanes down (row_shr:4)
  int rl = __builtin_amdgcn_update_dpp(0, (int)sl, 0x104, 0xf, 0x5, false); rl = __builtin_amdgcn_update_dpp(rl, (int)sl, 0x114, 0xf, 0xa, false);
  int rh = __builtin_amdgcn_update_dpp(0, sh, 0x104, 0xf, 0x5, false); rh = __builtin_amdgcn_update_dpp(rh, sh, 0x114, 0xf, 0xa, false);
  long long v = (long long)(kl + (uint32_t)rl) + (long long)(kh + rh) * (1ll << 23);
  v += LCR_DPP_LL(v, 0x128, 0xf);   // row_ror:8 = lane ^ 8
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ int mad24(int a, int b, int c) {   // (the compiler makes v_mul_i32_i24 + v_add3_u32 of a * b + c)
  int d;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// tie_row_decide_g (k4_dev.h) for ONE (row, state) by a whole wave: a lane per entry forms the entry's two terms (one trip to memory for
// the whole row instead of one per entry), lane order = list order, and the two f64 sums are taken in that order through v_readlane.
// Same census, same doubles, same verdict.  get(i, &d, &eta) as there.
template <class Get>
__device__ __forceinline__ bool tie_row_decide_wave(const PhaseDev& P, const int32_t* rp, const int32_t* pc, const uint8_t* pv, int row, Get get,
                                                    int sigma, const double* le, const double* l1e, int lane) {
  const int e0 = rp[row], e1 = rp[row + 1];
  double lp = 0.0, lm = 0.0;
  bool h = false;
  for (int eb = e0; eb < e1; eb += 64) {
    const int e = eb + lane;
    double tp = 0.0, tm = 0.0;
    bool hh = false;
    if (e < e1) {
      const int i = pc[e];
      const uint8_t vb = pv[e];
      int d, eta;
      get(i, d, eta);
      const int p = (vb & 32) ? 1 : -1, q = vb & 31;
      const int xp = eta == 0 ? d : eta, xm = eta == 0 ? -d : eta;
      hh = eta == 0;
      tp = p == xp ? l1e[q] : le[q];
      tm = p == xm ? l1e[q] : le[q];
    }
    h = h || __ballot(hh) != 0ull;
    const int n = min(64, e1 - eb);
    for (int k = 0; k < n; k++) {
      const double a = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tp), k), __builtin_amdgcn_readlane(__double2loint(tp), k));
      const double b = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tm), k), __builtin_amdgcn_readlane(__double2loint(tm), k));
      lp += a; lm += b;
    }
  }
  bool f = false;
  if (P.tie_arith < 2) {
    if (h && lane == 0) TIE_COUNT(P.tie_ctr, TIE_SIGMA_UNRES, 1ull);
    return false;
  }
  if (e1 - e0 > 2 && h && lp != lm) {   // (two entries: a + b against b + a -- the same double)
    const double l1 = sigma == 1 ? lp : lm, l1n = sigma == 1 ? lm : lp, den = lp + lm;
    const double q = 1.0 - l1 / den, qn = 1.0 - l1n / den;
    f = q < qn;
  }
  if (h && lane == 0) { TIE_COUNT(P.tie_ctr, TIE_SIGMA_F64, 1ull); if (f) TIE_COUNT(P.tie_ctr, TIE_SIGMA_FLIPS, 1ull); }
  return f;
}
__device__ __forceinline__ int bitrev3(int x) { return ((x & 1) << 2) | (x & 2) | ((x >> 2) & 1); }

__device__ __forceinline__ bool chain_rounds_batch(GridScope& sc, const ChainDev& C, const RegionDev& rd, const ChainView& v, const long long* wl,
                                                   uint8_t* dyn, long long best, int slot, const FlipLut& FL) {
  const int S = rd.S, R = rd.R, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  BatchCtl* const B = reinterpret_cast<BatchCtl*>(C.bt_ctl);
  uint4* const pk4 = reinterpret_cast<uint4*>(C.bt_pk4);
  int2* const blk_tab = reinterpret_cast<int2*>(C.bt_up4);        // per block of 64 sorted rows: {first group, rounds}
  unsigned long long* const bs64 = reinterpret_cast<unsigned long long*>(C.bt_bs32);   // best sigma, a bit per row: one word per block
  int32_t* const perm = C.bt_perm; int32_t* const inv = C.bt_inv; // sorted position -> row, row -> sorted position
  uint8_t* const sig8 = C.bt_sig8; uint32_t* const m32 = C.bt_m32;
  const int32_t* rp = v.mv.rp; const int32_t* cp = v.mv.cp;
  uint32_t* const pkc = C.pk_csc;
  // ---- LDS
  uint2* const s_m = reinterpret_cast<uint2*>(dyn);                                  // S + 1: spread masks of every SNP (+ the null SNP)
  uint2* const s_w = s_m + ((S + 2) & ~1);                                           // w = .x (low 23 bits) + 2^23 .y (signed)
  unsigned long long* const t_sum = reinterpret_cast<unsigned long long*>(s_w + 32); // [team][slot][state]
  unsigned* const t_cnt = reinterpret_cast<unsigned*>(t_sum + BATCH_TEAMS * 8 * 8);  // [team][slot]
  unsigned long long* const s_acc = reinterpret_cast<unsigned long long*>(t_cnt + BATCH_TEAMS * 8);
  unsigned long long* const s_out = s_acc + 8;
  unsigned* const s_misc = reinterpret_cast<unsigned*>(s_out + 8);                   // [0] changed bits of this workgroup, [1] of the grid, [2] next block, [3] tied (row, state) pairs
  unsigned* const s_hist = s_misc + 4;                                               // 256: set-up histogram; then the sigma step's list of tied (position | state << 24)
  constexpr unsigned TIE_CAP = 256;
  BatchCol* const s_col = reinterpret_cast<BatchCol*>(s_hist + 256);                 // [team][round]
  uint32_t* const s_raw = reinterpret_cast<uint32_t*>(s_col + BATCH_TEAMS * BATCH_ROUNDS);     // S: m32 as staged for the iteration
  if (threadIdx.x < 32) { const long long w = wl[threadIdx.x]; s_w[threadIdx.x] = make_uint2((uint32_t)(w & 0x7FFFFF), (uint32_t)(w >> 23)); }
  for (int k = threadIdx.x; k < BATCH_TEAMS * 8 * 8; k += blockDim.x) t_sum[k] = 0;
  if (threadIdx.x < BATCH_TEAMS * 8) t_cnt[threadIdx.x] = 0;
  if (threadIdx.x < 8) s_acc[threadIdx.x] = 0;
  if (threadIdx.x < 4) s_misc[threadIdx.x] = 0;
  if (threadIdx.x < 256) s_hist[threadIdx.x] = 0;
  __syncthreads();
  // ---- the rows sorted by their number of groups (four het-site entries each), longest first
  const int n_blocks = (R + 63) >> 6;
  const int n_rounds = (S + BATCH_TEAMS * (int)sc.nblk() - 1) / (BATCH_TEAMS * (int)sc.nblk());   // columns per team of the delta step
  if (n_rounds > BATCH_ROUNDS) return false;
  {
    const int32_t* pc = v.mv.pc; const uint8_t* pv = v.mv.pv; const int32_t* cr = v.mv.cr; const uint8_t* cv = v.mv.cv;
    const int E = cp[S];
    int32_t* const nh = v.flipcol;   // (free after the block flip) groups of every row
    for (int row = sc.tid(); row < R; row += sc.nt()) {
      int n = 0;
      for (int e = rp[row]; e < rp[row + 1]; e++) n += v.bet[pc[e]] == 0 ? 1 : 0;
      n = (n + 3) >> 2;
      nh[row] = n;
      atomicAdd(&s_hist[min(n, 255)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 256 && s_hist[threadIdx.x]) atomicAdd(&B->hist[threadIdx.x], s_hist[threadIdx.x]);
    sc.sync();
    if (threadIdx.x < 256) {   // where the bin of this length begins: the rows of all longer bins come first
      unsigned o = 0;
      for (int l = 255; l > (int)threadIdx.x; l--) o += cload(&B->hist[l]);
      s_hist[threadIdx.x] = o;
    }
    __syncthreads();
    for (int r0 = 64 * sc.wave(); r0 < R; r0 += 64 * sc.nwaves()) {   // a wave asks once per distinct length among its 64 rows
      const int row = r0 + lane;
      const int key = row < R ? min(nh[row], 255) : -1;
      unsigned long long todo = __ballot(key >= 0);
      int pos = -1;
      while (todo) {
        const int lead = __ffsll((long long)todo) - 1;
        const int kk = __shfl(key, lead, 64);
        const unsigned long long same = __ballot(key == kk);
        unsigned base = 0;
        if (lane == lead) base = atomicAdd(&B->cursor[kk], (unsigned)__popcll(same));
        base = (unsigned)__shfl((int)base, lead, 64);
        if (key == kk) pos = (int)(s_hist[kk] + base) + __popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
      }
      if (row < R) { perm[pos] = row; inv[row] = pos; }
    }
    sc.sync();
    // column order as in chain_rounds_fast, the row numbers as sorted positions: value byte << 24 | position
    for (int e = sc.tid(); e < E + 8; e += sc.nt()) pkc[e] = e < E ? ((uint32_t)cv[e] << 24) | (uint32_t)inv[cr[e]] : 0u;
    // a lane per block: its rounds = its longest row; a wave's blocks get one run of the array (one atomic per 64 blocks)
    for (int b0 = 64 * sc.wave(); b0 < n_blocks; b0 += 64 * sc.nwaves()) {
      const int b = b0 + lane;
      int len = 0;
      if (b < n_blocks) for (int p = 64 * b; p < min(64 * b + 64, R); p++) len = max(len, nh[perm[p]]);   // (bin 255 is not sorted inside)
      const int n4 = 64 * len;
      int incl = n4;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
      const int total = __shfl(incl, 63, 64);
      unsigned base = 0;
      if (lane == 63 && total) base = atomicAdd(&B->alloc4, (unsigned)total);
      base = (unsigned)__shfl((int)base, 63, 64);
      if (b < n_blocks) blk_tab[b] = make_int2((int)base + incl - n4, len);
    }
    sc.sync();
    if ((int64_t)cload(&B->alloc4) > C.bt_cap4) return false;   // (uniform; never with the host's sizing)
    for (int b = sc.wave(); b < n_blocks; b += sc.nwaves()) {   // a wave per block, a lane per row: round-major groups
      const int2 bt = blk_tab[b];
      const int p = 64 * b + lane;
      const int row = p < R ? perm[p] : -1;
      const int e0 = row >= 0 ? rp[row] : 0, e1 = row >= 0 ? rp[row + 1] : 0;
      const uint32_t nul = (uint32_t)S;
      uint32_t en[4] = {nul, nul, nul, nul};
      int j = 0, fill = 0;
      for (int e = e0; e < e1; e++) {
        const int i = pc[e];
        if (v.bet[i] != 0) continue;
        en[fill++] = ((uint32_t)pv[e] << 24) | (uint32_t)i;
        if (fill == 4) { pk4[bt.x + 64 * j + lane] = make_uint4(en[0], en[1], en[2], en[3]); j++; fill = 0; en[0] = en[1] = en[2] = en[3] = nul; }
      }
      if (fill) { pk4[bt.x + 64 * j + lane] = make_uint4(en[0], en[1], en[2], en[3]); j++; }
      for (; j < bt.y; j++) pk4[bt.x + 64 * j + lane] = make_uint4(nul, nul, nul, nul);
    }
    if (C.dbg && sc.tid() == 0) C.dbg[14] = (long long)cload(&B->alloc4);
  }
  const uint8_t* fp = v.mv.fp;
  const long long* scn = C.P.snp_const + 4ll * rd.snp_off;
  const PhaseLutDev& lut = C.P.lut;
  const int nw = sc.nwaves(), nblk = sc.nblk(), blk = sc.blk();
  const int wj0 = wv * nblk + blk;          // load / commit of the blocks wj0 + k nw is this wave's (any fixed owner will do)
  // ---- delta step order (as in chain_rounds_fast): SNPs by column length, dealt to the teams in serpentine order
  int32_t* const ord = v.queue;
  for (int i = sc.tid(); i < S; i += sc.nt()) {   // (a hom site's column is not swept: length 0 here)
    const int len = v.bet[i] == 0 ? cp[i + 1] - cp[i] : 0;
    int rank = 0;
    for (int j = 0; j < S; j++) { const int lj = v.bet[j] == 0 ? cp[j + 1] - cp[j] : 0; rank += (lj > len || (lj == len && j < i)) ? 1 : 0; }
    ord[rank] = i;
  }
  // ---- best sigma as bits (best == working here)
  for (int b = wj0; b < n_blocks; b += nw) {
    const int p = 64 * b + lane;
    const unsigned long long bb = __ballot(p < R && v.bsg[perm[p]] == 1);
    if (lane == 0) cstore(&bs64[b], bb);
  }
  sc.sync();   // (fenced: everything above was written with plain stores)
  // the columns of this workgroup's four teams, round by round (fixed for the whole launch), with their constants
  for (int t = threadIdx.x; t < BATCH_TEAMS * BATCH_ROUNDS; t += blockDim.x) {
    const int team = t / BATCH_ROUNDS, r = t % BATCH_ROUNDS, nteams = nblk * BATCH_TEAMS, gteam = team * nblk + blk;
    const int pl = r * nteams + ((r & 1) ? nteams - 1 - gteam : gteam);
    BatchCol c{-1, 0, 0, 0, 0, 0, 0, 0};
    if (r < n_rounds && pl < S) {
      const int i = ord[pl];
      c.i = i; c.c0 = cp[i]; c.c1 = cp[i + 1]; c.fp = fp[i];
      c.F = scn[4 * i]; c.Wt = scn[4 * i + 1]; c.D2 = scn[4 * i + 2]; c.D3 = scn[4 * i + 3];
    }
    s_col[t] = c;
  }
  __syncthreads();
  const uint64_t SF = (uint64_t)S + (uint64_t)R;
  const int H = 2 * (S / 4 + 1);
  unsigned bstamp = 0;   // barriers passed (the words are zero at the launch)
  // light barrier (see BatchCtl); `flags` are ORed over the grid.  need(changed) = mask of the states whose sums the caller wants (<- s_out)
  auto bar = [&](uint32_t flags, auto need) -> uint32_t {
    bstamp++;
    __builtin_amdgcn_s_waitcnt(0);   // this wave's coherent stores are acknowledged ...
    __syncthreads();                 // ... all of the workgroup's are, before it counts as arrived
    if (threadIdx.x == 0) cstore(&B->arr[blk][0], ((unsigned long long)bstamp << 32) | flags);
    if (blk == 0 && wv == 0) {
      uint32_t fl = 0;
      for (;;) {
        bool ok = true;
        fl = 0;
        for (int b = lane; b < nblk; b += 64) { const unsigned long long x = cload(&B->arr[b][0]); ok = ok && (uint32_t)(x >> 32) == bstamp; fl |= (uint32_t)x; }
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) fl |= (uint32_t)__shfl_xor((int)fl, d, 64);
      const uint32_t want = need(fl);
      if (want) {
        for (int k = 0; k < 8; k++)
          if ((want >> k) & 1u) {
            long long x = 0;
            for (int b = lane; b < nblk; b += 64) x += (long long)cload(&B->arr[b][1 + k]);
            x = wave_sum_ll_dpp(x);
            if (lane == 0) cstore(&B->res[1 + k], (unsigned long long)x);
          }
        __builtin_amdgcn_s_waitcnt(0);
      }
      if (lane == 0) cstore(&B->res[0], ((unsigned long long)bstamp << 32) | fl);
    }
    if (threadIdx.x == 0) {
      unsigned long long x;
      while ((uint32_t)((x = cload(&B->res[0])) >> 32) != bstamp) __builtin_amdgcn_s_sleep(1);
      s_misc[1] = (uint32_t)x;
    }
    __syncthreads();
    const uint32_t all = s_misc[1];
    const uint32_t want = need(all);
    if (want) {
      if (threadIdx.x < 8 && ((want >> threadIdx.x) & 1u)) s_out[threadIdx.x] = cload(&B->res[1 + threadIdx.x]);
    }
    __syncthreads();   // (s_misc[1] / s_out are read; the next barrier may write them)
    return all;
  };
  auto bar_light = [&]() { (void)bar(0u, [](uint32_t) { return 0u; }); };
  long long it_total = 0;
  for (int h = 0; h < H;) {
    const int nact = min(8, H - h);
    uint32_t act = (1u << nact) - 1u;
    // ---- the batch's start states: best, with the perturbation of half-round h + s in state s (owner-local)
    for (int i = blk + nblk * (int)threadIdx.x; i < S; i += sc.nt()) {
      const int d = cload(&v.bdl[i]), e = cload(&v.bet[i]);
      uint32_t het = 0, dn = 0, dz = 0, hv = 0;
#pragma unroll
      for (int s = 0; s < 8; s++) {
        int ds = d;
        const int hh = h + s;
        if (hh < H && (hh & 1) == 0) {
          const int tidx = hh >> 1;
          const bool flip = (tidx & 1) == 1;
          const double rg = u01(rd.seed, 2 * SF + (uint64_t)tidx * SF + (uint64_t)i);
          if (rg < 0.1) ds = flip ? 1 : -1;
          else if (rg >= 0.9) ds = flip ? -1 : 1;
        }
        het |= (uint32_t)(e == 0) << s; hv |= (uint32_t)(e == -1) << s; dn |= (uint32_t)(ds < 0) << s; dz |= (uint32_t)(ds == 0) << s;
      }
      cstore(&m32[i], het | (dn << 8) | (dz << 16) | (hv << 24));
    }
    for (int b = wj0; b < n_blocks; b += nw) {
      const int p = 64 * b + lane;
      if (p < R) {
        const int row = perm[p];
        uint32_t x = ((cload(&bs64[b]) >> lane) & 1ull) ? 0xFFu : 0u;
#pragma unroll
        for (int s = 0; s < 8; s++) {   // (the sigma half-rounds are the odd ones)
          const int hh = h + s;
          if (hh < H && (hh & 1) && u01(rd.seed, 2 * SF + (uint64_t)(hh >> 1) * SF + (uint64_t)S + (uint64_t)row) < 0.1) x ^= 1u << s;
        }
        cstore(&sig8[p], (uint8_t)x);
      }
    }
    bar_light();
    long long obj[8];
#pragma unroll
    for (int s = 0; s < 8; s++) obj[s] = LLONG_MIN;
    int iters = 0;
    while (act) {
      long long tk0 = 0;
      const bool tk = C.dbg && sc.tid() == 0;
      auto tick = [&](int slot_) { if (tk) { const long long t = (long long)wall_clock64(); C.dbg[slot_] += t - tk0; tk0 = t; } };
      if (tk) tk0 = (long long)wall_clock64();
      // ---- sigma step
      if (threadIdx.x == 0) { s_misc[2] = 0; s_misc[3] = 0; }
      for (int i = threadIdx.x; i <= S; i += blockDim.x) {
        const uint32_t m = i < S ? cload(&m32[i]) : 0u;
        if (i < S) s_raw[i] = m;
        s_m[i] = make_uint2(spread8(m & 0xFFu) | (spread8((m >> 8) & 0xFFu) << 16), spread8((m >> 16) & 0xFFu) | (spread8(m >> 24) << 16));
      }
      __syncthreads();
      tick(8);
      const long long wg_t0 = C.dbg ? (long long)wall_clock64() : 0;
      uint32_t anyb = 0;
      for (;;) {
        // the workgroup's blocks j nblk + (blk, or mirrored in odd j): longest first, a wave takes the next one
        unsigned jn = 0;
        if (lane == 0) jn = atomicAdd(&s_misc[2], 1u);
        const int j = __builtin_amdgcn_readfirstlane((int)jn);
        const int b = j * nblk + ((j & 1) ? nblk - 1 - blk : blk);
        if (b >= n_blocks) break;
        const int2 bt = blk_tab[b];
        const int p = 64 * b + lane;
        const uint32_t sgb = p < R ? (uint32_t)cload(&sig8[p]) : 0u;
        const uint32_t sg16 = spread8(sgb);
        int lo[8], hi[8];
        long long tot[8];
#pragma unroll
        for (int s = 0; s < 8; s++) { lo[s] = 0; hi[s] = 0; tot[s] = 0; }
        uint32_t hetor = 0;
        uint4 tn = bt.y > 0 ? pk4[bt.x + lane] : make_uint4(0, 0, 0, 0);
        for (int r = 0; r < bt.y; r++) {
          const uint4 t = tn;
          if (r + 1 < bt.y) tn = pk4[bt.x + 64 * (r + 1) + lane];
          const uint32_t en[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
          for (int x4 = 0; x4 < 4; x4++) {
            const uint32_t i = en[x4] & 0x3FFFFu, x = en[x4] >> 24;
            const uint2 mm = s_m[i];
            const uint2 w2 = s_w[x & 31u];
            const uint32_t het16 = mm.x & 0xFFFFu;
            // the allele equals sigma * delta: +w; it does not, or delta is 0: -w; hom site: nothing
            const uint32_t neg16 = ((x & 32u) ? 0x5555u : 0u) ^ sg16 ^ (mm.x >> 16);
            const uint32_t code = het16 | (((neg16 | mm.y) & het16) << 1);   // state s at bits 2 s, 2 s + 1: 01 = +1, 11 = -1
            hetor |= het16;
#pragma unroll
            for (int s = 0; s < 8; s++) {
              const int sgn = __builtin_amdgcn_sbfe((int)code, 2 * s, 2);
              lo[s] = mad24(sgn, (int)w2.x, lo[s]); hi[s] = mad24(sgn, (int)w2.y, hi[s]);
            }
          }
          if ((r & 15) == 15) {   // (the limb sums hold 64 entries)
#pragma unroll
            for (int s = 0; s < 8; s++) { tot[s] += (long long)lo[s] + (long long)hi[s] * (1ll << 23); lo[s] = 0; hi[s] = 0; }
          }
        }
        uint32_t fb = 0;
        if (p < R) {
          uint32_t tm = 0;   // states in which the row's sums tie exactly and the row has an entry at a het site
#pragma unroll
          for (int s = 0; s < 8; s++) {
            const long long diff = tot[s] + (long long)lo[s] + (long long)hi[s] * (1ll << 23);
            if (diff < 0) fb |= 1u << s;
            else if (diff == 0 && ((hetor >> (2 * s)) & 1u)) tm |= 1u << s;
          }
          fb &= act; tm &= act;
          anyb |= fb;
          while (tm) {   // the reference-order f64 scores decide (a tie flip is no improvement: not in anyb): by a wave, behind the blocks
            const int s = __ffs((int)tm) - 1;
            tm &= tm - 1u;
            const unsigned at = atomicAdd(&s_misc[3], 1u);
            if (at < TIE_CAP) { s_hist[at] = (unsigned)p | ((unsigned)s << 24); continue; }
            auto get = [&](int i, int& d, int& eta) {
              const uint2 mm = s_m[i];
              eta = ((mm.x >> (2 * s)) & 1u) ? 0 : (((mm.y >> (16 + 2 * s)) & 1u) ? -1 : 1);
              d = ((mm.y >> (2 * s)) & 1u) ? 0 : (((mm.x >> (16 + 2 * s)) & 1u) ? -1 : 1);
            };
            if (tie_row_decide_g(C.P, rp, v.mv.pc, v.mv.pv, perm[p], get, ((sgb >> s) & 1u) ? 1 : -1, FL.le, FL.l1e)) fb |= 1u << s;
          }
          if (fb) cstore(&sig8[p], (uint8_t)(sgb ^ fb));
        }
      }
      __syncthreads();
      {   // the tied (row, state) pairs of this workgroup's blocks, a wave each (the state's bit of sig8 is still the step's input)
        const unsigned nt = min(s_misc[3], TIE_CAP);
        for (unsigned t = wv; t < nt; t += CH_WAVES) {
          const unsigned item = s_hist[t];
          const int p = (int)(item & 0xFFFFFFu), s = (int)(item >> 24);
          const uint32_t sgb = (uint32_t)cload(&sig8[p]);
          auto get = [&](int i, int& d, int& eta) {
            const uint2 mm = s_m[i];
            eta = ((mm.x >> (2 * s)) & 1u) ? 0 : (((mm.y >> (16 + 2 * s)) & 1u) ? -1 : 1);
            d = ((mm.y >> (2 * s)) & 1u) ? 0 : (((mm.x >> (16 + 2 * s)) & 1u) ? -1 : 1);
          };
          const bool f = tie_row_decide_wave(C.P, rp, v.mv.pc, v.mv.pv, perm[p], get, ((sgb >> s) & 1u) ? 1 : -1, FL.le, FL.l1e, lane);
          if (f && lane == 0)
            __hip_atomic_fetch_xor(reinterpret_cast<unsigned*>(sig8 + (p & ~3)), 1u << (8 * (p & 3) + s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __syncthreads();
      if (C.dbg && threadIdx.x == 0 && blk < 1024) C.dbg[16 + blk] += (long long)wall_clock64() - wg_t0;
      tick(9);
      bar_light();
      tick(10);
      // sig8 is read with plain (cached) loads below: drop what the L2 / L1 hold of it
      if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __syncthreads();
      tick(11);
      // ---- delta / eta step
      const long long wg_t1 = C.dbg ? (long long)wall_clock64() : 0;
      long long acc = 0;   // lane s < 8: the objective terms of state s this wave decided
      {
        const int team = wv / BATCH_TW, wt = wv % BATCH_TW;
        constexpr int PASS = 256 * BATCH_TW;   // entries of a column a team takes per load
        const BatchCol* const cols = s_col + team * BATCH_ROUNDS;
        // a column's first 2 048 entries are requested while the column before it is worked on (the matrix comes from beyond
        // L2 after the invalidate: one exposed trip per step instead of one per column)
        uint4 ta_n = make_uint4(0, 0, 0, 0), tb_n = ta_n;
        auto request = [&](int r) {
          const BatchCol& c = cols[r];
          const bool sw = c.i >= 0 && (s_raw[max(c.i, 0)] & 0xFFu) != 0;
          const int ea = c.c0 + 4 * (64 * wt + lane), eb2 = ea + PASS;
          ta_n = (sw && ea < c.c1) ? *reinterpret_cast<const uint4*>(pkc + ea) : make_uint4(0, 0, 0, 0);
          tb_n = (sw && eb2 < c.c1) ? *reinterpret_cast<const uint4*>(pkc + eb2) : make_uint4(0, 0, 0, 0);
        };
        request(0);
        for (int it = 0; it < n_rounds; it++) {
          if ((it & 7) == 0 && it) __syncthreads();   // the ring of 8 slots per team wraps (uniform trip count)
          const BatchCol col = cols[it];
          const uint4 ta = ta_n, tb = tb_n;
          if (it + 1 < n_rounds) request(it + 1);
          const int i = col.i, c0 = col.c0, c1 = col.c1;
          if (i < 0) continue;
          if (c0 == c1) continue;
          const uint32_t m = s_raw[i];
          const uint32_t dneg8 = (m >> 8) & 0xFFu, ndz8 = ~(m >> 16) & 0xFFu;
          const bool sweep = (m & 0xFFu) != 0;   // a het site (in every state); a hom site's decision needs no column sum
          if (!sweep && wt) continue;            // (its team's first wave decides it)
          uint32_t lo[8]; int hi[8];
#pragma unroll
          for (int s = 0; s < 8; s++) { lo[s] = 0; hi[s] = 0; }
          unsigned long long* const ts = &t_sum[(team * 8 + (it & 7)) * 8];
          unsigned* const tc = &t_cnt[team * 8 + (it & 7)];
          auto flush = [&]() {   // the wave's partial sums of the eight states into the team's slots (lane l < 8 holds state bitrev3(l))
            const long long M = wave_reduce8(lo, hi, lane);
#pragma unroll
            for (int s = 0; s < 8; s++) { lo[s] = 0; hi[s] = 0; }
            if (lane < 8 && M) __hip_atomic_fetch_add(&ts[bitrev3(lane)], (unsigned long long)M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          };
          auto run4 = [&](const uint4& t, int e) {
            const uint32_t en[4] = {t.x, t.y, t.z, t.w};
            uint32_t sg[4];
#pragma unroll
            for (int u = 0; u < 4; u++) sg[u] = sig8[en[u] & 0xFFFFFFu];   // (plain loads: behind the invalidate above)
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const uint32_t x = en[u] >> 24;
              const uint2 w2 = s_w[x & 31u];
              const bool valid = e + u < c1;
              const uint32_t wlo = valid ? w2.x : 0u; const int whi = valid ? (int)w2.y : 0;
              const uint32_t hit = ~(((x & 32u) ? 0xFFu : 0u) ^ sg[u] ^ dneg8) & ndz8;   // allele == sigma * delta, delta != 0
#pragma unroll
              for (int s = 0; s < 8; s++) {
                const uint32_t b = (hit >> s) & 1u;
                lo[s] += __umul24(b, wlo); hi[s] += __mul24((int)b, whi);
              }
            }
          };
          if (sweep) {
            const int ea = c0 + 4 * (64 * wt + lane), eb2 = ea + PASS;
            if (ea < c1) run4(ta, ea);
            if (eb2 < c1) run4(tb, eb2);
            int n = 0;   // (wave_reduce8 takes limb sums of up to 32 entries per lane: columns beyond 8 192 entries flush on the way)
            for (int eb = c0 + 2 * PASS; eb < c1; eb += PASS) {   // (wave-uniform trip count: flush() is a cross-lane step)
              const int e = eb + 4 * (64 * wt + lane);
              if (e < c1) run4(*reinterpret_cast<const uint4*>(pkc + e), e);
              if (++n == 6) { flush(); n = 0; }
            }
            flush();
          }
          long long M = 0;
          const long long F = col.F, Wt = col.Wt, D2 = col.D2, D3 = col.D3;
          const int fpi = col.fp;
          unsigned arrived = BATCH_TW - 1;
          if (sweep) {
            if (lane == 0) arrived = __hip_atomic_fetch_add(tc, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            arrived = (unsigned)__builtin_amdgcn_readfirstlane((int)arrived);
          }
          if (arrived == BATCH_TW - 1) {
            bool flipd = false, chg = false, to3 = false;
            if (lane < 8) {
              if (sweep) {
                M = (long long)__hip_atomic_load(&ts[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&ts[lane], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (lane == 0) __hip_atomic_store(tc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
              if ((act >> lane) & 1u) {
                const int hh = ((m >> lane) & 1u) ? 0 : (((m >> (24 + lane)) & 1u) ? -1 : 1);
                const bool dz = ((m >> (16 + lane)) & 1u) != 0;
                // data terms of (d,0) (-d,0) (d,+1) (d,-1); with_genotype is false: a het site stays het (it may flip), a hom
                // site may change between homref and homvar.  The priors are equal inside a class, so the data terms decide.
                const long long D0 = F + M, D1 = F + Wt - M;
                long long chosen = hh == 0 ? D0 : (hh == 1 ? D2 : D3);
                bool any = false;
                if (fpi) {
                  if (hh == 0) { if (D1 > D0) { chosen = D1; any = true; flipd = !dz; } }
                  else {
                    const long long n2 = D2 + lut.f_homref, n3 = D3 + lut.f_homvar;
                    to3 = n3 > n2;                       // first maximum wins: homref on a tie
                    const long long ncur = hh == 1 ? n2 : n3, nch = to3 ? n3 : n2;
                    if (nch > ncur) any = true;
                    chg = to3 != (hh == -1);
                    chosen = to3 ? D3 : D2;
                  }
                }
                acc += chosen;
                if (any) anyb |= 1u << lane;
              }
            }
            const uint32_t fd = (uint32_t)__ballot(flipd) & 0xFFu, cm = (uint32_t)__ballot(chg) & 0xFFu, tv = (uint32_t)__ballot(to3) & 0xFFu;
            uint32_t nm = m ^ (fd << 8);
            nm = (nm & ~(cm << 24)) | ((tv & cm) << 24);
            if (lane == 0 && nm != m) cstore(&m32[i], nm);
          }
        }
      }
      __syncthreads();
      if (C.dbg && threadIdx.x == 0 && blk < 1024) C.dbg[16 + 1024 + blk] += (long long)wall_clock64() - wg_t1;
      tick(12);
      // ---- barrier with the states' objective sums and "changed" bits
      if (lane < 8 && acc) atomicAdd(&s_acc[lane], (unsigned long long)acc);
      if (anyb) atomicOr(&s_misc[0], anyb);
      __syncthreads();
      if (threadIdx.x < 8) { cstore(&B->arr[blk][1 + threadIdx.x], s_acc[threadIdx.x]); s_acc[threadIdx.x] = 0; }
      const uint32_t mine = s_misc[0];
      __syncthreads();
      if (threadIdx.x == 0) s_misc[0] = 0;
      iters++;
      // a state has settled when nothing changed in this iteration, and stops after 21 iterations (phase.rs:967-972)
      const uint32_t act0 = act;
      const bool last_it = iters > 20;
      const uint32_t changed = bar(mine, [&](uint32_t ch) { return last_it ? act0 : (act0 & ~ch); });
      const uint32_t done = last_it ? act0 : (act0 & ~changed);
#pragma unroll
      for (int s = 0; s < 8; s++) if ((done >> s) & 1u) obj[s] = (long long)s_out[s];
      act &= ~done;
      __syncthreads();   // (s_out is the next barrier's)
      it_total++;
      tick(13);
    }
    // ---- commit in order: the first half-round of the batch that raises the best objective (owner-local)
    int win = -1;
#pragma unroll
    for (int s = 0; s < 8; s++) if (win < 0 && s < nact && obj[s] > best) { win = s; best = obj[s]; }
    if (win >= 0) {
      for (int i = blk + nblk * (int)threadIdx.x; i < S; i += sc.nt()) {
        const uint32_t m = cload(&m32[i]);
        cstore(&v.bdl[i], (int8_t)(((m >> (16 + win)) & 1u) ? 0 : (((m >> (8 + win)) & 1u) ? -1 : 1)));
        cstore(&v.bet[i], (int8_t)(((m >> win) & 1u) ? 0 : (((m >> (24 + win)) & 1u) ? -1 : 1)));
      }
      for (int b = wj0; b < n_blocks; b += nw) {
        const int p = 64 * b + lane;
        const uint32_t x = p < R ? (uint32_t)cload(&sig8[p]) : 0u;
        const unsigned long long bb = __ballot((x >> win) & 1u);
        if (lane == 0) cstore(&bs64[b], bb);
      }
      h += win + 1;
    } else h += 8;
  }
  if (C.dbg && sc.tid() == 0) C.dbg[15] += it_total;
  // ---- result: best sigma bits back to bytes (the best delta / eta arrays are up to date)
  for (int b = wj0; b < n_blocks; b += nw) {
    const int p = 64 * b + lane;
    if (p < R) v.bsg[perm[p]] = ((cload(&bs64[b]) >> lane) & 1ull) ? 1 : -1;
  }
  if (sc.tid() == 0) C.P.st_obj[slot] = best;
  return true;
}
