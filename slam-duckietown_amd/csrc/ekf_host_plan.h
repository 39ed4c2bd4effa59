// Host-side planning logic of the EKF-SLAM core: everything that decides WHAT is launched -- the covariance pass's kernel and
// launch shape, its work queues and static shares, how many steps of an uploaded stream form a fused cadence, the step
// records and their active bound, the validation of observation lists -- as plain C++ on plain data (no HIP type, no device
// call).  ekf_api.hip's handle derives from HostPlan and calls these; the same header compiles with plain g++
// (-DEKF_HOST_ONLY), and tests/host_plan_check.cpp runs it under -fsanitize=address,undefined: enumerations of the queue /
// share arithmetic plus randomised invariants of the planning functions (tests/test_cpu_host.py builds and runs it).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "ekf_device.h"

namespace ekf {

constexpr int RS_ROWS = 128;            // rows of a slab of the row-slab pass = 8 waves x 16

// What the planning functions read of a handle (ekf_handle derives from this).
struct HostPlan {
  int device = 0, n_max = 0, ld = 0, rows = 0, batch = 0;
  long pstride = 0;
  ekf_config cfg{};
  int cu_count = 0;
  int pending_k = 0, pending_steps = 0;   // ranks / steps appended to (V, W) since the last flush
  std::vector<int> n;             // state size per trajectory
  std::vector<int> neff_enq;      // active bound of the last ENQUEUED step (what dso[b].neff holds)
  std::vector<int> neff;          // active bound per trajectory (<= n): indices beyond were never correlated
  std::vector<int> floor_host;    // what dfloor holds (see push_floor)
  bool sizes_dirty = false;       // the device grew the state: n / neff must be read back before use
  int stream_steps = 0;
  std::vector<int> stream_mhi;    // per step: most observations of any trajectory
  std::vector<unsigned char> stream_m;   // per (step, trajectory): observations the kernels will process (0 with the measurement model off)
  std::vector<int> stream_own;    // per (step, trajectory): active bound from the stream's OWN observations up to that step
  std::vector<int> stream_maxlm;  // per trajectory: landmarks the stream needs in the state (largest index + 1)
  int opt_active_bound = 1;       // 0 = always treat the whole state as active
  int opt_rank_limit = KTOT;      // automatic cadence: flush when the next step would exceed this many ranks
  int opt_pass_kernel = -1;       // -1 = auto, 0 = k_flush (column strips), 2 = k_flush_rs (row slabs)
  int opt_fused_step = 1;         // 1 = one launch per step where the launch is small (k_step_split), 0 = always two
  int opt_fused_cadence = 1;      // 1 = uploaded streams run whole cadences as one solve + one panel launch (ekf_cadence.hip)
  int opt_lookahead = 1;
  int opt_small_state = 1;        // 1 = small filters (n_max <= 79: up to 38 landmarks) run in ONE workgroup, P in LDS (ekf_small.hip)
  int opt_rows_per_block = 0;     // 0 = auto (flush kernel: rows per workgroup, multiple of 16)
  int opt_pass_chunk = 0;         // 0 = auto (k_flush_rs: strips per unit)
  int opt_share_order = 1;        // 1 = static shares dealt to the XCDs by starting column (order_pass_shares), 0 = as built
  int opt_pass_workgroups = 0;    // 0 = one per CU (k_flush_rs: persistent workgroups; fewer leaves CUs to other streams)
  int opt_flush_every = 0;        // 0 = auto; k = flush the pending low-rank update after k steps
  int opt_streaming = -1;         // -1 = auto (by working-set size), 0 = resident kernel, 1 = nontemporal kernel
};

// ---- the work queues of k_flush_rs (one per XCD): how many units queue g2 holds and which unit its u-th one is ----
// By `mode`:
//   0  uniform ("pass_chunk" set, or fewer than 8 trajectories): trajectories g2, g2 + 8, ..., every slab in `nch`
//      chunks, chunk-major;
//   1  pairs (batch a multiple of 8): trajectories g2, g2 + 8, ... one after the other, whole slabs longest
//      first -- the workgroup that got the longest slab of one trajectory gets the shortest of the next; the last of
//      an odd number (8 trajectories: the only one) has no partner and its slabs, only they, are cut into `nch` chunks;
//   2  dealt (any other batch): the queue's own trajectories among the first 8 * (batch / 8), plus the slabs
//      rb = (g2 - j) mod 8, + 8, ... of each of the batch-modulo-8 last trajectories j -- every queue carries the same
//      work -- whole slabs, longest first over ALL of them (slab index major): list scheduling in that order is
//      as good as the longest slab allows; the price is that an XCD walks the V strips of several trajectories at
//      once (1-3 % on the batches where mode 1 applies, hence not used there).  N=2000, 20 trajectories: 496 us
//      against 524 us with mode 1, 28: 662 against 700; 24 (mode 1): 560 against 584 with mode 2.
//   3  dealt halves (8 < batch <= 12, where one trajectory per queue leaves a workgroup less than two slabs): as mode 2,
//      but every slab in two chunks of cs = 2 h strips; chunk 1 of slab rb is as long as slab rb + h, so handing out
//      "chunk 0 of slab v, chunk 1 of slab v - h" for v = 0, 1, ... is again longest first.  (`nch` carries h.)
// A unit is (trajectory * nrb + slab) * 1024 + chunk, chunk = 1023 for a whole slab.  Plain integer functions, also
// compiled for the host: tests/test_cpu_host.py enumerates them through ekf_debug_pass_units and checks that every
// (trajectory, slab, chunk) comes exactly once.
__host__ __device__ inline int rs_queue_count(int g2, int batch, int nrb, int nch, int mode) {
  const int upt = nrb * nch;
  if (mode == 3) return 2 * rs_queue_count(g2, batch, nrb, 1, 2);
  if (mode == 2) {
    const int nfull = batch >> 3, nleft = batch & 7;
    int dealt = 0;                                     // slabs rb < nrb with ((g2 - rb) & 7) < nleft
    for (int j = 0; j < nleft; ++j) {
      const int r0 = (g2 - j) & 7;
      dealt += (r0 < nrb) ? ((nrb - r0 + 7) >> 3) : 0;
    }
    return nfull * nrb + dealt;
  }
  const int tq = (g2 < batch) ? ((batch - g2 + 7) >> 3) : 0;
  if (mode == 0) return tq * upt;
  const int lone = tq & 1;
  return (tq - lone) * nrb + lone * upt;
}
__host__ __device__ inline int rs_queue_unit(int g2, int u, int batch, int nrb, int nch, int mode) {
  const int upt = nrb * nch;
  int r = u;
  if (mode == 3) {
    const int nfull = batch >> 3, nleft = batch & 7, h = nch;
    for (int v = 0; v < nrb + h; ++v) {                // (a few dozen iterations, once per unit, one thread)
      for (int chunk = 0; chunk < 2; ++chunk) {
        const int rb = v - chunk * h;
        if (rb < 0 || rb >= nrb) continue;
        const int j = (g2 - rb) & 7;
        const int ci = nfull + (j < nleft ? 1 : 0);
        if (r < ci) return ((r < nfull ? g2 + 8 * r : 8 * nfull + j) * nrb + rb) * 1024 + chunk;
        r -= ci;
      }
    }
    return -1;                                         // (not reached for u < rs_queue_count)
  }
  if (mode == 0) {
    const int t = r / upt;
    r -= t * upt;
    return ((g2 + 8 * t) * nrb + r % nrb) * 1024 + (nch > 1 ? r / nrb : 1023);
  }
  if (mode == 2) {
    // slab-index major: a block of 8 consecutive slabs holds 8 * nfull own units and nleft dealt ones
    const int nfull = batch >> 3, nleft = batch & 7;
    const int per = 8 * nfull + nleft, blk = r / per;
    r -= blk * per;
    for (int i = 0; i < 8; ++i) {
      const int rb = 8 * blk + i, j = (g2 - rb) & 7;
      const int ci = nfull + (j < nleft ? 1 : 0);
      if (r < ci) return ((r < nfull ? g2 + 8 * r : 8 * nfull + j) * nrb + rb) * 1024 + 1023;
      r -= ci;
    }
    return -1;                                         // (not reached for u < rs_queue_count)
  }
  const int tq = (g2 < batch) ? ((batch - g2 + 7) >> 3) : 0;
  const int whole = (tq - (tq & 1)) * nrb;
  if (r < whole) return ((g2 + 8 * (r / nrb)) * nrb + r % nrb) * 1024 + 1023;
  r -= whole;
  return ((g2 + 8 * (tq - 1)) * nrb + r % nrb) * 1024 + r / nrb;
}
// ---- mode 4: equal static shares (a few LONG trajectories, e.g. N = 8000 x 1: 126 slabs for 256 CUs) ----
// Whole slabs cannot balance 256 workgroups there, and dynamically handed-out chunks end in a tail as long as a chunk
// while every unit boundary costs about two strips' worth (pipeline fill and drain).  So the batch's strips -- trajectory
// by trajectory, slab by slab, each slab from its right end to the diagonal -- are cut into one contiguous share per
// workgroup of equal COST (strips + RS_PIECE_COST per piece): a share is a handful of pieces (trajectory, slab, first
// strip, strips), at most RS_PIECES.  No queue, no atomics; the table depends on (batch, n_hi, workgroups) only and is
// cached on the device.  Returns the pieces of the longest share, 0 if some share would need more than RS_PIECES.
// (Groups of 2 / 4 / 8 workgroups walking ADJACENT strips of the same rows in step -- longer contiguous row segments in
//  flight at any time -- were measured at N = 8000 x 1: 401 / 439 / 471 us against 391 us: not adopted.)
constexpr int RS_PIECES = 16;
constexpr int RS_PIECE_COST = 2;
inline int build_pass_shares(int batch, int n_hi, int workgroups, int* out /* workgroups x RS_PIECES x 4 */) {
  const int nrb = (n_hi + RS_ROWS - 1) / RS_ROWS, s_last = (n_hi - 1) >> 6;
  long rem_strips = 0;
  for (int rb = 0; rb < nrb; ++rb) rem_strips += s_last - 2 * rb + 1;
  rem_strips *= batch;
  long rem_slabs = (long)batch * nrb;                  // slabs not yet started
  for (int i = 0; i < workgroups * RS_PIECES * 4; ++i) out[i] = 0;
  int w = 0, k = 0, longest = 0;
  // what a share may cost: what is left (strips + a piece per slab still to start + a piece per share still to open,
  // the continuation of a slab cut by a share boundary) over the shares left -- recomputed whenever a share is opened
  auto budget_now = [&](int slab_left) {
    const long left = rem_strips + RS_PIECE_COST * (rem_slabs + (slab_left > 0 ? 1 : 0) + (workgroups - w - 1));
    return (double)left / (double)(workgroups - w);
  };
  double budget = budget_now(0), used = 0.0;
  for (int b = 0; b < batch; ++b)
    for (int v = 0; v < nrb; ++v) {
      // slabs of a trajectory alternately from both ends (longest, shortest, second longest, ...): the many short slabs
      // near the diagonal's end do not pile up in one share
      const int rb = (v & 1) ? nrb - 1 - (v >> 1) : (v >> 1);
      int S = s_last - 2 * rb + 1, start = 0;
      --rem_slabs;
      while (S > 0) {
        if (k > 0 && used + RS_PIECE_COST + 1 > budget && w + 1 < workgroups) {   // no room for even one strip: next share
          ++w;
          k = 0;
          used = 0.0;
          budget = budget_now(S);
        }
        const int room = w + 1 < workgroups ? (int)(budget - used - RS_PIECE_COST + 0.5) : S;
        const int cnt = room < 1 ? 1 : (room < S ? room : S);
        if (k >= RS_PIECES) return 0;
        int* pc = out + ((long)w * RS_PIECES + k) * 4;
        pc[0] = b;
        pc[1] = rb;
        pc[2] = start;
        pc[3] = cnt;
        ++k;
        longest = k > longest ? k : longest;
        used += cnt + RS_PIECE_COST;
        start += cnt;
        S -= cnt;
        rem_strips -= cnt;
        if (S > 0 && w + 1 < workgroups) {             // the slab goes on in the next share
          ++w;
          k = 0;
          used = 0.0;
          budget = budget_now(S);
        }
      }
    }
  return longest;
}
inline int pass_share_pieces() { return RS_PIECES; }

// (test hook) all units of all queues in hand-out order; returns their number (may exceed cap)
inline int debug_pass_units(int batch, int nrb, int nch, int mode, int* out, int cap) {
  int total = 0;
  for (int g2 = 0; g2 < 8; ++g2) {
    const int cnt = rs_queue_count(g2, batch, nrb, nch, mode);
    for (int u = 0; u < cnt; ++u, ++total)
      if (total < cap) out[total] = rs_queue_unit(g2, u, batch, nrb, nch, mode);
  }
  return total;
}


// ---- step machinery -------------------------------------------------------------------------
inline int cap_for(int m) { return m <= 1 ? 1 : m <= 2 ? 2 : m <= 4 ? 4 : m <= 8 ? 8 : 16; }

// Rows per workgroup of k_flush: every wave re-reads its V strip (K x 1 KiB, from L2) per row block, so
// the block must be long where many ranks are pending, and short enough to give every CU several waves.
// Streaming launches with at least four 256-row workgroups per CU (big batches when k_flush is forced; N=8000): 256
// rows, tuned in round 1 (N=8000, 1 trajectory: 445 us against 488 us with 96 rows).
// Launches of several rounds of workgroups per CU: 96 rows.  Small launches leave the CUs with one to three workgroups
// each (two resident at a time) and the pass takes as long as the busiest CU, roughly (rows of a block) x (0.2 + load),
// load = workgroups per CU, rounded up to the next half where it is below that: the block height minimising it is
// taken.  N=2000, 1 trajectory: 80 rows (441 workgroups) 46 us, against 52 us with 96 rows (367) and 55 us with 64
// (543); N=500, 1 trajectory: 64 rows, 23 us against 30 us; N=2000, 2 / 4 / 6 trajectories (streaming): 96 rows 84 /
// 135 / 199 us against 108 / 164 / 208 us with 256 (profiles/r02_rows_per_block.txt).
inline int flush_workgroups(int n_hi, int rows_per_block) {
  const int gx = (n_hi + 255) / 256, gy = (n_hi + rows_per_block - 1) / rows_per_block;
  int total = 0;                                       // (the launcher's count: workgroups that reach the upper triangle)
  for (int by = 0; by < gy; ++by) total += std::max(0, gx - (by * rows_per_block) / 256);
  return total;
}
inline int flush_rows_per_block(const HostPlan* h, bool streaming, int n_hi) {
  if (h->opt_rows_per_block > 0) return (h->opt_rows_per_block + 15) / 16 * 16;
  const long cus = h->cu_count;
  if (streaming && (long)flush_workgroups(n_hi, 256) * h->batch >= 4 * cus) {
    // 256 rows, or 512 where that fills its rounds of 2 x CUs workgroups better (N=8000, 1 trajectory: 1024 workgroups
    // = two full rounds, 421 us against 454 us with 256 rows = 2016 workgroups; 2 trajectories 840 / 852 us)
    const long slots = 2 * cus;
    auto fill = [&](int r) {
      const long w = (long)flush_workgroups(n_hi, r) * h->batch;
      return (double)w / (double)((w + slots - 1) / slots * slots);
    };
    if ((long)flush_workgroups(n_hi, 512) * h->batch >= 2 * slots && fill(512) > fill(256) + 0.01) return 512;
    return 256;
  }
  if ((long)flush_workgroups(n_hi, 96) * h->batch > 5 * cus / 2) return 96;
  int best = 96;
  double best_cost = 0.0;
  for (int r = 64; r <= 256; r += 16) {
    const double load = (double)flush_workgroups(n_hi, r) * h->batch / (double)cus;
    const double cost = r * (0.2 + std::max(load, std::ceil(load) - 0.5));
    if (best_cost == 0.0 || cost < best_cost) {
      best_cost = cost;
      best = r;
    }
  }
  return best;
}

// The covariances of the batch stream through HBM when they cannot stay in the 256 MiB Infinity Cache.
inline bool streaming_pass(const HostPlan* h, int n_hi) {
  if (h->opt_streaming >= 0) return h->opt_streaming != 0;
  return (double)h->batch * 8.0 * n_hi * n_hi > 192.0e6;
}

// What the next covariance pass will launch (decided from the handle's state alone, so that the caller can ask before
// it launches).
struct PassPlan {
  int n_hi, e_hi, nkt, kernel, rs_workgroups;
  bool streaming, long_few, beside;                    // beside: the row-slab pass leaves CUs free for a solve beside it
};
inline PassPlan plan_pass(const HostPlan* h) {
  PassPlan p;
  p.n_hi = h->sizes_dirty ? h->n_max : *std::max_element(h->n.begin(), h->n.end());
  p.e_hi = 3;                                          // the grid covers the largest active bound of the batch
  for (int b = 0; b < h->batch; ++b) p.e_hi = std::max(p.e_hi, std::min(h->n[b], h->neff_enq[b]));
  if (h->sizes_dirty) p.e_hi = h->n_max;
  p.streaming = streaming_pass(h, p.n_hi);
  p.nkt = (h->pending_k + 3) / 4;
  p.kernel = h->opt_pass_kernel;
  p.rs_workgroups = h->opt_pass_workgroups > 0 ? std::min(h->opt_pass_workgroups, h->cu_count) : h->cu_count;
  // A few LONG trajectories (N = 8000 x 1: 126 slabs of up to 251 strips for 256 CUs): the row-slab pass with one equal
  // static share of the strips per workgroup (build_pass_shares) -- where a share is long enough (>= 40 strips) for
  // the pipeline fills at its piece boundaries not to matter.  The same for 10 .. 14 trajectories, where the queues
  // hold one to two whole slabs per workgroup and cannot balance them (N = 2000, 80 ranks, queues -> shares: x 10
  // 290 -> 261 us, x 11 321 -> 289, x 12 334 -> 303, x 13 350 -> 335, x 14 352 -> 346; N = 3000 x 12 753 -> 706;
  // 8, 9, 15 - 17 and from 23 on the queues are as good or better, 18 - 22 gain 2 - 5 % at N = 2000 but lose at N = 3000:
  // profiles/r03_pass_vs_batch.txt).
  const long slabs = (p.e_hi + 127) / 128, s_last = (p.e_hi - 1) >> 6;
  const long strips = (long)h->batch * (slabs * (s_last + 1) - slabs * (slabs - 1));
  p.long_few = (h->batch < 8 || (h->batch >= 10 && h->batch <= 14)) && h->opt_pass_chunk == 0 && strips >= 40L * p.rs_workgroups;
  // auto: the row-slab form where the batch streams through HBM and has at least one 128-row slab per CU (below
  // three per CU the slabs are cut into chunks of strips) or is a few long trajectories; measured at N=2000: 8
  // trajectories 256 us against 266 us with k_flush, 4 trajectories 166 / 164 us, 1 trajectory 97 / 52 us (pipeline
  // fills dominate)
  // SHORT slabs (n < 3000: fewer than 24 slabs of at most 47 strips) need more of them before the row-slab form's pipeline fills
  // are paid for -- 2.35 slabs per CU up to n = 2048, one per CU from n = 3072 on (N = 500 x 32, 64: the column strips +10 %, +3 %,
  // x 128: the row slabs +9 %; N = 1000 x 16, 32: strips +20 %, +7 %, x 64: slabs +2.5 %; N = 1500 x 16 and N = 2000 x 8: slabs
  // +4 %, +12 %; tools/mid_size_probe.sh, profiles/r04_n_sweep.txt)
  const double per_cu = slabs >= 24 ? 1.0 : slabs <= 16 ? 2.35 : 2.35 - (slabs - 16) * (1.35 / 8.0);
  if (p.kernel < 0)
    p.kernel = (p.streaming && ((double)h->batch * slabs >= per_cu * h->cu_count || p.long_few)) ? 2 : 0;
  // A few long trajectories on static shares: the pass leaves one CU per trajectory free, so that the next cadence's solve
  // (one workgroup per trajectory) can run beside it (the look-ahead of ekf_stream_run); always, not only when a solve
  // follows: the share table is built per workgroup count (N = 8000 x 1: 255 instead of 256 workgroups, 0.4 %).
  p.beside = p.kernel == 2 && p.long_few && h->batch < 8 && h->opt_lookahead && h->opt_pass_workgroups == 0 &&
             h->cu_count > 8 * h->batch;
  if (p.beside) p.rs_workgroups = h->cu_count - h->batch;
  return p;
}

// Which workgroup gets which static share.  build_pass_shares cuts the strips slab by slab, so consecutive shares are
// consecutive pieces of the same rows: at any time the workgroups of an XCD (equal blockIdx % 8) sit on 32 different
// column strips, every V strip they stage is used by one workgroup only, and V (80 ranks x ld doubles: 10 MB at N = 8000)
// does not fit an XCD's 4 MB L2 -- each of the 15 876 strip visits of a pass fetches its 40 KB from the Infinity Cache
// (650 MB per pass beside the 4.1 GB of P).  Workgroups advance at the same rate, so shares that START on the same column
// stay on the same column: the shares are sorted by (trajectory, first column) and dealt to the XCDs in runs, and the 32
// workgroups of an XCD walk (nearly) the same V strips together -- one fetch per XCD instead of one per workgroup.
inline void order_pass_shares(int workgroups, int pieces, int* table_ptr, size_t words) {
  std::vector<int> table(table_ptr, table_ptr + words);
  std::vector<int> order(workgroups);
  for (int w = 0; w < workgroups; ++w) order[w] = w;
  auto key = [&](int w) { return ((long)table[(size_t)w * pieces * 4] << 32) + table[(size_t)w * pieces * 4 + 2]; };
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key(a) < key(b); });
  std::vector<int> slots;                              // blockIdx values XCD by XCD (workgroups go round-robin over the 8 XCDs)
  slots.reserve(workgroups);
  for (int x = 0; x < 8; ++x)
    for (int w = x; w < workgroups; w += 8) slots.push_back(w);
  std::vector<int> out(table.size(), 0);
  for (int q = 0; q < workgroups; ++q)
    std::copy_n(table.begin() + (size_t)order[q] * pieces * 4, (size_t)pieces * 4, out.begin() + (size_t)slots[q] * pieces * 4);
  std::copy(out.begin(), out.end(), table_ptr);
}

// ---- the packed cadences of ekf_stream_run (ekf_device.h: CadPlan) ----
// Steps [k, end) of the uploaded stream as a sequence of fused cadences, planned in one go.  Every trajectory walks its own
// flat sequence of predictions and landmark updates; a cadence gives it
//   * whole steps while their landmarks fit the slots that are left (steps that observe nothing are free),
//   * then, if slots are left and the next step does not fit, that step's prediction and as many of its landmarks as do fit
//     (the rest open the trajectory's next cadence -- no second prediction),
// within `slot_limit` landmark updates ("rank_limit" / 2, at most CAD_SLOTS) and `step_limit` touched steps ("flush_every",
// at most CAD_SLOTS: the kernels' per-step arrays).  The covariance pass follows every cadence but possibly the last, with
// as many ranks as the busiest trajectory appended (the others zero-fill): the number of passes of a run is what the
// trajectory with the most landmark updates needs at 40 per pass, however the counts are spread over steps and trajectories.
// A trajectory that has reached `end` idles (ns = 0).  The plan of a trajectory depends on its own observations only.
struct RunPlan {
  int ncad = 0;
  std::vector<CadPlan> entries;          // ncad x batch
  std::vector<int> slots_hi;             // per cadence: most landmark updates of any trajectory
  std::vector<int> steps_hi;             // per cadence: most steps completed by any trajectory
  std::vector<long> steps_sum;           // per cadence: steps completed, summed over the trajectories
};
inline int cadence_slot_limit(const HostPlan* h) { return std::max(1, std::min(CAD_SLOTS, h->opt_rank_limit / 2)); }
inline int cadence_step_limit(const HostPlan* h) { return h->opt_flush_every > 0 ? std::min(CAD_SLOTS, h->opt_flush_every) : CAD_SLOTS; }
inline bool cadences_possible(const HostPlan* h) {
  return h->opt_fused_cadence && h->pending_k == 0 && !h->sizes_dirty && (int)h->stream_m.size() == h->stream_steps * h->batch;
}
inline void plan_cadences(const HostPlan* h, int k, int end, RunPlan& rp) {
  const int B = h->batch, slot_limit = cadence_slot_limit(h), step_limit = cadence_step_limit(h);
  rp.ncad = 0;
  rp.entries.clear();
  rp.slots_hi.clear();
  rp.steps_hi.clear();
  rp.steps_sum.clear();
  std::vector<int> ct(B, k), cj(B, 0);   // cursor per trajectory: next step, next landmark of it (> 0: the step is cut)
  for (;;) {
    bool any = false;
    for (int b = 0; b < B; ++b) any = any || ct[b] < end;
    if (!any) break;
    const size_t base = rp.entries.size();
    rp.entries.resize(base + B);
    int slots_hi = 0, steps_hi = 0;
    long steps_sum = 0;
    for (int b = 0; b < B; ++b) {
      CadPlan& e = rp.entries[base + b];
      e = CadPlan{};
      e.t0 = ct[b];
      e.j0 = cj[b];
      int t = ct[b], j = cj[b], slots = 0, ns = 0, done = 0;
      e.jend = 0;
      while (t < end && ns < step_limit) {
        const int m = h->stream_m[(size_t)t * B + b], left = m - j;
        if (slots + left <= slot_limit) {              // the whole (rest of the) step
          slots += left;
          e.jend = m;
          ++ns;
          ++done;
          ++t;
          j = 0;
          continue;
        }
        const int room = slot_limit - slots;
        if (room > 0) {                                // cut: its prediction (if not yet done) and `room` landmarks
          slots += room;
          j += room;
          e.jend = j;
          ++ns;
        }
        break;
      }
      e.ns = ns;
      e.nslots = slots;
      const int t_last = ns > 0 ? e.t0 + ns - 1 : std::max(e.t0 - 1, 0);   // (an idle trajectory keeps its last step's bound)
      e.neff = std::min(h->n[b], std::max(h->floor_host[b], h->stream_own[(size_t)t_last * B + b]));
      ct[b] = t;
      cj[b] = j;
      slots_hi = std::max(slots_hi, slots);
      steps_hi = std::max(steps_hi, done);
      steps_sum += done;
    }
    rp.slots_hi.push_back(slots_hi);
    rp.steps_hi.push_back(steps_hi);
    rp.steps_sum.push_back(steps_sum);
    rp.ncad += 1;
  }
}

// Validate one trajectory's whole observation list (all device passes of it) before any handle state changes:
// indices inside the current state, no index twice (the reference keys observations by landmark index,
// replay_no_ros.py:312-313).
// Returns nullptr when the list is good, else what is wrong with it.
inline const char* validate_obs(const HostPlan* h, int b, const int* idx, int m, std::vector<unsigned char>& seen) {
  const int n_lm = (h->n[b] - 3) / 2;
  seen.assign((size_t)std::max(n_lm, 1), 0);
  for (int i = 0; i < m; ++i) {
    const int id = idx[i];
    if (id < 0 || id >= n_lm) return "landmark index outside the current state (add_landmarks first)";
    if (seen[id]) return "duplicate landmark index in one update (the reference keys observations by index, replay_no_ros.py:312-313)";
    seen[id] = 1;
  }
  return nullptr;
}

// Fill StepIn for pass `p` (landmarks [p*MMAX, ...)) of a validated list; `bound` is the trajectory's running
// active bound (monotone): an observed landmark and everything below it may be correlated from now on.
inline void fill_step(StepIn& s, int n_b, int& bound, double lin, double ang, int flags, const int* idx,
                      const double* range, const double* bearing, int m, int p) {
  s.lin = lin;
  s.ang = ang;
  s.flags = flags;
  const int lo = p * MMAX, cnt = std::max(0, std::min(m - lo, MMAX));
  s.m = cnt;
  for (int i = 0; i < cnt; ++i) {
    s.idx[i] = idx[lo + i];
    s.range[i] = range[lo + i];
    s.bearing[i] = bearing[lo + i];
  }
  for (int i = cnt; i < MMAX; ++i) { s.idx[i] = 0; s.range[i] = 0.0; s.bearing[i] = 0.0; }
  for (int i = 0; i < cnt; ++i) bound = std::max(bound, 3 + 2 * (s.idx[i] + 1));
  bound = std::min(bound, n_b);
  s.neff = bound;                                      // (k_solve raises it to the handle's floor, see push_floor)
  s.pad = 0;
}

}  // namespace ekf
