// Sanitizer build of the host-side planning logic ("-fsanitize=address,undefined host build"):
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -DEKF_HOST_ONLY
//       -I slam-duckietown_amd/csrc -I include tests/host_plan_check.cpp -o host_plan_check
// Everything below runs the SAME source the library ships (slam-duckietown_amd/csrc/ekf_host_plan.h, ekf_device.h): the
// work queues and equal static shares of the row-slab covariance pass, the cadence / pass planning, the step records and
// their active bound, the validation of observation lists, the covariance's device layout.  Enumerations (every unit /
// strip / matrix entry exactly once) plus randomised invariants; any sanitizer report or failed check ends the run with a
// non-zero status.  tests/test_cpu_host.py::test_host_planning_logic_under_the_sanitizers builds and runs it (CPU only).
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <set>
#include <vector>

#include "ekf_host_plan.h"

using namespace ekf;

static long checks = 0;
#define CHECK(cond, ...)                                              \
  do {                                                                \
    ++checks;                                                         \
    if (!(cond)) {                                                    \
      std::fprintf(stderr, "FAILED %s:%d: %s  [", __FILE__, __LINE__, #cond); \
      std::fprintf(stderr, __VA_ARGS__);                              \
      std::fprintf(stderr, "]\n");                                    \
      std::exit(1);                                                   \
    }                                                                 \
  } while (0)

// ---- the covariance's device layout: an injection into the allocation, row-major inside a panel ----
static void check_layout() {
  for (int ld : {64, 128, 1024, 4096, 4160, 8064, 8192, 16064, 21824}) {
    const int rows = ld > PPW ? ld : std::min(ld, 4032 + 64);
    const long alloc = p_alloc(rows, ld);
    CHECK(p_lds(ld) == std::min(ld, PPW), "ld %d", ld);
    CHECK(p_panels(ld) == (ld <= PPW ? 1 : (ld + PPW - 1) / PPW), "ld %d", ld);
    // a band of rows x every column: distinct offsets inside the allocation, consecutive along a row inside a panel
    std::set<long> seen;
    for (int i : std::set<int>{0, 1, 2, 63, rows / 2, rows - 1}) {
      for (int j = 0; j < ld; ++j) {
        const long o = p_index(ld, i, j);
        CHECK(o >= 0 && o < alloc, "ld %d (%d, %d) -> %ld of %ld", ld, i, j, o, alloc);
        CHECK(seen.insert(o).second, "ld %d (%d, %d) collides", ld, i, j);
        if (j % PPW != 0 && j > 0) CHECK(o == p_index(ld, i, j - 1) + 1, "ld %d (%d, %d) not contiguous", ld, i, j);
        CHECK((unsigned long long)p_col8(ld, (unsigned)j) == (unsigned long long)p_col(ld, j) * 8ull, "ld %d col %d", ld, j);
      }
    }
    // a 64-column strip never straddles a panel
    for (int j0 = 0; j0 + 64 <= ld; j0 += 64) CHECK(p_col(ld, j0 + 63) == p_col(ld, j0) + 63, "ld %d strip %d", ld, j0);
    // the largest byte offset fits 32 bits at the documented limit
    CHECK((unsigned long long)alloc * 8ull < (1ull << 32), "ld %d allocation", ld);
  }
  const int rows_limit = (EKF_N_MAX_LIMIT + 63) / 64 * 64;
  CHECK((unsigned long long)p_alloc(rows_limit, rows_limit) * 8ull < (1ull << 32), "limit");
  CHECK((unsigned long long)p_alloc(rows_limit + 64, rows_limit + 64) * 8ull >= (1ull << 32), "limit + 64");
}

// ---- work queues of the row-slab pass: every (trajectory, slab) once, whole or as all of its chunks ----
static void check_queues() {
  std::vector<int> buf(70000);
  for (int batch = 1; batch <= 100; batch = batch < 41 ? batch + 1 : batch + 23)
    for (int nrb : {1, 2, 5, 7, 8, 9, 15, 16, 17, 32, 126})
      for (auto [nch_in, mode] : std::vector<std::pair<int, int>>{{1, 0}, {2, 0}, {3, 0}, {2, 1}, {1, 2}, {1, 3}, {3, 3}, {8, 3}, {16, 3}}) {
        const int total = debug_pass_units(batch, nrb, nch_in, mode, buf.data(), (int)buf.size());
        CHECK(total > 0 && total <= (int)buf.size(), "batch %d nrb %d mode %d", batch, nrb, mode);
        const int nch = mode == 3 ? 2 : nch_in;
        std::map<int, int> whole, chunks;
        std::set<long> chunk_keys;
        int per_queue = 0;
        for (int g2 = 0; g2 < 8; ++g2) per_queue += rs_queue_count(g2, batch, nrb, nch_in, mode);
        CHECK(per_queue == total, "queue counts %d != %d", per_queue, total);
        for (int u = 0; u < total; ++u) {
          const int unit = buf[u];
          CHECK(unit >= 0, "negative unit");
          const int code = unit & 1023, slab = unit >> 10;
          CHECK(slab / nrb < batch, "trajectory %d of %d", slab / nrb, batch);
          if (code == 1023) ++whole[slab];
          else {
            ++chunks[slab];
            CHECK(code < nch, "chunk %d of %d", code, nch);
            CHECK(chunk_keys.insert((long)slab * 1024 + code).second, "chunk handed out twice");
          }
        }
        for (int slab = 0; slab < batch * nrb; ++slab) {
          const int w = whole.count(slab) ? whole[slab] : 0, c = chunks.count(slab) ? chunks[slab] : 0;
          CHECK((w == 1 && c == 0) || (w == 0 && c == nch), "batch %d nrb %d nch %d mode %d slab %d: %d whole, %d chunks", batch, nrb,
                nch, mode, slab, w, c);
        }
      }
}

// ---- equal static shares: every strip of every slab in exactly one piece, equal cost, any order of the shares ----
static void check_shares(std::mt19937& rng) {
  struct Case { int batch, n, wgs; };
  std::vector<Case> cases = {{1, 16003, 256}, {2, 16003, 256}, {1, 16003, 240}, {3, 12003, 256}, {7, 16003, 256}, {1, 21823, 256},
                             {3, 1403, 8},    {3, 1403, 5},    {1, 4003, 8},    {1, 16003, 255}, {4, 8003, 252},  {12, 4003, 256}};
  for (int r = 0; r < 40; ++r)
    cases.push_back({1 + (int)(rng() % 14), 3 + 2 * (int)(200 + rng() % 9000), 1 + (int)(rng() % 256)});
  for (const Case& c : cases) {
    const int pieces = pass_share_pieces();
    std::vector<int> table((size_t)c.wgs * pieces * 4, -7);
    const int longest = build_pass_shares(c.batch, c.n, c.wgs, table.data());
    if (longest == 0) continue;                        // (no table for this shape: the queue modes are used)
    CHECK(longest >= 1 && longest <= pieces, "longest %d", longest);
    for (int order = 0; order < 2; ++order) {
      if (order) order_pass_shares(c.wgs, pieces, table.data(), table.size());
      const int nrb = (c.n + RS_ROWS - 1) / RS_ROWS, s_last = (c.n - 1) >> 6;
      std::vector<int> seen((size_t)c.batch * nrb * (s_last + 1), 0);
      int cmin = 1 << 30, cmax = 0;
      for (int w = 0; w < c.wgs; ++w) {
        int cost = 0;
        bool ended = false;
        for (int k = 0; k < pieces; ++k) {
          const int* pc = &table[((size_t)w * pieces + k) * 4];
          if (pc[3] <= 0) { ended = true; continue; }
          CHECK(!ended, "a piece behind the end of share %d", w);
          CHECK(pc[0] >= 0 && pc[0] < c.batch && pc[1] >= 0 && pc[1] < nrb && pc[2] >= 0, "piece out of range");
          CHECK(pc[2] + pc[3] <= s_last - 2 * pc[1] + 1, "piece beyond its slab");
          for (int t = pc[2]; t < pc[2] + pc[3]; ++t) ++seen[((size_t)pc[0] * nrb + pc[1]) * (s_last + 1) + t];
          cost += pc[3] + 2;
        }
        cmin = std::min(cmin, cost);
        cmax = std::max(cmax, cost);
      }
      for (int b = 0; b < c.batch; ++b)
        for (int rb = 0; rb < nrb; ++rb)
          for (int t = 0; t <= s_last; ++t)
            CHECK(seen[((size_t)b * nrb + rb) * (s_last + 1) + t] == (t <= s_last - 2 * rb ? 1 : 0), "batch %d n %d wgs %d: strip (%d, %d, %d)",
                  c.batch, c.n, c.wgs, b, rb, t);
      if (c.wgs >= 5 && (long)c.batch * nrb * 4 <= (long)s_last * c.wgs) CHECK(cmax - cmin <= 4, "batch %d n %d wgs %d: costs %d..%d", c.batch, c.n, c.wgs, cmin, cmax);
    }
  }
}

// ---- planning: randomised handles ----
static HostPlan random_plan(std::mt19937& rng) {
  HostPlan h;
  h.batch = 1 + rng() % 40;
  const int N = 1 + rng() % (rng() % 4 == 0 ? 9000 : 2500);
  h.n_max = 3 + 2 * N;
  h.rows = (h.n_max + 63) / 64 * 64;
  h.ld = h.rows;
  if (h.n_max <= PPW) {
    int p2 = 64;
    while (p2 < h.n_max) p2 *= 2;
    h.ld = p2;
  }
  h.pstride = p_alloc(h.rows, h.ld);
  h.cu_count = rng() % 5 == 0 ? 64 + 8 * (int)(rng() % 25) : 256;
  h.n.resize(h.batch);
  h.neff.resize(h.batch);
  h.neff_enq.resize(h.batch);
  h.floor_host.assign(h.batch, 3);
  for (int b = 0; b < h.batch; ++b) {
    h.n[b] = 3 + 2 * (int)(rng() % (N + 1));
    h.neff[b] = std::min(h.n[b], 3 + 2 * (int)(rng() % (N + 1)));
    h.neff_enq[b] = rng() % 2 ? h.neff[b] : h.n[b];
  }
  h.sizes_dirty = rng() % 16 == 0;
  h.pending_k = 4 * (int)(rng() % 21);
  h.opt_pass_kernel = (int[]){-1, -1, -1, 0, 2}[rng() % 5];
  h.opt_pass_workgroups = rng() % 4 == 0 ? 1 + (int)(rng() % 300) : 0;
  h.opt_pass_chunk = rng() % 6 == 0 ? 1 + (int)(rng() % 80) : 0;
  h.opt_lookahead = rng() % 2;
  h.opt_streaming = (int)(rng() % 3) - 1;
  h.opt_rows_per_block = rng() % 5 == 0 ? (int)(rng() % 600) : 0;
  h.opt_flush_every = rng() % 3 == 0 ? (int)(rng() % 12) : 0;
  h.opt_rank_limit = rng() % 3 == 0 ? 2 + (int)(rng() % 79) : KTOT;
  h.opt_fused_cadence = rng() % 8 != 0;
  return h;
}

static void check_planning(std::mt19937& rng) {
  for (int it = 0; it < 4000; ++it) {
    HostPlan h = random_plan(rng);
    const PassPlan p = plan_pass(&h);
    CHECK(p.kernel == 0 || p.kernel == 2, "kernel %d", p.kernel);
    CHECK(p.nkt == (h.pending_k + 3) / 4 && p.nkt <= NKT, "nkt %d", p.nkt);
    CHECK(p.rs_workgroups >= 1 && p.rs_workgroups <= h.cu_count, "workgroups %d of %d", p.rs_workgroups, h.cu_count);
    CHECK(p.e_hi >= 3 && p.e_hi <= h.n_max && p.n_hi <= h.n_max && p.e_hi <= std::max(p.n_hi, 3), "sizes %d %d", p.e_hi, p.n_hi);
    if (h.opt_pass_kernel >= 0) CHECK(p.kernel == h.opt_pass_kernel, "forced kernel");
    if (p.beside) CHECK(p.kernel == 2 && p.long_few && p.rs_workgroups == h.cu_count - h.batch, "beside");
    const int rpb = flush_rows_per_block(&h, p.streaming, p.e_hi);
    CHECK(rpb >= 16 && rpb % 16 == 0 && rpb <= 4096 + 16, "rows per block %d", rpb);
    CHECK(flush_workgroups(p.e_hi, rpb) >= 1, "k_flush launches no workgroup");
    // the static-share table exists whenever the plan asks for it, for the size the pass covers
    if (p.kernel == 2 && p.long_few) {
      std::vector<int> table((size_t)h.cu_count * pass_share_pieces() * 4);
      const int longest = build_pass_shares(h.batch, p.e_hi, p.rs_workgroups, table.data());
      CHECK(longest >= 0 && longest <= pass_share_pieces(), "shares %d", longest);
      if (longest > 0) order_pass_shares(p.rs_workgroups, pass_share_pieces(), table.data(), table.size());
    }
    // the packed cadences of a random uploaded stream (plan_cadences): every trajectory's flat sequence of predictions and
    // landmark updates is covered exactly once, in order, within the slot and step limits
    const int steps = 1 + rng() % 120, B = h.batch;
    h.stream_steps = steps;
    h.stream_m.assign((size_t)steps * B, 0);
    h.stream_own.assign((size_t)steps * B, 3);
    h.stream_mhi.assign(steps, 0);
    const int shape = rng() % 4;                       // constant m / runs of a size / anything / mostly nothing
    int run_m = 1 + rng() % 16;
    for (int k = 0; k < steps; ++k) {
      if (rng() % 7 == 0) run_m = (int)(rng() % 17);
      for (int b = 0; b < B; ++b) {
        int m = run_m;
        if (shape == 2) m = (int)(rng() % 17);
        if (shape == 3) m = rng() % 5 == 0 ? (int)(rng() % 4) : 0;
        if (shape == 1 && b > 0) m = (int)(rng() % (run_m + 1));
        h.stream_m[(size_t)k * B + b] = (unsigned char)m;
        h.stream_own[(size_t)k * B + b] = std::min(h.n[b], 3 + 2 * (int)(rng() % ((h.n_max - 3) / 2 + 1)));
        h.stream_mhi[k] = std::max(h.stream_mhi[k], m);
      }
    }
    const int pend = h.pending_k;
    h.pending_k = 0;
    const bool dirty = h.sizes_dirty;
    h.sizes_dirty = false;
    CHECK(cadences_possible(&h) == (h.opt_fused_cadence != 0), "cadences_possible");
    const int k0 = (int)(rng() % steps), end = k0 + 1 + (int)(rng() % (steps - k0));
    RunPlan rp;
    plan_cadences(&h, k0, end, rp);
    const int slot_limit = cadence_slot_limit(&h), step_limit = cadence_step_limit(&h);
    CHECK(slot_limit >= 1 && slot_limit <= CAD_SLOTS && step_limit >= 1 && step_limit <= CAD_SLOTS, "limits %d %d", slot_limit, step_limit);
    CHECK(rp.ncad >= 1 && rp.entries.size() == (size_t)rp.ncad * B && (int)rp.slots_hi.size() == rp.ncad, "plan shape");
    long most = 0;
    for (int b = 0; b < B; ++b) {
      int t = k0, j = 0;                               // the trajectory's cursor
      long total = 0, done = 0;
      for (int c = 0; c < rp.ncad; ++c) {
        const CadPlan& e = rp.entries[(size_t)c * B + b];
        CHECK(e.t0 == t && e.j0 == j, "cadence %d of trajectory %d starts at (%d, %d), cursor (%d, %d)", c, b, e.t0, e.j0, t, j);
        CHECK(e.ns >= 0 && e.ns <= step_limit && e.nslots >= 0 && e.nslots <= slot_limit, "cadence of %d steps, %d slots", e.ns, e.nslots);
        CHECK(e.nslots <= rp.slots_hi[c], "slots_hi");
        CHECK(e.neff >= 3 && e.neff <= h.n[b], "bound %d of %d", e.neff, h.n[b]);
        if (e.ns == 0) {
          CHECK(t == end && e.nslots == 0, "an idle trajectory that is not at the end");
          continue;
        }
        CHECK(t < end, "work behind the end");
        int slots = 0;
        for (int p = 0; p < e.ns; ++p) {
          const int m = h.stream_m[(size_t)(e.t0 + p) * B + b];
          const int lo = p == 0 ? e.j0 : 0, hi = p == e.ns - 1 ? std::min(e.jend, m) : m;
          CHECK(e.t0 + p < end && lo <= hi && hi <= m, "step %d of a cadence: landmarks [%d, %d) of %d", p, lo, hi, m);
          if (p < e.ns - 1) CHECK(hi == m, "a step in the middle of a cadence is cut");
          slots += hi - lo;
        }
        CHECK(slots == e.nslots, "slots %d against %d", slots, e.nslots);
        const int m_last = h.stream_m[(size_t)(e.t0 + e.ns - 1) * B + b];
        const bool cut = e.jend < m_last;
        if (cut) CHECK(e.nslots == slot_limit, "a step is cut although %d of %d slots are free", slot_limit - e.nslots, slot_limit);
        // greedy: the cadence stops because a limit is reached or the range ends
        if (!cut && e.t0 + e.ns < end && e.ns < step_limit)
          CHECK(e.nslots == slot_limit, "cadence stops early: %d of %d slots, %d of %d steps", e.nslots, slot_limit, e.ns, step_limit);
        done += e.ns - (cut ? 1 : 0);
        t = e.t0 + e.ns - (cut ? 1 : 0);
        j = cut ? e.jend : 0;
        total += slots;
      }
      CHECK(t == end && j == 0, "trajectory %d ends at (%d, %d), not at %d", b, t, j, end);
      long want = 0;
      for (int k = k0; k < end; ++k) want += h.stream_m[(size_t)k * B + b];
      CHECK(total == want, "landmark updates %ld of %ld", total, want);
      most = std::max(most, want);
    }
    // as many cadences as the busiest trajectory needs, when only the slots limit them
    if (step_limit == CAD_SLOTS && most > 0) {
      long steps_bound = (end - k0 + CAD_SLOTS - 1) / CAD_SLOTS;
      CHECK(rp.ncad <= std::max((most + slot_limit - 1) / slot_limit, 1L) + steps_bound, "%d cadences for %ld updates", rp.ncad, most);
    }
    h.sizes_dirty = dirty;
    h.pending_k = pend;
  }
}

static void check_step_records(std::mt19937& rng) {
  for (int it = 0; it < 3000; ++it) {
    HostPlan h;
    h.batch = 1;
    const int n_lm = 1 + rng() % 300;
    h.n = {3 + 2 * n_lm};
    const int m = rng() % 40;
    std::vector<int> idx(m);
    std::vector<double> r(m), bq(m);
    bool dup = false, oob = false;
    std::set<int> used;
    for (int i = 0; i < m; ++i) {
      idx[i] = (int)(rng() % (n_lm + (rng() % 9 == 0 ? 3 : 0))) - (rng() % 50 == 0 ? 1 : 0);
      if (idx[i] < 0 || idx[i] >= n_lm) oob = true;
      else if (!used.insert(idx[i]).second) dup = true;
      r[i] = 0.1 * i;
      bq[i] = -0.01 * i;
    }
    std::vector<unsigned char> seen;
    const char* why = validate_obs(&h, 0, idx.data(), m, seen);
    CHECK((why != nullptr) == (dup || oob), "validate_obs: %s with dup=%d oob=%d", why ? why : "accepted", dup, oob);
    if (why) continue;
    int bound = 3 + 2 * (int)(rng() % (n_lm + 1)), prev = bound;
    int taken = 0;
    for (int p = 0; p < std::max(1, (m + MMAX - 1) / MMAX); ++p) {
      StepIn s;
      std::memset(&s, 0xAB, sizeof s);
      fill_step(s, h.n[0], bound, 0.004, 0.02, FLAG_PREDICT | FLAG_UPDATE, idx.data(), r.data(), bq.data(), m, p);
      CHECK(s.m == std::max(0, std::min(m - p * MMAX, MMAX)), "pass %d takes %d of %d", p, s.m, m);
      for (int i = 0; i < MMAX; ++i) {
        if (i < s.m) CHECK(s.idx[i] == idx[p * MMAX + i] && s.range[i] == r[p * MMAX + i] && s.bearing[i] == bq[p * MMAX + i], "record entry %d", i);
        else CHECK(s.idx[i] == 0 && s.range[i] == 0.0 && s.bearing[i] == 0.0, "padding entry %d", i);
      }
      CHECK(bound >= prev && bound <= h.n[0] && s.neff == bound, "bound %d -> %d (n %d)", prev, bound, h.n[0]);
      for (int i = 0; i < s.m; ++i) CHECK(bound >= std::min(h.n[0], 3 + 2 * (s.idx[i] + 1)), "observed landmark beyond the bound");
      prev = bound;
      taken += s.m;
    }
    CHECK(taken == m, "passes took %d of %d", taken, m);
  }
}

int main() {
  std::mt19937 rng(20261004);
  check_layout();
  check_queues();
  check_shares(rng);
  check_planning(rng);
  check_step_records(rng);
  std::printf("host_plan_check: %ld checks passed\n", checks);
  return 0;
}
