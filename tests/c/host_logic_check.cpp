// host_logic_check.cpp -- the pure host-side arithmetic of libnka_hip.so (nka_amd/csrc/host_logic.hpp: the very text the
// library compiles) against brute-force models, built with g++ -fsanitize=address,undefined (make -C nka_amd/csrc hostcheck)
// and run on the CPU by tests/test_sanitizers_cpu.py.  Exit 0 = every check passed and no sanitizer report.
//   host_logic_check            all checks
//   host_logic_check plant      the same, then a deliberately PLANTED heap overflow: the run must die with an AddressSanitizer
//                               report -- the proof that the build really is instrumented (the test asserts that it does)
#include "../../nka_amd/csrc/host_logic.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

using namespace nka_host;

static int failures = 0;
#define CHECK(cond, ...)                                                          \
  do {                                                                            \
    if (!(cond)) {                                                                \
      if (failures++ < 20) { std::fprintf(stderr, "FAILED %s:%d: %s  ", __FILE__, __LINE__, #cond); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } \
    }                                                                             \
  } while (0)

static void check_pass_widths() {
  for (int total = 1; total <= 4000; total++) {
    const int np = balanced_passes(total);
    CHECK(np >= 1 && (long long)np * kMaxPerPass >= total && (long long)(np - 1) * kMaxPerPass < total, "total %d np %d", total, np);
    std::vector<int> w((size_t)np + 2, -12345);          // (guards either side: balanced_widths must write w[0..np) only)
    balanced_widths(total, np, w.data() + 1);
    CHECK(w[0] == -12345 && w[(size_t)np + 1] == -12345, "total %d: wrote outside its array", total);
    int sum = 0, lo = 1 << 30, hi = 0, primes = 0, primes_balanced = 0;
    for (int p = 0; p < np; p++) {
      const int x = w[(size_t)p + 1];
      CHECK(x >= 1 && x <= kMaxPerPass, "total %d pass %d width %d", total, p, x);
      sum += x;
      lo = std::min(lo, x);
      hi = std::max(hi, x);
      primes += heavy_prime(x);
      primes_balanced += heavy_prime(total / np + (p < total % np ? 1 : 0));
    }
    CHECK(sum == total, "total %d: widths add up to %d", total, sum);
    CHECK(hi - lo <= 1 + np, "total %d: widths %d..%d are not balanced", total, lo, hi);      // (each move for a prime shifts one vector)
    if (np <= 2) CHECK(hi - lo <= 3, "total %d: widths %d..%d", total, lo, hi);
    CHECK(primes <= primes_balanced, "total %d: more heavy primes (%d) than the plain balanced split (%d)", total, primes, primes_balanced);
  }
  int w2[2];
  balanced_widths(62, 2, w2);
  CHECK(w2[0] + w2[1] == 62 && !heavy_prime(w2[0]) && !heavy_prime(w2[1]), "62 = %d + %d", w2[0], w2[1]);      // (62 = 32 + 30)
  for (int x = -5; x <= 200; x++) CHECK(round_up4(x) % 4 == 0 && round_up4(x) >= std::max(x, 1) && round_up4(x) < std::max(x, 1) + 4, "round_up4(%d)", x);
  for (int count = 1; count <= 3000; count++) {
    const int g = many_groups(count);
    int sum = 0;
    for (int p = 0; p < g; p++) {
      const int x = many_group_width(count, p);
      CHECK(x >= 1 && x <= kManyMax, "count %d group %d width %d", count, p, x);
      sum += x;
    }
    CHECK(sum == count && (long long)(g - 1) * kManyMax < count, "count %d groups %d sum %d", count, g, sum);
  }
}

// The list word against a model of device and host: updates (with dependence drops the host cannot see), relax, restart, and
// a word that reaches host memory whenever the device gets that far (any published word not older than the last one seen).
// SAFETY: the bound is never below the true list length at the entry of an update (the passes are launched at its width);
// EXACTNESS: it equals the true length whenever the newest word has arrived (a caller that synchronises once per iteration).
static void check_list_word() {
  std::mt19937_64 rng(12345);
  for (int trial = 0; trial < 4000; trial++) {
    const int mvec = 1 + (int)(rng() % 40);
    int64_t seq = 0, valid_after = 0;
    std::vector<int64_t> relaxed_after;
    int list_ub = 0;                          // the host's own count (nka_hip_state::list_ub)
    int L = 0;                                // the device's list length, pending pair included
    bool pending = false;
    std::vector<std::pair<int64_t, int>> published;      // (update number, list length at its exit), in order
    size_t seen = 0;                          // index + 1 of the newest word that has reached host memory (0 = none)
    for (int step = 0; step < 300; step++) {
      const unsigned r = (unsigned)(rng() % 100);
      if (r < 70) {                           // accel_update
        if (seen < published.size() && rng() % 2) seen += 1 + (size_t)(rng() % (published.size() - seen));     // the device got further
        const unsigned long long word = seen ? (((unsigned long long)published[seen - 1].first << kListWordLenBits) | (unsigned long long)published[seen - 1].second) : 0ull;
        const int ub = list_bound_from_word(list_ub, word, seq, valid_after, relaxed_after);
        CHECK(ub >= L && ub <= list_ub, "trial %d step %d: bound %d, true length %d, host count %d", trial, step, ub, L, list_ub);
        if (seen && published[seen - 1].first == seq && published[seen - 1].first > valid_after)
          CHECK(ub == L, "trial %d step %d: the newest word has arrived, bound %d != true length %d", trial, step, ub, L);
        const int d = L > 0 && rng() % 3 == 0 ? (int)(rng() % (unsigned)(L + 1)) : 0;      // dependence drops / s == 0 (device only)
        const int ncomb = std::min(L - d, mvec);
        L = ncomb + 1;
        const int comb_ub = pending ? std::min(ub, mvec) : ub;           // update_impl
        list_ub = comb_ub + 1;
        CHECK(list_ub >= L, "trial %d step %d: host count %d below the true length %d after an update", trial, step, list_ub, L);
        seq++;
        pending = true;
        published.push_back({seq, L});
      } else if (r < 85) {                    // relax (nka_hip_relax)
        if (pending) {
          pending = false;
          list_ub = std::max(list_ub - 1, 0);
          L--;
          relaxed_after.push_back(seq);
        }
      } else if (r < 93) {                    // restart (nka_hip_restart): words of updates enqueued so far are stale
        pending = false;
        list_ub = 0;
        L = 0;
        valid_after = seq;
        relaxed_after.clear();
      } else if (!published.empty()) {        // the caller synchronises: the newest word is there
        seen = published.size();
      }
      CHECK(relaxed_after.size() <= 300, "relaxed_after grows without bound");
    }
    // stale words and words from the future are ignored
    std::vector<int64_t> none;
    CHECK(list_bound_from_word(7, ((unsigned long long)(seq + 5) << kListWordLenBits) | 3ull, seq, valid_after, none) == 7, "a word from the future was used");
    CHECK(list_bound_from_word(7, ((unsigned long long)valid_after << kListWordLenBits) | 3ull, seq, valid_after, none) == 7, "a stale word was used");
  }
  // update numbers up to 2^43: no shift overflow
  std::vector<int64_t> none;
  const int64_t big = (1ll << 43) - 1;
  CHECK(list_bound_from_word(21, ((unsigned long long)big << kListWordLenBits) | 20ull, big, 0, none) == 20, "large update number");
}

// BufferBook::held against a brute-force interval model on a synthetic address space.
static void check_buffer_book() {
  std::mt19937_64 rng(777);
  std::vector<double> arena(1 << 16);         // real addresses, never dereferenced by held()
  for (int trial = 0; trial < 3000; trial++) {
    const int64_t n = 1 + (int64_t)(rng() % 97);
    const int64_t block = n * (int64_t)(2 + rng() % 5);
    const double *w = arena.data() + 1000, *v = w + block + (int64_t)(rng() % 50);
    BufferBook book;
    std::vector<const double *> all;
    for (int k = 0; k < 12; k++) {
      const double *q = arena.data() + 20000 + (int64_t)(rng() % 30000);
      book.taken.insert(q);
      all.push_back(q);
    }
    for (int k = 0; k < 4; k++) {
      const double *q = (rng() & 1) ? all[(size_t)(rng() % all.size())] : w + (int64_t)(rng() % (uint64_t)block);
      book.lent.insert(q);
    }
    for (int probe = 0; probe < 200; probe++) {
      const double *p = (rng() % 3 == 0) ? all[(size_t)(rng() % all.size())] + (int64_t)(rng() % (uint64_t)(2 * n)) - n
                                         : arena.data() + (int64_t)(rng() % 60000);
      bool want = false;
      if (!book.lent.count(p)) {
        auto hit = [&](const double *q, int64_t len) { return p < q + len && q < p + n; };
        want = hit(w, block) || hit(v, block);
        for (const double *q : book.taken) want = want || hit(q, n);
      }
      CHECK(book.held(p, n, w, v, block) == want, "trial %d probe at %td: held() says %d", trial, p - arena.data(), (int)!want);
    }
    // an empty slice (vlen 0) hands over a null buffer every time: nothing is held at the null address
    BufferBook none;
    CHECK(!none.held(nullptr, 0, w, v, block), "null buffer of an empty slice");
  }
}

int main(int argc, char **argv) {
  check_pass_widths();
  check_list_word();
  check_buffer_book();
  if (failures) {
    std::fprintf(stderr, "host_logic_check: %d check(s) FAILED\n", failures);
    return 1;
  }
  std::printf("host_logic_check: pass widths, launch groups, list word, buffer book: OK\n");
  if (argc > 1 && !std::strcmp(argv[1], "plant")) {
    // PLANTED: balanced_widths asked for one pass more than its array holds
    int *w = static_cast<int *>(std::malloc(sizeof(int) * 2));
    balanced_widths(70, 3, w);
    std::printf("planted overflow was NOT caught: %d\n", w[0]);
    std::free(w);
    return 0;
  }
  return 0;
}
