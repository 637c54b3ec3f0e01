// host_logic.hpp -- the PURE host-side arithmetic of libnka_hip.so: pass widths, the decoding of the list word, who holds
// which buffer, launch groups of the abstract-vector hooks.  No HIP, no device state: nka_hip.hip / vec_ops.hip include it,
// and tests/c/host_logic_check.cpp compiles the same text with g++ -fsanitize=address,undefined and checks every function
// against a brute-force model on the CPU (`make -C nka_amd/csrc hostcheck`, tests/test_sanitizers_cpu.py) -- the part of the
// host code a sanitizer can see without a GPU (the reference's stand-in is its `defined` invariant check,
// /root/reference/src-F08/nka_type.F90:460-524).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <set>
#include <vector>

namespace nka_host {

constexpr int kMaxPerPass = 32;        // largest list width one launch of the streaming kernels takes (= nka_kernels.hpp)
constexpr int kListWordLenBits = 20;   // list word = update number << 20 | list length at its exit (= nka_kernels.hpp)
constexpr int kManyMax = 24;           // vectors per launch of the batched vector hooks (= vec_ops.hip)

inline int round_up4(int x) { return ((std::max(x, 1) + 3) / 4) * 4; }

// A list of `total` > kMaxPerPass entries in the fewest passes of BALANCED widths (33 -> 17 + 16, 70 -> 24 + 23 + 23): every
// pass then runs the rolling-window kernel at (nearly) its exact width instead of one full pass of 32 and one that is mostly
// padding (measured in round 5, profiles/r05/multipass.txt: mvec = 33 at 0.58 of the roofline against 0.76 at mvec = 32).
inline int balanced_passes(int total) { return (total + kMaxPerPass - 1) / kMaxPerPass; }
// The widths whose only ring is the whole width (win_ring / win_ring_pairs: primes) AND large: 23, 29, 31.
inline bool heavy_prime(int w) { return w == 23 || w == 29 || w == 31; }
// (22 and 26 -- twice a prime: no ring in win_ring's list either -- were tried the same way in round 5: padding to 24 / 28 makes PA
//  3-4 % SLOWER and the update +1...2 % (compact), -1 % (src-F08): not kept, profiles/r05/multipass.txt)
// widths[0..np): balanced, then one vector moved between two passes wherever that removes a heavy prime without making another
inline void balanced_widths(int total, int np, int *w) {
  for (int p = 0; p < np; p++) w[p] = total / np + (p < total % np ? 1 : 0);
  for (int i = 0; i < np; i++) {
    if (!heavy_prime(w[i])) continue;
    for (int j = 0; j < np; j++) {
      if (j == i) continue;
      if (w[i] + 1 <= kMaxPerPass && w[j] > 1 && !heavy_prime(w[j] - 1)) { w[i]++; w[j]--; break; }          // (w[i] + 1 is even)
      if (w[j] + 1 <= kMaxPerPass && !heavy_prime(w[j] + 1) && !heavy_prime(w[i] - 1)) { w[i]--; w[j]++; break; }
    }
  }
}

// Launch groups of the batched hooks of the abstract-vector flavour: balanced too (25 = 13 + 12, not 24 + 1).
inline int many_groups(int count) { return count <= kManyMax ? 1 : (count + kManyMax - 1) / kManyMax; }
inline int many_group_width(int count, int p) { const int np = many_groups(count); return count / np + (p < count % np ? 1 : 0); }

// THE LIST WORD (include/nka_hip_ext.h: nka_hip_list_bound).  PB of update number u publishes (u, list length at its exit).
// The host's upper bound on the list length at the entry of the next update: its own count `ub`, tightened by the word if the
// word is fresh enough to be trusted -- newer than the last restart (`valid_after`) and not from the future (`seq` = updates
// enqueued so far) -- plus one per update enqueued since, minus one per relax() that dropped a pending pair since
// (`relaxed_after`: the update number before each such relax; entries older than the word are dropped: they are part of
// every later word).  Never above `ub`, never negative.
inline int list_bound_from_word(int ub, unsigned long long word, int64_t seq, int64_t valid_after, std::vector<int64_t> &relaxed_after) {
  const int64_t u = (int64_t)(word >> kListWordLenBits);
  if (u <= valid_after || u > seq) return ub;
  int64_t len = (int64_t)(word & ((1ull << kListWordLenBits) - 1)) + (seq - u);
  size_t keep = 0;
  for (int64_t r : relaxed_after)
    if (r >= u) {
      len--;
      relaxed_after[keep++] = r;
    }
  relaxed_after.resize(keep);
  return (int)std::min<int64_t>(ub, std::max<int64_t>(len, 0));
}

// WHO HOLDS WHICH BUFFER, as far as the host can know it (the out-of-place entry, nka_hip_accel_update_swap).  `taken` = every
// buffer outside the two slot-major allocations that a slot table entry may name; `lent` = the free buffers handed to the
// caller for its next input (they may lie anywhere, the slot-major allocations included).  Every such buffer holds n doubles.
struct BufferBook {
  std::set<const double *> taken, lent;
  // Does [p, p + n) touch memory the library holds -- one of the two slot-major blocks [w, w + block), [v, v + block) or a taken
  // buffer -- that it has NOT lent to the caller?
  bool held(const double *p, int64_t n, const double *w, const double *v, int64_t block) const {
    if (lent.count(p)) return false;
    n = std::max<int64_t>(n, 1);
    // (address arithmetic on integers: p - n may lie before any object)
    const uintptr_t P = reinterpret_cast<uintptr_t>(p), N = (uintptr_t)n * sizeof(double), B = (uintptr_t)block * sizeof(double);
    auto overlaps = [&](const double *q, uintptr_t len) {
      const uintptr_t Q = reinterpret_cast<uintptr_t>(q);
      return P < Q + len && Q < P + N;
    };
    if (overlaps(w, B) || overlaps(v, B)) return true;
    // every taken buffer holds n doubles: one of them overlaps [p, p + n) iff its base lies in (p - n, p + n)
    const uintptr_t lo = P >= N ? P - N : 0;
    auto it = taken.upper_bound(reinterpret_cast<const double *>(lo));
    if (P < N && !taken.empty() && reinterpret_cast<uintptr_t>(*taken.begin()) == 0) it = taken.begin();
    return it != taken.end() && reinterpret_cast<uintptr_t>(*it) < P + N;
  }
};

}  // namespace nka_host
