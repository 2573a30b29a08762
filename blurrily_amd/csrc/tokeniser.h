// tokeniser.h -- string -> sorted distinct base-28 trigram codes (host side).
//
// Restates the behaviour of the reference's blurrily_tokeniser_parse_string
// (ext/blurrily/tokeniser.c:59-119, string_to_code :21-31): the string is
// framed as "**" + s + "*", every byte outside 'a'..'z' is the epsilon symbol 0,
// letters are 1..26, and the trigram starting at frame position k has the code
//   sym(p[k]) + 28*sym(p[k+1]) + 784*sym(p[k+2])      (first char = low digit).
// A string of length L yields L+1 codes, returned ascending and de-duplicated.
// The same function runs on the device for batched queries (find_kernels.hip).
#pragma once
#include <cstddef>
#include <cstdint>

namespace blurrily {

constexpr int      kBase     = 28;               // tokeniser.h:22 TRIGRAM_BASE
constexpr uint32_t kNumCodes = 28u * 28u * 28u;  // storage.c:30   TRIGRAM_COUNT (21952)

inline uint32_t symbol_of(unsigned char c) {
  return (c >= 'a' && c <= 'z') ? uint32_t(c - 'a' + 1) : 0u;
}

// Tokenise s[0..len).  `out` must hold len+1 codes.  Returns the number of
// distinct codes (>= 1).
inline int tokenise(const char* s, size_t len, uint16_t* out) {
  // rolling window over the framed string: a = sym(p[k]), b = sym(p[k+1])
  uint32_t a = 0, b = 0;
  const size_t n = len + 1;
  for (size_t k = 0; k < n; ++k) {
    const uint32_t c = (k < len) ? symbol_of((unsigned char)s[k]) : 0u;  // p[k+2]
    out[k] = uint16_t(a + kBase * b + kBase * kBase * c);
    a = b; b = c;
  }
  // insertion sort: n is a handful to a few dozen for real needles
  if (n <= 64) {
    for (size_t i = 1; i < n; ++i) {
      const uint16_t v = out[i];
      size_t j = i;
      while (j > 0 && out[j - 1] > v) { out[j] = out[j - 1]; --j; }
      out[j] = v;
    }
  } else {
    // long needles: counting sort over the code space is overkill; heap sort in place
    auto sift = [&](size_t start, size_t end) {
      size_t root = start;
      while (2 * root + 1 < end) {
        size_t child = 2 * root + 1;
        if (child + 1 < end && out[child] < out[child + 1]) ++child;
        if (out[root] >= out[child]) return;
        const uint16_t t = out[root]; out[root] = out[child]; out[child] = t;
        root = child;
      }
    };
    for (size_t i = n / 2; i-- > 0;) sift(i, n);
    for (size_t end = n; end-- > 1;) {
      const uint16_t t = out[0]; out[0] = out[end]; out[end] = t;
      sift(0, end);
    }
  }
  size_t m = 0;
  for (size_t k = 0; k < n; ++k)
    if (m == 0 || out[m - 1] != out[k]) out[m++] = out[k];
  return int(m);
}

}  // namespace blurrily
