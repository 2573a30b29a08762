// synth.cpp -- deterministic synthetic workloads for tests and bench.py
// (libblurrily_synth.so; bench/test infrastructure, not part of the product ABI).
//
// Restates SURVEY.md section 8(d): /usr/share/dict/words and the Geonames
// dumps the reference's bin/bench downloads (bin/bench:44-71) are not
// available offline, so haystacks are generated from seeded pseudo-words:
//   words      N distinct lowercase pseudo-words, length 2..22 (mean ~9.6),
//              letters from an order-1 Markov chain with English-like
//              frequencies                        -> BASELINE.json configs[0..1]
//   geonames   strings of 1..4 words drawn Zipf(s=1) from a pseudo-word
//              vocabulary, ~14 distinct trigrams per string
//                                                 -> configs[2..3]
//   skewed     hot-trigram haystack: most strings share a few prefix/suffix
//              words and a handful of lengths     -> configs[4]
//   queries    haystack samples with 0..2 random edits (insert / delete /
//              substitute / transpose)
// One splitmix64 stream per call; integer arithmetic except the Zipf table
// (IEEE doubles, same toolchain and image on both boxes), so this container and
// the GPU box produce identical bytes.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_set>
#include <vector>

namespace {

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull) { next(); next(); }
  uint64_t next() {                       // splitmix64
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  uint32_t below(uint32_t n) { return uint32_t((uint64_t(uint32_t(next() >> 32)) * n) >> 32); }
  double unit() { return double(next() >> 11) * (1.0 / 9007199254740992.0); }
};

// English letter frequencies (per 10 000), a..z
const uint32_t kFreq[26] = {817, 149, 278, 425, 1270, 223, 202, 609, 697, 15, 77, 403, 241,
                            675, 751, 193, 10,  599,  633, 906, 276, 98,  236, 15, 197, 7};
inline bool is_vowel(int c) { return c == 0 || c == 4 || c == 8 || c == 14 || c == 20; }

struct Markov {
  uint32_t cum[27][26];                   // row 26 = word start
  Markov() {
    for (int p = 0; p <= 26; ++p) {
      uint32_t acc = 0;
      for (int c = 0; c < 26; ++c) {
        uint32_t w = kFreq[c] * 4;
        if (p < 26) {
          if (is_vowel(p)) w = is_vowel(c) ? w / 2 : w * 3 / 2;      // vowel -> consonant
          else             w = is_vowel(c) ? w * 2 : w * 3 / 4;      // consonant -> vowel
          if (p == c) w /= 3;                                        // few doubled letters
        }
        acc += w ? w : 1;
        cum[p][c] = acc;
      }
    }
  }
  char draw(Rng& r, int prev) const {
    const uint32_t* row = cum[prev];
    const uint32_t x = r.below(row[25]);
    int c = 0;
    while (row[c] <= x) ++c;
    return char('a' + c);
  }
};
const Markov kMarkov;

uint32_t binom(Rng& r, uint32_t n, uint32_t p_per_1024) {
  uint32_t k = 0;
  for (uint32_t i = 0; i < n; ++i) k += (r.below(1024) < p_per_1024) ? 1u : 0u;
  return k;
}

std::string make_word(Rng& r, uint32_t min_len, uint32_t spread, uint32_t p_per_1024) {
  const uint32_t len = min_len + binom(r, spread, p_per_1024);
  std::string w;
  int prev = 26;
  for (uint32_t i = 0; i < len; ++i) { const char c = kMarkov.draw(r, prev); w.push_back(c); prev = c - 'a'; }
  return w;
}

std::vector<std::string> distinct_words(Rng& r, uint32_t n, uint32_t min_len, uint32_t spread, uint32_t p) {
  std::vector<std::string> out;
  out.reserve(n);
  std::unordered_set<std::string> seen;
  seen.reserve(size_t(n) * 2);
  while (out.size() < n) {
    std::string w = make_word(r, min_len, spread, p);
    if (seen.insert(w).second) out.push_back(std::move(w));
  }
  return out;
}

size_t emit(const std::vector<std::string>& v, char* out, uint64_t* offsets) {
  uint64_t pos = 0;
  for (size_t i = 0; i < v.size(); ++i) {
    offsets[i] = pos;
    std::memcpy(out + pos, v[i].data(), v[i].size());
    pos += v[i].size();
  }
  offsets[v.size()] = pos;
  return size_t(pos);
}

}  // namespace

extern "C" {

// Upper bound of bytes any generator below writes for n strings.
uint64_t synth_max_bytes(uint32_t n) { return uint64_t(n) * 96 + 64; }

// configs[0..1]: n distinct pseudo-words, length 2..22, mean ~9.6.
uint64_t synth_words(uint64_t seed, uint32_t n, char* out, uint64_t* offsets) {
  Rng r(seed);
  return emit(distinct_words(r, n, 2, 20, 389), out, offsets);     // 2 + Binomial(20, 0.38)
}

// configs[2..3]: Geonames-like multi-word strings.
uint64_t synth_geonames(uint64_t seed, uint32_t n, uint32_t vocab, char* out, uint64_t* offsets) {
  Rng r(seed);
  const std::vector<std::string> words = distinct_words(r, vocab, 2, 12, 330);   // mean ~5.9 letters
  // Zipf(s=1) over vocabulary ranks via inverse CDF on a cumulative table
  std::vector<double> cdf(vocab);
  double acc = 0;
  for (uint32_t i = 0; i < vocab; ++i) { acc += 1.0 / double(i + 1); cdf[i] = acc; }
  uint64_t pos = 0;
  for (uint32_t i = 0; i < n; ++i) {
    offsets[i] = pos;
    const uint32_t x = r.below(100);
    const uint32_t nw = x < 31 ? 1 : x < 72 ? 2 : x < 93 ? 3 : 4;   // mean 2.04 words
    for (uint32_t k = 0; k < nw; ++k) {
      const double u = r.unit() * acc;
      const uint32_t wi = uint32_t(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
      const std::string& w = words[std::min(wi, vocab - 1)];
      if (k) out[pos++] = ' ';
      std::memcpy(out + pos, w.data(), w.size());
      pos += w.size();
    }
  }
  offsets[n] = pos;
  return pos;
}

// configs[4]: heavy bucket skew and massive (matches, weight) ties.  `hot_pct` (synth_skewed: 100) scales how
// often a string takes one of the hot prefixes / suffixes -- 0 leaves the bare stems -- which moves the haystack's
// mean posting-list size per window between a plain and a hot-trigram one (tools/gate_probe.py).
uint64_t synth_skewed_mix(uint64_t seed, uint32_t n, uint32_t hot_pct, char* out, uint64_t* offsets);
uint64_t synth_skewed(uint64_t seed, uint32_t n, char* out, uint64_t* offsets) {
  return synth_skewed_mix(seed, n, 100, out, offsets);
}
uint64_t synth_skewed_mix(uint64_t seed, uint32_t n, uint32_t hot_pct, char* out, uint64_t* offsets) {
  Rng r(seed);
  static const char* kPre[] = {"san ", "new ", "saint ", "el ", "la "};
  static const char* kSuf[] = {"ville", " city", "ton", "burg"};
  const std::vector<std::string> stems = distinct_words(r, 50000, 5, 2, 512);    // length 5..7
  uint64_t pos = 0;
  for (uint32_t i = 0; i < n; ++i) {
    offsets[i] = pos;
    const uint32_t x = r.below(100);
    if (x < 70) {                                                  // (the draws are hot_pct-independent: one stream of strings)
      const uint32_t pick_pre = r.below(x < 45 ? 2 : 5);
      if (x * 100 < 70 * hot_pct) { const char* p = kPre[pick_pre]; const size_t l = std::strlen(p); std::memcpy(out + pos, p, l); pos += l; }
    }
    const std::string& w = stems[r.below(uint32_t(stems.size()))];
    std::memcpy(out + pos, w.data(), w.size());
    pos += w.size();
    const uint32_t y = r.below(100);
    if (y < 60) {
      const uint32_t pick_suf = r.below(r.below(100) < 70 ? 1 : 4);
      if (y * 100 < 60 * hot_pct) { const char* s = kSuf[pick_suf]; const size_t l = std::strlen(s); std::memcpy(out + pos, s, l); pos += l; }
    }
  }
  offsets[n] = pos;
  return pos;
}

// Queries: uniform samples of the haystack with 0..2 random edits.
uint64_t synth_queries(uint64_t seed, const char* hay, const uint64_t* hay_off, uint32_t n_hay, uint32_t n_q,
                       char* out, uint64_t* offsets) {
  Rng r(seed);
  uint64_t pos = 0;
  std::string s;
  for (uint32_t i = 0; i < n_q; ++i) {
    offsets[i] = pos;
    const uint32_t h = r.below(n_hay);
    s.assign(hay + hay_off[h], size_t(hay_off[h + 1] - hay_off[h]));
    const uint32_t edits = r.below(3);
    for (uint32_t e = 0; e < edits; ++e) {
      const uint32_t kind = r.below(4);
      const char c = char('a' + r.below(26));
      if (kind == 0 || s.empty()) {                       // insert
        s.insert(s.begin() + r.below(uint32_t(s.size()) + 1), c);
      } else if (kind == 1) {                             // delete
        if (s.size() > 1) s.erase(s.begin() + r.below(uint32_t(s.size())));
      } else if (kind == 2) {                             // substitute
        s[r.below(uint32_t(s.size()))] = c;
      } else if (s.size() > 1) {                          // transpose
        const uint32_t k = r.below(uint32_t(s.size()) - 1);
        std::swap(s[k], s[k + 1]);
      }
    }
    std::memcpy(out + pos, s.data(), s.size());
    pos += s.size();
  }
  offsets[n_q] = pos;
  return pos;
}

// Sum over needles of the number of distinct trigrams (T of tokeniser.c:119),
// for the 8 B x T term of the algorithmic byte count.
uint64_t synth_count_trigrams(const char* packed, const uint64_t* offsets, uint32_t n) {
  uint64_t total = 0;
  std::vector<uint16_t> codes;
  for (uint32_t i = 0; i < n; ++i) {
    const char* s = packed + offsets[i];
    const size_t len = size_t(offsets[i + 1] - offsets[i]);
    codes.clear();
    uint32_t a = 0, b = 0;
    for (size_t k = 0; k <= len; ++k) {
      const unsigned char ch = k < len ? (unsigned char)s[k] : 0;
      const uint32_t c = (ch >= 'a' && ch <= 'z') ? uint32_t(ch - 'a' + 1) : 0u;
      codes.push_back(uint16_t(a + 28u * b + 784u * c));
      a = b; b = c;
    }
    std::sort(codes.begin(), codes.end());
    total += uint64_t(std::unique(codes.begin(), codes.end()) - codes.begin());
  }
  return total;
}

}  // extern "C"
