/*
 * normalize_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of Blurrily::Map#normalize_string
 * (/root/reference/lib/blurrily/map.rb:40-47), written from the Ruby text --
 * not from blurrily_amd/map.py and not from normalise_kernel -- as the
 * independent checker of both.  It follows the Ruby method chain step by step,
 * each step a separate pass over the string:
 *
 *   map.rb:41  result = needle.downcase
 *   map.rb:42  unless result =~ /^([a-z ])+$/
 *   map.rb:43    result = ...mb_chars.normalize(:kd).gsub(/[^\x00-\x7F]/,'').to_s.gsub(/[^a-z]/,' ')
 *   map.rb:46  result.gsub(/\s+/,' ').strip
 *
 * Scope: ASCII input only.  For bytes >= 0x80 the Ruby goes through
 * ActiveSupport 4.2's NFKD tables (Gemfile.lock:11), a third-party dependency
 * that is not under /root/reference: this file returns -1 for such input and
 * parity there stays UNPINNED beyond the one vector the reference's spec holds
 * ('@€%é' -> 'e', spec/blurrily/map_spec.rb:55-59).
 *
 * Ruby semantics restated (MRI 1.9.3 - 2.2, the versions of .travis.yml:1-5):
 *   String#downcase   ASCII-only: 'A'..'Z' -> 'a'..'z'.
 *   /^...$/           ^ and $ are LINE anchors in Ruby: ^ matches at the start
 *                     of the string and after every "\n"; $ matches at the end
 *                     of the string and before every "\n".  So the test is
 *                     "some line is one or more of [a-z ]".
 *   /\s/              [ \t\r\n\f\v]
 *   String#strip      leading whitespace [ \t\n\v\f\r]; trailing whitespace
 *                     AND trailing NULs (rb_str_rstrip: "remove trailing
 *                     spaces or '\0's").
 * For ASCII input the NFKD step and the "drop non-ASCII" step are the identity.
 */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

static int nrm_is_s(unsigned char c) {             /* Ruby's \s */
  return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\f' || c == '\v';
}

/* in[0..len) -> out (capacity >= len + 1, NUL-terminated for convenience).
 * Returns the length of the result, or -1 if the input holds a byte >= 0x80. */
long oracle_normalize_ascii(const char* in, size_t len, char* out) {
  unsigned char* a = (unsigned char*)malloc(len + 1);
  unsigned char* b = (unsigned char*)malloc(len + 1);
  size_t n = len, i, m;
  for (i = 0; i < len; ++i)
    if ((unsigned char)in[i] >= 0x80) { free(a); free(b); return -1; }

  /* map.rb:41 downcase */
  for (i = 0; i < n; ++i) {
    unsigned char c = (unsigned char)in[i];
    a[i] = (unsigned char)((c >= 'A' && c <= 'Z') ? c - 'A' + 'a' : c);
  }

  /* map.rb:42 =~ /^([a-z ])+$/ : split at "\n" and test every line */
  int matched = 0;
  size_t line = 0;
  while (line <= n && !matched) {
    size_t end = line;
    while (end < n && a[end] != '\n') ++end;       /* the line is a[line..end) */
    if (end > line) {
      int ok = 1;
      for (i = line; i < end; ++i)
        if (!((a[i] >= 'a' && a[i] <= 'z') || a[i] == ' ')) { ok = 0; break; }
      matched = ok;
    }
    line = end + 1;
  }

  /* map.rb:43 (ASCII): gsub(/[^a-z]/, ' ') */
  if (!matched)
    for (i = 0; i < n; ++i)
      if (!(a[i] >= 'a' && a[i] <= 'z')) a[i] = ' ';

  /* map.rb:46 gsub(/\s+/, ' ') */
  m = 0;
  for (i = 0; i < n;) {
    if (nrm_is_s(a[i])) {
      while (i < n && nrm_is_s(a[i])) ++i;
      b[m++] = ' ';
    } else {
      b[m++] = a[i++];
    }
  }

  /* map.rb:46 .strip */
  size_t lo = 0, hi = m;
  while (lo < hi && nrm_is_s(b[lo])) ++lo;
  while (hi > lo && (b[hi - 1] == 0 || nrm_is_s(b[hi - 1]))) --hi;
  memcpy(out, b + lo, hi - lo);
  out[hi - lo] = 0;
  free(a); free(b);
  return (long)(hi - lo);
}
