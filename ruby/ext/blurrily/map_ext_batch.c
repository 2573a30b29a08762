/*
 * map_ext_batch.c -- batched methods for Blurrily::RawMap over libblurrily_hip.so.
 *
 * Compiled BESIDE the blurrily gem's own glue, ext/blurrily/map_ext.c (mezis/blurrily v1.0.2), by
 * ruby/ext/blurrily/extconf.rb of this repository.  The gem's glue is taken as it is: it defines
 * Blurrily::RawMap with new/load/put/delete/save/find/stats/close (map_ext.c:206-229) over the nine
 * blurrily_storage_* functions of storage.h:36-117, which libblurrily_hip.so exports with the same
 * signatures (include/blurrily_storage.h, part 1).  This file adds what the reference has no
 * counterpart for (include/blurrily_storage.h, part 2):
 *
 *   RawMap#find_batch(needles, limit)       -> [[[ref, matches, weight], ...], ...]   one list per needle
 *   RawMap#find_batch_raw(needles, limit)   -> [lists, non_ascii_flags]   needles BEFORE normalize_string
 *   RawMap#put_many(needles, refs, weights) -> trigrams added
 *   RawMap#sync_device                      -> nil        build the device image now
 *   RawMap#set_option(key, value) / #get_option(key)      tunables, "devices" among them (multi-GPU)
 *
 * Each element of a batch is defined as exactly one RawMap#find (map_ext.c:131-162): limit <= 0 means
 * LIMIT_DEFAULT, rows are [reference, matches, weight].  The batched calls run WITHOUT the GVL
 * (rb_thread_call_without_gvl): the reference holds it for every call (SURVEY.md section 8(b),
 * "Threading"), which is right for a 100 us find and wrong for a 500 ms batch.
 *
 * extconf.rb compiles every source with -DInit_map_ext=Init_map_ext_reference, so the gem's
 * initialiser gets that name; the Init_map_ext Ruby calls on `require 'blurrily/map_ext'` is the one
 * at the end of this file, which runs the gem's first.
 *
 * No Ruby toolchain exists in the image this repository is built in: this file is written against the
 * documented C API of Ruby >= 2.0 and is not compiled by tests/ (tests/test_header_compat.py checks the
 * header pairing it relies on).
 */
#include <ruby.h>
#include <ruby/thread.h>
#include <errno.h>
#include <stdint.h>
#include <string.h>

#include "storage.h"            /* the gem's own header: trigram_map, trigram_match_t, the nine functions */
#include "blurrily_storage.h"   /* this repository's: re-declares those nine (compatibly) and adds part 2 */

#undef Init_map_ext
void Init_map_ext_reference(void);          /* the gem's Init_map_ext under the name extconf.rb gave it */
void Init_map_ext(void);

static VALUE mBlurrily = Qnil, cRawMap = Qnil;

static trigram_map map_of(VALUE self)
{
  trigram_map haystack = NULL;
  if (rb_ivar_get(self, rb_intern("@closed")) == Qtrue)          /* map_ext.c:11-16 */
    rb_raise(rb_const_get(cRawMap, rb_intern("ClosedError")), "Map was freed");
  Data_Get_Struct(self, struct trigram_map_t, haystack);
  return haystack;
}

static uint16_t limit_of(VALUE rb_limit)
{
  int limit = NUM2UINT(rb_limit);                                  /* map_ext.c:135,142-146 */
  if (limit <= 0) limit = NUM2UINT(rb_const_get(mBlurrily, rb_intern("LIMIT_DEFAULT")));
  return (uint16_t)limit;                                          /* storage.h:110: uint16_t */
}

/* needles -> one packed buffer + n+1 offsets (what the library's batched entry points take) */
typedef struct {
  char*     packed;
  uint64_t* offsets;
  size_t    n;
} packed_needles;

static void pack_needles(VALUE rb_needles, packed_needles* out)
{
  long   n = RARRAY_LEN(rb_needles), i;
  size_t bytes = 0;
  Check_Type(rb_needles, T_ARRAY);
  for (i = 0; i < n; ++i) {
    VALUE s = rb_ary_entry(rb_needles, i);
    StringValue(s);
    bytes += (size_t)RSTRING_LEN(s);
  }
  out->n = (size_t)n;
  out->offsets = ALLOC_N(uint64_t, n + 1);
  out->packed  = ALLOC_N(char, bytes + 1);
  out->offsets[0] = 0;
  for (i = 0; i < n; ++i) {
    VALUE s = rb_ary_entry(rb_needles, i);
    memcpy(out->packed + out->offsets[i], RSTRING_PTR(s), (size_t)RSTRING_LEN(s));
    out->offsets[i + 1] = out->offsets[i] + (uint64_t)RSTRING_LEN(s);
  }
}

typedef struct {
  trigram_map     map;
  packed_needles  in;
  uint16_t        limit;
  trigram_match   rows;
  uint32_t*       counts;
  uint32_t*       non_ascii;      /* find_batch_raw only */
  int             res, err;
} batch_call;

static void* batch_call_run(void* p)
{
  batch_call* c = (batch_call*)p;
  c->res = c->non_ascii
    ? blurrily_storage_find_batch_raw(c->map, c->in.packed, c->in.offsets, c->in.n, c->limit, c->rows, c->counts, c->non_ascii)
    : blurrily_storage_find_batch(c->map, c->in.packed, c->in.offsets, c->in.n, c->limit, c->rows, c->counts);
  c->err = errno;
  return NULL;
}

static VALUE rows_to_ruby(const batch_call* c)
{
  VALUE  out = rb_ary_new2((long)c->in.n);
  size_t i;
  for (i = 0; i < c->in.n; ++i) {
    VALUE    per = rb_ary_new2(c->counts[i]);
    uint32_t k;
    for (k = 0; k < c->counts[i]; ++k) {                           /* map_ext.c:153-160 */
      const trigram_match_t* m = c->rows + i * c->limit + k;
      rb_ary_push(per, rb_ary_new3(3, rb_uint_new(m->reference), rb_uint_new(m->matches), rb_uint_new(m->weight)));
    }
    rb_ary_push(out, per);
  }
  return out;
}

static VALUE find_batch_common(VALUE self, VALUE rb_needles, VALUE rb_limit, int raw)
{
  batch_call c;
  VALUE      out = Qnil, flags = Qnil;
  memset(&c, 0, sizeof c);
  c.map   = map_of(self);
  c.limit = limit_of(rb_limit);
  pack_needles(rb_needles, &c.in);
  c.rows   = ALLOC_N(trigram_match_t, c.in.n * c.limit + 1);
  c.counts = ALLOC_N(uint32_t, c.in.n + 1);
  if (raw) c.non_ascii = ALLOC_N(uint32_t, c.in.n + 1);
  rb_thread_call_without_gvl(batch_call_run, &c, RUBY_UBF_IO, NULL);
  if (c.res >= 0) {
    out = rows_to_ruby(&c);
    if (raw) {
      size_t i;
      flags = rb_ary_new2((long)c.in.n);
      for (i = 0; i < c.in.n; ++i) rb_ary_push(flags, c.non_ascii[i] ? Qtrue : Qfalse);
      out = rb_ary_new3(2, out, flags);
    }
  }
  xfree(c.in.packed); xfree(c.in.offsets); xfree(c.rows); xfree(c.counts);
  if (c.non_ascii) xfree(c.non_ascii);
  if (c.res < 0) { errno = c.err; rb_sys_fail("blurrily_storage_find_batch"); }   /* ENODEV: no GPU, no CPU fallback */
  return out;
}

static VALUE blurrily_find_batch(VALUE self, VALUE rb_needles, VALUE rb_limit)
{
  return find_batch_common(self, rb_needles, rb_limit, 0);
}

static VALUE blurrily_find_batch_raw(VALUE self, VALUE rb_needles, VALUE rb_limit)
{
  return find_batch_common(self, rb_needles, rb_limit, 1);
}

static VALUE blurrily_put_many(VALUE self, VALUE rb_needles, VALUE rb_refs, VALUE rb_weights)
{
  trigram_map    map = map_of(self);
  packed_needles in;
  uint32_t      *refs, *weights = NULL;
  long           i, added;
  Check_Type(rb_refs, T_ARRAY);
  pack_needles(rb_needles, &in);
  if ((size_t)RARRAY_LEN(rb_refs) != in.n || (!NIL_P(rb_weights) && (size_t)RARRAY_LEN(rb_weights) != in.n)) {
    xfree(in.packed); xfree(in.offsets);
    rb_raise(rb_eArgError, "needles, references and weights differ in length");
  }
  refs = ALLOC_N(uint32_t, in.n + 1);
  if (!NIL_P(rb_weights)) weights = ALLOC_N(uint32_t, in.n + 1);
  for (i = 0; i < (long)in.n; ++i) {
    refs[i] = NUM2UINT(rb_ary_entry(rb_refs, i));
    if (weights) weights[i] = NUM2UINT(rb_ary_entry(rb_weights, i));
  }
  added = blurrily_storage_put_many(map, in.packed, in.offsets, refs, weights, in.n);
  xfree(in.packed); xfree(in.offsets); xfree(refs);
  if (weights) xfree(weights);
  if (added < 0) rb_sys_fail("blurrily_storage_put_many");
  return LONG2NUM(added);
}

static VALUE blurrily_sync_device(VALUE self)
{
  if (blurrily_storage_sync_device(map_of(self)) < 0) rb_sys_fail("blurrily_storage_sync_device");
  return Qnil;
}

static VALUE blurrily_set_option(VALUE self, VALUE rb_key, VALUE rb_value)
{
  if (blurrily_storage_set_option(map_of(self), StringValueCStr(rb_key), NUM2LL(rb_value)) < 0)
    rb_sys_fail("blurrily_storage_set_option");
  return rb_value;
}

static VALUE blurrily_get_option(VALUE self, VALUE rb_key)
{
  long long v = 0;
  if (blurrily_storage_get_option(map_of(self), StringValueCStr(rb_key), &v) < 0)
    rb_sys_fail("blurrily_storage_get_option");
  return LL2NUM(v);
}

void Init_map_ext(void)
{
  Init_map_ext_reference();                                        /* Blurrily::RawMap as the gem defines it */
  mBlurrily = rb_define_module("Blurrily");
  cRawMap   = rb_const_get(mBlurrily, rb_intern("RawMap"));
  rb_define_method(cRawMap, "find_batch",     blurrily_find_batch,     2);
  rb_define_method(cRawMap, "find_batch_raw", blurrily_find_batch_raw, 2);
  rb_define_method(cRawMap, "put_many",       blurrily_put_many,       3);
  rb_define_method(cRawMap, "sync_device",    blurrily_sync_device,    0);
  rb_define_method(cRawMap, "set_option",     blurrily_set_option,     2);
  rb_define_method(cRawMap, "get_option",     blurrily_get_option,     1);
}
