/*
 * tests/c/mock_ruby/ruby.h -- TEST INFRASTRUCTURE: declarations only.
 *
 * The image this repository is built in has no Ruby (no ruby.h).  This header declares -- types, macros and
 * prototypes, no definitions -- the part of Ruby's documented C API (ruby.h of Ruby >= 2.0) that the blurrily
 * gem's glue (ext/blurrily/map_ext.c) and ruby/ext/blurrily/map_ext_batch.c use, so that
 * tests/test_ruby_glue_syntax.py can run the compiler's FRONT END over those sources:
 *     gcc -fsyntax-only -std=c99 -Wall -Wextra -Werror
 * Nothing is compiled to object code, linked or run with it; it is not a stand-in for a Ruby build and no
 * oracle, baseline or product path depends on it.  rb_define_method takes its function as `VALUE (*)()`
 * (what ANYARGS is in C), as the real header does.
 */
#ifndef BLURRILY_MOCK_RUBY_H
#define BLURRILY_MOCK_RUBY_H 1
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

typedef uintptr_t VALUE;
typedef uintptr_t ID;

#define Qfalse ((VALUE)0)
#define Qtrue  ((VALUE)20)
#define Qnil   ((VALUE)8)
#define NIL_P(v) ((VALUE)(v) == Qnil)
#define ANYARGS
enum ruby_value_type { T_ARRAY = 7, T_STRING = 5, T_DATA = 12 };

extern VALUE rb_cObject, rb_eRuntimeError, rb_eArgError;

ID    rb_intern(const char*);
VALUE rb_ivar_get(VALUE, ID);
VALUE rb_ivar_set(VALUE, ID, VALUE);
VALUE rb_const_get(VALUE, ID);
void  rb_raise(VALUE, const char*, ...) __attribute__((noreturn));
void  rb_sys_fail(const char*) __attribute__((noreturn));
void  rb_check_type(VALUE, int);
#define Check_Type(v, t) rb_check_type((VALUE)(v), (t))

VALUE rb_define_module(const char*);
VALUE rb_define_class_under(VALUE, const char*, VALUE);
void  rb_define_method(VALUE, const char*, VALUE (*)(ANYARGS), int);
void  rb_define_singleton_method(VALUE, const char*, VALUE (*)(ANYARGS), int);
void  rb_define_alias(VALUE, const char*, const char*);
void  rb_obj_call_init(VALUE, int, const VALUE*);
VALUE rb_funcallv(VALUE, ID, int, const VALUE*);
VALUE rb_ensure(VALUE (*)(VALUE), VALUE, VALUE (*)(VALUE), VALUE);

/* wrapped C pointers */
typedef void (*RUBY_DATA_FUNC)(void*);
struct RData { VALUE flags, klass; RUBY_DATA_FUNC dmark, dfree; void* data; };
VALUE rb_data_object_wrap(VALUE, void*, RUBY_DATA_FUNC, RUBY_DATA_FUNC);
#define Data_Wrap_Struct(klass, mark, free, sval) \
  rb_data_object_wrap((klass), (sval), (RUBY_DATA_FUNC)(mark), (RUBY_DATA_FUNC)(free))
#define DATA_PTR(obj) (((struct RData*)(obj))->data)
#define Data_Get_Struct(obj, type, sval) ((sval) = (type*)DATA_PTR(obj))

/* strings and arrays */
VALUE rb_string_value(volatile VALUE*);
char* rb_string_value_ptr(volatile VALUE*);
char* rb_string_value_cstr(volatile VALUE*);
#define StringValue(v)     rb_string_value(&(v))
#define StringValuePtr(v)  rb_string_value_ptr(&(v))
#define StringValueCStr(v) rb_string_value_cstr(&(v))
char* rb_mock_rstring_ptr(VALUE);
long  rb_mock_rstring_len(VALUE);
long  rb_mock_rarray_len(VALUE);
#define RSTRING_PTR(s) rb_mock_rstring_ptr(s)
#define RSTRING_LEN(s) rb_mock_rstring_len(s)
#define RARRAY_LEN(a)  rb_mock_rarray_len(a)
VALUE rb_ary_new(void);
VALUE rb_ary_new2(long);
VALUE rb_ary_new3(long, ...);
VALUE rb_ary_push(VALUE, VALUE);
VALUE rb_ary_entry(VALUE, long);
VALUE rb_hash_new(void);
VALUE rb_hash_aset(VALUE, VALUE, VALUE);
#define ID2SYM(id) ((VALUE)(id))

/* numbers */
unsigned long rb_num2uint(VALUE);
long long     rb_num2ll(VALUE);
VALUE rb_uint_new(unsigned long);
VALUE rb_int_new(long);
VALUE rb_ll2inum(long long);
#define NUM2UINT(v) ((unsigned int)rb_num2uint(v))
#define NUM2LL(v)   rb_num2ll(v)
#define UINT2NUM(v) rb_uint_new(v)
#define INT2NUM(v)  rb_int_new(v)
#define LONG2NUM(v) rb_int_new(v)
#define LL2NUM(v)   rb_ll2inum(v)

/* memory */
void* ruby_xmalloc2(size_t, size_t);
void  ruby_xfree(void*);
#define ALLOC_N(type, n) ((type*)ruby_xmalloc2((size_t)(n), sizeof(type)))
#define xfree ruby_xfree

#endif
