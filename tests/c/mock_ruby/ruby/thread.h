/* tests/c/mock_ruby/ruby/thread.h -- TEST INFRASTRUCTURE, declarations only (see ../ruby.h). */
#ifndef BLURRILY_MOCK_RUBY_THREAD_H
#define BLURRILY_MOCK_RUBY_THREAD_H 1
typedef void rb_unblock_function_t(void*);
void* rb_thread_call_without_gvl(void* (*func)(void*), void* data1, rb_unblock_function_t* ubf, void* data2);
#define RUBY_UBF_IO ((rb_unblock_function_t*)-1)
#endif
