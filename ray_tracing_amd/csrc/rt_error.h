// rt_error.h -- thread-local error text behind rt_last_error() (include/rt_amd.h)
#ifndef RT_ERROR_H_
#define RT_ERROR_H_
// records a printf-style message and returns `status` so call sites can `return rt_fail(...)`
int rt_fail(int status, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#endif
