// rsx_internal.h — shared between the host builders and the device side of librsx (not installed).
#ifndef RSX_INTERNAL_H
#define RSX_INTERNAL_H

#include <cstdarg>
#include <cstdio>

#include "../../include/rsx.h"

// records the thread-local error text and returns `code` (so call sites can `return rsx_fail(...)`)
int rsx_fail(int code, const char *fmt, ...);

#endif
