// xpbd_internal.h -- shared by the translation units behind the C ABI (not installed).
#pragma once

#include <cstdint>

namespace xpbd {

// Records the thread's last error message (xpbd_last_error) and returns `code`.
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

} // namespace xpbd
