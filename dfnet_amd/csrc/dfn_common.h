// dfn_common.h — error reporting shared by the translation units of libdfnet_hip.so.
#pragma once

namespace dfn {
// Records a message for dfn_last_error() (thread-local) and returns `code`.
int set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// Compute units of the current device (256 on MI355X).
int device_cu_count();
}  // namespace dfn
