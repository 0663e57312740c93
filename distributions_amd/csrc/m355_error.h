// Thread-local last-error record shared by every translation unit of libm355dist.so.
#pragma once
namespace m355 {
// Records a printf-style message for m355_last_error() and returns `code`.
int fail(int code, const char* fmt, ...);
}  // namespace m355
