// host_guard.h - the exception guard of libprt_host.so's C entry points (include/prt_host.h).
//
// The host mirror is C++ (std::vector, std::string, new); an exception that leaves an extern "C" function is
// std::terminate, i.e. an abort of the caller's process.  Every entry point that can allocate runs its body inside
// HOST_API_TRY ... HOST_API_CATCH_*: the exception becomes the function's error value and a message.
#pragma once

#include <cstdio>
#include <exception>
#include <new>
#include <string>

inline void HostApiException(std::string * err, const char * where) noexcept {
    char what[256] = "unknown C++ exception";
    try { throw; }
    catch (const std::bad_alloc &) { snprintf(what, sizeof(what), "out of host memory (std::bad_alloc)"); }
    catch (const std::exception & e) { snprintf(what, sizeof(what), "C++ exception: %s", e.what()); }
    catch (...) {}
    try { if (err) *err = std::string(where) + ": " + what; } catch (...) {}
}

#define HOST_API_TRY try {
#define HOST_API_CATCH(err, where, value) } catch (...) { HostApiException(err, where); return value; }
#define HOST_API_CATCH_VOID(err, where) } catch (...) { HostApiException(err, where); }
