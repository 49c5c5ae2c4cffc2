// Minimal stand-in for the subset of spdlog the reference viewer uses (apps/viewer/main.cpp:10,52,69,101):
// set_pattern, set_level, level::debug, debug/info/warn/error/critical with "{}" placeholders.
// spdlog is not vendored by the reference (it is fetched by CMake) and is absent here; the reference
// ships the same kind of shim for its Xcode build (apps/apple/VulkanSplatting/include/spdlog/spdlog.h).
#pragma once
#include <cstdio>
#include <sstream>
#include <string>

namespace spdlog {
namespace level {
enum level_enum { trace, debug, info, warn, err, critical, off };
}
namespace detail {
inline level::level_enum& threshold() {
    static level::level_enum lvl = level::info;
    return lvl;
}
inline void substitute(std::ostringstream& os, const char* fmt) { os << fmt; }
template <class T, class... Rest>
void substitute(std::ostringstream& os, const char* fmt, const T& value, const Rest&... rest) {
    for (; *fmt; ++fmt) {
        if (fmt[0] == '{' && fmt[1] == '}') {
            os << value;
            substitute(os, fmt + 2, rest...);
            return;
        }
        os << *fmt;
    }
}
template <class... Args>
void log(level::level_enum lvl, const char* tag, const char* fmt, const Args&... args) {
    if (lvl < threshold()) return;
    std::ostringstream os;
    substitute(os, fmt, args...);
    std::fprintf(stderr, "[%s] %s\n", tag, os.str().c_str());
}
template <class... Args>
void log(level::level_enum lvl, const char* tag, const std::string& fmt, const Args&... args) {
    log(lvl, tag, fmt.c_str(), args...);
}
}  // namespace detail
inline void set_pattern(const std::string&) {}
inline void set_level(level::level_enum lvl) { detail::threshold() = lvl; }
template <class F, class... A> void debug(const F& f, const A&... a) { detail::log(level::debug, "D", f, a...); }
template <class F, class... A> void info(const F& f, const A&... a) { detail::log(level::info, "I", f, a...); }
template <class F, class... A> void warn(const F& f, const A&... a) { detail::log(level::warn, "W", f, a...); }
template <class F, class... A> void error(const F& f, const A&... a) { detail::log(level::err, "E", f, a...); }
template <class F, class... A> void critical(const F& f, const A&... a) { detail::log(level::critical, "C", f, a...); }
}  // namespace spdlog
