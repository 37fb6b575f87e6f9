// ungar_amd :: UNGAR_LOG(level, "format {} {:.3f}", args...) without spdlog/fmt (absent from this image):
// a small `{}` formatter covering what the reference's headers and examples use -- `{}`, `{:.Nf}`,
// `{:>W}`, `{:<W}`, `{:>W.Nf}` and `fmt::join(range, separator)` (reference include/ungar/io/logging.hpp).
#pragma once

#include <cstdio>
#include <sstream>
#include <string>
#include <string_view>
#include <type_traits>

#include "../hana_support.hpp"

namespace Ungar {
namespace Log {

enum class Level { trace, debug, info, warn, error, critical };

template <class Range>
struct Joined {
    const Range& range;
    std::string_view separator;
};

namespace Internal {

struct Spec {
    char align = 0;
    int width = 0, precision = -1;
    char type = 0;
};

inline Spec ParseSpec(std::string_view s) {
    Spec sp;
    std::size_t i = 0;
    if (i < s.size() && (s[i] == '>' || s[i] == '<' || s[i] == '^')) sp.align = s[i++];
    while (i < s.size() && s[i] >= '0' && s[i] <= '9') sp.width = 10 * sp.width + (s[i++] - '0');
    if (i < s.size() && s[i] == '.') {
        ++i;
        sp.precision = 0;
        while (i < s.size() && s[i] >= '0' && s[i] <= '9') sp.precision = 10 * sp.precision + (s[i++] - '0');
    }
    if (i < s.size()) sp.type = s[i];
    return sp;
}

template <class T>
std::string ToText(const T& v, const Spec& sp) {
    if constexpr (std::is_floating_point_v<T>) {
        char buf[64];
        if (sp.precision >= 0) std::snprintf(buf, sizeof buf, sp.type == 'e' ? "%.*e" : "%.*f", sp.precision, static_cast<double>(v));
        else std::snprintf(buf, sizeof buf, "%g", static_cast<double>(v));
        return buf;
    } else if constexpr (std::is_same_v<T, bool>) {
        return v ? "true" : "false";
    } else if constexpr (std::is_arithmetic_v<T> || std::is_convertible_v<T, std::string_view>) {
        std::ostringstream os;
        os << v;
        return os.str();
    } else if constexpr (requires { v.range; v.separator; }) {
        std::string out;
        bool first = true;
        for (const auto& el : v.range) {
            if (!first) out += v.separator;
            first = false;
            out += ToText(el, sp);
        }
        return out;
    } else if constexpr (requires { v.IsLiteral(); }) {
        return ToText(Value(v), sp);
#if defined(UNGAR_AMD_HAS_HANA)
    } else if constexpr (boost::hana::Struct<T>::value) {
        // the reference's formatter for Boost.Hana structs (io/logging.hpp:93-145): 'c' = "{ m0, m1 }", default 'v' = one "key = value" per line
        const bool compact = sp.type == 'c';
        std::string body;
        bool first = true;
        boost::hana::for_each(boost::hana::accessors<T>(), [&](auto accessor) {
            body += first ? (compact ? "" : "\t") : (compact ? ", " : ",\n\t");
            first = false;
            if (!compact) body += std::string(boost::hana::to<const char*>(boost::hana::first(accessor))) + " = ";
            body += ToText(boost::hana::second(accessor)(v), Spec{});
        });
        return compact ? "{ " + body + " }" : "\n{\n" + body + "\n}";
#endif
    } else {
        std::ostringstream os;
        os << v;
        return os.str();
    }
}

inline void Pad(std::string& text, const Spec& sp, bool numeric) {
    if (static_cast<int>(text.size()) >= sp.width) return;
    const std::size_t fill = static_cast<std::size_t>(sp.width) - text.size();
    const char align = sp.align ? sp.align : (numeric ? '>' : '<');
    if (align == '>') text.insert(0, fill, ' ');
    else if (align == '<') text.append(fill, ' ');
    else {
        text.insert(0, fill / 2, ' ');
        text.append(fill - fill / 2, ' ');
    }
}

inline void FormatTo(std::string& out, std::string_view fmt) {
    for (std::size_t i = 0; i < fmt.size(); ++i) {
        if ((fmt[i] == '{' || fmt[i] == '}') && i + 1 < fmt.size() && fmt[i + 1] == fmt[i]) ++i;
        out += fmt[i];
    }
}

template <class T, class... Rest>
void FormatTo(std::string& out, std::string_view fmt, const T& first, const Rest&... rest) {
    for (std::size_t i = 0; i < fmt.size(); ++i) {
        if (fmt[i] == '{' && i + 1 < fmt.size() && fmt[i + 1] == '{') {
            out += '{';
            ++i;
        } else if (fmt[i] == '}' && i + 1 < fmt.size() && fmt[i + 1] == '}') {
            out += '}';
            ++i;
        } else if (fmt[i] == '{') {
            const std::size_t close = fmt.find('}', i);
            if (close == std::string_view::npos) break;
            std::string_view inner = fmt.substr(i + 1, close - i - 1);
            const std::size_t colon = inner.find(':');
            const Spec sp = colon == std::string_view::npos ? Spec{} : ParseSpec(inner.substr(colon + 1));
            std::string text = ToText(first, sp);
            Pad(text, sp, std::is_arithmetic_v<T>);
            out += text;
            FormatTo(out, fmt.substr(close + 1), rest...);
            return;
        } else {
            out += fmt[i];
        }
    }
}

}  // namespace Internal

template <class... Args>
std::string Format(std::string_view fmt, const Args&... args) {
    std::string out;
    Internal::FormatTo(out, fmt, args...);
    return out;
}

inline Level& Threshold() {
    static Level level = Level::info;
    return level;
}

template <class... Args>
void Write(Level level, std::string_view fmt, const Args&... args) {
    if (level < Threshold()) return;
    static constexpr const char* kNames[] = {"trace", "debug", "info", "warning", "error", "critical"};
    std::fprintf(level >= Level::warn ? stderr : stdout, "[ungar] [%s] %s\n", kNames[static_cast<int>(level)], Format(fmt, args...).c_str());
}

}  // namespace Log
}  // namespace Ungar

// the reference's examples spell `fmt::join(range, ", ")` directly
namespace fmt {
template <class Range>
::Ungar::Log::Joined<Range> join(const Range& range, std::string_view separator) {
    return {range, separator};
}
}  // namespace fmt

#define UNGAR_LOG(level, ...) ::Ungar::Log::Write(::Ungar::Log::Level::level, __VA_ARGS__)
