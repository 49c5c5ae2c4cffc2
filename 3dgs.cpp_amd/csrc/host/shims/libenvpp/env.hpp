// Minimal stand-in for the subset of libenvpp the reference viewer uses (apps/viewer/main.cpp:46-50,57-62):
// env::prefix(name), register_variable<T>(name), parse_and_validate(), get(id) -> optional<T>, get_or(id, dflt).
// Variables are read as <PREFIX>_<NAME>.  libenvpp is not vendored by the reference and is absent here.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <optional>
#include <sstream>
#include <string>
#include <vector>

namespace env {
template <class T>
struct variable_id {
    size_t index;
};

class parsed {
public:
    explicit parsed(std::vector<std::optional<std::string>> values) : values_(std::move(values)) {}
    template <class T>
    std::optional<T> get(const variable_id<T>& id) const {
        const auto& raw = values_[id.index];
        if (!raw) return std::nullopt;
        return convert<T>(*raw);
    }
    template <class T>
    T get_or(const variable_id<T>& id, const T& dflt) const {
        auto v = get(id);
        return v ? *v : dflt;
    }
    bool ok() const { return true; }

private:
    template <class T>
    static std::optional<T> convert(const std::string& s) {
        if constexpr (std::is_same_v<T, bool>) {
            if (s == "1" || s == "true" || s == "TRUE" || s == "on" || s == "yes") return true;
            if (s == "0" || s == "false" || s == "FALSE" || s == "off" || s == "no") return false;
            return std::nullopt;
        } else if constexpr (std::is_integral_v<T>) {
            long long v = 0;
            std::istringstream is(s);
            if (!(is >> v)) return std::nullopt;
            return static_cast<T>(v);
        } else {
            T v{};
            std::istringstream is(s);
            if (!(is >> v)) return std::nullopt;
            return v;
        }
    }
    std::vector<std::optional<std::string>> values_;
};

class prefix {
public:
    explicit prefix(std::string name) : name_(std::move(name)) {}
    template <class T>
    variable_id<T> register_variable(const std::string& var) {
        names_.push_back(name_ + "_" + var);
        return variable_id<T>{names_.size() - 1};
    }
    parsed parse_and_validate() const {
        std::vector<std::optional<std::string>> values;
        for (const auto& n : names_) {
            const char* v = std::getenv(n.c_str());
            values.push_back(v ? std::optional<std::string>(v) : std::nullopt);
        }
        return parsed(std::move(values));
    }

private:
    std::string name_;
    std::vector<std::string> names_;
};
}  // namespace env
