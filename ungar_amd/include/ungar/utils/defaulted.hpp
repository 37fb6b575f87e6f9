// ungar_amd :: `defaulted<V>` constructor arguments (reference include/ungar/utils/defaulted.hpp:33-67):
// an argument that is either given or takes the value V named in its type; `default_value` selects V
// explicitly so that later positional arguments can still be passed.
#pragma once

#include <optional>

namespace Ungar {

template <auto DEFAULT>
class defaulted {
  public:
    using value_type = decltype(DEFAULT);
    constexpr defaulted() = default;
    constexpr defaulted(std::nullopt_t) {  // NOLINT: implicit on purpose, as in the reference
    }
    template <class T>
        requires std::is_convertible_v<T, value_type>
    constexpr defaulted(T v) : given_{static_cast<value_type>(v)} {  // NOLINT
    }
    constexpr value_type value() const {
        return given_ ? *given_ : DEFAULT;
    }
    constexpr operator value_type() const {  // NOLINT
        return value();
    }

  private:
    std::optional<value_type> given_;
};

inline constexpr auto default_value = std::nullopt;

}  // namespace Ungar
