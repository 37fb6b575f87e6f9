// ungar_amd :: AD scalar and vector typedefs of the host API
// (reference include/ungar/autodiff/data_types.hpp:39-111).
#pragma once

#include <functional>
#include <type_traits>

#include "../../../csrc/tape/derive.hpp"
#include "../data_types.hpp"

namespace Ungar {

/// Recorded scalar: replaces CppAD::AD<CppAD::cg::CG<real_t>> (data_types.hpp:39-41).
using ad_scalar_t = ::ungar_amd::tape::AD;

using VectorXad = VectorX<ad_scalar_t>;
using Vector2ad = Vector2<ad_scalar_t>;
using Vector3ad = Vector3<ad_scalar_t>;
using Vector4ad = Vector4<ad_scalar_t>;
using Quaternionad = Quaternion<ad_scalar_t>;
using MapToQuaternionad = Eigen::Map<Quaternionad>;
using MapToConstQuaternionad = Eigen::Map<const Quaternionad>;

/// y = f([x^T p^T]^T) on AD scalars (data_types.hpp:82-91).
using ADFunction = std::function<void(const VectorXad& xp, VectorXad& y)>;

/// Same odd encoding as the reference (data_types.hpp:95-111): NONE = 1, `&` yields bool.
enum class EnabledDerivatives : unsigned { NONE = 1U << 0, JACOBIAN = 1U << 1, HESSIAN = 1U << 2, ALL = JACOBIAN | HESSIAN };
constexpr auto operator|(const EnabledDerivatives a, const EnabledDerivatives b) {
    return static_cast<EnabledDerivatives>(static_cast<unsigned>(a) | static_cast<unsigned>(b));
}
constexpr bool operator&(const EnabledDerivatives a, const EnabledDerivatives b) {
    return static_cast<bool>(static_cast<unsigned>(a) & static_cast<unsigned>(b));
}

}  // namespace Ungar

namespace Eigen {
template <>
struct NumTraits<::ungar_amd::tape::AD> {
    static ::ungar_amd::tape::AD epsilon() {
        return ::ungar_amd::tape::AD{std::numeric_limits<double>::epsilon()};
    }
};
}  // namespace Eigen
