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

#if defined(UNGAR_AMD_USE_SYSTEM_EIGEN)
// ---- hooks real Eigen documents for custom scalar types (https://eigen.tuxfamily.org/dox/TopicCustomizing_CustomScalar.html)
namespace Eigen {
template <>
struct NumTraits<::ungar_amd::tape::AD> : GenericNumTraits<::ungar_amd::tape::AD> {
    using Real = ::ungar_amd::tape::AD;
    using NonInteger = ::ungar_amd::tape::AD;
    using Nested = ::ungar_amd::tape::AD;
    using Literal = double;
    enum { IsComplex = 0, IsInteger = 0, IsSigned = 1, RequireInitialization = 1, ReadCost = 1, AddCost = 2, MulCost = 2 };
    static Real epsilon() { return Real{std::numeric_limits<double>::epsilon()}; }
    static Real dummy_precision() { return Real{1e-12}; }
    static Real highest() { return Real{std::numeric_limits<double>::max()}; }
    static Real lowest() { return Real{std::numeric_limits<double>::lowest()}; }
    static int digits10() { return std::numeric_limits<double>::digits10; }
};
// mixed recorded / real arithmetic keeps the recorded type (Eigen asks for this per binary operation)
template <class BinaryOp>
struct ScalarBinaryOpTraits<::ungar_amd::tape::AD, double, BinaryOp> {
    using ReturnType = ::ungar_amd::tape::AD;
};
template <class BinaryOp>
struct ScalarBinaryOpTraits<double, ::ungar_amd::tape::AD, BinaryOp> {
    using ReturnType = ::ungar_amd::tape::AD;
};
}  // namespace Eigen

namespace ungar_amd::tape {
// the coefficient-wise helpers Eigen looks up by argument-dependent lookup on a custom scalar
inline const AD& conj(const AD& x) { return x; }
inline const AD& real(const AD& x) { return x; }
inline AD imag(const AD&) { return AD{0.0}; }
inline AD abs2(const AD& x) { return x * x; }
inline bool isfinite(const AD&) { return true; }
}  // namespace ungar_amd::tape

#include "support/quaternion.hpp"  // AD-safe inverse / normalize(d) / slerp, as in the reference's autodiff/support/quaternion.hpp
#else
namespace Eigen {
template <>
struct NumTraits<::ungar_amd::tape::AD> {
    static ::ungar_amd::tape::AD epsilon() {
        return ::ungar_amd::tape::AD{std::numeric_limits<double>::epsilon()};
    }
};
}  // namespace Eigen
#endif
