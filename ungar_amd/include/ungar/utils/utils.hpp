// ungar_amd :: scalar / quaternion helpers that user model lambdas call while a tape is recorded.
//
// Reference include/ungar/utils/utils.hpp: ApproximateNorm :731-736, ApproximateExponentialMap
// :738-749, Pow :820-837, Sqrt :839-852, Min :969-982, SmoothMin :984-989, Sign :991-999,
// Abs :1001-1015, SmoothAbs :1017-1021, ToRealFunction :1040-1059, CompareMatrices :1062-1137.
// These define the exact expressions parity is measured against: e.g. the
// exponential map recorded on a tape is always the *approximate*, branch-free one.
#pragma once

#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <string_view>
#include <type_traits>

#include "../autodiff/data_types.hpp"
#include "../io/logging.hpp"

namespace Ungar {
namespace Utils {

inline constexpr index_t Q = ::Ungar::Q;

template <class T>
inline constexpr bool is_ad_v = std::is_same_v<std::remove_cvref_t<T>, ad_scalar_t>;

template <class V>
inline std::remove_const_t<typename V::Scalar> ApproximateNorm(const Eigen::MatrixBase<V>& v) {
    using S = std::remove_const_t<typename V::Scalar>;
    using std::sqrt;
    return sqrt(v.squaredNorm() + Eigen::NumTraits<S>::epsilon());
}

template <class V>
inline Quaternion<std::remove_const_t<typename V::Scalar>> ApproximateExponentialMap(const Eigen::MatrixBase<V>& v) {
    using S = std::remove_const_t<typename V::Scalar>;
    using std::cos;
    using std::sin;
    Quaternion<S> q;
    const S n = ApproximateNorm(v);
    q.vec() = v * sin(0.5 * n) / n;
    q.w() = cos(0.5 * n);
    return q;
}

/// Exact exponential map; real scalars only (the reference static_asserts on AD, utils.hpp:704-707).
template <class V>
inline Quaternion<real_t> ExponentialMap(const Eigen::MatrixBase<V>& v) {
    static_assert(std::is_same_v<std::remove_const_t<typename V::Scalar>, real_t>, "ExponentialMap is not implemented for AD scalars; use ApproximateExponentialMap");
    const real_t n = v.norm();
    if (n == 0.0) return Quaternion<real_t>::Identity();
    const real_t s = std::sin(0.5 * n) / n;
    return Quaternion<real_t>{std::cos(0.5 * n), v[0] * s, v[1] * s, v[2] * s};
}

template <class B, class E>
inline auto Pow(const B& base, const E& exponent) {
    if constexpr (is_ad_v<B>) {
        if constexpr (std::is_integral_v<E>) return ::ungar_amd::tape::pow(base, static_cast<int>(exponent));
        else return ::ungar_amd::tape::pow(base, ad_scalar_t{exponent});
    } else {
        return std::pow(base, exponent);
    }
}
template <class S>
inline S Sqrt(const S& a) {
    using std::sqrt;
    return sqrt(a);
}
template <class S>
inline S Min(const S& a, const std::type_identity_t<S>& b) {
    if constexpr (is_ad_v<S>) return ::ungar_amd::tape::CondExpGt(a, b, b, a);
    else return std::min(a, b);
}
template <class S>
inline S SmoothMin(const S& a, const std::type_identity_t<S>& b, const std::type_identity_t<S>& alpha = S{8.0}) {
    using std::exp;
    return (a * exp(-alpha * a) + b * exp(-alpha * b)) / (exp(-alpha * a) + exp(-alpha * b));
}
template <class S>
inline S Sign(const S& a) {
    if constexpr (is_ad_v<S>)
        return ::ungar_amd::tape::CondExpGt(a, S{0.0}, S{1.0}, S{0.0}) - ::ungar_amd::tape::CondExpLt(a, S{0.0}, S{1.0}, S{0.0});
    else return static_cast<real_t>(a > 0.0) - static_cast<real_t>(a < 0.0);
}
template <class S>
inline S Abs(const S& a) {
    using std::abs;
    return abs(a);
}
template <class S>
inline S SmoothAbs(const S& a, const S& epsilon = S{std::numeric_limits<double>::epsilon()}) {
    return Sqrt(Pow(a, 2) + epsilon);
}

/// Rotations about the coordinate axes as unit quaternions (reference utils.hpp:906-925).
template <class S>
inline Quaternion<S> ElementaryXQuaternion(const S& angle) {
    using std::cos;
    using std::sin;
    return Quaternion<S>{cos(angle / S{2.0}), sin(angle / S{2.0}), S{0.0}, S{0.0}};
}
template <class S>
inline Quaternion<S> ElementaryYQuaternion(const S& angle) {
    using std::cos;
    using std::sin;
    return Quaternion<S>{cos(angle / S{2.0}), S{0.0}, sin(angle / S{2.0}), S{0.0}};
}
template <class S>
inline Quaternion<S> ElementaryZQuaternion(const S& angle) {
    using std::cos;
    using std::sin;
    return Quaternion<S>{cos(angle / S{2.0}), S{0.0}, S{0.0}, sin(angle / S{2.0})};
}

/// [roll, pitch, yaw] of a unit quaternion (yaw = .z()).  Real scalars follow the reference's
/// `toRotationMatrix().eulerAngles(2, 1, 0).reverse()` (utils.hpp:949-953), i.e. Eigen's convention
/// with the yaw folded into [0, pi]; tape scalars use the branch-free atan2 form (utils.hpp:956-966).
template <class Q>
inline auto QuaternionToYawPitchRoll(const Eigen::QuaternionBase<Q>& q) {
    using S = std::remove_const_t<typename Eigen::QuaternionBase<Q>::Scalar>;
    using std::atan2;
    using std::cos;
    using std::sin;
    using std::sqrt;
    const S x = q.x(), y = q.y(), z = q.z(), w = q.w();
    const S r00 = 1.0 - 2.0 * (y * y + z * z), r01 = 2.0 * (x * y - w * z), r02 = 2.0 * (x * z + w * y);
    const S r10 = 2.0 * (x * y + w * z), r11 = 1.0 - 2.0 * (x * x + z * z), r12 = 2.0 * (y * z - w * x);
    const S r20 = 2.0 * (x * z - w * y), r21 = 2.0 * (y * z + w * x), r22 = 1.0 - 2.0 * (x * x + y * y);
    if constexpr (is_ad_v<S>) {
        (void)r01, (void)r02, (void)r11, (void)r12;
        return Vector3<S>{atan2(r21, r22), atan2(-r20, sqrt(r21 * r21 + r22 * r22)), atan2(r10, r00)};
    } else {
        constexpr real_t pi = 3.14159265358979323846;
        real_t yaw = std::atan2(r10, r00), pitch;
        const real_t c2 = std::sqrt(r22 * r22 + r21 * r21);
        if (yaw < 0.0) {
            yaw += pi;
            pitch = std::atan2(-r20, -c2);
        } else {
            pitch = std::atan2(-r20, c2);
        }
        const real_t s1 = std::sin(yaw), c1 = std::cos(yaw);
        const real_t roll = std::atan2(s1 * r02 - c1 * r12, c1 * r11 - s1 * r01);
        return Vector3<S>{roll, pitch, yaw};
    }
}

/// The single coefficient of a 1-vector (reference utils.hpp:1026-1035).
template <class V>
inline auto Squeeze(const Eigen::MatrixBase<V>& m) {
    if (m.size() != 1) throw std::invalid_argument("Utils::Squeeze: the argument must have exactly one coefficient");
    return m[0];
}

/// Runs an AD lambda on doubles: arguments are cast to un-recorded AD literals and the result is
/// read back with Value() (reference utils.hpp:1040-1059).
template <class F>
struct RealFunctionHelper {
    F autodiffFunction;
    template <class... Vs>
    VectorXr operator()(const Vs&... realVectors) const {
        return autodiffFunction(realVectors.template cast<ad_scalar_t>()...).unaryExpr([](const ad_scalar_t& el) { return ::ungar_amd::tape::Value(el); });
    }
};
template <class F>
inline auto ToRealFunction(const F& f) {
    return RealFunctionHelper<F>{f};
}

/// Pass criterion of the reference's self-tests (utils.hpp:1070-1071, 1089): an entry FAILS only if
/// BOTH its relative error > 1e-2 and its absolute error > 1e-3.
inline bool CompareMatrices(const real_t* a, std::string_view nameA, const real_t* b, std::string_view nameB, index_t n,
                            real_t relTol = 1e-2, real_t absTol = 1e-3, bool verbose = true) {
    bool ok = true;
    for (index_t i = 0; i < n; ++i) {
        const real_t abs = std::fabs(a[i] - b[i]);
        const real_t rel = abs / std::max(std::fabs(b[i]), std::numeric_limits<real_t>::min());
        if (rel > relTol && abs > absTol) {
            ok = false;
            if (verbose) std::fprintf(stderr, "[ungar] mismatch at %td: %.*s = %.12g, %.*s = %.12g\n", i, static_cast<int>(nameA.size()),
                                      nameA.data(), a[i], static_cast<int>(nameB.size()), nameB.data(), b[i]);
        }
    }
    return ok;
}

}  // namespace Utils
}  // namespace Ungar
