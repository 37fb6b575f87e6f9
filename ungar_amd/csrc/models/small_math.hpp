// ungar_amd :: scalar-generic 3-vector / quaternion helpers used by the built-in node models.
//
// These restate the *formulas* Eigen 3.4.0 evaluates for the expressions in the reference's model
// lambdas, so that a recorded tape contains the same real-valued function:
//   quaternion product  (xyzw storage)                    Eigen/src/Geometry/Quaternion.h  quat_product
//   q * v  = v + w t + u x t,  t = 2 (u x v)              Eigen QuaternionBase::_transformVector
//   ApproximateNorm / ApproximateExponentialMap           include/ungar/utils/utils.hpp:731-749
#pragma once

#include <array>
#include <cmath>
#include <type_traits>
#include <limits>

namespace ungar_amd::models {

template <class S>
using Vec3 = std::array<S, 3>;
/// Unit quaternion stored in Eigen coefficient order x, y, z, w.
template <class S>
using Quat = std::array<S, 4>;

template <class S>
inline Vec3<S> Cross(const Vec3<S>& a, const Vec3<S>& b) {
    return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
}
template <class S>
inline Vec3<S> Add(const Vec3<S>& a, const Vec3<S>& b) {
    return {a[0] + b[0], a[1] + b[1], a[2] + b[2]};
}
template <class S>
inline Vec3<S> Sub(const Vec3<S>& a, const Vec3<S>& b) {
    return {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
}
template <class S, class T>
inline Vec3<S> Scale(const T& s, const Vec3<S>& a) {
    return {s * a[0], s * a[1], s * a[2]};
}
/// min(a, b); on the tape a conditional expression, as Utils::Min (reference utils.hpp:969-982).
template <class S>
inline S Min(const S& a, const S& b) {
    if constexpr (std::is_arithmetic_v<S>) return a < b ? a : b;
    else return CondExpGt(a, b, b, a);  // found by ADL in ungar_amd::tape
}

template <class S>
inline S SquaredNorm(const Vec3<S>& a) {
    return a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
}

/// Eigen's quaternion-times-vector: v + w*t + u x t with t = 2 (u x v).
template <class S>
inline Vec3<S> Rotate(const Quat<S>& q, const Vec3<S>& v) {
    const Vec3<S> u{q[0], q[1], q[2]};
    Vec3<S> t = Cross(u, v);
    t = {t[0] + t[0], t[1] + t[1], t[2] + t[2]};
    const Vec3<S> ut = Cross(u, t);
    return {v[0] + q[3] * t[0] + ut[0], v[1] + q[3] * t[1] + ut[1], v[2] + q[3] * t[2] + ut[2]};
}

/// Eigen's Hamilton product a * b (xyzw storage).
template <class S>
inline Quat<S> QuatMul(const Quat<S>& a, const Quat<S>& b) {
    return {a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1],
            a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2],
            a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0],
            a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]};
}

/// utils.hpp:731-736 -- sqrt(|v|^2 + eps), eps = machine epsilon of double.
template <class S>
inline S ApproximateNorm(const Vec3<S>& v) {
    using std::sqrt;
    return sqrt(SquaredNorm(v) + S{std::numeric_limits<double>::epsilon()});
}

/// utils.hpp:738-749 -- q.vec = v sin(|v|~/2) / |v|~ ;  q.w = cos(|v|~/2).
template <class S>
inline Quat<S> ApproximateExponentialMap(const Vec3<S>& v) {
    using std::cos;
    using std::sin;
    const S n = ApproximateNorm(v);
    const S s = sin(S{0.5} * n);
    return {v[0] * s / n, v[1] * s / n, v[2] * s / n, cos(S{0.5} * n)};
}

}  // namespace ungar_amd::models
