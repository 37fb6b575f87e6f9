// ungar_amd :: AD-safe quaternion / vector operations on REAL Eigen types (UNGAR_AMD_USE_SYSTEM_EIGEN builds only).
//
// Eigen's stock normalize(), normalized(), inverse() and slerp() branch on the VALUE of a norm or a dot product, which a
// recorded scalar does not have.  The reference replaces them by explicit specialisations whose branches are conditional
// expressions on the tape (include/ungar/autodiff/support/quaternion.hpp:34-192) and forbids the two operations that cannot
// be recorded that way (:120-129 rotation matrix -> quaternion, :194-222 setFromTwoVectors).  Same behaviour here, for the
// same plain-object and Map types, written against this project's tape scalar.  (With the built-in algebra the guarded
// formulas are the only ones, linalg.hpp.)
#pragma once

#include <type_traits>

namespace Ungar::Detail {
template <class...>
inline constexpr bool dependent_false = false;
/// sqrt(z) where z > 0, else 1: the divisor that leaves the null vector untouched.
inline ::ungar_amd::tape::AD GuardedNorm(const ::ungar_amd::tape::AD& z) {
    return ::ungar_amd::tape::CondExpGt(z, ::ungar_amd::tape::AD{0.0}, ::ungar_amd::tape::sqrt(z), ::ungar_amd::tape::AD{1.0});
}
template <class Self, class Other>
inline Eigen::Quaternion<::ungar_amd::tape::AD> Slerp(const Self& self, const ::ungar_amd::tape::AD& t, const Other& other) {
    namespace tape = ::ungar_amd::tape;
    using S = tape::AD;
    const S one = S{1.0} - S{std::numeric_limits<double>::epsilon()};
    const S d = self.dot(other);
    const S absD = tape::abs(d);
    const S theta = tape::acos(absD), sinTheta = tape::sin(theta);
    const S scale0 = tape::CondExpGe(absD, one, S{1.0} - t, tape::sin((S{1.0} - t) * theta) / sinTheta);
    S scale1 = tape::CondExpGe(absD, one, t, tape::sin(t * theta) / sinTheta);
    scale1 = tape::CondExpLt(d, S{0.0}, -scale1, scale1);
    return Eigen::Quaternion<S>{scale0 * self.coeffs() + scale1 * other.coeffs()};
}
}  // namespace Ungar::Detail

namespace Eigen {

#define UNGAR_AMD_AD_QUATERNION(QUAT)                                                                                                  \
    template <>                                                                                                                        \
    inline Quaternion<::ungar_amd::tape::AD> QuaternionBase<QUAT>::inverse() const {                                                    \
        const Scalar n2 = this->squaredNorm();                                                                                          \
        return Quaternion<Scalar>{conjugate().coeffs() / ::ungar_amd::tape::CondExpGt(n2, Scalar{0.0}, n2, Scalar{1.0})};               \
    }                                                                                                                                  \
    template <>                                                                                                                        \
    template <class OtherDerived>                                                                                                      \
    inline Quaternion<::ungar_amd::tape::AD> QuaternionBase<QUAT>::slerp(const Scalar& t, const QuaternionBase<OtherDerived>& other) const { \
        return ::Ungar::Detail::Slerp(*this, t, other);                                                                                \
    }                                                                                                                                  \
    template <>                                                                                                                        \
    template <typename Derived1, typename Derived2>                                                                                    \
    inline QUAT& QuaternionBase<QUAT>::setFromTwoVectors(const MatrixBase<Derived1>&, const MatrixBase<Derived2>&) {                    \
        static_assert(::Ungar::Detail::dependent_false<Derived1, Derived2>,                                                             \
                      "The construction of unit quaternions with scalar type 'ad_scalar_t' from two vectors is not implemented.");     \
        return derived();                                                                                                              \
    }
UNGAR_AMD_AD_QUATERNION(Quaternion<::ungar_amd::tape::AD>)
UNGAR_AMD_AD_QUATERNION(Map<Quaternion<::ungar_amd::tape::AD>>)
UNGAR_AMD_AD_QUATERNION(Map<const Quaternion<::ungar_amd::tape::AD>>)
#undef UNGAR_AMD_AD_QUATERNION

#define UNGAR_AMD_AD_NORMALIZED(VEC)                                                                        \
    template <>                                                                                             \
    inline const typename MatrixBase<VEC>::PlainObject MatrixBase<VEC>::normalized() const {                 \
        const PlainObject n(derived());                                                                     \
        return n / ::Ungar::Detail::GuardedNorm(n.squaredNorm());                                           \
    }
#define UNGAR_AMD_AD_NORMALIZE(VEC)                                                 \
    template <>                                                                     \
    inline void MatrixBase<VEC>::normalize() {                                      \
        derived() /= ::Ungar::Detail::GuardedNorm(squaredNorm());                   \
    }
#define UNGAR_AMD_AD_VECTOR(N)                                                      \
    UNGAR_AMD_AD_NORMALIZED(Matrix<::ungar_amd::tape::AD UNGAR_AMD_COMMA N UNGAR_AMD_COMMA 1>)             \
    UNGAR_AMD_AD_NORMALIZED(Map<Matrix<::ungar_amd::tape::AD UNGAR_AMD_COMMA N UNGAR_AMD_COMMA 1>>)        \
    UNGAR_AMD_AD_NORMALIZED(Map<const Matrix<::ungar_amd::tape::AD UNGAR_AMD_COMMA N UNGAR_AMD_COMMA 1>>)  \
    UNGAR_AMD_AD_NORMALIZE(Matrix<::ungar_amd::tape::AD UNGAR_AMD_COMMA N UNGAR_AMD_COMMA 1>)              \
    UNGAR_AMD_AD_NORMALIZE(Map<Matrix<::ungar_amd::tape::AD UNGAR_AMD_COMMA N UNGAR_AMD_COMMA 1>>)
#define UNGAR_AMD_COMMA ,
UNGAR_AMD_AD_VECTOR(3)
UNGAR_AMD_AD_VECTOR(4)
#undef UNGAR_AMD_COMMA
#undef UNGAR_AMD_AD_VECTOR
#undef UNGAR_AMD_AD_NORMALIZE
#undef UNGAR_AMD_AD_NORMALIZED

namespace internal {
/// Rotation matrix -> quaternion needs value-dependent branches (:120-129 of the reference header).
template <>
struct quaternionbase_assign_impl<Matrix<::ungar_amd::tape::AD, 3, 3>, 3, 3> {
    template <class Derived>
    static inline void run(QuaternionBase<Derived>&, const Matrix<::ungar_amd::tape::AD, 3, 3>&) {
        static_assert(::Ungar::Detail::dependent_false<Derived>,
                      "The construction of unit quaternions from rotation matrices with scalar type 'ad_scalar_t' is not implemented.");
    }
};
}  // namespace internal

}  // namespace Eigen
