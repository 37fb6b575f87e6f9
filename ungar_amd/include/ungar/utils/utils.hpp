// ungar_amd :: scalar / quaternion helpers that user model lambdas call while a tape is recorded.
//
// Reference include/ungar/utils/utils.hpp: ApproximateNorm :731-736, ApproximateExponentialMap
// :738-749, Pow :820-837, Sqrt :839-852, Min :969-982, SmoothMin :984-989, Sign :991-999,
// Abs :1001-1015, SmoothAbs :1017-1021, ToRealFunction :1040-1059, CompareMatrices :1062-1137.
// These define the exact expressions parity is measured against: e.g. the
// exponential map recorded on a tape is always the *approximate*, branch-free one.
#pragma once

#include <numbers>
#include <random>
#include <string>
#include <tuple>
#include <vector>
#include <utility>

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

/// The reference's form (utils.hpp:1062-1137): two dense matrix / vector expressions of equal shape.
template <class A, class B>
    requires requires(const A& a, const B& b) { a.size(); b.size(); a.rows(); b.cols(); }
inline bool CompareMatrices(const A& a, std::string_view nameA, const B& b, std::string_view nameB, real_t relTol = 1e-2, real_t absTol = 1e-3, bool verbose = true) {
    if (a.rows() != b.rows() || a.cols() != b.cols()) {
        if (verbose) std::fprintf(stderr, "[ungar] size mismatch: %.*s is %td x %td, %.*s is %td x %td\n", static_cast<int>(nameA.size()), nameA.data(), static_cast<index_t>(a.rows()),
                                  static_cast<index_t>(a.cols()), static_cast<int>(nameB.size()), nameB.data(), static_cast<index_t>(b.rows()), static_cast<index_t>(b.cols()));
        return false;
    }
    std::vector<real_t> va(static_cast<std::size_t>(a.size())), vb(static_cast<std::size_t>(b.size()));
    for (index_t c = 0; c < static_cast<index_t>(a.cols()); ++c)
        for (index_t r = 0; r < static_cast<index_t>(a.rows()); ++r) {
            va[static_cast<std::size_t>(c * a.rows() + r)] = a(r, c);
            vb[static_cast<std::size_t>(c * a.rows() + r)] = b(r, c);
        }
    return CompareMatrices(va.data(), nameA, vb.data(), nameB, static_cast<index_t>(va.size()), relTol, absTol, verbose);
}

/// CamelCase -> snake_case; digits kept, any other non-letter becomes '_' (reference utils.hpp:499-522: "CCWord" -> "cc_word").
inline std::string ToSnakeCase(std::string_view in) {
    std::string out;
    const auto lower = [](char c) { return c >= 'a' && c <= 'z'; };
    const auto upper = [](char c) { return c >= 'A' && c <= 'Z'; };
    for (std::size_t i = 0; i < in.size(); ++i) {
        const char c = in[i];
        if (c >= '0' && c <= '9') {
            out += c;
        } else if (lower(c) || upper(c)) {
            out += static_cast<char>(upper(c) ? c - 'A' + 'a' : c);
            const bool wordEnds = i + 1 < in.size() && lower(c) && upper(in[i + 1]);                               // "lC" -> "l_c"
            const bool acronymEnds = i + 2 < in.size() && upper(c) && upper(in[i + 1]) && lower(in[i + 2]);       // "CCW" + "o" -> "cc_w"
            if (wordEnds || acronymEnds) out += '_';
        } else {
            out += '_';
        }
    }
    return out;
}

#if defined(UNGAR_AMD_USE_SYSTEM_EIGEN)
// Rotation helpers on matrix types and the small block inverses (reference utils.hpp:765-960); real Eigen only -- the built-in
// algebra has no fixed-size matrices.
template <class S>
inline Matrix3<S> ElementaryXRotationMatrix(const S& angle) {
    const S c = cos(angle), s = sin(angle);
    Matrix3<S> R;
    R << S{1.0}, S{0.0}, S{0.0}, S{0.0}, c, -s, S{0.0}, s, c;
    return R;
}
template <class S>
inline Matrix3<S> ElementaryYRotationMatrix(const S& angle) {
    const S c = cos(angle), s = sin(angle);
    Matrix3<S> R;
    R << c, S{0.0}, s, S{0.0}, S{1.0}, S{0.0}, -s, S{0.0}, c;
    return R;
}
template <class S>
inline Matrix3<S> ElementaryZRotationMatrix(const S& angle) {
    const S c = cos(angle), s = sin(angle);
    Matrix3<S> R;
    R << c, -s, S{0.0}, s, c, S{0.0}, S{0.0}, S{0.0}, S{1.0};
    return R;
}
/// ypr = (roll, pitch, yaw): R = Rz(yaw) Ry(pitch) Rx(roll).
template <class V>
inline Matrix3<typename V::Scalar> RotationMatrixFromYawPitchRoll(const Eigen::MatrixBase<V>& ypr) {
    return ElementaryZRotationMatrix(ypr.z()) * ElementaryYRotationMatrix(ypr.y()) * ElementaryXRotationMatrix(ypr.x());
}
template <class V>
inline Quaternion<typename V::Scalar> QuaternionFromYawPitchRoll(const Eigen::MatrixBase<V>& ypr) {
    return ElementaryZQuaternion(ypr.z()) * ElementaryYQuaternion(ypr.y()) * ElementaryXQuaternion(ypr.x());
}
/// Inverse of an invertible 3 x 3 matrix by cofactors.
template <class M>
inline Matrix3<typename M::Scalar> Inverse3(const Eigen::MatrixBase<M>& m) {
    using S = typename M::Scalar;
    UNGAR_ASSERT(m.rows() == 3 && m.cols() == 3);
    Matrix3<S> adj;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {  // adj(j, i) = cofactor(i, j), cyclic index form (sign included)
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            adj(j, i) = m(i1, j1) * m(i2, j2) - m(i1, j2) * m(i2, j1);
        }
    const S det = m(0, 0) * adj(0, 0) + m(0, 1) * adj(1, 0) + m(0, 2) * adj(2, 0);
    return adj / det;
}
/// Inverse of a 6 x 6 block upper-triangular matrix [A B; 0 C] with invertible 3 x 3 blocks A, C.
template <class M>
inline Eigen::Matrix<typename M::Scalar, 6, 6> Inverse6(const Eigen::MatrixBase<M>& m) {
    using S = typename M::Scalar;
    UNGAR_ASSERT(m.rows() == 6 && m.cols() == 6);
    const Matrix3<S> Ai = Inverse3(m.derived().template topLeftCorner<3, 3>()), Ci = Inverse3(m.derived().template bottomRightCorner<3, 3>());
    Eigen::Matrix<S, 6, 6> inv;
    inv.template topLeftCorner<3, 3>() = Ai;
    inv.template topRightCorner<3, 3>() = -(Ai * m.derived().template topRightCorner<3, 3>() * Ci);
    inv.template bottomLeftCorner<3, 3>().setZero();
    inv.template bottomRightCorner<3, 3>() = Ci;
    return inv;
}

namespace detail {
/// Sparse matrices (any storage order / expression) stacked on top of each other (reference utils.hpp:140-230).
template <class... Ms>
class SparseStack {
  public:
    using ScalarType = std::common_type_t<typename std::remove_cvref_t<Ms>::Scalar...>;
    explicit SparseStack(Ms&&... ms) : parts_{std::forward<Ms>(ms)...} {}
    template <class Target>
    void In(const Eigen::SparseMatrixBase<Target>& target) {
        Fill(const_cast<Target&>(target.derived()));
    }
    SparseMatrix<ScalarType> ToSparse() {
        SparseMatrix<ScalarType> out;
        Fill(out);
        return out;
    }

  private:
    template <class Target>
    void Fill(Target& out) {
        std::vector<Eigen::Triplet<ScalarType>> entries;
        index_t rows = 0, cols = -1;
        std::apply(
            [&](const auto&... m) {
                (Append(m, entries, rows, cols), ...);
            },
            parts_);
        out.resize(rows, cols < 0 ? 0 : cols);
        out.setFromTriplets(entries.begin(), entries.end());
    }
    template <class Mat>
    static void Append(const Mat& expr, std::vector<Eigen::Triplet<ScalarType>>& entries, index_t& rows, index_t& cols) {
        const Eigen::SparseMatrix<ScalarType> m = expr;  // evaluates sparse views / products
        UNGAR_ASSERT(cols < 0 || cols == m.cols());
        cols = m.cols();
        for (int k = 0; k < m.outerSize(); ++k)
            for (typename Eigen::SparseMatrix<ScalarType>::InnerIterator it(m, k); it; ++it) entries.emplace_back(rows + it.row(), it.col(), it.value());
        rows += m.rows();
    }
    std::tuple<Ms&&...> parts_;
};
}  // namespace detail

template <class... Ms>
    requires(sizeof...(Ms) > 0)
inline auto VerticallyStackSparseMatrices(Ms&&... ms) {
    return detail::SparseStack<Ms&&...>{std::forward<Ms>(ms)...};
}
#endif  // UNGAR_AMD_USE_SYSTEM_EIGEN

inline constexpr std::string_view DASH_LINE_SEPARATOR = "----------------------------------------------------------------";
inline constexpr std::string_view STAR_LINE_SEPARATOR = "****************************************************************";

// ---- Decompose / Compose (reference utils/utils.hpp:232-480) -----------------------------------------------------------
// Decompose<SIZES...>(v) splits a vector into a tuple of views, in order: a size-1 piece is a REFERENCE to the scalar, a piece of
// size Q an Eigen::Map of a quaternion over 4 coefficients, any other piece an Eigen::Map of a fixed-size vector; the views are
// const when the vector is (a const lvalue, or an Eigen::Map / Eigen::Ref / block of const data).  Compose(a, b, ...) is the
// inverse: scalars and vectors concatenated `.In(target)`, `.ToDynamic()` or (all sizes fixed) `.ToFixed()`.
namespace detail {

constexpr int PieceSize(int size) { return size == static_cast<int>(Q) ? 4 : size; }

template <class S, int SIZE, bool CONST>
struct Piece {
    using Data = std::conditional_t<CONST, const S, S>;
    using type = Eigen::Map<std::conditional_t<CONST, const Vector<S, SIZE>, Vector<S, SIZE>>>;
    static type Make(Data* p) { return type{p}; }
};
template <class S, bool CONST>
struct Piece<S, 1, CONST> {
    using Data = std::conditional_t<CONST, const S, S>;
    using type = Data&;
    static type Make(Data* p) { return *p; }
};
template <class S, bool CONST>
struct Piece<S, static_cast<int>(Q), CONST> {
    using Data = std::conditional_t<CONST, const S, S>;
    using type = Eigen::Map<std::conditional_t<CONST, const Quaternion<S>, Quaternion<S>>>;
    static type Make(Data* p) { return type{p}; }
};

template <bool CONST, int... SIZES, class V, std::size_t... IS>
auto DecomposeAt(V& v, std::index_sequence<IS...>) {
    using S = std::remove_const_t<typename std::remove_cvref_t<V>::Scalar>;
    constexpr int sizes[] = {PieceSize(SIZES)..., 0};
    constexpr auto offset = [](std::size_t i) {
        int o = 0;
        for (std::size_t k = 0; k < i; ++k) o += sizes[k];
        return o;
    };
    UNGAR_ASSERT(static_cast<int>(v.size()) == offset(sizeof...(SIZES)));
    auto* base = const_cast<std::conditional_t<CONST, const S, S>*>(v.data());
    return std::tuple<typename Piece<S, SIZES, CONST>::type...>{Piece<S, SIZES, CONST>::Make(base + offset(IS))...};
}

/// Does this vector expression only give const access to its coefficients?
template <class V>
inline constexpr bool kReadOnly = std::is_const_v<std::remove_reference_t<V>> ||
                                  std::is_const_v<std::remove_pointer_t<decltype(std::declval<std::remove_cvref_t<V>&>().data())>>;

template <class T>
inline constexpr bool kIsVector = requires(const std::remove_cvref_t<T>& t) { t.size(); t.data(); typename std::remove_cvref_t<T>::Scalar; } ||
                                  requires(const std::remove_cvref_t<T>& t) { t.size(); t[0]; typename std::remove_cvref_t<T>::Scalar; };

template <class T>
struct ComposableTraits {  // scalars
    using Scalar = std::remove_cvref_t<T>;
    static constexpr int size = 1;
};
template <class T>
    requires kIsVector<T>
struct ComposableTraits<T> {
    using Scalar = std::remove_const_t<typename std::remove_cvref_t<T>::Scalar>;
    static constexpr int size = static_cast<int>(std::remove_cvref_t<T>::RowsAtCompileTime);
};

template <class... Ts>
class Composition {
  public:
    static constexpr bool kAllFixed = ((ComposableTraits<Ts>::size >= 0) && ...);
    static constexpr int kSize = kAllFixed ? (ComposableTraits<Ts>::size + ... + 0) : -1;
    using ScalarType = std::common_type_t<typename ComposableTraits<Ts>::Scalar...>;

    explicit Composition(Ts&&... parts) : parts_{std::forward<Ts>(parts)...} {}

    /// Writes the concatenation into `target` (resized when dynamic; a fixed-size target must have the composed size).
    template <class Target>
    void In(Target&& target) {
        auto& out = const_cast<std::remove_cvref_t<Target>&>(static_cast<const std::remove_cvref_t<Target>&>(target));
        const index_t total = std::apply([](const auto&... p) { return (index_t{0} + ... + Extent(p)); }, parts_);
        if constexpr (std::remove_cvref_t<Target>::RowsAtCompileTime < 0) {
            if (static_cast<index_t>(out.size()) != total) out.resize(total);
        } else {
            UNGAR_ASSERT(static_cast<index_t>(out.size()) == total);
        }
        index_t at = 0;
        std::apply([&](const auto&... p) { (Put(out, at, p), ...); }, parts_);
    }
    auto ToDynamic() {
        VectorX<ScalarType> v;
        In(v);
        return v;
    }
    auto ToFixed()
        requires kAllFixed
    {
        Vector<ScalarType, kSize> v;
        In(v);
        return v;
    }

  private:
    template <class P>
    static index_t Extent(const P& p) {
        if constexpr (kIsVector<P>) return static_cast<index_t>(p.size());
        else return 1;
    }
    template <class Out, class P>
    static void Put(Out& out, index_t& at, const P& p) {
        if constexpr (kIsVector<P>) {
            for (index_t i = 0; i < static_cast<index_t>(p.size()); ++i) out[at + i] = p[i];
            at += static_cast<index_t>(p.size());
        } else {
            out[at++] = p;
        }
    }
    std::tuple<Ts&&...> parts_;
};

}  // namespace detail

template <int... SIZES, class V>
inline auto Decompose(V&& vector) {
    return detail::DecomposeAt<detail::kReadOnly<V>, SIZES...>(vector, std::make_index_sequence<sizeof...(SIZES)>{});
}

template <class... Ts>
inline detail::Composition<Ts&&...> Compose(Ts&&... parts) {
    return detail::Composition<Ts&&...>{std::forward<Ts>(parts)...};
}

}  // namespace Utils
}  // namespace Ungar
