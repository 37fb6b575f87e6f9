// ungar_amd :: minimal dense linear algebra for the host side (vectors, maps, quaternions).
//
// The reference writes its models against Eigen 3.4 (bundled as a zip that its CMake unpacks,
// external/config/eigen/).  Eigen is not installed on the MI355X boxes and is not vendored here;
// the model lambdas only touch a narrow slice of it, which this header
// provides under the same spellings so that the example problems read unchanged:
//   Vector<S,N> / VectorX<S>, Map<...>, Quaternion<S>, cross, dot, squaredNorm, cwiseProduct,
//   cwiseInverse, asDiagonal, array(), comma initialisation, setZero/Ones/Constant/LinSpaced/Unit,
//   head/tail/segment, cast, unaryExpr, UnitX/Y/Z, Zero, Ones, Random, Identity.
// Evaluation is eager (no expression templates): the scalar is usually an AD handle whose
// operations are recorded once, so laziness buys nothing.  Formulas that define recorded
// arithmetic (quaternion product, quaternion * vector) follow Eigen's, see models/small_math.hpp.
//
// UNGAR_AMD_USE_SYSTEM_EIGEN: a project that already builds against the real Eigen (every real Ungar installation does:
// the reference bundles Eigen 3.4, external/config/eigen/) defines this macro and gets linalg_system_eigen.hpp instead --
// the facade's types then ARE Eigen's, `namespace Eigen` is not touched by this project beyond the scalar-type hooks Eigen
// documents for custom scalars, and Function::Jacobian returns the reference's own
// Eigen::Map<const Eigen::SparseMatrix<real_t, Eigen::RowMajor>> (function.hpp:217, 237).
#pragma once

#if defined(UNGAR_AMD_USE_SYSTEM_EIGEN)
#include "linalg_system_eigen.hpp"
#else

#include <array>
#include <ostream>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <initializer_list>
#include <limits>
#include <type_traits>
#include <vector>

namespace Eigen {

using Index = std::ptrdiff_t;
inline constexpr Index Dynamic = -1;
inline constexpr int Infinity = -1;  // lpNorm<Eigen::Infinity>()

template <class S>
struct NumTraits {
    static constexpr double epsilon() {
        return std::numeric_limits<double>::epsilon();
    }
};

namespace internal {

template <class S, Index N>
struct Storage {
    std::array<S, static_cast<std::size_t>(N)> v{};
    explicit Storage(Index = N) {
    }
    static constexpr Index size() {
        return N;
    }
    S* data() {
        return v.data();
    }
    const S* data() const {
        return v.data();
    }
    void resize(Index n) {
        assert(n == N);
        (void)n;
    }
};
template <class S>
struct Storage<S, Dynamic> {
    std::vector<S> v;
    explicit Storage(Index n = 0) : v(static_cast<std::size_t>(n)) {
    }
    Index size() const {
        return static_cast<Index>(v.size());
    }
    S* data() {
        return v.data();
    }
    const S* data() const {
        return v.data();
    }
    void resize(Index n) {
        v.resize(static_cast<std::size_t>(n));
    }
};

template <class T>
using remove_cvref_t = std::remove_cv_t<std::remove_reference_t<T>>;

// Conditional expressions on REAL scalars.  Recorded scalars bring their own CondExp* overloads (found by
// argument-dependent lookup, tape/scalar.hpp), which is how the guarded formulas below stay on the tape as
// selections instead of host branches (reference autodiff/support/quaternion.hpp:34-192).
template <class S, std::enable_if_t<std::is_arithmetic_v<S>, int> = 0>
inline S CondExpGt(S a, S b, S c, S d) { return a > b ? c : d; }
template <class S, std::enable_if_t<std::is_arithmetic_v<S>, int> = 0>
inline S CondExpGe(S a, S b, S c, S d) { return a >= b ? c : d; }
template <class S, std::enable_if_t<std::is_arithmetic_v<S>, int> = 0>
inline S CondExpLt(S a, S b, S c, S d) { return a < b ? c : d; }

/// sqrt(z) where z > 0, else 1: the divisor of a normalisation that leaves the null vector untouched (Eigen 3.4's
/// `if (z > 0) derived() /= sqrt(z)`, and the reference's AD specialisations, support/quaternion.hpp:56-116).
template <class S>
inline S GuardedNorm(const S& squaredNorm) {
    using std::sqrt;
    return CondExpGt(squaredNorm, S{0.0}, sqrt(squaredNorm), S{1.0});
}

}  // namespace internal

template <class S, Index N>
class Matrix;  // column vector only: Matrix<S, N, 1>
template <class S, Index N>
using Vector = Matrix<S, N>;
template <class S>
using VectorX = Matrix<S, Dynamic>;

template <class Derived>
class MatrixBase;

template <class Derived>
class CommaInitializer {
  public:
    CommaInitializer(Derived& d, Index filled) : d_{d}, i_{filled} {
    }
    template <class T>
    CommaInitializer& operator,(const T& v) {
        Append(v);
        return *this;
    }
    Derived& finished() {
        assert(i_ == d_.size());
        return d_;
    }
    template <class T>
    void Append(const T& v) {
        if constexpr (std::is_base_of_v<MatrixBase<internal::remove_cvref_t<T>>, internal::remove_cvref_t<T>>) {
            for (Index k = 0; k < v.size(); ++k) d_[i_++] = v[k];
        } else {
            d_[i_++] = v;
        }
    }

  private:
    Derived& d_;
    Index i_;
};

template <class Derived>
struct Traits;

/// Array view used for coefficient-wise `/` and `*` (`a.array() / b.array()`).
template <class S, Index N>
struct ArrayWrap {
    Matrix<S, N> v;
    friend Matrix<S, N> operator/(const ArrayWrap& a, const ArrayWrap& b) {
        Matrix<S, N> r(a.v.size());
        for (Index i = 0; i < r.size(); ++i) r[i] = a.v[i] / b.v[i];
        return r;
    }
    friend Matrix<S, N> operator*(const ArrayWrap& a, const ArrayWrap& b) {
        Matrix<S, N> r(a.v.size());
        for (Index i = 0; i < r.size(); ++i) r[i] = a.v[i] * b.v[i];
        return r;
    }
};

template <class S, Index N>
struct DiagonalWrap {
    Matrix<S, N> d;
    template <class D2>
    Matrix<S, N> operator*(const MatrixBase<D2>& v) const {
        Matrix<S, N> r(d.size());
        for (Index i = 0; i < r.size(); ++i) r[i] = d[i] * v[i];
        return r;
    }
};

template <class Derived>
class MatrixBase {
  public:
    using Scalar = typename Traits<Derived>::Scalar;
    static constexpr Index RowsAtCompileTime = Traits<Derived>::Size;
    static constexpr Index ColsAtCompileTime = 1;
    static constexpr Index SizeAtCompileTime = Traits<Derived>::Size;
    using PlainObject = Matrix<std::remove_const_t<Scalar>, RowsAtCompileTime>;
    using S = std::remove_const_t<Scalar>;

    Derived& derived() {
        return static_cast<Derived&>(*this);
    }
    const Derived& derived() const {
        return static_cast<const Derived&>(*this);
    }
    Derived& const_cast_derived() const {
        return const_cast<Derived&>(derived());
    }
    Index size() const {
        return derived().size();
    }
    Index rows() const {
        return size();
    }
    Index cols() const {
        return 1;
    }
    decltype(auto) operator[](Index i) {
        return derived().data()[i];
    }
    const S& operator[](Index i) const {
        return derived().data()[i];
    }
    decltype(auto) operator()(Index i) {
        return derived().data()[i];
    }
    const S& operator()(Index i) const {
        return derived().data()[i];
    }
    decltype(auto) x() {
        return (*this)[0];
    }
    decltype(auto) y() {
        return (*this)[1];
    }
    decltype(auto) z() {
        return (*this)[2];
    }
    decltype(auto) w() {
        return (*this)[3];
    }
    const S& x() const {
        return (*this)[0];
    }
    const S& y() const {
        return (*this)[1];
    }
    const S& z() const {
        return (*this)[2];
    }
    const S& w() const {
        return (*this)[3];
    }

    PlainObject eval() const {
        PlainObject r(size());
        for (Index i = 0; i < size(); ++i) r[i] = (*this)[i];
        return r;
    }

    // ---- assignment-style mutators (available on Matrix and mutable Map) ---------------------
    template <class D2>
    Derived& assign(const MatrixBase<D2>& o) {
        if constexpr (RowsAtCompileTime == Dynamic) derived().resize(o.size());
        assert(size() == o.size());
        for (Index i = 0; i < size(); ++i) (*this)[i] = o[i];
        return derived();
    }
    Derived& setZero() {
        return setConstant(S{0.0});
    }
    Derived& setOnes() {
        return setConstant(S{1.0});
    }
    Derived& setConstant(const S& v) {
        for (Index i = 0; i < size(); ++i) (*this)[i] = v;
        return derived();
    }
    Derived& setLinSpaced(const S& lo, const S& hi) {
        const Index n = size();
        for (Index i = 0; i < n; ++i) (*this)[i] = n == 1 ? hi : lo + (hi - lo) * (static_cast<double>(i) / static_cast<double>(n - 1));
        return derived();
    }
    Derived& setUnit(Index k) {
        setZero();
        (*this)[k] = S{1.0};
        return derived();
    }
    Derived& setRandom() {
        for (Index i = 0; i < size(); ++i) (*this)[i] = S{2.0 * (static_cast<double>(std::rand()) / RAND_MAX) - 1.0};
        return derived();
    }
    template <class T>
    CommaInitializer<Derived> operator<<(const T& v) {
        CommaInitializer<Derived> ci{derived(), 0};
        ci.Append(v);
        return ci;
    }
    template <class D2>
    Derived& operator+=(const MatrixBase<D2>& o) {
        for (Index i = 0; i < size(); ++i) (*this)[i] = (*this)[i] + o[i];
        return derived();
    }
    template <class D2>
    Derived& operator-=(const MatrixBase<D2>& o) {
        for (Index i = 0; i < size(); ++i) (*this)[i] = (*this)[i] - o[i];
        return derived();
    }
    Derived& operator*=(const S& s) {
        for (Index i = 0; i < size(); ++i) (*this)[i] = (*this)[i] * s;
        return derived();
    }
    Derived& operator/=(const S& s) {
        for (Index i = 0; i < size(); ++i) (*this)[i] = (*this)[i] / s;
        return derived();
    }

    // ---- reductions / products --------------------------------------------------------------------
    S squaredNorm() const {
        S acc = (*this)[0] * (*this)[0];
        for (Index i = 1; i < size(); ++i) acc = acc + (*this)[i] * (*this)[i];
        return acc;
    }
    S norm() const {
        using std::sqrt;
        return sqrt(squaredNorm());
    }
    S sum() const {
        S acc = (*this)[0];
        for (Index i = 1; i < size(); ++i) acc = acc + (*this)[i];
        return acc;
    }
    S maxCoeff() const {
        S best = (*this)[0];
        for (Index i = 1; i < size(); ++i)
            if ((*this)[i] > best) best = (*this)[i];
        return best;
    }
    template <class I>
    S maxCoeff(I* index) const {
        *index = 0;
        for (Index i = 1; i < size(); ++i)
            if ((*this)[i] > (*this)[*index]) *index = static_cast<I>(i);
        return (*this)[*index];
    }
    S minCoeff() const {
        S best = (*this)[0];
        for (Index i = 1; i < size(); ++i)
            if ((*this)[i] < best) best = (*this)[i];
        return best;
    }
    /// P = 1, 2 or Eigen::Infinity.
    template <int P>
    S lpNorm() const {
        using std::abs;
        if constexpr (P == 2) {
            return norm();
        } else {
            S acc{0};
            for (Index i = 0; i < size(); ++i) {
                const S a = abs((*this)[i]);
                if constexpr (P == 1) acc = acc + a;
                else if (a > acc) acc = a;
            }
            return acc;
        }
    }
    /// Forward iteration over the coefficients (range-for, fmt::join).
    struct ConstIterator {
        const MatrixBase* m;
        Index i;
        S operator*() const { return (*m)[i]; }
        ConstIterator& operator++() {
            ++i;
            return *this;
        }
        bool operator!=(const ConstIterator& o) const { return i != o.i; }
    };
    ConstIterator begin() const { return {this, 0}; }
    ConstIterator end() const { return {this, size()}; }
    template <class D2>
    auto dot(const MatrixBase<D2>& o) const {
        auto acc = (*this)[0] * o[0];
        for (Index i = 1; i < size(); ++i) acc = acc + (*this)[i] * o[i];
        return acc;
    }
    template <class D2>
    auto cross(const MatrixBase<D2>& b) const {
        const auto& a = *this;
        Matrix<internal::remove_cvref_t<decltype((*this)[0] * b[0])>, RowsAtCompileTime> r(3);
        r[0] = a[1] * b[2] - a[2] * b[1];
        r[1] = a[2] * b[0] - a[0] * b[2];
        r[2] = a[0] * b[1] - a[1] * b[0];
        return r;
    }
    template <class D2>
    auto cwiseProduct(const MatrixBase<D2>& o) const {
        Matrix<internal::remove_cvref_t<decltype((*this)[0] * o[0])>, RowsAtCompileTime> r(size());
        for (Index i = 0; i < size(); ++i) r[i] = (*this)[i] * o[i];
        return r;
    }
    PlainObject cwiseInverse() const {
        PlainObject r(size());
        for (Index i = 0; i < size(); ++i) r[i] = S{1.0} / (*this)[i];
        return r;
    }
    /// Guarded like Eigen 3.4 (null vector unchanged); on recorded scalars the guard is a CondExpGt on the tape
    /// (reference autodiff/support/quaternion.hpp:56-116).
    PlainObject normalized() const {
        return *this / internal::GuardedNorm(squaredNorm());
    }
    void normalize() {
        derived() /= internal::GuardedNorm(squaredNorm());
    }
    ArrayWrap<S, RowsAtCompileTime> array() const {
        return {eval()};
    }
    DiagonalWrap<S, RowsAtCompileTime> asDiagonal() const {
        return {eval()};
    }
    template <class T>
    Matrix<T, RowsAtCompileTime> cast() const {
        Matrix<T, RowsAtCompileTime> r(size());
        for (Index i = 0; i < size(); ++i) r[i] = T{(*this)[i]};
        return r;
    }
    template <class F>
    auto unaryExpr(F&& f) const {
        using R = internal::remove_cvref_t<decltype(f((*this)[0]))>;
        Matrix<R, RowsAtCompileTime> r(size());
        for (Index i = 0; i < size(); ++i) r[i] = f((*this)[i]);
        return r;
    }
    VectorX<S> head(Index n) const {
        return segment(0, n);
    }
    VectorX<S> tail(Index n) const {
        return segment(size() - n, n);
    }
    VectorX<S> segment(Index start, Index n) const {
        VectorX<S> r(n);
        for (Index i = 0; i < n; ++i) r[i] = (*this)[start + i];
        return r;
    }
    template <Index K>
    Matrix<S, K> head() const {
        Matrix<S, K> r;
        for (Index i = 0; i < K; ++i) r[i] = (*this)[i];
        return r;
    }
    template <Index K>
    Matrix<S, K> tail() const {
        Matrix<S, K> r;
        for (Index i = 0; i < K; ++i) r[i] = (*this)[size() - K + i];
        return r;
    }
    template <class D2>
    bool isApprox(const MatrixBase<D2>& o, double prec = 1e-12) const {
        if (size() != o.size()) return false;
        double d2 = 0, a2 = 0, b2 = 0;
        for (Index i = 0; i < size(); ++i) {
            const double a = static_cast<double>((*this)[i]), b = static_cast<double>(o[i]);
            d2 += (a - b) * (a - b);
            a2 += a * a;
            b2 += b * b;
        }
        return d2 <= prec * prec * std::min(a2, b2);
    }

    PlainObject operator-() const {
        PlainObject r(size());
        for (Index i = 0; i < size(); ++i) r[i] = -(*this)[i];
        return r;
    }
};

/// Coefficients separated by blanks (Eigen prints column vectors one coefficient per line; one line reads
/// better in a log message).
template <class D>
std::ostream& operator<<(std::ostream& os, const MatrixBase<D>& v) {
    for (Index i = 0; i < v.size(); ++i) os << (i ? " " : "") << v[i];
    return os;
}

namespace internal {
template <class T>
inline constexpr bool is_scalar_like_v = std::is_arithmetic_v<remove_cvref_t<T>> || requires(const remove_cvref_t<T>& t) { t.IsLiteral(); };
/// Plain vector type of a binary operation between scalars SA and SB (double x tape scalar -> tape scalar).
template <class A, class SB>
using Promoted = Matrix<remove_cvref_t<decltype(std::declval<typename MatrixBase<A>::S>() * std::declval<SB>())>, MatrixBase<A>::RowsAtCompileTime>;
}  // namespace internal

template <class A, class B>
auto operator+(const MatrixBase<A>& a, const MatrixBase<B>& b) {
    internal::Promoted<A, typename MatrixBase<B>::S> r(a.size());
    for (Index i = 0; i < a.size(); ++i) r[i] = a[i] + b[i];
    return r;
}
template <class A, class B>
auto operator-(const MatrixBase<A>& a, const MatrixBase<B>& b) {
    internal::Promoted<A, typename MatrixBase<B>::S> r(a.size());
    for (Index i = 0; i < a.size(); ++i) r[i] = a[i] - b[i];
    return r;
}
template <class A, class T, std::enable_if_t<internal::is_scalar_like_v<T>, int> = 0>
auto operator*(const MatrixBase<A>& a, const T& s) {
    internal::Promoted<A, T> r(a.size());
    for (Index i = 0; i < a.size(); ++i) r[i] = a[i] * s;
    return r;
}
template <class A, class T, std::enable_if_t<internal::is_scalar_like_v<T>, int> = 0>
auto operator*(const T& s, const MatrixBase<A>& a) {
    internal::Promoted<A, T> r(a.size());
    for (Index i = 0; i < a.size(); ++i) r[i] = s * a[i];
    return r;
}
template <class A, class T, std::enable_if_t<internal::is_scalar_like_v<T>, int> = 0>
auto operator/(const MatrixBase<A>& a, const T& s) {
    internal::Promoted<A, T> r(a.size());
    for (Index i = 0; i < a.size(); ++i) r[i] = a[i] / s;
    return r;
}

template <class S_, Index N>
struct Traits<Matrix<S_, N>> {
    using Scalar = S_;
    static constexpr Index Size = N;
};

template <class S_, Index N>
class Matrix : public MatrixBase<Matrix<S_, N>> {
  public:
    using Base = MatrixBase<Matrix<S_, N>>;
    using Scalar = S_;
    Matrix() : st_{N == Dynamic ? 0 : N} {
    }
    template <class T, std::enable_if_t<std::is_integral_v<T> && N == Dynamic, int> = 0>
    explicit Matrix(T n) : st_{static_cast<Index>(n)} {
    }
    template <class T, std::enable_if_t<std::is_integral_v<T> && N != Dynamic && N != 1, int> = 0>
    explicit Matrix(T n) : st_{N} {
        assert(static_cast<Index>(n) == N);
        (void)n;
    }
    /// Coefficient constructors: Vector2(a,b), Vector3(a,b,c), Vector4(a,b,c,d), and braces.
    template <class A, class B, class... R, std::enable_if_t<(N == Dynamic || static_cast<Index>(2 + sizeof...(R)) == N), int> = 0>
    Matrix(const A& a, const B& b, const R&... r) : st_{static_cast<Index>(2 + sizeof...(R))} {
        Index i = 0;
        ((*this)[i++] = S_{a});
        ((*this)[i++] = S_{b});
        (((*this)[i++] = S_{r}), ...);
    }
    template <class D2>
    Matrix(const MatrixBase<D2>& o) : st_{o.size()} {  // NOLINT
        for (Index i = 0; i < o.size(); ++i) (*this)[i] = o[i];
    }
    template <class D2>
    Matrix& operator=(const MatrixBase<D2>& o) {
        return this->assign(o);
    }
    Index size() const {
        return st_.size();
    }
    S_* data() {
        return st_.data();
    }
    const S_* data() const {
        return st_.data();
    }
    void resize(Index n) {
        st_.resize(n);
    }

    static Matrix Zero(Index n = N) {
        Matrix r(Sized{}, n);
        r.setZero();
        return r;
    }
    static Matrix Ones(Index n = N) {
        Matrix r(Sized{}, n);
        r.setOnes();
        return r;
    }
    static Matrix Constant(Index n, const S_& v) {
        Matrix r(Sized{}, n);
        r.setConstant(v);
        return r;
    }
    static Matrix Constant(const S_& v) {
        return Constant(N, v);
    }
    static Matrix Random(Index n = N) {
        Matrix r(Sized{}, n);
        r.setRandom();
        return r;
    }
    static Matrix Unit(Index n, Index k) {
        Matrix r(Sized{}, n);
        r.setUnit(k);
        return r;
    }
    static Matrix Unit(Index k) {
        return Unit(N, k);
    }
    static Matrix UnitX() {
        return Unit(N, 0);
    }
    static Matrix UnitY() {
        return Unit(N, 1);
    }
    static Matrix UnitZ() {
        return Unit(N, 2);
    }
    static Matrix LinSpaced(Index n, const S_& lo, const S_& hi) {
        Matrix r(Sized{}, n);
        r.setLinSpaced(lo, hi);
        return r;
    }

  private:
    struct Sized {};
    Matrix(Sized, Index n) : st_{n} {
    }
    internal::Storage<S_, N> st_;
};

// ---- Map ----------------------------------------------------------------------------------------------
template <class Plain>
class Map;

template <class S_, Index N>
struct Traits<Map<Matrix<S_, N>>> {
    using Scalar = S_;
    static constexpr Index Size = N;
};
template <class S_, Index N>
struct Traits<Map<const Matrix<S_, N>>> {
    using Scalar = const S_;
    static constexpr Index Size = N;
};

template <class S_, Index N>
class Map<Matrix<S_, N>> : public MatrixBase<Map<Matrix<S_, N>>> {
  public:
    using Scalar = S_;
    Map(S_* p, Index n = N) : p_{p}, n_{n} {
    }
    Map(const Map&) = default;
    template <class D2>
    Map& operator=(const MatrixBase<D2>& o) {
        return this->assign(o);
    }
    Map& operator=(const Map& o) {
        return this->assign(o);
    }
    Index size() const {
        return n_;
    }
    S_* data() const {
        return p_;
    }
    void resize(Index n) {
        assert(n == n_);
        (void)n;
    }

  private:
    S_* p_;
    Index n_;
};

template <class S_, Index N>
class Map<const Matrix<S_, N>> : public MatrixBase<Map<const Matrix<S_, N>>> {
  public:
    using Scalar = S_;
    Map(const S_* p, Index n = N) : p_{p}, n_{n} {
    }
    Index size() const {
        return n_;
    }
    const S_* data() const {
        return p_;
    }

  private:
    const S_* p_;
    Index n_;
};

// ---- Quaternion (coefficients stored x, y, z, w) -----------------------------------------------------------
/// Coefficient-wise equality of two vector expressions (Eigen's operator==: sizes must agree).
template <class A, class B>
inline bool operator==(const MatrixBase<A>& a, const MatrixBase<B>& b) {
    assert(a.size() == b.size());
    for (Index i = 0; i < a.size(); ++i)
        if (!(a.derived().data()[i] == b.derived().data()[i])) return false;
    return true;
}
template <class A, class B>
inline bool operator!=(const MatrixBase<A>& a, const MatrixBase<B>& b) {
    return !(a == b);
}

template <class Derived>
class QuaternionBase {
  public:
    using Scalar = typename Traits<Derived>::Scalar;
    using S = std::remove_const_t<Scalar>;
    const Derived& derived() const {
        return static_cast<const Derived&>(*this);
    }
    Derived& derived() {
        return static_cast<Derived&>(*this);
    }
    template <class Other>
    bool operator==(const QuaternionBase<Other>& o) const {  // Eigen 3.4: coeffs() == o.coeffs()
        for (int i = 0; i < 4; ++i)
            if (!(derived().data()[i] == o.derived().data()[i])) return false;
        return true;
    }
    template <class Other>
    bool operator!=(const QuaternionBase<Other>& o) const {
        return !(*this == o);
    }
    const S& x() const { return derived().data()[0]; }
    const S& y() const { return derived().data()[1]; }
    const S& z() const { return derived().data()[2]; }
    const S& w() const { return derived().data()[3]; }
    decltype(auto) x() { return derived().data()[0]; }
    decltype(auto) y() { return derived().data()[1]; }
    decltype(auto) z() { return derived().data()[2]; }
    decltype(auto) w() { return derived().data()[3]; }
    auto coeffs() const {
        return Map<const Matrix<S, 4>>{derived().data()};
    }
    auto coeffs() {
        if constexpr (std::is_const_v<Scalar>) return Map<const Matrix<S, 4>>{derived().data()};
        else return Map<Matrix<S, 4>>{derived().data()};
    }
    auto vec() const {
        return Map<const Matrix<S, 3>>{derived().data()};
    }
    auto vec() {
        if constexpr (std::is_const_v<Scalar>) return Map<const Matrix<S, 3>>{derived().data()};
        else return Map<Matrix<S, 3>>{derived().data()};
    }
    Derived& setIdentity() {
        x() = S{0.0};
        y() = S{0.0};
        z() = S{0.0};
        w() = S{1.0};
        return derived();
    }
    S squaredNorm() const {
        return coeffs().squaredNorm();
    }
    S norm() const {
        return coeffs().norm();
    }
    /// Rotate a 3-vector: v + w t + u x t with t = 2 (u x v)  (Eigen's _transformVector).
    template <class D2>
    auto operator*(const MatrixBase<D2>& v) const {
        using R = internal::remove_cvref_t<decltype(x() * v[0])>;  // real rotation of a tape vector and vice versa
        const Matrix<S, 3> u{x(), y(), z()};
        Matrix<R, 3> t = u.cross(v);
        t = t + t;
        const Matrix<R, 3> ut = u.cross(t);
        return Matrix<R, 3>{v[0] + w() * t[0] + ut[0], v[1] + w() * t[1] + ut[1], v[2] + w() * t[2] + ut[2]};
    }
    template <class D2>
    S dot(const QuaternionBase<D2>& o) const {
        return x() * o.x() + y() * o.y() + z() * o.z() + w() * o.w();
    }
    void normalize() {
        const S n = internal::GuardedNorm(squaredNorm());
        x() = x() / n;
        y() = y() / n;
        z() = z() / n;
        w() = w() / n;
    }
    template <class T>
    auto cast() const;
    auto conjugate() const;
    /// Inverse rotation: conjugate / squaredNorm, the null quaternion mapping to itself -- on every scalar type through
    /// a conditional expression (reference autodiff/support/quaternion.hpp:34-54; Eigen 3.4 branches on n2 > 0).
    auto inverse() const;
    auto normalized() const;
    /// Spherical linear interpolation, Eigen 3.4's formula with its two branches (|d| >= 1 - eps: linear; d < 0: the
    /// shorter arc) expressed as conditional expressions so that it can be recorded
    /// (reference autodiff/support/quaternion.hpp:133-192).
    template <class D2>
    auto slerp(const S& t, const QuaternionBase<D2>& other) const;
    /// Rotation taking a to b.  Real scalars only: the reference static_asserts on recorded scalars
    /// (autodiff/support/quaternion.hpp:194-222).
    template <class D1, class D2>
    Derived& setFromTwoVectors(const MatrixBase<D1>& a, const MatrixBase<D2>& b) {
        static_assert(std::is_arithmetic_v<S> && sizeof(D1) != 0 && sizeof(D2) != 0,
                      "The construction of unit quaternions with scalar type 'ad_scalar_t' from two vectors is not implemented.");
        Matrix<S, 3> v0{a[0], a[1], a[2]}, v1{b[0], b[1], b[2]};
        v0.normalize();
        v1.normalize();
        const S c = v0.dot(v1);
        if (c < S{-1.0} + NumTraits<S>::epsilon()) {  // opposite vectors: any axis orthogonal to a
            Matrix<S, 3> axis = std::abs(v0[0]) < S{0.9} ? Matrix<S, 3>{S{0}, -v0[2], v0[1]} : Matrix<S, 3>{-v0[2], S{0}, v0[0]};
            axis.normalize();
            x() = axis[0];
            y() = axis[1];
            z() = axis[2];
            w() = S{0};
            return derived();
        }
        const Matrix<S, 3> axis = v0.cross(v1);
        const S s2 = std::sqrt((S{1} + c) * S{2});
        x() = axis[0] / s2;
        y() = axis[1] / s2;
        z() = axis[2] / s2;
        w() = s2 * S{0.5};
        return derived();
    }
    /// Assignment from a 3 x 3 rotation matrix (Shepperd's method, as Eigen).  Real scalars only: the reference
    /// static_asserts on recorded scalars (autodiff/support/quaternion.hpp:120-129).
    template <class M, std::enable_if_t<std::is_same_v<decltype(std::declval<const M&>().rows()), Index> && !std::is_base_of_v<MatrixBase<M>, M>, int> = 0>
    Derived& operator=(const M& m) {
        static_assert(std::is_arithmetic_v<S> && sizeof(M) != 0,
                      "The construction of unit quaternions from rotation matrices with scalar type 'ad_scalar_t' is not implemented.");
        assert(m.rows() == 3 && m.cols() == 3);
        S t = m(0, 0) + m(1, 1) + m(2, 2);
        S q[4];  // x, y, z, w
        if (t > S{0}) {
            t = std::sqrt(t + S{1.0});
            q[3] = S{0.5} * t;
            t = S{0.5} / t;
            q[0] = (m(2, 1) - m(1, 2)) * t;
            q[1] = (m(0, 2) - m(2, 0)) * t;
            q[2] = (m(1, 0) - m(0, 1)) * t;
        } else {
            Index i = 0;
            if (m(1, 1) > m(0, 0)) i = 1;
            if (m(2, 2) > m(i, i)) i = 2;
            const Index j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + S{1.0});
            q[i] = S{0.5} * t;
            t = S{0.5} / t;
            q[3] = (m(k, j) - m(j, k)) * t;
            q[j] = (m(j, i) + m(i, j)) * t;
            q[k] = (m(k, i) + m(i, k)) * t;
        }
        x() = q[0];
        y() = q[1];
        z() = q[2];
        w() = q[3];
        return derived();
    }
};

template <class S_>
class Quaternion;
template <class S_>
struct Traits<Quaternion<S_>> {
    using Scalar = S_;
    static constexpr Index Size = 4;
};
template <class S_>
struct Traits<Map<Quaternion<S_>>> {
    using Scalar = S_;
    static constexpr Index Size = 4;
};
template <class S_>
struct Traits<Map<const Quaternion<S_>>> {
    using Scalar = const S_;
    static constexpr Index Size = 4;
};

template <class S_>
class Quaternion : public QuaternionBase<Quaternion<S_>> {
  public:
    using Scalar = S_;
    Quaternion() = default;
    /// Eigen's constructor order: (w, x, y, z).
    Quaternion(const S_& w, const S_& x, const S_& y, const S_& z) : c_{x, y, z, w} {
    }
    template <class D2>
    Quaternion(const QuaternionBase<D2>& o) : c_{o.x(), o.y(), o.z(), o.w()} {  // NOLINT
    }
    template <class D2>
    Quaternion& operator=(const QuaternionBase<D2>& o) {
        c_ = {o.x(), o.y(), o.z(), o.w()};
        return *this;
    }
    static Quaternion Identity() {
        return Quaternion{S_{1.0}, S_{0.0}, S_{0.0}, S_{0.0}};
    }
    S_* data() {
        return c_.data();
    }
    const S_* data() const {
        return c_.data();
    }
    Quaternion conjugate() const {
        return Quaternion{this->w(), -this->x(), -this->y(), -this->z()};
    }
    using QuaternionBase<Quaternion<S_>>::operator=;

  private:
    std::array<S_, 4> c_{};
};

template <class S_>
class Map<Quaternion<S_>> : public QuaternionBase<Map<Quaternion<S_>>> {
  public:
    using Scalar = S_;
    explicit Map(S_* p) : p_{p} {
    }
    Map(const Map&) = default;
    template <class D2>
    Map& operator=(const QuaternionBase<D2>& o) {
        const S_ x = o.x(), y = o.y(), z = o.z(), w = o.w();
        p_[0] = x;
        p_[1] = y;
        p_[2] = z;
        p_[3] = w;
        return *this;
    }
    Map& operator=(const Map& o) {
        return operator=<Map>(o);
    }
    S_* data() const {
        return p_;
    }

  private:
    S_* p_;
};

template <class S_>
class Map<const Quaternion<S_>> : public QuaternionBase<Map<const Quaternion<S_>>> {
  public:
    using Scalar = S_;
    explicit Map(const S_* p) : p_{p} {
    }
    const S_* data() const {
        return p_;
    }

  private:
    const S_* p_;
};

/// Hamilton product, Eigen's formula.
template <class A, class B>
auto operator*(const QuaternionBase<A>& a, const QuaternionBase<B>& b) {
    using S = typename QuaternionBase<A>::S;
    return Quaternion<S>{a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                         a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                         a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                         a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x()};
}

template <class Derived>
auto QuaternionBase<Derived>::conjugate() const {
    return Quaternion<S>{w(), -x(), -y(), -z()};
}
template <class Derived>
auto QuaternionBase<Derived>::inverse() const {
    using internal::CondExpGt;
    const S n2 = squaredNorm();
    const S den = CondExpGt(n2, S{0.0}, n2, S{1.0});
    return Quaternion<S>{w() / den, -x() / den, -y() / den, -z() / den};
}
template <class Derived>
auto QuaternionBase<Derived>::normalized() const {
    const S n = internal::GuardedNorm(squaredNorm());
    return Quaternion<S>{w() / n, x() / n, y() / n, z() / n};
}
template <class Derived>
template <class D2>
auto QuaternionBase<Derived>::slerp(const S& t, const QuaternionBase<D2>& other) const {
    using internal::CondExpGe;
    using internal::CondExpLt;
    using std::abs;
    using std::acos;
    using std::sin;
    const S one = S{1.0} - NumTraits<S>::epsilon();
    const S d = dot(other);
    const S absD = abs(d);
    const S theta = acos(absD), sinTheta = sin(theta);
    const S scale0 = CondExpGe(absD, one, S{1.0} - t, sin((S{1.0} - t) * theta) / sinTheta);
    S scale1 = CondExpGe(absD, one, t, sin(t * theta) / sinTheta);
    scale1 = CondExpLt(d, S{0.0}, -scale1, scale1);
    return Quaternion<S>{scale0 * w() + scale1 * other.w(), scale0 * x() + scale1 * other.x(), scale0 * y() + scale1 * other.y(),
                         scale0 * z() + scale1 * other.z()};
}
template <class Derived>
template <class T>
auto QuaternionBase<Derived>::cast() const {
    return Quaternion<T>{T{w()}, T{x()}, T{y()}, T{z()}};
}

/// Minimal dense matrix (column-major like Eigen's default): storage + element access + matrix-vector
/// product; what the rigid-body quantities M(q) and M(q)^-1 are returned in.
template <class S_>
class DenseMatrix {
  public:
    using Scalar = S_;
    DenseMatrix() = default;
    DenseMatrix(Index rows, Index cols) : rows_{rows}, cols_{cols}, v_(static_cast<std::size_t>(rows * cols), S_{0.0}) {
    }
    Index rows() const { return rows_; }
    Index cols() const { return cols_; }
    Index size() const { return rows_ * cols_; }
    S_& operator()(Index r, Index c) { return v_[static_cast<std::size_t>(c * rows_ + r)]; }
    const S_& operator()(Index r, Index c) const { return v_[static_cast<std::size_t>(c * rows_ + r)]; }
    S_* data() { return v_.data(); }
    const S_* data() const { return v_.data(); }
    void resize(Index rows, Index cols) {
        rows_ = rows;
        cols_ = cols;
        v_.assign(static_cast<std::size_t>(rows * cols), S_{0.0});
    }
    template <class D>
    auto operator*(const MatrixBase<D>& x) const {
        Matrix<remove_cvref_helper<decltype(std::declval<S_>() * x[0])>, Dynamic> y(rows_);
        for (Index r = 0; r < rows_; ++r) {
            auto acc = (*this)(r, 0) * x[0];
            for (Index c = 1; c < cols_; ++c) acc = acc + (*this)(r, c) * x[c];
            y[r] = acc;
        }
        return y;
    }
    DenseMatrix transpose() const {
        DenseMatrix t(cols_, rows_);
        for (Index r = 0; r < rows_; ++r)
            for (Index c = 0; c < cols_; ++c) t(c, r) = (*this)(r, c);
        return t;
    }

  private:
    template <class T>
    using remove_cvref_helper = std::remove_cv_t<std::remove_reference_t<T>>;
    Index rows_ = 0, cols_ = 0;
    std::vector<S_> v_;
};
template <class S_>
std::ostream& operator<<(std::ostream& os, const DenseMatrix<S_>& m) {
    for (Index r = 0; r < m.rows(); ++r) {
        for (Index c = 0; c < m.cols(); ++c) os << (c ? " " : "") << m(r, c);
        if (r + 1 < m.rows()) os << "\n";
    }
    return os;
}

/// Minimal row-major compressed sparse matrix (what Function::Jacobian/Hessian return a view of,
/// reference function.hpp:375-383).
template <class S_>
class SparseMatrixCsr {
  public:
    SparseMatrixCsr() = default;
    SparseMatrixCsr(Index rows, Index cols, const int* innerStarts, const int* outerIndices, const S_* values)
        : rows_{rows}, cols_{cols}, starts_{innerStarts}, idx_{outerIndices}, val_{values} {
    }
    Index rows() const { return rows_; }
    Index cols() const { return cols_; }
    Index nonZeros() const { return starts_ ? starts_[rows_] : 0; }
    const int* outerIndexPtr() const { return starts_; }
    const int* innerIndexPtr() const { return idx_; }
    const S_* valuePtr() const { return val_; }
    S_ coeff(Index r, Index c) const {
        for (int k = starts_[r]; k < starts_[r + 1]; ++k)
            if (idx_[k] == c) return val_[k];
        return S_{0};
    }
    /// Row-major dense copy, rows() x cols().
    std::vector<S_> toDense() const {
        std::vector<S_> d(static_cast<std::size_t>(rows_ * cols_), S_{0});
        for (Index r = 0; r < rows_; ++r)
            for (int k = starts_[r]; k < starts_[r + 1]; ++k) d[static_cast<std::size_t>(r * cols_ + idx_[k])] = val_[k];
        return d;
    }

  private:
    Index rows_ = 0, cols_ = 0;
    const int* starts_ = nullptr;
    const int* idx_ = nullptr;
    const S_* val_ = nullptr;
};

}  // namespace Eigen

namespace Ungar::Linalg {
/// Names the facade uses for the two matrix types whose spelling differs between the built-in algebra and real Eigen.
template <class S>
using DenseMatrix = Eigen::DenseMatrix<S>;
template <class S>
using SparseView = Eigen::SparseMatrixCsr<S>;  // row-major compressed view over (row starts, column indices, values)
template <class S>
inline SparseView<S> MakeSparseView(std::ptrdiff_t rows, std::ptrdiff_t cols, std::ptrdiff_t /*nnz*/, const int* starts, const int* indices, const S* values) {
    return SparseView<S>{rows, cols, starts, indices, values};
}
}  // namespace Ungar::Linalg

#endif  // UNGAR_AMD_USE_SYSTEM_EIGEN
