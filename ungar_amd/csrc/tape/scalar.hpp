// ungar_amd :: AD scalar recorded into the expression tape.
//
// Mirrors the slice of CppAD::AD<CG<double>> the reference uses (SURVEY.md Appendix B;
// include/ungar/autodiff/data_types.hpp:39-41, include/ungar/utils/utils.hpp:820-852, 962-1059,
// include/ungar/autodiff/support/quaternion.hpp:34-192): arithmetic, elementary functions,
// pow(x,int) by repeated multiplication, CondExp*, and Value() on constants.  A scalar is either a
// *literal* (plain double, nothing recorded -- what VectorXad::Random yields in
// Function::Blueprint's sizing run, function.hpp:53-58) or a handle to a DAG node.
#pragma once

#include <type_traits>

#include "graph.hpp"

namespace ungar_amd::tape {

class AD {
  public:
    constexpr AD() = default;
    template <typename T, std::enable_if_t<std::is_arithmetic_v<T>, int> = 0>
    constexpr AD(T v) : val_{static_cast<double>(v)} {  // NOLINT(google-explicit-constructor)
    }

    static AD FromId(Id id) {
        Graph& g = CurrentGraph();
        if (g.IsConst(id)) return AD{g.ConstValue(id)};
        AD x;
        x.id_ = id;
        return x;
    }

    bool IsLiteral() const {
        return id_ == kNoId;
    }
    /// Value of a constant (CppAD::Value(x).getValue(), utils.hpp:1049).  Throws on variables.
    double Literal() const {
        if (!IsLiteral()) throw std::logic_error("ungar_amd::tape::AD: value of a recorded variable requested");
        return val_;
    }
    Id Node() const {
        return IsLiteral() ? CurrentGraph().Constant(val_) : id_;
    }

    AD operator-() const {
        if (IsLiteral()) return AD{-val_};
        return FromId(CurrentGraph().Unary(Op::Neg, id_));
    }
    AD operator+() const {
        return *this;
    }

    AD& operator+=(const AD& o) {
        return *this = Bin(Op::Add, *this, o);
    }
    AD& operator-=(const AD& o) {
        return *this = Bin(Op::Sub, *this, o);
    }
    AD& operator*=(const AD& o) {
        return *this = Bin(Op::Mul, *this, o);
    }
    AD& operator/=(const AD& o) {
        return *this = Bin(Op::Div, *this, o);
    }

    friend AD operator+(const AD& a, const AD& b) {
        return Bin(Op::Add, a, b);
    }
    friend AD operator-(const AD& a, const AD& b) {
        return Bin(Op::Sub, a, b);
    }
    friend AD operator*(const AD& a, const AD& b) {
        return Bin(Op::Mul, a, b);
    }
    friend AD operator/(const AD& a, const AD& b) {
        return Bin(Op::Div, a, b);
    }

    // Comparisons are only defined between constants (CppAD would record a compare op that the
    // reference discards with optimize("no_compare_op"), function.hpp:466).  Use CondExp* on
    // variables.
    friend bool operator<(const AD& a, const AD& b) {
        return a.Literal() < b.Literal();
    }
    friend bool operator>(const AD& a, const AD& b) {
        return a.Literal() > b.Literal();
    }
    friend bool operator<=(const AD& a, const AD& b) {
        return a.Literal() <= b.Literal();
    }
    friend bool operator>=(const AD& a, const AD& b) {
        return a.Literal() >= b.Literal();
    }
    friend bool operator==(const AD& a, const AD& b) {
        if (!a.IsLiteral() && !b.IsLiteral()) return a.id_ == b.id_;
        if (a.IsLiteral() != b.IsLiteral()) return false;
        return a.val_ == b.val_;
    }
    friend bool operator!=(const AD& a, const AD& b) {
        return !(a == b);
    }

    static AD Un(Op op, const AD& a) {
        if (a.IsLiteral()) return AD{EvalUnary(op, a.val_)};
        return FromId(CurrentGraph().Unary(op, a.id_));
    }
    static AD Bin(Op op, const AD& a, const AD& b) {
        if (a.IsLiteral() && b.IsLiteral()) return AD{EvalBinary(op, a.val_, b.val_)};
        return FromId(CurrentGraph().Binary(op, a.Node(), b.Node()));
    }
    static AD Cond(Op op, const AD& a, const AD& b, const AD& c, const AD& d) {
        if (a.IsLiteral() && b.IsLiteral()) return EvalCompare(op, a.val_, b.val_) ? c : d;
        return FromId(CurrentGraph().Cond(op, a.Node(), b.Node(), c.Node(), d.Node()));
    }

  private:
    Id id_ = kNoId;
    double val_ = 0.0;
};

inline AD sin(const AD& a) { return AD::Un(Op::Sin, a); }
inline AD cos(const AD& a) { return AD::Un(Op::Cos, a); }
inline AD tan(const AD& a) { return AD::Un(Op::Tan, a); }
inline AD asin(const AD& a) { return AD::Un(Op::Asin, a); }
inline AD acos(const AD& a) { return AD::Un(Op::Acos, a); }
inline AD atan(const AD& a) { return AD::Un(Op::Atan, a); }
inline AD exp(const AD& a) { return AD::Un(Op::Exp, a); }
inline AD log(const AD& a) { return AD::Un(Op::Log, a); }
inline AD sqrt(const AD& a) { return AD::Un(Op::Sqrt, a); }
inline AD abs(const AD& a) { return AD::Un(Op::Abs, a); }
inline AD fabs(const AD& a) { return AD::Un(Op::Abs, a); }
inline AD sign(const AD& a) { return AD::Un(Op::Sign, a); }
inline AD atan2(const AD& y, const AD& x) { return AD::Bin(Op::Atan2, y, x); }
inline AD pow(const AD& x, const AD& y) { return AD::Bin(Op::Pow, x, y); }
inline AD pow(const AD& x, double y) { return AD::Bin(Op::Pow, x, AD{y}); }
inline AD pow(double x, const AD& y) { return AD::Bin(Op::Pow, AD{x}, y); }

/// pow(x, int): repeated multiplication (division for negative exponents), the scheme CppAD uses
/// for integer exponents and what utils.hpp:827-829 reaches through `CppAD::pow(base, int)`.
inline AD pow(const AD& x, int n) {
    AD p{1.0};
    const int m = n < 0 ? -n : n;
    for (int i = 0; i < m; ++i) p = p * x;
    return n < 0 ? AD{1.0} / p : p;
}

inline AD CondExpLt(const AD& a, const AD& b, const AD& t, const AD& f) { return AD::Cond(Op::CondLt, a, b, t, f); }
inline AD CondExpLe(const AD& a, const AD& b, const AD& t, const AD& f) { return AD::Cond(Op::CondLe, a, b, t, f); }
inline AD CondExpEq(const AD& a, const AD& b, const AD& t, const AD& f) { return AD::Cond(Op::CondEq, a, b, t, f); }
inline AD CondExpGe(const AD& a, const AD& b, const AD& t, const AD& f) { return AD::Cond(Op::CondGe, a, b, t, f); }
inline AD CondExpGt(const AD& a, const AD& b, const AD& t, const AD& f) { return AD::Cond(Op::CondGt, a, b, t, f); }

inline AD QuadSum(const AD& a) { return AD::Un(Op::QuadSum, a); }
inline AD QuadRot(const AD& a, int r) { return r == 0 ? a : AD::Un(r == 1 ? Op::QuadRot1 : r == 2 ? Op::QuadRot2 : Op::QuadRot3, a); }

inline double Value(const AD& a) { return a.Literal(); }

}  // namespace ungar_amd::tape
