// ungar_amd :: expression tape (recorder half of the CppAD / CppADCodeGen replacement).
//
// What it replaces in the reference:
//   CppAD::Independent / ADFun / optimize("no_compare_op")    include/ungar/autodiff/function.hpp:456-466
//   CppAD::AD<CG<double>> scalar                               include/ungar/autodiff/data_types.hpp:39-41
//   CondExp*, pow(int), sqrt, abs, atan2                       include/ungar/utils/utils.hpp:820-852, 962-1015
//
// Design (MI355X-first, not a CppAD clone): the tape is a hash-consed scalar expression DAG that is
// *built already optimised* (constant folding, algebraic identities, common-subexpression sharing),
// because its only consumers are source-to-source transforms (derive.hpp) and straight-line code
// emitters (emit.hpp) that lower it to one HIP kernel body per model.  There is no interpreter and
// no CPU evaluation path: values are only ever produced by the generated HIP code.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace ungar_amd::tape {

using Id = std::int32_t;
inline constexpr Id kNoId = -1;

enum class Op : std::uint8_t {
    Const,
    Input,
    Add,
    Sub,
    Mul,
    Div,
    Neg,
    Sin,
    Cos,
    Tan,
    Asin,
    Acos,
    Atan,
    Exp,
    Log,
    Sqrt,
    Abs,
    Sign,
    Pow,
    Atan2,
    // (a cmp b) ? c : d   -- CppAD::CondExp{Lt,Le,Eq,Ge,Gt}
    CondLt,
    CondLe,
    CondEq,
    CondGe,
    CondGt,
    // SPMD communication inside a quad of 4 adjacent lanes (one lane per leg of a quadruped): opaque to
    // the derivative transforms, emitted as io.quad_sum / io.quad_rotN (DPP quad_perm on the GPU).
    QuadSum,   // every lane receives the sum over the 4 lanes of its quad
    QuadRot1,  // lane l receives the value of lane (l + 1) & 3 of its quad
    QuadRot2,
    QuadRot3,
    // a * b + c with ONE rounding.  Never recorded and never differentiated: FuseMultiplyAdd (emit.hpp) rewrites a finished program into it, so that which products
    // are contracted is decided by the generated source and not by the compiler (the same program then gives the same bits in every kernel it is compiled into).
    Fma,
};

inline bool IsQuad(Op op) {
    return op >= Op::QuadSum && op <= Op::QuadRot3;
}
inline bool IsCond(Op op) {
    return op >= Op::CondLt && op <= Op::CondGt;
}
inline bool IsUnary(Op op) {
    return (op >= Op::Neg && op <= Op::Sign) || IsQuad(op);
}
inline bool IsBinary(Op op) {
    return (op >= Op::Add && op <= Op::Div) || op == Op::Pow || op == Op::Atan2;
}
inline int Arity(Op op) {
    if (op == Op::Const || op == Op::Input) return 0;
    if (IsUnary(op)) return 1;
    if (IsBinary(op)) return 2;
    if (op == Op::Fma) return 3;
    return 4;
}

struct Node {
    Op op = Op::Const;
    Id a = kNoId, b = kNoId, c = kNoId, d = kNoId;
    double value = 0.0;  // Const: the constant.  Input: unused (index is in `a`).
};

inline double EvalUnary(Op op, double a) {
    switch (op) {
        case Op::Neg: return -a;
        case Op::Sin: return std::sin(a);
        case Op::Cos: return std::cos(a);
        case Op::Tan: return std::tan(a);
        case Op::Asin: return std::asin(a);
        case Op::Acos: return std::acos(a);
        case Op::Atan: return std::atan(a);
        case Op::Exp: return std::exp(a);
        case Op::Log: return std::log(a);
        case Op::Sqrt: return std::sqrt(a);
        case Op::Abs: return std::fabs(a);
        case Op::Sign: return static_cast<double>(a > 0.0) - static_cast<double>(a < 0.0);
        case Op::QuadSum: return 4.0 * a;  // a constant is the same in all four lanes
        case Op::QuadRot1:
        case Op::QuadRot2:
        case Op::QuadRot3: return a;
        default: throw std::logic_error("EvalUnary: not a unary op");
    }
}

inline double EvalBinary(Op op, double a, double b) {
    switch (op) {
        case Op::Add: return a + b;
        case Op::Sub: return a - b;
        case Op::Mul: return a * b;
        case Op::Div: return a / b;
        case Op::Pow: return std::pow(a, b);
        case Op::Atan2: return std::atan2(a, b);
        default: throw std::logic_error("EvalBinary: not a binary op");
    }
}

inline bool EvalCompare(Op op, double a, double b) {
    switch (op) {
        case Op::CondLt: return a < b;
        case Op::CondLe: return a <= b;
        case Op::CondEq: return a == b;
        case Op::CondGe: return a >= b;
        case Op::CondGt: return a > b;
        default: throw std::logic_error("EvalCompare: not a conditional op");
    }
}

/// Hash-consed expression DAG.  Node ids are creation-ordered, hence always topologically sorted:
/// a node only references smaller ids.
class Graph {
  public:
    Graph() {
        nodes_.reserve(1 << 16);
    }

    void Clear() {
        nodes_.clear();
        cse_.clear();
        constants_.clear();
        numInputs_ = 0;
    }

    std::size_t Size() const {
        return nodes_.size();
    }
    const Node& At(Id id) const {
        return nodes_[static_cast<std::size_t>(id)];
    }
    const std::vector<Node>& Nodes() const {
        return nodes_;
    }
    int NumInputs() const {
        return numInputs_;
    }

    bool IsConst(Id id) const {
        return At(id).op == Op::Const;
    }
    bool IsConst(Id id, double v) const {
        return At(id).op == Op::Const && At(id).value == v;
    }
    double ConstValue(Id id) const {
        return At(id).value;
    }

    Id Constant(double v) {
        if (v == 0.0) v = 0.0;  // collapse -0.0 onto +0.0 so that "is zero" is a bit test
        std::uint64_t bits;
        std::memcpy(&bits, &v, sizeof bits);
        auto [it, inserted] = constants_.try_emplace(bits, static_cast<Id>(nodes_.size()));
        if (inserted) {
            Node n;
            n.op = Op::Const;
            n.value = v;
            nodes_.push_back(n);
        }
        return it->second;
    }

    /// Declares the next independent variable (index = order of declaration).
    Id Input() {
        Node n;
        n.op = Op::Input;
        n.a = numInputs_++;
        nodes_.push_back(n);
        return static_cast<Id>(nodes_.size() - 1);
    }

    Id Unary(Op op, Id a) {
        const Node na = At(a);
        if (na.op == Op::Const) return Constant(EvalUnary(op, na.value));
        switch (op) {
            case Op::Neg:
                if (na.op == Op::Neg) return na.a;
                if (na.op == Op::Sub) return Binary(Op::Sub, na.b, na.a);
                break;
            case Op::Abs:
                if (na.op == Op::Abs) return a;
                if (na.op == Op::Neg) return Unary(Op::Abs, na.a);
                break;
            case Op::Cos:
                if (na.op == Op::Neg) return Unary(Op::Cos, na.a);
                break;
            case Op::Sin:
                if (na.op == Op::Neg) return Unary(Op::Neg, Unary(Op::Sin, na.a));
                break;
            default: break;
        }
        return Intern(op, a, kNoId, kNoId, kNoId);
    }

    Id Binary(Op op, Id a, Id b) {
        const Node na = At(a);
        const Node nb = At(b);
        const bool ca = na.op == Op::Const, cb = nb.op == Op::Const;
        if (ca && cb) return Constant(EvalBinary(op, na.value, nb.value));
        switch (op) {
            case Op::Add:
                if (ca && na.value == 0.0) return b;
                if (cb && nb.value == 0.0) return a;
                if (nb.op == Op::Neg) return Binary(Op::Sub, a, nb.a);
                if (na.op == Op::Neg) return Binary(Op::Sub, b, na.a);
                if (a > b) std::swap(a, b);
                break;
            case Op::Sub:
                if (cb && nb.value == 0.0) return a;
                if (ca && na.value == 0.0) return Unary(Op::Neg, b);
                if (a == b) return Constant(0.0);
                if (nb.op == Op::Neg) return Binary(Op::Add, a, nb.a);
                break;
            case Op::Mul:
                if ((ca && na.value == 0.0) || (cb && nb.value == 0.0)) return Constant(0.0);
                if (ca && na.value == 1.0) return b;
                if (cb && nb.value == 1.0) return a;
                if (ca && na.value == -1.0) return Unary(Op::Neg, b);
                if (cb && nb.value == -1.0) return Unary(Op::Neg, a);
                if (na.op == Op::Neg && nb.op == Op::Neg) return Binary(Op::Mul, na.a, nb.a);
                if (na.op == Op::Neg) return Unary(Op::Neg, Binary(Op::Mul, na.a, b));
                if (nb.op == Op::Neg) return Unary(Op::Neg, Binary(Op::Mul, a, nb.a));
                if (a > b) std::swap(a, b);
                break;
            case Op::Div:
                if (ca && na.value == 0.0) return Constant(0.0);
                if (cb && nb.value == 1.0) return a;
                if (cb && nb.value == -1.0) return Unary(Op::Neg, a);
                if (a == b) return Constant(1.0);
                if (na.op == Op::Neg && nb.op == Op::Neg) return Binary(Op::Div, na.a, nb.a);
                if (na.op == Op::Neg) return Unary(Op::Neg, Binary(Op::Div, na.a, b));
                if (nb.op == Op::Neg) return Unary(Op::Neg, Binary(Op::Div, a, nb.a));
                break;
            case Op::Pow:
                if (cb && nb.value == 1.0) return a;
                if (cb && nb.value == 0.0) return Constant(1.0);
                break;
            default: break;
        }
        return Intern(op, a, b, kNoId, kNoId);
    }

    /// a * b + c, fused (no algebraic rewriting: the caller decides what is contracted).
    Id Fma(Id a, Id b, Id c) {
        if (a > b) std::swap(a, b);
        return Intern(Op::Fma, a, b, c, kNoId);
    }

    Id Cond(Op op, Id a, Id b, Id c, Id d) {
        if (c == d) return c;
        if (IsConst(a) && IsConst(b)) return EvalCompare(op, ConstValue(a), ConstValue(b)) ? c : d;
        return Intern(op, a, b, c, d);
    }

    // Convenience wrappers used by the derivative transforms.
    Id Add(Id a, Id b) {
        return Binary(Op::Add, a, b);
    }
    Id Sub(Id a, Id b) {
        return Binary(Op::Sub, a, b);
    }
    Id Mul(Id a, Id b) {
        return Binary(Op::Mul, a, b);
    }
    Id Div(Id a, Id b) {
        return Binary(Op::Div, a, b);
    }
    Id Neg(Id a) {
        return Unary(Op::Neg, a);
    }

  private:
    struct Key {
        std::uint64_t k0, k1;
        bool operator==(const Key& o) const {
            return k0 == o.k0 && k1 == o.k1;
        }
    };
    struct KeyHash {
        std::size_t operator()(const Key& k) const {
            std::uint64_t h = k.k0 * 0x9E3779B97F4A7C15ULL;
            h ^= (k.k1 + 0xC2B2AE3D27D4EB4FULL + (h << 6) + (h >> 2));
            return static_cast<std::size_t>(h ^ (h >> 29));
        }
    };

    Id Intern(Op op, Id a, Id b, Id c, Id d) {
        const Key key{(static_cast<std::uint64_t>(static_cast<std::uint32_t>(a)) << 32) |
                          static_cast<std::uint32_t>(b),
                      (static_cast<std::uint64_t>(static_cast<std::uint32_t>(c)) << 32) ^
                          (static_cast<std::uint64_t>(static_cast<std::uint32_t>(d)) << 8) ^
                          static_cast<std::uint64_t>(op)};
        // k1 packs (c,d,op); collisions across distinct (c,d) pairs are resolved by the full check.
        auto range = cse_.equal_range(key);
        for (auto it = range.first; it != range.second; ++it) {
            const Node& n = At(it->second);
            if (n.op == op && n.a == a && n.b == b && n.c == c && n.d == d) return it->second;
        }
        Node n;
        n.op = op;
        n.a = a;
        n.b = b;
        n.c = c;
        n.d = d;
        nodes_.push_back(n);
        const Id id = static_cast<Id>(nodes_.size() - 1);
        cse_.emplace(key, id);
        return id;
    }

    std::vector<Node> nodes_;
    std::unordered_multimap<Key, Id, KeyHash> cse_;
    std::unordered_map<std::uint64_t, Id> constants_;
    int numInputs_ = 0;
};

/// The graph the AD scalar records into.  One per host thread: recording, like the reference's
/// (function.hpp:377 "not thread-safe"), is a single-threaded affair per Function.
inline Graph& CurrentGraph() {
    thread_local Graph graph;
    return graph;
}

}  // namespace ungar_amd::tape
