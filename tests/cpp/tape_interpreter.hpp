// TEST INFRASTRUCTURE: host interpreter of the expression DAG (ungar_amd/csrc/tape/graph.hpp).  The product has no CPU
// evaluation path -- values are only ever produced by generated HIP code; this exists so that recorded programs and
// their derivative transforms can be pinned against closed forms without a GPU (tests/cpp/tape_test.cpp,
// tests/cpp/helpers_test.cpp).
#pragma once

#include <vector>

#include "../../ungar_amd/csrc/tape/derive.hpp"

namespace ungar_amd::tape {

/// Evaluates every node of `g` for the input vector `in` (test infrastructure, not a product path).
inline std::vector<double> Interpret(const Graph& g, const std::vector<double>& in) {
    std::vector<double> v(g.Size(), 0.0);
    for (std::size_t i = 0; i < g.Size(); ++i) {
        const Node& n = g.At(static_cast<Id>(i));
        switch (Arity(n.op)) {
            case 0: v[i] = n.op == Op::Const ? n.value : in[static_cast<std::size_t>(n.a)]; break;
            case 1: v[i] = EvalUnary(n.op, v[static_cast<std::size_t>(n.a)]); break;
            case 2: v[i] = EvalBinary(n.op, v[static_cast<std::size_t>(n.a)], v[static_cast<std::size_t>(n.b)]); break;
            default:
                v[i] = EvalCompare(n.op, v[static_cast<std::size_t>(n.a)], v[static_cast<std::size_t>(n.b)]) ? v[static_cast<std::size_t>(n.c)]
                                                                                                            : v[static_cast<std::size_t>(n.d)];
        }
    }
    return v;
}


}  // namespace ungar_amd::tape
