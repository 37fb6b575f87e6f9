// ungar_amd :: straight-line code emission from the expression tape.
//
// The reference lowers its tapes to C through CppADCodeGen and compiles them at run time with gcc
// (include/ungar/autodiff/function.hpp:468-503).  Here a tape is lowered to the body of a HIP
// device function, generic over an I/O policy (`io.x(i)`, `io.f(i, v)`, `io.j(k, r, c, v)`, ...)
// so that the hand-written kernel skeletons (csrc/kernels/node_kernel.hpp) decide how a wavefront
// maps onto shooting nodes, how operands are staged and how results are stored.
//
// A second dialect emits plain C with the reference's calling style (one instance per call); it
// exists ONLY to build the CPU baseline / checker under oracle/_gen (see oracle/README.md) and is
// never linked into the product library.
#pragma once

#include <cstdio>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "derive.hpp"

namespace ungar_amd::tape {

/// One value the generated code must deliver: `sink` is the store statement with "%s" standing
/// for the value expression.
struct OutputSlot {
    Id value;
    std::string sink;
};

/// How input `i` of the tape is spelled in the generated code.
using InputNamer = std::string (*)(int index, const void* ctx);

struct EmitStats {
    std::size_t statements = 0, flops = 0, transcendentals = 0, divisions = 0;
};

class Emitter {
  public:
    Emitter(const Graph& g, std::vector<std::string> inputExpr) : g_{g}, inputExpr_{std::move(inputExpr)} {
    }

    /// Emits statements computing every slot, in slot order; each value's not-yet-emitted
    /// dependencies are emitted depth-first immediately before its first use, which keeps live
    /// ranges short (matters on the GPU where the "stack" is the VGPR file).
    std::string Emit(const std::vector<OutputSlot>& slots, const char* indent = "    ") {
        std::ostringstream os;
        name_.assign(g_.Size(), -1);
        for (const OutputSlot& s : slots) {
            EmitNode(s.value, os, indent);
            char buf[64];
            const std::string v = Ref(s.value);
            std::string line = s.sink;
            const std::size_t pos = line.find("%s");
            if (pos != std::string::npos) line.replace(pos, 2, v);
            (void)buf;
            os << indent << line << "\n";
        }
        return os.str();
    }

    const EmitStats& Stats() const {
        return stats_;
    }

  private:
    static std::string Lit(double v) {
        char buf[64];
        if (v == static_cast<double>(static_cast<long long>(v)) && std::fabs(v) < 1e15) {
            std::snprintf(buf, sizeof buf, "%.1f", v);
        } else {
            std::snprintf(buf, sizeof buf, "%.17g", v);
        }
        std::string s{buf};
        if (v < 0) s = "(" + s + ")";
        return s;
    }

    std::string Ref(Id id) const {
        const Node& nd = g_.At(id);
        if (nd.op == Op::Const) return Lit(nd.value);
        if (nd.op == Op::Input) return inputExpr_[static_cast<std::size_t>(nd.a)];
        return "v" + std::to_string(name_[static_cast<std::size_t>(id)]);
    }

    void EmitNode(Id root, std::ostringstream& os, const char* indent) {
        // Iterative post-order DFS (tapes are deep: thousands of chained adds).
        std::vector<std::pair<Id, int>> stack{{root, 0}};
        while (!stack.empty()) {
            auto& [id, state] = stack.back();
            const Node& nd = g_.At(id);
            if (Arity(nd.op) == 0 || name_[static_cast<std::size_t>(id)] >= 0) {
                stack.pop_back();
                continue;
            }
            const Id ops[4] = {nd.a, nd.b, nd.c, nd.d};
            bool pushed = false;
            while (state < 4) {
                const Id o = ops[state++];
                if (o == kNoId) continue;
                const Node& no = g_.At(o);
                if (Arity(no.op) != 0 && name_[static_cast<std::size_t>(o)] < 0) {
                    stack.emplace_back(o, 0);
                    pushed = true;
                    break;
                }
            }
            if (pushed) continue;
            const Id me = id;
            stack.pop_back();
            name_[static_cast<std::size_t>(me)] = next_++;
            os << indent << "const double v" << name_[static_cast<std::size_t>(me)] << " = " << Expr(me) << ";\n";
            ++stats_.statements;
        }
    }

    std::string Expr(Id id) {
        const Node& nd = g_.At(id);
        auto A = [&] { return Ref(nd.a); };
        auto B = [&] { return Ref(nd.b); };
        switch (nd.op) {
            case Op::Add: ++stats_.flops; return A() + " + " + B();
            case Op::Sub: ++stats_.flops; return A() + " - " + B();
            case Op::Mul: ++stats_.flops; return A() + " * " + B();
            case Op::Div: ++stats_.flops; ++stats_.divisions; return A() + " / " + B();
            case Op::Neg: return "-" + A();
            case Op::Sin: ++stats_.transcendentals; return "sin(" + A() + ")";
            case Op::Cos: ++stats_.transcendentals; return "cos(" + A() + ")";
            case Op::Tan: ++stats_.transcendentals; return "tan(" + A() + ")";
            case Op::Asin: ++stats_.transcendentals; return "asin(" + A() + ")";
            case Op::Acos: ++stats_.transcendentals; return "acos(" + A() + ")";
            case Op::Atan: ++stats_.transcendentals; return "atan(" + A() + ")";
            case Op::Exp: ++stats_.transcendentals; return "exp(" + A() + ")";
            case Op::Log: ++stats_.transcendentals; return "log(" + A() + ")";
            case Op::Sqrt: ++stats_.transcendentals; return "sqrt(" + A() + ")";
            case Op::Abs: return "fabs(" + A() + ")";
            case Op::Sign: return "(double)((" + A() + " > 0.0) - (" + A() + " < 0.0))";
            case Op::Pow: ++stats_.transcendentals; return "pow(" + A() + ", " + B() + ")";
            case Op::Atan2: ++stats_.transcendentals; return "atan2(" + A() + ", " + B() + ")";
            case Op::CondLt: return "(" + A() + " < " + B() + " ? " + Ref(nd.c) + " : " + Ref(nd.d) + ")";
            case Op::CondLe: return "(" + A() + " <= " + B() + " ? " + Ref(nd.c) + " : " + Ref(nd.d) + ")";
            case Op::CondEq: return "(" + A() + " == " + B() + " ? " + Ref(nd.c) + " : " + Ref(nd.d) + ")";
            case Op::CondGe: return "(" + A() + " >= " + B() + " ? " + Ref(nd.c) + " : " + Ref(nd.d) + ")";
            case Op::CondGt: return "(" + A() + " > " + B() + " ? " + Ref(nd.c) + " : " + Ref(nd.d) + ")";
            default: throw std::logic_error("Emitter: unexpected op");
        }
    }

    const Graph& g_;
    std::vector<std::string> inputExpr_;
    std::vector<int> name_;
    int next_ = 0;
    EmitStats stats_;
};

}  // namespace ungar_amd::tape
