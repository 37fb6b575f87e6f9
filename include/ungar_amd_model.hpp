// ungar_amd -- C++ adapter over the C ABI for applications that KEEP the reference's headers (real Eigen, Boost.Hana,
// CppAD recording) and swap only the object `Ungar::Autodiff::Function` owns.
//
// In the reference that object is `std::unique_ptr<CppAD::cg::GenericModel<real_t>> _model`
// (include/ungar/autodiff/function.hpp:364-365); Function reaches it through ForwardZero / SparseJacobian / SparseHessian /
// JacobianSparsity / HessianSparsity (:98-105, 135-145, 186-189, 224-228, 252-257).  `ungar_amd::AmdModel` offers the same
// calls on top of `ungar_function` (include/ungar_amd.h), and `ungar_amd::TapeBuilder` is what a walker over the CppAD
// operation sequence fills: one `ungar_tape_node` per CppAD operator, operands referring to earlier nodes
//     CppAD InvOp / ParOp            -> Input(i) / Constant(v)
//     AddvvOp SubvvOp MulvvOp DivvvOp (and the pv / vp forms with a Constant operand) -> Add Sub Mul Div
//     NegOp SinOp CosOp TanOp AsinOp AcosOp AtanOp ExpOp LogOp SqrtOp AbsOp SignOp   -> the unary ops of the same name
//     PowvvOp / PowvpOp / PowpvOp -> Pow;  atan2 (CppAD expands it)                 -> Atan2 where recorded directly
//     CExpOp with CompareLt / Le / Eq / Ge / Gt                                       -> CondLt / CondLe / CondEq / CondGe / CondGt
// Header-only, no Eigen, no HIP headers: it compiles wherever the C header does (tests/cpp/amd_model_test.cpp builds a tape by
// hand and checks values, Jacobian and Hessian against closed forms on the GPU, and the sparsity without one).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "ungar_amd.h"

namespace ungar_amd {

/// Op codes of `ungar_tape_node::op` (ungar_amd/csrc/tape/graph.hpp; listed in ungar_amd.h).
enum TapeOp : int32_t {
    kConst = 0, kInput = 1, kAdd = 2, kSub = 3, kMul = 4, kDiv = 5, kNeg = 6, kSin = 7, kCos = 8, kTan = 9, kAsin = 10, kAcos = 11, kAtan = 12, kExp = 13,
    kLog = 14, kSqrt = 15, kAbs = 16, kSign = 17, kPow = 18, kAtan2 = 19, kCondLt = 20, kCondLe = 21, kCondEq = 22, kCondGe = 23, kCondGt = 24
};

/// Topologically ordered expression tape; every method returns the index of the node it appended.
class TapeBuilder {
  public:
    int32_t Constant(double v) { return Push({kConst, -1, -1, -1, -1, 0, v}); }
    int32_t Input(int32_t index) { return Push({kInput, index, -1, -1, -1, 0, 0.0}); }  // index into [x; p]
    int32_t Unary(TapeOp op, int32_t a) { return Push({op, a, -1, -1, -1, 0, 0.0}); }
    int32_t Binary(TapeOp op, int32_t a, int32_t b) { return Push({op, a, b, -1, -1, 0, 0.0}); }
    /// (a cmp b) ? c : d
    int32_t Conditional(TapeOp op, int32_t a, int32_t b, int32_t c, int32_t d) { return Push({op, a, b, c, d, 0, 0.0}); }
    const std::vector<ungar_tape_node>& Nodes() const { return _nodes; }

  private:
    int32_t Push(const ungar_tape_node& n) {
        _nodes.push_back(n);
        return static_cast<int32_t>(_nodes.size() - 1);
    }
    std::vector<ungar_tape_node> _nodes;
};

/// What `Function` calls on its model object, on top of the C ABI.  enabled: UNGAR_ENABLE_* of ungar_amd.h.
class AmdModel {
  public:
    AmdModel(const std::vector<ungar_tape_node>& tape, const std::vector<int32_t>& outputs, int64_t n, int64_t p, const std::string& name, unsigned enabled,
             const std::string& folder = "", bool recompile = false) {
        if (ungar_function_make(tape.data(), static_cast<int64_t>(tape.size()), outputs.data(), static_cast<int64_t>(outputs.size()), n, p, name.c_str(), enabled,
                                folder.c_str(), recompile ? 1 : 0, &_fn) != UNGAR_OK)
            throw std::runtime_error(ungar_last_error());
        Check(ungar_function_get_info(_fn, &_info));
    }
    ~AmdModel() { ungar_function_free(_fn); }
    AmdModel(const AmdModel&) = delete;
    AmdModel& operator=(const AmdModel&) = delete;

    void ForwardZero(const double* xp, double* y) { Check(ungar_function_eval_host(_fn, 0, xp, y)); }             // function.hpp:186-189
    void SparseJacobian(const double* xp, double* values) { Check(ungar_function_eval_host(_fn, 1, xp, values)); }  // :224-228
    void SparseHessian(const double* xp, double* values) { Check(ungar_function_eval_host(_fn, 2, xp, values)); }   // :252-257
    void JacobianSparsity(const int32_t** rows, const int32_t** cols, int64_t* nnz) const { Check(ungar_function_jacobian_sparsity(_fn, rows, cols, nnz)); }  // :98-105
    void HessianSparsity(const int32_t** rows, const int32_t** cols, int64_t* nnz) const { Check(ungar_function_hessian_sparsity(_fn, rows, cols, nnz)); }    // :135-145
    const ungar_function_info& Info() const { return _info; }
    ungar_function* Handle() const { return _fn; }  // batched device entry points: ungar_function_forward_zero / _sparse_jacobian / _sparse_hessian

  private:
    static void Check(int rc) {
        if (rc != UNGAR_OK) throw std::runtime_error(ungar_last_error());
    }
    ungar_function* _fn = nullptr;
    ungar_function_info _info{};
};

}  // namespace ungar_amd
