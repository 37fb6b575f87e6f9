// ungar_amd :: Ungar::Autodiff::Function on the MI355X engine.
//
// Drop-in for reference include/ungar/autodiff/function.hpp: same Blueprint (:44-75), same member
// signatures (:180-361), same MakeFunction (:607-613), modelling the three optimisation concepts of
// include/ungar/optimization/concepts.hpp:38-76.  What differs is underneath: instead of a
// dlopen'ed CppADCodeGen library the object owns an `ungar_function` handle of the C ABI
// (include/ungar_amd.h); every evaluation runs a gfx950 kernel (single-instance host calls are
// batch-1 launches; batched device entry points are additive).  There is no CPU evaluation path.
#pragma once

#include <filesystem>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

#if __has_include(<ungar_amd.h>)  // installed layout: the C ABI header is on the include path
#include <ungar_amd.h>
#else  // in-tree layout: <repo>/include/ungar_amd.h next to <repo>/ungar_amd/include/ungar/
#include "../../../../include/ungar_amd.h"
#endif
#include "../utils/utils.hpp"
#include "data_types.hpp"

#ifndef UNGAR_CODEGEN_FOLDER
#define UNGAR_CODEGEN_FOLDER ""
#endif

namespace Ungar {
namespace Autodiff {

using SparseMatrix = Linalg::SparseView<real_t>;  // Eigen::Map<const Eigen::SparseMatrix<real_t, RowMajor>> on real Eigen (reference :217, :237)

class Function {
  public:
    struct Blueprint {
        Blueprint(const ADFunction& functionImpl_, const index_t independentVariableSize_, const index_t parameterSize_,
                  std::string_view name_, const EnabledDerivatives enabledDerivatives_ = EnabledDerivatives::ALL,
                  const std::filesystem::path& folder_ = UNGAR_CODEGEN_FOLDER)
            : independentVariableSize{independentVariableSize_},
              parameterSize{parameterSize_},
              // sizing run on un-recorded random values (reference function.hpp:53-58)
              dependentVariableSize{[&] {
                  VectorXad y;
                  functionImpl_(VectorXad::Random(independentVariableSize_ + parameterSize_), y);
                  return y.size();
              }()},
              name{name_},
              folder{folder_},
              enabledDerivatives{enabledDerivatives_},
              functionImpl{functionImpl_} {
        }
        index_t independentVariableSize, parameterSize, dependentVariableSize;
        std::string name;
        std::filesystem::path folder;
        EnabledDerivatives enabledDerivatives;
        ADFunction functionImpl;
    };

    Function(Function&&) = default;
    Function& operator=(Function&&) = default;

    template <class XP, class Y>
    void Evaluate(const Eigen::MatrixBase<XP>& xp, const Eigen::MatrixBase<Y>& y) const {
        CheckInput(xp.size());
        if (y.size() != _m) throw std::invalid_argument("Function::Evaluate: wrong output size");
        const auto& in = xp.derived().eval();  // contiguous storage (the reference reads xp.derived().data(), function.hpp:187)
        Check(ungar_function_eval_host(_fn.get(), 0, in.data(), y.const_cast_derived().data()));
    }
    template <class XP>
    VectorXr operator()(const Eigen::MatrixBase<XP>& xp) const {
        VectorXr y{_m};
        Evaluate(xp, y);
        return y;
    }
    /// m x n row-major sparse Jacobian over the decision variables (parameters trimmed); the returned
    /// reference aliases internal storage overwritten by the next call (reference :216-230, :380-381).
    template <class XP>
    const SparseMatrix& Jacobian(const Eigen::MatrixBase<XP>& xp) const {
        if (!ImplementsJacobian()) throw std::logic_error("Function::Jacobian: function was made without JACOBIAN");
        CheckInput(xp.size());
        const auto& in = xp.derived().eval();
        Check(ungar_function_eval_host(_fn.get(), 1, in.data(), _jacData.data()));
        return *_jac;
    }
    /// Upper-triangular n x n Hessian of dependent variable `i` (scalar functions only, :236-259).
    template <class XP>
    const SparseMatrix& Hessian(const index_t dependentVariableIndex, const Eigen::MatrixBase<XP>& xp) const {
        if (!ImplementsHessian()) throw std::logic_error("Function::Hessian: function was made without HESSIAN");
        if (dependentVariableIndex != 0 || _m != 1) throw std::logic_error("The Hessian is implemented only for scalar functions.");
        CheckInput(xp.size());
        const auto& in = xp.derived().eval();
        Check(ungar_function_eval_host(_fn.get(), 2, in.data(), _hesData.data()));
        return *_hes;
    }
    template <class XP>
    const SparseMatrix& Hessian(const Eigen::MatrixBase<XP>& xp) const {
        return Hessian(0, xp);
    }

    /// Self-checks of the reference (:276-337): value vs a ground-truth functor; derivatives vs
    /// second-order central differences, compared with Utils::CompareMatrices' loose criterion.
    template <class XP>
    [[nodiscard]] bool TestFunction(const Eigen::MatrixBase<XP>& xp, const std::function<VectorXr(const VectorXr&)>& groundTruth) const {
        const VectorXr y = (*this)(xp), g = groundTruth(VectorXr{xp});
        return y.size() == g.size() && Utils::CompareMatrices(y.data(), "Autodiff function", g.data(), "Ground truth", y.size());
    }
    template <class XP>
    [[nodiscard]] bool TestJacobian(const Eigen::MatrixBase<XP>& xp, const real_t epsilon = 1e-6) const {
        VectorXr z{xp};
        std::vector<real_t> fd(static_cast<std::size_t>(_m * _n));
        for (index_t j = 0; j < _n; ++j) {
            const real_t x0 = z[j];
            z[j] = x0 + epsilon;
            const VectorXr yp = (*this)(z);
            z[j] = x0 - epsilon;
            const VectorXr ym = (*this)(z);
            z[j] = x0;
            for (index_t i = 0; i < _m; ++i) fd[static_cast<std::size_t>(i * _n + j)] = (yp[i] - ym[i]) / (2 * epsilon);
        }
        const std::vector<real_t> ad = Linalg::ToDense(Jacobian(xp));
        return Utils::CompareMatrices(ad.data(), "Autodiff Jacobian", fd.data(), "FD Jacobian", _m * _n);
    }
    template <class XP>
    [[nodiscard]] bool TestHessian(const Eigen::MatrixBase<XP>& xp, const real_t epsilon = 1e-4) const {
        VectorXr z{xp};
        std::vector<real_t> fd(static_cast<std::size_t>(_n * _n), 0.0);
        auto f = [&](index_t a, real_t da, index_t b, real_t db) {
            const real_t xa = z[a], xb = z[b];
            z[a] += da;
            z[b] += db;
            const real_t v = (*this)(z)[0];
            z[a] = xa;
            z[b] = xb;
            if (a == b) z[a] = xa;
            return v;
        };
        for (index_t r = 0; r < _n; ++r)
            for (index_t c = r; c < _n; ++c)
                fd[static_cast<std::size_t>(r * _n + c)] =
                    (f(r, epsilon, c, epsilon) - f(r, epsilon, c, -epsilon) - f(r, -epsilon, c, epsilon) + f(r, -epsilon, c, -epsilon)) /
                    (4 * epsilon * epsilon);
        const std::vector<real_t> ad = Linalg::ToDense(Hessian(xp));  // upper triangle only
        return Utils::CompareMatrices(ad.data(), "Autodiff Hessian", fd.data(), "FD Hessian", _n * _n);
    }

    bool ImplementsFunction() const {
        return true;
    }
    bool ImplementsJacobian() const {
        return _hasJac;
    }
    bool ImplementsHessian() const {
        return _hasHes;
    }
    index_t IndependentVariableSize() const {
        return _n;
    }
    index_t ParameterSize() const {
        return _p;
    }
    index_t DependentVariableSize() const {
        return _m;
    }

    /// Additive, MI355X-specific: the C-ABI handle for batched, stream-ordered device evaluation
    /// (ungar_function_forward_zero / _sparse_jacobian / _sparse_hessian).
    ungar_function* Handle() const {
        return _fn.get();
    }
    bool LoadedFromCache() const {
        return _cacheHit;
    }

  private:
    friend class FunctionFactory;
    struct Deleter {
        void operator()(ungar_function* f) const {
            ungar_function_free(f);
        }
    };
    Function() = default;
    static void Check(int code) {
        if (code != UNGAR_OK) throw std::runtime_error(std::string("ungar_amd: ") + ungar_last_error());
    }
    void CheckInput(index_t size) const {
        if (size != _n + _p) throw std::invalid_argument("Function: xp must hold independent variables followed by parameters");
    }
    void BuildCsr(const int32_t* rows, const int32_t* cols, int64_t nnz, index_t nRows, std::vector<int>& starts, std::vector<int>& idx) {
        starts.assign(static_cast<std::size_t>(nRows) + 1, 0);
        idx.assign(cols, cols + nnz);
        for (int64_t k = 0; k < nnz; ++k) ++starts[static_cast<std::size_t>(rows[k]) + 1];
        for (index_t r = 0; r < nRows; ++r) starts[static_cast<std::size_t>(r) + 1] += starts[static_cast<std::size_t>(r)];
    }

    std::unique_ptr<ungar_function, Deleter> _fn;
    index_t _n = 0, _p = 0, _m = 0;
    bool _hasJac = false, _hasHes = false, _cacheHit = false;
    std::vector<int> _jacStarts, _jacIdx, _hesStarts, _hesIdx;
    mutable std::vector<real_t> _jacData, _hesData;
    std::unique_ptr<SparseMatrix> _jac, _hes;  // views over the index / value arrays above (an Eigen::Map is not assignable)
};

class FunctionFactory {
  public:
    static Function Make(const Function::Blueprint& bp, const bool recompileLibraries, const std::vector<std::string>& compilerFlags) {
        namespace tape = ::ungar_amd::tape;
        // The reference forwards `compilerFlags` to gcc (function.hpp:475-480); kernels here are built by hipcc for gfx950
        // and gcc options such as -march=native do not apply.  Anything other than the reference's default set is reported
        // once instead of being dropped silently; UNGAR_AMD_JIT_FLAGS replaces the device compiler's optimisation flags.
        static bool warned = false;
        static const std::vector<std::string> kReferenceDefaults{"-O3", "-g", "-march=native", "-mtune=native", "-ffast-math"};
        if (!warned && compilerFlags != kReferenceDefaults) {
            warned = true;
            UNGAR_LOG(warn, "Autodiff::MakeFunction: compilerFlags are host-compiler (gcc) options and do not apply to gfx950 kernels; ignored. "
                            "Set UNGAR_AMD_JIT_FLAGS to change the hipcc optimisation flags.");
        }
        const index_t nIn = bp.independentVariableSize + bp.parameterSize;
        // record (reference CreateModelsImpl, function.hpp:453-466)
        const std::vector<tape::AD> in = tape::Independent(static_cast<int>(nIn));
        VectorXad xp{nIn};
        for (index_t i = 0; i < nIn; ++i) xp[i] = in[static_cast<std::size_t>(i)];
        VectorXad y;
        bp.functionImpl(xp, y);
        if (y.size() == 0 || y.size() != bp.dependentVariableSize) throw std::logic_error("Function: the AD function changed its output size");
        std::vector<int32_t> outputs(static_cast<std::size_t>(y.size()));
        for (index_t i = 0; i < y.size(); ++i) outputs[static_cast<std::size_t>(i)] = y[i].Node();
        const tape::Graph& g = tape::CurrentGraph();
        std::vector<ungar_tape_node> nodes(g.Size());
        for (std::size_t i = 0; i < g.Size(); ++i) {
            const tape::Node& nd = g.At(static_cast<tape::Id>(i));
            nodes[i] = {static_cast<int32_t>(nd.op), nd.a, nd.b, nd.c, nd.d, 0, nd.value};
        }
        const std::string folder = bp.folder.string();
        return MakeFromTape(nodes, outputs, bp.independentVariableSize, bp.parameterSize, bp.name, bp.enabledDerivatives, folder, recompileLibraries);
    }

    /// A function from a tape that is already recorded (C-ABI node format): what Make ends in, and how several functions over shared inputs become ONE
    /// (ungar_function_get_tape; ungar/optimization/batched_soft_sqp.hpp evaluates all stage values of a shooting problem in one launch that way).
    static Function MakeFromTape(const std::vector<ungar_tape_node>& nodes, const std::vector<int32_t>& outputs, const index_t independentVariableSize,
                                 const index_t parameterSize, const std::string& name, const EnabledDerivatives enabledDerivatives, const std::string& folder,
                                 const bool recompileLibraries = false) {
        struct {
            index_t independentVariableSize, parameterSize;
            EnabledDerivatives enabledDerivatives;
        } bp{independentVariableSize, parameterSize, enabledDerivatives};
        struct {
            index_t n;
            index_t size() const { return n; }
        } y{static_cast<index_t>(outputs.size())};
        ungar_function* raw = nullptr;
        Function::Check(ungar_function_make(nodes.data(), static_cast<int64_t>(nodes.size()), outputs.data(), y.size(), bp.independentVariableSize,
                                            bp.parameterSize, name.c_str(), static_cast<uint32_t>(bp.enabledDerivatives), folder.c_str(),
                                            recompileLibraries ? 1 : 0, &raw));
        Function f;
        f._fn.reset(raw);
        f._n = bp.independentVariableSize;
        f._p = bp.parameterSize;
        f._m = y.size();
        ungar_function_info info{};
        Function::Check(ungar_function_get_info(raw, &info));
        f._cacheHit = info.cache_hit != 0;
        f._hasJac = bp.enabledDerivatives & EnabledDerivatives::JACOBIAN;
        f._hasHes = bp.enabledDerivatives & EnabledDerivatives::HESSIAN;
        const int32_t *rows = nullptr, *cols = nullptr;
        int64_t nnz = 0;
        if (f._hasJac) {
            Function::Check(ungar_function_jacobian_sparsity(raw, &rows, &cols, &nnz));
            f.BuildCsr(rows, cols, nnz, f._m, f._jacStarts, f._jacIdx);
            f._jacData.assign(static_cast<std::size_t>(nnz), 0.0);
            f._jac = std::make_unique<SparseMatrix>(Linalg::MakeSparseView<real_t>(f._m, f._n, nnz, f._jacStarts.data(), f._jacIdx.data(), f._jacData.data()));
        }
        if (f._hasHes) {
            Function::Check(ungar_function_hessian_sparsity(raw, &rows, &cols, &nnz));
            f.BuildCsr(rows, cols, nnz, f._n, f._hesStarts, f._hesIdx);
            f._hesData.assign(static_cast<std::size_t>(nnz), 0.0);
            f._hes = std::make_unique<SparseMatrix>(Linalg::MakeSparseView<real_t>(f._n, f._n, nnz, f._hesStarts.data(), f._hesIdx.data(), f._hesData.data()));
        }
        return f;
    }
};

/// The reference's default flags are gcc's (function.hpp:610-611); here kernels are compiled by
/// hipcc for gfx950 with fixed flags, so `compilerFlags` is accepted for source compatibility only.
inline Function MakeFunction(const Function::Blueprint& blueprint, const bool recompileLibraries = false,
                             std::vector<std::string> compilerFlags = {"-O3", "-g", "-march=native", "-mtune=native", "-ffast-math"}) {
    return FunctionFactory::Make(blueprint, recompileLibraries, compilerFlags);
}

}  // namespace Autodiff
}  // namespace Ungar
