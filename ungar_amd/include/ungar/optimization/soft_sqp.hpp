// ungar_amd :: soft sequential quadratic programming -- the caller of the derivative hot path
// (SURVEY.md section 8(f) N1/N2).  Same interface and iteration as the reference's SoftSQPOptimizer
// (include/ungar/optimization/soft_sqp.hpp:42-283):
//
//   repeat up to maxIterations:
//     QP   min_d 1/2 d^T H d + g^T d   s.t.  J_g d = -g(x)
//          H = hess f  +  J_h^T diag(b''(-h)) J_h  +  1e-6 I        (:143-152, :257-264)
//          g = grad f  -  J_h^T b'(-h)                               (:153-155, :247-253)
//     filter-style backtracking line search on  phi = f + sum b(-h),  theta = c ||g(x)||   (:68-87)
//     stop when the step is rejected or f decreases by less than 1e-6                      (:88-99)
//
// where b is the relaxed POLY or LOG barrier of soft_inequality_constraint.hpp applied to -h >= 0.
// Differences, by construction: the QP is solved exactly by a sparse L D L^T of the KKT system
// (kkt_solver.hpp) instead of OSQP's ADMM (whose `polish` flag is accepted and ignored), and the
// barrier derivatives are closed forms instead of a second generated function; f, g, h and their
// derivatives are Autodiff::Function objects, i.e. evaluated by the device engine.
#pragma once

#include <algorithm>
#include <cmath>
#include <memory>
#include <string>
#include <vector>

#include "../io/logging.hpp"
#include "../utils/defaulted.hpp"
#include "backtracking_line_search.hpp"
#include "concepts.hpp"
#include "kkt_solver.hpp"
#include "soft_inequality_constraint.hpp"

namespace Ungar {

enum class RelaxedBarrierType { LOG, POLY };

class SoftSQPOptimizer {
  public:
    SoftSQPOptimizer(const bool verbose, const defaulted<1.0> constraintViolationMultiplier = {}, const defaulted<index_t{10}> maxIterations = {},
                     const defaulted<100.0> stiffness = {}, const defaulted<2e-5> epsilon = {},
                     const defaulted<RelaxedBarrierType::POLY> relaxedBarrierType = {}, const defaulted<false> polish = {})
        : _verbose{verbose},
          _constraintViolationMultiplier{constraintViolationMultiplier},
          _maxIterations{maxIterations},
          _stiffness{stiffness},
          _epsilon{epsilon},
          _relaxedBarrierType{relaxedBarrierType} {
        (void)polish;
    }

    /// xp = [decision variables | parameters]; returns the optimised decision variables.
    template <class XP>
    VectorXr Optimize(const Concepts::NLPProblem auto& nlp, const XP& xp) {
        const index_t n = nlp.objective.IndependentVariableSize();
        if (xp.size() != n + nlp.objective.ParameterSize()) throw std::invalid_argument("SoftSQPOptimizer: xp must hold decision variables followed by parameters");
        _xp = VectorXr{xp};
        VectorXr probe = _xp;
        _iterations = 0;

        for (index_t it = 0; it < _maxIterations; ++it) {
            if (_verbose) UNGAR_LOG(trace, "Starting soft SQP iteration {}...", it);
            const real_t objective = FunctionInterface::Invoke(nlp.objective, _xp)[0];

            SolveLocalQP(nlp);
            ++_iterations;
            const std::vector<real_t> gradient = DenseRow(FunctionInterface::Jacobian(nlp.objective, _xp), n);
            auto withHead = [&](const VectorXr& x) -> const VectorXr& {
                for (index_t i = 0; i < n; ++i) probe[i] = x[i];
                return probe;
            };
            Head w{_xp.data(), n};
            const bool accepted = BacktrackingLineSearch{_verbose}.Do(
                gradient, _step,
                [&](const VectorXr& x) -> real_t {
                    const VectorXr& z = withHead(x);
                    return FunctionInterface::Invoke(nlp.objective, z)[0] + EvaluateSoftInequalityConstraints(nlp, z);
                },
                [&](const VectorXr& x) -> real_t {
                    const VectorXr c = FunctionInterface::Invoke(nlp.equalityConstraints, withHead(x));
                    return _constraintViolationMultiplier * std::sqrt(c.size() ? c.squaredNorm() : 0.0);
                },
                w);
            if (!accepted) break;

            const real_t difference = FunctionInterface::Invoke(nlp.objective, _xp)[0] - objective;
            if (difference < 0.0 && std::abs(difference) < 1e-6) {
                if (_verbose) UNGAR_LOG(trace, "Soft SQP convergence criterion met.");
                break;
            }
        }
        return _xp.head(n);
    }

    /// The scalar function z -> sum_i b(-z_i) of the inequality values as an Autodiff::Function
    /// (reference soft_sqp.hpp:105-130), e.g. to inspect the barrier or to batch it on the device.
    Autodiff::Function MakeSoftInequalityConstraintFunction(const Concepts::NLPProblem auto& nlp) const {
        const index_t size = nlp.inequalityConstraints.DependentVariableSize();
        const bool log = _relaxedBarrierType == RelaxedBarrierType::LOG;
        const real_t k = _stiffness, eps = _epsilon;
        const ADFunction zsoft = [log, k, eps](const VectorXad& z, VectorXad& out) {
            ad_scalar_t acc{0.0};
            for (index_t i = 0; i < z.size(); ++i)
                acc += log ? RelaxedLogBarrierFunction{0.0, k, eps}.Evaluate(ad_scalar_t{-z[i]}) : RelaxedPolyBarrierFunction{0.0, k, eps}.Evaluate(ad_scalar_t{-z[i]});
            out.resize(1);
            out[0] = acc;
        };
        const std::string name = std::string("soft_sqp_relaxed_") + (log ? "log" : "poly") + "_sz_" + std::to_string(size) + "_k_" + Tag(k) + "_eps_" + Tag(eps);
        return Autodiff::MakeFunction({zsoft, size, index_t{0}, name, EnabledDerivatives::ALL}, false);
    }

    /// QP solves performed by the last Optimize call.
    index_t Iterations() const {
        return _iterations;
    }
    /// Search direction of the last QP.
    const std::vector<real_t>& LastStep() const {
        return _step;
    }

  private:
    struct Head {  // writable view of the decision variables inside [x | p]
        real_t* p;
        index_t n;
        index_t size() const { return n; }
        real_t& operator[](index_t i) const { return p[i]; }
    };

    static std::string Tag(real_t v) {
        std::string s = std::to_string(v);
        for (char& c : s)
            if (c == '.' || c == '-') c = '_';
        return s;
    }
    template <class Sparse>
    static std::vector<real_t> DenseRow(const Sparse& jac, index_t n) {
        std::vector<real_t> row(static_cast<std::size_t>(n), 0.0);
        if (jac.rows() > 0)
            for (int k = jac.outerIndexPtr()[0]; k < jac.outerIndexPtr()[1]; ++k) row[static_cast<std::size_t>(jac.innerIndexPtr()[k])] = jac.valuePtr()[k];
        return row;
    }
    real_t Barrier(real_t z) const {
        return _relaxedBarrierType == RelaxedBarrierType::LOG ? RelaxedLogBarrierFunction{0.0, _stiffness, _epsilon}.Evaluate(z)
                                                              : RelaxedPolyBarrierFunction{0.0, _stiffness, _epsilon}.Evaluate(z);
    }
    real_t BarrierD1(real_t z) const {
        return _relaxedBarrierType == RelaxedBarrierType::LOG ? RelaxedLogBarrierFunction{0.0, _stiffness, _epsilon}.FirstDerivative(z)
                                                              : RelaxedPolyBarrierFunction{0.0, _stiffness, _epsilon}.FirstDerivative(z);
    }
    real_t BarrierD2(real_t z) const {
        return _relaxedBarrierType == RelaxedBarrierType::LOG ? RelaxedLogBarrierFunction{0.0, _stiffness, _epsilon}.SecondDerivative(z)
                                                              : RelaxedPolyBarrierFunction{0.0, _stiffness, _epsilon}.SecondDerivative(z);
    }

    template <class XP>
    real_t EvaluateSoftInequalityConstraints(const Concepts::NLPProblem auto& nlp, const XP& xp) const {
        if (nlp.inequalityConstraints.DependentVariableSize() == 0) return 0.0;
        const VectorXr h = FunctionInterface::Invoke(nlp.inequalityConstraints, xp);
        real_t acc = 0.0;
        for (index_t i = 0; i < h.size(); ++i) acc += Barrier(-h[i]);
        return acc;
    }

    void SolveLocalQP(const Concepts::NLPProblem auto& nlp) {
        const index_t n = nlp.objective.IndependentVariableSize();
        // ---- H (upper triangle, row-wise lists) and g
        std::vector<std::vector<std::pair<int, real_t>>> rows(static_cast<std::size_t>(n));
        std::vector<real_t> g = DenseRow(FunctionInterface::Jacobian(nlp.objective, _xp), n);
        {
            const auto& H = FunctionInterface::Hessian(nlp.objective, 0, _xp);
            for (index_t r = 0; r < H.rows(); ++r)
                for (int k = H.outerIndexPtr()[r]; k < H.outerIndexPtr()[r + 1]; ++k)
                    if (H.innerIndexPtr()[k] >= r) rows[static_cast<std::size_t>(r)].emplace_back(H.innerIndexPtr()[k], H.valuePtr()[k]);
        }
        if (nlp.inequalityConstraints.DependentVariableSize() > 0) {
            const VectorXr h = FunctionInterface::Invoke(nlp.inequalityConstraints, _xp);
            const auto& J = FunctionInterface::Jacobian(nlp.inequalityConstraints, _xp);
            for (index_t i = 0; i < J.rows(); ++i) {
                const real_t d1 = BarrierD1(-h[i]), d2 = BarrierD2(-h[i]);
                const int begin = J.outerIndexPtr()[i], end = J.outerIndexPtr()[i + 1];
                for (int a = begin; a < end; ++a) {
                    const int ca = J.innerIndexPtr()[a];
                    const real_t va = J.valuePtr()[a];
                    g[static_cast<std::size_t>(ca)] -= d1 * va;  // d/dx b(-h) = -b'(-h) dh/dx
                    if (d2 == 0.0) continue;
                    for (int b = begin; b < end; ++b) {
                        const int cb = J.innerIndexPtr()[b];
                        if (cb >= ca) rows[static_cast<std::size_t>(ca)].emplace_back(cb, d2 * va * J.valuePtr()[b]);
                    }
                }
            }
        }
        std::vector<int> hStarts(static_cast<std::size_t>(n) + 1, 0), hCols;
        std::vector<real_t> hValues;
        for (index_t r = 0; r < n; ++r) {
            auto& row = rows[static_cast<std::size_t>(r)];
            row.emplace_back(static_cast<int>(r), 1e-6);  // regularisation of the reference (:149-151)
            std::sort(row.begin(), row.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
            for (std::size_t k = 0; k < row.size(); ++k) {
                if (k && row[k].first == row[k - 1].first) hValues.back() += row[k].second;
                else {
                    hCols.push_back(row[k].first);
                    hValues.push_back(row[k].second);
                }
            }
            hStarts[static_cast<std::size_t>(r) + 1] = static_cast<int>(hCols.size());
        }
        // ---- equality constraints
        const index_t m = nlp.equalityConstraints.DependentVariableSize();
        static const int kEmpty[1] = {0};
        std::vector<real_t> b(static_cast<std::size_t>(m));
        const int *aStarts = kEmpty, *aCols = nullptr;
        const real_t* aValues = nullptr;
        if (m > 0) {
            const VectorXr c = FunctionInterface::Invoke(nlp.equalityConstraints, _xp);
            for (index_t i = 0; i < m; ++i) b[static_cast<std::size_t>(i)] = -c[i];
            const auto& A = FunctionInterface::Jacobian(nlp.equalityConstraints, _xp);
            aStarts = A.outerIndexPtr();
            aCols = A.innerIndexPtr();
            aValues = A.valuePtr();
        }
        for (real_t v : hValues)
            if (!std::isfinite(v)) throw std::runtime_error("SoftSQPOptimizer: non-finite entry in the QP objective matrix");
        _kkt.Solve(n, hStarts, hCols, hValues, g.data(), m, aStarts, aCols, aValues, b.data(), _step, _multipliers);
    }

    bool _verbose;
    real_t _constraintViolationMultiplier;
    index_t _maxIterations;
    real_t _stiffness, _epsilon;
    RelaxedBarrierType _relaxedBarrierType;
    index_t _iterations = 0;
    VectorXr _xp;
    std::vector<real_t> _step, _multipliers;
    KktSolver _kkt;
};

}  // namespace Ungar
