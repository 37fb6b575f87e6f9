// ungar_amd :: backtracking line search with the three-way acceptance test of the reference
// (include/ungar/optimization/backtracking_line_search.hpp:56-178; parameters :58-77):
//   theta = constraint violation, phi = cost, alpha halves (gammaAlpha) from 1 down to alphaMin;
//   1. theta_next > thetaMax        -> accept iff the violation shrinks by the factor (1 - gammaTheta);
//   2. both violations < thetaMin and the step is a descent direction -> Armijo on phi with slope eta;
//   3. otherwise                    -> accept iff phi OR theta decreases by its gamma factor.
// On acceptance w += alpha dw.
#pragma once

#include <algorithm>

#include "../data_types.hpp"
#include "../io/logging.hpp"

namespace Ungar {

class BacktrackingLineSearch {
  public:
    struct Parameters {
        real_t alphaMin = 1e-4;
        real_t thetaMin = 1e-6;
        real_t thetaMax = 1e-2;
        real_t eta = 1e-4;
        real_t gammaPhi = 1e-6;
        real_t gammaTheta = 1e-6;
        real_t gammaAlpha = 0.5;
    };

    constexpr explicit BacktrackingLineSearch(const bool verbose) : _verbose{verbose}, _parameters{} {
    }
    constexpr BacktrackingLineSearch(const bool verbose, const Parameters& parameters) : _verbose{verbose}, _parameters{parameters} {
    }

    /// `costFunction(w)` and `constraintViolation(w)` take a VectorXr and return real_t.
    template <class Gradient, class Direction, class Cost, class Violation, class W>
    [[nodiscard]] bool Do(const Gradient& costFunctionGradient, const Direction& dw, const Cost& costFunction, const Violation& constraintViolation,
                          W&& w) const {
        const index_t n = static_cast<index_t>(dw.size());
        if (static_cast<index_t>(costFunctionGradient.size()) != n || static_cast<index_t>(w.size()) != n) throw std::invalid_argument("BacktrackingLineSearch: size mismatch");
        real_t slope = 0.0;
        for (index_t i = 0; i < n; ++i) slope += costFunctionGradient[i] * dw[i];

        VectorXr start{n}, trial{n};
        for (index_t i = 0; i < n; ++i) start[i] = w[i];
        const real_t theta = constraintViolation(start);
        const real_t phi = costFunction(start);
        if (_verbose) UNGAR_LOG(trace, "Backtracking line search: initial constraint violation {}, initial cost {}", theta, phi);

        const Parameters& P = _parameters;
        for (real_t alpha = 1.0; alpha >= P.alphaMin; alpha *= P.gammaAlpha) {
            for (index_t i = 0; i < n; ++i) trial[i] = start[i] + alpha * dw[i];
            const real_t thetaNext = constraintViolation(trial);
            const real_t phiNext = costFunction(trial);
            if (_verbose) UNGAR_LOG(trace, "\tstep size {:>14}  constraint violation {:>14}  cost {:>14}", alpha, thetaNext, phiNext);

            bool accepted;
            if (thetaNext > P.thetaMax) {
                accepted = thetaNext < (1.0 - P.gammaTheta) * theta;
            } else if (std::max(theta, thetaNext) < P.thetaMin && slope < 0.0) {
                accepted = phiNext < phi + P.eta * alpha * slope;
            } else {
                accepted = phiNext < (1.0 - P.gammaPhi) * phi || thetaNext < (1.0 - P.gammaTheta) * theta;
            }
            if (accepted) {
                for (index_t i = 0; i < n; ++i) w[i] = trial[i];
                return true;
            }
        }
        if (_verbose) UNGAR_LOG(trace, "The line search found no acceptable step size; the solution was not updated.");
        return false;
    }

    const Parameters& GetParameters() const {
        return _parameters;
    }
    void SetParameters(const Parameters& parameters) {
        _parameters = parameters;
    }

  private:
    bool _verbose;
    Parameters _parameters;
};

}  // namespace Ungar
