// ungar_amd :: soft ("relaxed barrier") inequality constraints, generic over the scalar so that the same
// expression is evaluated on doubles and recorded on a derivative tape.
//
// Mirrors the classes of reference include/ungar/optimization/soft_inequality_constraint.hpp:
//   LogisticFunction :40-65, RelaxedLogBarrierFunction :77-124 (Grandia et al., IROS 2019),
//   RelaxedPolyBarrierFunction :135-205 (Bern et al., TOG 2017), SoftBoundConstraint :207-236.
// All barriers penalise violations of  lhs >= rhs.  On the tape the branches become conditional
// expressions (CondExpGe / CondExpLt), so derivatives are those of the active polynomial / log piece.
// Additionally (not in the reference, used by the SQP caller): closed-form first and second derivatives
// of the scalar barriers on doubles, FirstDerivative / SecondDerivative.
#pragma once

#include <cmath>

#include "../autodiff/data_types.hpp"
#include "../utils/utils.hpp"

namespace Ungar {

class LogisticFunction {
  public:
    constexpr LogisticFunction(const real_t midpoint = 0.0, const real_t steepness = 1.0, const real_t maximumValue = 1.0)
        : _midpoint{midpoint}, _steepness{steepness}, _maximumValue{maximumValue} {
    }

    template <Concepts::Scalar S>
    S Evaluate(const S& x) const {
        using std::exp;
        return _maximumValue / (1.0 + exp(-_steepness * (x - _midpoint)));
    }

    /// 1 where x is (smoothly) greater / less than lhs, 0 elsewhere; default steepness and the `1 - ...` form of the
    /// reference (soft_inequality_constraint.hpp:47-59).
    template <Concepts::Scalar S>
    static S SmoothGreaterThan(const S& x, const real_t lhs, const real_t steepness = 1024.0) {
        return LogisticFunction{lhs, steepness}.Evaluate(x);
    }
    template <Concepts::Scalar S>
    static S SmoothLessThan(const S& x, const real_t lhs, const real_t steepness = 1024.0) {
        return 1.0 - SmoothGreaterThan(x, lhs, steepness);
    }

  private:
    real_t _midpoint, _steepness, _maximumValue;
};

/// -mu log(x) for x >= epsilon, continued quadratically below (x = lhs - rhs).
class RelaxedLogBarrierFunction {
  public:
    constexpr RelaxedLogBarrierFunction(const real_t rhs, const real_t stiffness = 1e-5, const real_t epsilon = 5.0)
        : _rhs{rhs}, _epsilon{epsilon}, _mu{stiffness} {
    }

    template <Concepts::Scalar S>
    S Evaluate(const S& lhs) const {
        return Piece(S{lhs - _rhs});
    }
    /// Sum of the barrier over the coefficients of a vector.
    template <class V>
        requires requires(const V& v) { v.size(); v[0]; }
    auto Evaluate(const V& lhs) const {
        using S = std::remove_cvref_t<decltype(lhs[0] + lhs[0])>;
        S acc{0.0};
        for (index_t i = 0; i < lhs.size(); ++i) acc += Piece(S{lhs[i] - _rhs});
        return acc;
    }
    /// The reference's spelling with the scalar named explicitly, `barrier.Evaluate<ad_scalar_t>(vector)` (:90-95).
    template <Concepts::Scalar S, class V>
        requires(!Concepts::Scalar<V>) && requires(const V& v) { v.size(); v[0]; }
    S Evaluate(const V& lhs) const {
        S acc{0.0};
        for (index_t i = 0; i < lhs.size(); ++i) acc += Piece(S{lhs[i] - _rhs});
        return acc;
    }

    real_t FirstDerivative(const real_t lhs) const {
        const real_t x = lhs - _rhs;
        return x >= _epsilon ? -_mu / x : _mu * (x - 2.0 * _epsilon) / (_epsilon * _epsilon);
    }
    real_t SecondDerivative(const real_t lhs) const {
        const real_t x = lhs - _rhs;
        return x >= _epsilon ? _mu / (x * x) : _mu / (_epsilon * _epsilon);
    }

  private:
    real_t Piece(const real_t& x) const {
        if (x >= _epsilon) return -_mu * std::log(x);
        const real_t t = (x - 2.0 * _epsilon) / _epsilon;
        return 0.5 * _mu * (t * t - 1.0) - _mu * std::log(_epsilon);
    }
    ad_scalar_t Piece(const ad_scalar_t& x) const {
        namespace tape = ::ungar_amd::tape;
        const ad_scalar_t t = (x - ad_scalar_t{2.0 * _epsilon}) / ad_scalar_t{_epsilon};
        return tape::CondExpGe(x, ad_scalar_t{_epsilon}, -_mu * tape::log(x), ad_scalar_t{0.5 * _mu} * (t * t - ad_scalar_t{1.0}) - ad_scalar_t{_mu * std::log(_epsilon)});
    }

    real_t _rhs, _epsilon, _mu;
};

/// 0 for x >= epsilon, a cubic on [0, epsilon), a quadratic for x < 0; C2 at both joints (x = lhs - rhs).
class RelaxedPolyBarrierFunction {
  public:
    constexpr RelaxedPolyBarrierFunction(const real_t rhs, const real_t stiffness = 1.0, const real_t epsilon = 2e-5)
        : _rhs{rhs},
          _epsilon{epsilon},
          _a1{stiffness},
          _b1{-0.5 * stiffness * epsilon},
          _c1{-(1.0 / 3.0) * (-_b1 - _a1 * epsilon) * epsilon - 0.5 * _a1 * epsilon * epsilon - _b1 * epsilon},
          _a2{(-_b1 - _a1 * epsilon) / (epsilon * epsilon)},
          _b2{_a1},
          _c2{_b1},
          _d2{_c1} {
    }

    template <Concepts::Scalar S, bool NO_CONDITIONAL_APPROXIMATION = false>
    S Evaluate(const S& lhs) const {
        if constexpr (NO_CONDITIONAL_APPROXIMATION) return Smooth(S{lhs - _rhs});
        else return Piece(S{lhs - _rhs});
    }
    template <class V, bool NO_CONDITIONAL_APPROXIMATION = false>
        requires requires(const V& v) { v.size(); v[0]; }
    auto Evaluate(const V& lhs) const {
        using S = std::remove_cvref_t<decltype(lhs[0] + lhs[0])>;
        S acc{0.0};
        for (index_t i = 0; i < lhs.size(); ++i) {
            if constexpr (NO_CONDITIONAL_APPROXIMATION) acc += Smooth(S{lhs[i] - _rhs});
            else acc += Piece(S{lhs[i] - _rhs});
        }
        return acc;
    }
    /// `barrier.Evaluate<ad_scalar_t>(vector)` / `Evaluate<ad_scalar_t, true>(vector)` as in the reference (:157-168).
    template <Concepts::Scalar S, bool NO_CONDITIONAL_APPROXIMATION = false, class V>
        requires(!Concepts::Scalar<V>) && requires(const V& v) { v.size(); v[0]; }
    S Evaluate(const V& lhs) const {
        return S{this->template Evaluate<V, NO_CONDITIONAL_APPROXIMATION>(lhs)};
    }

    real_t FirstDerivative(const real_t lhs) const {
        const real_t x = lhs - _rhs;
        if (x < 0.0) return _a1 * x + _b1;
        if (x < _epsilon) return _a2 * x * x + _b2 * x + _c2;
        return 0.0;
    }
    real_t SecondDerivative(const real_t lhs) const {
        const real_t x = lhs - _rhs;
        if (x < 0.0) return _a1;
        if (x < _epsilon) return 2.0 * _a2 * x + _b2;
        return 0.0;
    }

  private:
    template <class S>
    S Quadratic(const S& x) const {
        return 0.5 * _a1 * x * x + _b1 * x + _c1;
    }
    template <class S>
    S Cubic(const S& x) const {
        return (1.0 / 3.0) * _a2 * x * x * x + 0.5 * _b2 * x * x + _c2 * x + _d2;
    }
    real_t Piece(const real_t& x) const {
        return x < 0.0 ? Quadratic(x) : x < _epsilon ? Cubic(x) : 0.0;
    }
    ad_scalar_t Piece(const ad_scalar_t& x) const {
        namespace tape = ::ungar_amd::tape;
        return tape::CondExpLt(x, ad_scalar_t{0.0}, Quadratic(x), tape::CondExpLt(x, ad_scalar_t{_epsilon}, Cubic(x), ad_scalar_t{0.0}));
    }
    template <class S>
    S Smooth(const S& x) const {
        const real_t steepness = 1.0 / _epsilon;
        return LogisticFunction::SmoothLessThan(x, 0.0, steepness) * Quadratic(x) +
               LogisticFunction::SmoothGreaterThan(x, 0.0, steepness) * LogisticFunction::SmoothLessThan(x, _epsilon, steepness) * Cubic(x);
    }

    real_t _rhs, _epsilon;
    real_t _a1, _b1, _c1;
    real_t _a2, _b2, _c2, _d2;
};

/// lowerBound <= x <= upperBound as two relaxed poly barriers whose transition width is a fraction
/// `relativeEpsilon` of the admissible interval.
class SoftBoundConstraint {
  public:
    constexpr SoftBoundConstraint(const real_t lowerBound, const real_t upperBound, const real_t stiffness = 1.0, const real_t relativeEpsilon = 1e-1)
        : _epsilon{(upperBound - lowerBound) * relativeEpsilon}, _lower{lowerBound, stiffness, _epsilon}, _upper{-upperBound, stiffness, _epsilon} {
    }

    template <Concepts::Scalar S, bool NO_CONDITIONAL_APPROXIMATION = false>
    S Evaluate(const S& x) const {
        return _lower.Evaluate<S, NO_CONDITIONAL_APPROXIMATION>(x) + _upper.Evaluate<S, NO_CONDITIONAL_APPROXIMATION>(S{-x});
    }
    template <class V, bool NO_CONDITIONAL_APPROXIMATION = false>
        requires requires(const V& v) { v.size(); v[0]; }
    auto Evaluate(const V& x) const {
        using S = std::remove_cvref_t<decltype(x[0] + x[0])>;
        S acc{0.0};
        for (index_t i = 0; i < x.size(); ++i) acc += Evaluate<S, NO_CONDITIONAL_APPROXIMATION>(S{x[i]});
        return acc;
    }
    template <Concepts::Scalar S, bool NO_CONDITIONAL_APPROXIMATION = false, class V>
        requires(!Concepts::Scalar<V>) && requires(const V& v) { v.size(); v[0]; }
    S Evaluate(const V& x) const {
        return S{this->template Evaluate<V, NO_CONDITIONAL_APPROXIMATION>(x)};
    }

  private:
    real_t _epsilon;
    RelaxedPolyBarrierFunction _lower, _upper;
};

}  // namespace Ungar
