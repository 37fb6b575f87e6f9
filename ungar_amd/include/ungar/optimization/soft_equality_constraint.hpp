// ungar_amd :: quadratic penalty 0.5 k (lhs - rhs)^2 for soft equality constraints, generic over the scalar
// (reference include/ungar/optimization/soft_equality_constraint.hpp:34-52).
#pragma once

#include "../autodiff/data_types.hpp"

namespace Ungar {

class SoftEqualityConstraint {
  public:
    constexpr SoftEqualityConstraint(const real_t rhs, const real_t stiffness = 1.0) : _rhs{rhs}, _stiffness{stiffness} {
    }

    template <Concepts::Scalar S>
    S Evaluate(const S& lhs) const {
        const S e = lhs - _rhs;
        return 0.5 * _stiffness * e * e;
    }
    template <class V>
        requires requires(const V& v) { v.size(); v[0]; }
    auto Evaluate(const V& lhs) const {
        using S = std::remove_cvref_t<decltype(lhs[0] + lhs[0])>;
        S acc{0.0};
        for (index_t i = 0; i < lhs.size(); ++i) acc += Evaluate(S{lhs[i]});
        return acc;
    }

  private:
    real_t _rhs, _stiffness;
};

}  // namespace Ungar
