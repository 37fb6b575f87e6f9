// ungar_amd :: typed views over a flat scalar buffer laid out by a Variable hierarchy.
//
// Reference: include/ungar/variable_lazy_map.hpp:87-307 (Get/GetTuple/GetImpl) and
// include/ungar/variable_map.hpp:41-221.  View rule: size 1 -> reference to the
// scalar; unit quaternion -> Map<Quaternion> over (x,y,z,w); size <= 32 -> fixed-size Map; larger ->
// dynamic-size Map.  No copies; works for Scalar in {real_t, ad_scalar_t}.
#pragma once

#include <tuple>

#include "data_types.hpp"

namespace Ungar {

template <class Scalar, Concepts::Variable Var, bool MUTABLE>
class VariableLazyMap {
  public:
    using Ptr = std::conditional_t<MUTABLE, Scalar*, const Scalar*>;
    constexpr VariableLazyMap(Ptr data, const Var& var) : _data{data}, _var{var} {
    }

    /// View of the sub-variable addressed by `args` (same grammar as Variable::operator()).
    template <class... Args>
    decltype(auto) Get(const Args&... args) const {
        return View(_var(args...));
    }
    template <class... Vars>
    auto GetTuple(const Vars&... vars) const {
        return std::tuple<decltype(Get(vars))...>{Get(vars)...};
    }
    Ptr Data() const {
        return _data;
    }

  private:
    template <class V>
    decltype(auto) View(const V& v) const {
        Ptr p = _data + (v.Index() - _var.Index());
        if constexpr (V::IsScalar()) {
            return (*p);  // Scalar& / const Scalar&
        } else if constexpr (V::IsQuaternion()) {
            return Eigen::Map<std::conditional_t<MUTABLE, Quaternion<Scalar>, const Quaternion<Scalar>>>{p};
        } else if constexpr (V::Size() <= 32) {
            return Eigen::Map<std::conditional_t<MUTABLE, Vector<Scalar, V::Size()>, const Vector<Scalar, V::Size()>>>{p};
        } else {
            return Eigen::Map<std::conditional_t<MUTABLE, VectorX<Scalar>, const VectorX<Scalar>>>{p, V::Size()};
        }
    }
    Ptr _data;
    Var _var;
};

/// reference variable_lazy_map.hpp:341-361
template <class Underlying, Concepts::Variable Var>
auto MakeVariableLazyMap(Underlying& underlying, const Var& var) {
    using S = std::remove_const_t<typename std::remove_cvref_t<Underlying>::Scalar>;
    constexpr bool mut = !std::is_const_v<Underlying> && !std::is_const_v<typename Eigen::Traits<std::remove_cvref_t<Underlying>>::Scalar>;
    assert(underlying.size() == Var::Size());
    return VariableLazyMap<S, Var, mut>{underlying.data(), var};
}
/// m-variable spelling (reference mvariable_lazy_map.hpp:339-359).
template <class Underlying, Concepts::Variable Var>
auto MakeMVariableLazyMap(Underlying& underlying, const Var& var) {
    return MakeVariableLazyMap(underlying, var);
}

/// Owning map (reference variable_map.hpp:41-221).
template <class Scalar, Concepts::Variable Var>
class VariableMap {
  public:
    explicit VariableMap(const Var& var) : _underlying(Var::Size()), _var{var} {
        _underlying.setZero();
    }
    template <class... Args>
    decltype(auto) Get(const Args&... args) {
        if constexpr (sizeof...(Args) == 0) return (_underlying);
        else return VariableLazyMap<Scalar, Var, true>{_underlying.data(), _var}.Get(args...);
    }
    template <class... Args>
    decltype(auto) Get(const Args&... args) const {
        if constexpr (sizeof...(Args) == 0) return (_underlying);
        else return VariableLazyMap<Scalar, Var, false>{_underlying.data(), _var}.Get(args...);
    }
    template <class... Vars>
    auto GetTuple(const Vars&... vars) {
        return VariableLazyMap<Scalar, Var, true>{_underlying.data(), _var}.GetTuple(vars...);
    }
    template <class... Vars>
    auto GetTuple(const Vars&... vars) const {
        return VariableLazyMap<Scalar, Var, false>{_underlying.data(), _var}.GetTuple(vars...);
    }

  private:
    VectorX<Scalar> _underlying;
    Var _var;
};

template <class Scalar, Concepts::Variable Var>
auto MakeVariableMap(const Var& var) {
    return VariableMap<Scalar, Var>{var};
}

}  // namespace Ungar
