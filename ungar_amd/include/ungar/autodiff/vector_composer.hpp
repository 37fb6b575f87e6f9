// ungar_amd :: concatenates AD sub-vectors / scalars into one output vector, in push order
// (reference include/ungar/autodiff/vector_composer.hpp:35-76) -- defines the row order of the
// equality / inequality constraint outputs.
#pragma once

#include <vector>

#include "data_types.hpp"

namespace Ungar {
namespace Autodiff {

class VectorComposer {
  public:
    template <class V>
        requires std::is_same_v<std::remove_const_t<typename V::Scalar>, ad_scalar_t>
    VectorComposer& operator<<(const Eigen::MatrixBase<V>& vector) {
        for (index_t i = 0; i < vector.size(); ++i) _impl.push_back(vector[i]);
        return *this;
    }
    VectorComposer& operator<<(const ad_scalar_t& scalar) {
        _impl.push_back(scalar);
        return *this;
    }
    void Clear() {
        _impl.clear();
    }
    index_t Size() const {
        return static_cast<index_t>(_impl.size());
    }
    VectorXad Compose() const {
        VectorXad out{Size()};
        for (index_t i = 0; i < Size(); ++i) out[i] = _impl[static_cast<std::size_t>(i)];
        return out;
    }

  private:
    std::vector<ad_scalar_t> _impl;
};

}  // namespace Autodiff
}  // namespace Ungar
