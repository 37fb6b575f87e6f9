// ungar_amd :: rigid-body quantities -- tags, evaluators and getters
// (reference include/ungar/rbd/quantity.hpp:42-44, evaluator.hpp:36-58, getter.hpp:36-55).
//
//   robot.Compute(quantity).At(q, v, ...)   runs the algorithm and stores its result in the robot's data;
//   robot.Get(quantity)                      returns a reference to that result.
// The reference forwards to Pinocchio (absent from this image); here the algorithms are ungar_amd's own
// scalar-generic rigid-body code (csrc/rbd: ABA, RNEA, CRBA, U D U^T), in Pinocchio's conventions, so that
// with Scalar = ad_scalar_t the whole recursion is recorded on the derivative tape exactly as the
// reference records pinocchio::aba (test/rbd/robot.test.cpp:124-135) -- and then runs batched on the GPU.
#pragma once

#include <array>
#include <cmath>
#include <ostream>
#include <string>
#include <vector>

#include "../../../csrc/rbd/rnea_crba.hpp"
#include "../autodiff/data_types.hpp"

namespace Ungar {
namespace RBD {

/// What Robot::Model() exposes (field names follow pinocchio::Model).
struct ModelInfo {
    std::string name;
    int nq = 0, nv = 0, njoints = 0, nframes = 0;
    std::vector<std::string> names;       // joint names, universe first
    std::vector<std::string> frameNames;  // universe, root joint, then every joint and link of the description in the reference's frame order (csrc/rbd/model.hpp)
    int getFrameId(const std::string& name) const {
        for (int i = 0; i < nframes; ++i)
            if (frameNames[static_cast<std::size_t>(i)] == name) return i;
        return nframes;  // as pinocchio: "not found" = nframes
    }
    bool existFrame(const std::string& name) const {
        return getFrameId(name) < nframes;
    }
    ::ungar_amd::rbd::Model impl;
};

/// Placement of a frame in the world: x_world = rotation * x_frame + translation (cf. pinocchio::SE3).
template <class S>
struct Pose {
    std::array<std::array<S, 3>, 3> rotationMatrix;
    Vector3<S> position;
    const Vector3<S>& translation() const {
        return position;
    }
    const std::array<std::array<S, 3>, 3>& rotation() const {
        return rotationMatrix;
    }
    /// cf. pinocchio::SE3::isEqual (example/rbd/quantity.example.cpp:209): every entry within `precision`
    bool isEqual(const Pose& other, const double precision = 1e-12) const {
        for (std::size_t r = 0; r < 3; ++r) {
            if (std::abs(static_cast<double>(position[static_cast<index_t>(r)] - other.position[static_cast<index_t>(r)])) > precision) return false;
            for (std::size_t c = 0; c < 3; ++c)
                if (std::abs(static_cast<double>(rotationMatrix[r][c] - other.rotationMatrix[r][c])) > precision) return false;
        }
        return true;
    }
    friend std::ostream& operator<<(std::ostream& os, const Pose& pose) {  // "  R =\n...\n  p = ..." as the reference's logs print an SE3
        os << "  R =\n";
        for (std::size_t r = 0; r < 3; ++r) os << "    " << pose.rotationMatrix[r][0] << " " << pose.rotationMatrix[r][1] << " " << pose.rotationMatrix[r][2] << "\n";
        return os << "  p = " << pose.position[0] << " " << pose.position[1] << " " << pose.position[2];
    }
};

/// [linear; angular] momentum about the centre of mass: a 6-vector that also answers to the one member of pinocchio::Force the reference's example
/// calls on it (example/rbd/robot.example.cpp:127-141: `.toVector_impl()`).
template <class S>
struct SpatialMomentum : VectorX<S> {
    using VectorX<S>::VectorX;
    using VectorX<S>::operator=;
    const VectorX<S>& toVector_impl() const {
        return *this;
    }
};

/// Results of the algorithms (field names follow pinocchio::Data).
template <class S>
struct Data {
    std::vector<Pose<S>> oMf;  // world placements of Model().frames
    VectorX<S> ddq, tau, nle, g;
    SpatialMomentum<S> hg;
    MatrixX<S> M, Minv, Ag, Ig;
    Vector3<S> com, vcom, acom;
    S kinetic_energy{0.0}, potential_energy{0.0};
};

template <auto QUANTITY, class S>
struct Evaluator;
template <auto QUANTITY, class S>
struct Getter;

namespace Quantities {
#define UNGAR_MAKE_QUANTITY(name)      \
    inline constexpr struct name##_t { \
    } name
}  // namespace Quantities

namespace Internal {
template <class S, class V>
std::vector<S> ToStd(const V& v) {
    std::vector<S> out(static_cast<std::size_t>(v.size()));
    for (index_t i = 0; i < v.size(); ++i) out[static_cast<std::size_t>(i)] = S{v[i]};
    return out;
}
template <class S>
VectorX<S> ToVector(const std::vector<S>& v) {
    VectorX<S> out{static_cast<index_t>(v.size())};
    for (std::size_t i = 0; i < v.size(); ++i) out[static_cast<index_t>(i)] = v[i];
    return out;
}
}  // namespace Internal

/// Getter returning a member of Data.
#define UNGAR_MAKE_GETTER(quantity, dataMember)                   \
    template <class S>                                            \
    struct Getter<::Ungar::RBD::Quantities::quantity, S> {        \
        const auto& Get() const {                                 \
            return data.dataMember;                               \
        }                                                         \
        auto& Get() {                                             \
            return data.dataMember;                               \
        }                                                         \
        const ::Ungar::RBD::ModelInfo& model;                     \
        ::Ungar::RBD::Data<S>& data;                              \
    }

}  // namespace RBD
}  // namespace Ungar
