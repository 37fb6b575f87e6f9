// ungar_amd :: Robot<Scalar> -- a kinematic tree loaded from a robot description plus the data its
// rigid-body algorithms write to (reference include/ungar/rbd/robot.hpp:39-104).
//
// The description is a URDF file (read by ungar_amd's own XML reader: revolute / fixed joints, inertial
// tags; fixed-joint links are lumped into their parents as Pinocchio does) or the `.robot` text form of
// the same content shipped under ungar_amd/data; the root is a free-flyer, q = [p, quaternion xyzw,
// joint angles], v = [body-frame linear, angular velocity, joint rates].
#pragma once

#include <memory>
#include <random>
#include <string>

#include "../utils/utils.hpp"  // (as the reference's rbd/robot.hpp: user code that includes only this header uses Utils::...)
#include "quantities/quantities.hpp"

namespace Ungar {
namespace RBD {
/// Free-flyer-rooted model of a robot description file (URDF or the `.robot` text form), with what Robot::Model() exposes.
inline ModelInfo BuildModelInfo(const std::string& descriptionFilename) {
    namespace rbd = ::ungar_amd::rbd;
    ModelInfo model;
    model.impl = rbd::BuildModel(rbd::ReadRobotDescription(descriptionFilename));
    model.name = model.impl.name;
    model.nq = model.impl.nq;
    model.nv = model.impl.nv;
    model.njoints = model.impl.NumJoints();
    for (const auto& j : model.impl.joints) model.names.push_back(j.name);
    for (const auto& f : model.impl.frames) model.frameNames.push_back(f.name);
    model.nframes = static_cast<int>(model.frameNames.size());
    return model;
}
}  // namespace RBD

template <Concepts::Scalar S = real_t>
class Robot {
  public:
    explicit Robot(const std::string& descriptionFilename) : _model{RBD::BuildModelInfo(descriptionFilename)}, _data{std::make_unique<RBD::Data<S>>()} {
    }
    Robot(const Robot& other) : _model{other._model}, _data{std::make_unique<RBD::Data<S>>()} {
    }
    Robot(Robot&&) = default;

    constexpr auto Compute(auto quantity) {
        return RBD::Evaluator<quantity, S>{_model, *_data};
    }
    decltype(auto) Get(auto quantity) const {
        return std::as_const(*_data).*Member(quantity);
    }
    decltype(auto) Get(auto quantity) {
        auto getter = RBD::Getter<quantity, S>{_model, *_data};
        return getter.Get();
    }
    /// Getters with arguments (user-defined quantities, example/rbd/quantity.example.cpp:118-163: a frame by index or by name).
    template <class... Args>
        requires(sizeof...(Args) > 0)
    decltype(auto) Get(auto quantity, Args&&... args) {
        auto getter = RBD::Getter<quantity, S>{_model, *_data};
        return getter.Get(std::forward<Args>(args)...);
    }

    const RBD::ModelInfo& Model() const {
        return _model;
    }
    const RBD::Data<S>& Data() const {
        return *_data;
    }
    RBD::Data<S>& Data() {
        return *_data;
    }

    /// Uniform position in [-1, 1]^3, uniform random unit quaternion, joint angles in [-pi, pi].
    VectorX<S> RandomConfiguration() const {
        static std::mt19937_64 rng{0x5EEDu};
        std::uniform_real_distribution<double> U{-1.0, 1.0};
        std::normal_distribution<double> G{0.0, 1.0};
        VectorX<S> q{static_cast<index_t>(_model.nq)};
        for (index_t i = 0; i < q.size(); ++i) q[i] = S{U(rng) * (i < 3 ? 1.0 : 3.14159265358979323846)};
        double quat[4], n = 0;
        for (double& c : quat) {
            c = G(rng);
            n += c * c;
        }
        for (int k = 0; k < 4; ++k) q[3 + k] = S{quat[k] / std::sqrt(n)};
        return q;
    }

  private:
    template <class Q>
    static constexpr auto Member(Q quantity) {
        // const access goes through the mutable getter's member: resolve which Data member it names
        return MemberOf(quantity);
    }
    template <class Q>
    static constexpr auto MemberOf(Q) {
        namespace qs = RBD::Quantities;
        if constexpr (std::is_same_v<Q, qs::generalized_accelerations_t>) return &RBD::Data<S>::ddq;
        else if constexpr (std::is_same_v<Q, qs::joint_torques_t>) return &RBD::Data<S>::tau;
        else if constexpr (std::is_same_v<Q, qs::nonlinear_effects_t>) return &RBD::Data<S>::nle;
        else if constexpr (std::is_same_v<Q, qs::generalized_gravity_t>) return &RBD::Data<S>::g;
        else if constexpr (std::is_same_v<Q, qs::joint_space_inertia_matrix_t>) return &RBD::Data<S>::M;
        else if constexpr (std::is_same_v<Q, qs::joint_space_inertia_matrix_inverse_t>) return &RBD::Data<S>::Minv;
        else if constexpr (std::is_same_v<Q, qs::com_position_t>) return &RBD::Data<S>::com;
        else if constexpr (std::is_same_v<Q, qs::com_velocity_t>) return &RBD::Data<S>::vcom;
        else if constexpr (std::is_same_v<Q, qs::com_acceleration_t>) return &RBD::Data<S>::acom;
        else if constexpr (std::is_same_v<Q, qs::kinetic_energy_t>) return &RBD::Data<S>::kinetic_energy;
        else if constexpr (std::is_same_v<Q, qs::frames_t>) return &RBD::Data<S>::oMf;
        else if constexpr (std::is_same_v<Q, qs::centroidal_momentum_t>) return &RBD::Data<S>::hg;
        else if constexpr (std::is_same_v<Q, qs::centroidal_momentum_matrix_t>) return &RBD::Data<S>::Ag;
        else if constexpr (std::is_same_v<Q, qs::composite_rigid_body_inertia_t>) return &RBD::Data<S>::Ig;
        else return &RBD::Data<S>::potential_energy;
    }

    RBD::ModelInfo _model;
    std::unique_ptr<RBD::Data<S>> _data;
};

}  // namespace Ungar

/// The handful of Pinocchio names the reference's own rbd test and examples spell out (test/rbd/robot.test.cpp:99-107, 118-121;
/// example/rbd/quantity.example.cpp:86-163), on ungar_amd's model: `pinocchio::Model` is what Robot::Model() returns, `DataTpl` what the evaluators write to.
namespace pinocchio {
using Model = ::Ungar::RBD::ModelInfo;
template <class S>
using ModelTpl = ::Ungar::RBD::ModelInfo;
template <class S>
using DataTpl = ::Ungar::RBD::Data<S>;
using SE3 = ::Ungar::RBD::Pose<::Ungar::real_t>;
struct JointModelFreeFlyer {};  // the root joint is always a free-flyer here
namespace urdf {
inline Model& buildModel(const std::string& filename, const JointModelFreeFlyer& /*rootJoint*/, Model& model) {
    model = ::Ungar::RBD::BuildModelInfo(filename);
    return model;
}
}  // namespace urdf
/// World placements of every frame (the joint placements of pinocchio::forwardKinematics and the frame update in one pass)
template <class S, class Q>
inline void forwardKinematics(const ModelTpl<S>& model, DataTpl<S>& data, const Q& q) {
    ::Ungar::RBD::Evaluator<::Ungar::RBD::Quantities::frames, S>{model, data}.At(q);
}
template <class S>
inline const ::Ungar::RBD::Pose<S>& updateFramePlacement(const ModelTpl<S>& /*model*/, DataTpl<S>& data, const std::size_t frameId) {
    return data.oMf[frameId];  // (already placed by forwardKinematics above)
}
}  // namespace pinocchio
