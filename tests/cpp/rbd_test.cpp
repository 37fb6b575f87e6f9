// Host-facade test of the rigid-body layer (ungar/rbd/robot.hpp): mirrors the reference's
// test/rbd/robot.test.cpp (model dimensions :103-106; forward dynamics recorded on the tape and checked
// against the real-valued algorithm :124-135) and adds identities that pin the individual quantities
// without Pinocchio:  RNEA(q, v, ABA(q, v, tau)) = tau,  M a + nle = tau,  M M^-1 = 1,  nle(v = 0) = g,
// d com/dt = vcom and d vcom/dt = acom along an integrated motion.
//   rbd_test cpu <robot file>    host only; prints the ABA result for the Python side (oracle comparison)
//   rbd_test gpu <robot file>    additionally generates the taped ABA as an Autodiff::Function (device)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "ungar/autodiff/function.hpp"
#include "ungar/rbd/quantities/centroidal_momentum.hpp"
#include "ungar/rbd/quantities/centroidal_momentum_matrix.hpp"
#include "ungar/rbd/quantities/composite_rigid_body_inertia.hpp"
#include "ungar/rbd/quantities/frames.hpp"
#include "ungar/rbd/quantities/generalized_accelerations.hpp"
#include "ungar/rbd/robot.hpp"

using namespace Ungar;
namespace qs = RBD::Quantities;

static int g_failures = 0;
#define EXPECT_TRUE(cond)                                               \
    do {                                                                \
        if (!(cond)) {                                                  \
            ++g_failures;                                               \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); \
        }                                                               \
    } while (0)

static real_t MaxAbsDiff(const VectorXr& a, const VectorXr& b) {
    real_t m = 0;
    for (index_t i = 0; i < a.size(); ++i) m = std::max(m, std::fabs(a[i] - b[i]));
    return m;
}

/// q (+) h v for a free-flyer root: p += h R v_lin, quat = quat * exp(h omega), joints += h v_j.
static VectorXr Integrate(const VectorXr& q, const VectorXr& v, real_t h) {
    VectorXr r = q;
    const Quaternionr quat{q[6], q[3], q[4], q[5]};
    const Vector3r dp = quat * Vector3r{v[0], v[1], v[2]};
    for (index_t k = 0; k < 3; ++k) r[k] += h * dp[k];
    const Quaternionr next = quat * Utils::ExponentialMap(Vector3r{h * v[3], h * v[4], h * v[5]});
    r[3] = next.x(), r[4] = next.y(), r[5] = next.z(), r[6] = next.w();
    for (index_t k = 7; k < q.size(); ++k) r[k] += h * v[k - 1];
    return r;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        std::printf("usage: rbd_test cpu|gpu <robot description>\n");
        return 2;
    }
    const std::string file = argv[2];
    Robot<real_t> robot{file};
    std::printf("model %s nq %d nv %d njoints %d\n", robot.Model().name.c_str(), robot.Model().nq, robot.Model().nv, robot.Model().njoints);
    EXPECT_TRUE(robot.Model().nq == 19 && robot.Model().nv == 18 && robot.Model().njoints == 14);  // universe + root + 12 (robot.test.cpp:103-106)

    std::mt19937 gen{42};
    std::uniform_real_distribution<real_t> U{-1.0, 1.0};
    const index_t nq = robot.Model().nq, nv = robot.Model().nv;
    VectorXr q = robot.RandomConfiguration(), v{nv}, tau{nv};
    for (index_t i = 0; i < nv; ++i) v[i] = U(gen), tau[i] = 20.0 * U(gen);

    robot.Compute(qs::generalized_accelerations).At(q, v, tau);
    const VectorXr a = robot.Get(qs::generalized_accelerations);
    robot.Compute(qs::joint_torques).At(q, v, a);
    EXPECT_TRUE(MaxAbsDiff(robot.Get(qs::joint_torques), tau) < 1e-9);

    robot.Compute(qs::joint_space_inertia_matrix).At(q);
    robot.Compute(qs::nonlinear_effects).At(q, v);
    const MatrixXr M = robot.Get(qs::joint_space_inertia_matrix);
    EXPECT_TRUE(MaxAbsDiff(VectorXr{M * a + robot.Get(qs::nonlinear_effects)}, tau) < 1e-9);
    robot.Compute(qs::joint_space_inertia_matrix_inverse).At(q);
    const MatrixXr Minv = robot.Get(qs::joint_space_inertia_matrix_inverse);
    real_t worst = 0;
    for (index_t c = 0; c < nv; ++c) {
        VectorXr e{nv};
        e.setZero();
        e[c] = 1.0;
        const VectorXr col = M * VectorXr{Minv * e};
        worst = std::max(worst, MaxAbsDiff(col, e));
    }
    EXPECT_TRUE(worst < 1e-9);
    VectorXr zero{nv};
    zero.setZero();
    robot.Compute(qs::nonlinear_effects).At(q, zero);
    robot.Compute(qs::generalized_gravity).At(q);
    EXPECT_TRUE(MaxAbsDiff(robot.Get(qs::nonlinear_effects), robot.Get(qs::generalized_gravity)) < 1e-12);

    // centre of mass along an integrated motion (central differences)
    const real_t h = 1e-5;
    robot.Compute(qs::com_acceleration).At(q, v, a);
    const Vector3r vcom = robot.Get(qs::com_velocity), acom = robot.Get(qs::com_acceleration);
    robot.Compute(qs::com_position).At(Integrate(q, v, h));
    const Vector3r cp = robot.Get(qs::com_position);
    robot.Compute(qs::com_position).At(Integrate(q, v, -h));
    const Vector3r cm = robot.Get(qs::com_position);
    EXPECT_TRUE(MaxAbsDiff(VectorXr{(cp - cm) / (2 * h)}, VectorXr{vcom}) < 1e-7);
    robot.Compute(qs::com_velocity).At(Integrate(q, v, h), VectorXr{v + a * h});
    const Vector3r vp = robot.Get(qs::com_velocity);
    robot.Compute(qs::com_velocity).At(Integrate(q, v, -h), VectorXr{v - a * h});
    const Vector3r vm = robot.Get(qs::com_velocity);
    EXPECT_TRUE(MaxAbsDiff(VectorXr{(vp - vm) / (2 * h)}, VectorXr{acom}) < 1e-5);
    // energies: d/dt (T + V) = v^T (tau_applied) with tau_applied = M a + nle - g ... checked through T = 1/2 v^T M v
    robot.Compute(qs::kinetic_energy).At(q, v);
    EXPECT_TRUE(std::fabs(robot.Get(qs::kinetic_energy) - 0.5 * v.dot(VectorXr{M * v})) < 1e-10);
    robot.Compute(qs::potential_energy).At(q);
    robot.Compute(qs::com_position).At(q);
    EXPECT_TRUE(std::fabs(robot.Get(qs::potential_energy) - 9.81 * robot.Model().impl.TotalMass() * robot.Get(qs::com_position)[2]) < 1e-9);

    // forward kinematics of the frames (one per link; the feet are links lumped through fixed joints)
    robot.Compute(qs::frames).At(q);
    EXPECT_TRUE(robot.Model().nframes == static_cast<int>(robot.Get(qs::frames).size()) && robot.Model().existFrame("LF_FOOT") && !robot.Model().existFrame("no_such_frame"));
    {
        // frame numbering of the reference's model builder: universe, root joint, root link, then (joint, child link) pairs depth first -- the feet of ANYmal B are
        // frames 12 / 22 / 32 / 42 (test/rbd/robot.test.cpp:49-52, example/rbd/quantity.example.cpp:44-45)
        EXPECT_TRUE(robot.Model().getFrameId("universe") == 0 && robot.Model().getFrameId("root_joint") == 1 && robot.Model().getFrameId("base") == 2);
        EXPECT_TRUE(robot.Model().getFrameId("LF_FOOT") == 12 && robot.Model().getFrameId("LH_FOOT") == 22 && robot.Model().getFrameId("RF_FOOT") == 32 &&
                    robot.Model().getFrameId("RH_FOOT") == 42);
        const auto& universe = robot.Get(qs::frames)[0];
        EXPECT_TRUE(universe.translation()[0] == 0.0 && universe.rotation()[0][0] == 1.0 && universe.rotation()[0][1] == 0.0);
        const auto& base = robot.Get(qs::frames)[2];  // the root link sits in the free-flyer joint frame
        EXPECT_TRUE(std::fabs(base.translation()[0] - q[0]) + std::fabs(base.translation()[1] - q[1]) + std::fabs(base.translation()[2] - q[2]) < 1e-15);
    }
    for (const char* name : {"base", "LF_FOOT", "LH_FOOT", "RF_FOOT", "RH_FOOT", "LF_SHANK"}) {
        const int id = robot.Model().getFrameId(name);
        if (id >= robot.Model().nframes) continue;
        const auto& pose = robot.Get(qs::frames)[static_cast<std::size_t>(id)];
        std::printf("frame %s", name);
        for (int k = 0; k < 3; ++k) std::printf(" %.17g", pose.translation()[k]);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) std::printf(" %.17g", pose.rotation()[static_cast<std::size_t>(r)][static_cast<std::size_t>(c)]);
        std::printf("\n");
    }

    // centroidal momentum: linear part = total mass * com velocity; printed for the oracle comparison
    robot.Compute(qs::centroidal_momentum).At(q, v);
    const VectorXr hg = robot.Get(qs::centroidal_momentum);
    robot.Compute(qs::com_velocity).At(q, v);
    for (index_t k = 0; k < 3; ++k) EXPECT_TRUE(std::fabs(hg[k] - robot.Model().impl.TotalMass() * robot.Get(qs::com_velocity)[k]) < 1e-10);
    // centroidal map: h_G = A_G v; composite inertia: symmetric, positive, and consistent with A_G for a rigid rotation
    robot.Compute(qs::centroidal_momentum_matrix).At(q);
    EXPECT_TRUE(MaxAbsDiff(VectorXr{robot.Get(qs::centroidal_momentum_matrix) * v}, hg) < 1e-10);
    robot.Compute(qs::composite_rigid_body_inertia).At(q, v);
    {
        const MatrixXr Ig = robot.Get(qs::composite_rigid_body_inertia);
        real_t asym = 0, minDiag = 1e9;
        for (index_t r = 3; r < 6; ++r) {
            minDiag = std::min(minDiag, Ig(r, r));
            for (index_t c = 3; c < 6; ++c) asym = std::max(asym, std::fabs(Ig(r, c) - Ig(c, r)));
        }
        EXPECT_TRUE(asym < 1e-10 && minDiag > 0.1 && std::fabs(Ig(0, 0) - robot.Model().impl.TotalMass()) < 1e-12);
        std::printf("Ig");
        for (index_t r = 3; r < 6; ++r)
            for (index_t c = 3; c < 6; ++c) std::printf(" %.17g", Ig(r, c));
        std::printf("\n");
    }
    std::printf("hg");
    for (index_t i = 0; i < 6; ++i) std::printf(" %.17g", hg[i]);
    std::printf("\n");

    std::printf("q");
    for (index_t i = 0; i < nq; ++i) std::printf(" %.17g", q[i]);
    std::printf("\nv");
    for (index_t i = 0; i < nv; ++i) std::printf(" %.17g", v[i]);
    std::printf("\ntau");
    for (index_t i = 0; i < nv; ++i) std::printf(" %.17g", tau[i]);
    std::printf("\nddq");
    for (index_t i = 0; i < nv; ++i) std::printf(" %.17g", a[i]);
    std::printf("\n");

    if (std::strcmp(argv[1], "gpu") == 0) {
        // forward dynamics recorded on the tape through Robot<ad_scalar_t> (robot.test.cpp:124-135)
        const auto impl = [&](const VectorXad& xp, VectorXad& y) {
            Robot<ad_scalar_t> robotAD{file};
            robotAD.Compute(qs::generalized_accelerations).At(xp.head(nq), xp.segment(nq, nv), xp.segment(nq + nv, nv));
            y = robotAD.Get(qs::generalized_accelerations);
        };
        Autodiff::Function::Blueprint bp{impl, nq + 2 * nv, 0, "rbd_test_forward_dynamics", EnabledDerivatives::JACOBIAN, argc > 3 ? argv[3] : ""};
        const Autodiff::Function f = Autodiff::MakeFunction(bp, true);
        VectorXr xp{nq + 2 * nv};
        for (index_t i = 0; i < nq; ++i) xp[i] = q[i];
        for (index_t i = 0; i < nv; ++i) xp[nq + i] = v[i], xp[nq + nv + i] = tau[i];
        EXPECT_TRUE(f.DependentVariableSize() == nv);
        EXPECT_TRUE(MaxAbsDiff(f(xp), a) < 1e-9);
        EXPECT_TRUE(f.TestJacobian(xp));
        std::printf("taped ABA: jacobian nnz %td of %td\n", f.Jacobian(xp).nonZeros(), nv * (nq + 2 * nv));
    }
    std::printf(g_failures ? "FAILED %d\n" : "PASSED\n", g_failures);
    return g_failures ? 1 : 0;
}
