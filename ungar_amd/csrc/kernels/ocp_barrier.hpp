// ungar_amd :: device helpers shared by the batched SQP kernels (ocp_riccati.hip, ocp_shooting.hip): the relaxed barrier of the soft
// inequality constraints and its derivatives (reference include/ungar/optimization/soft_inequality_constraint.hpp:77-205), a wavefront sum.
#pragma once

#include <hip/hip_runtime.h>

#include "ocp_sqp.hpp"

namespace ungar_amd::kernels {

/// Wavefront sum (64 lanes), result in every lane.
__device__ __forceinline__ double WaveSum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ double Barrier(const BarrierParams& p, double z) {
    const double k = p.stiffness, eps = p.epsilon;
    if (p.type == 1) {  // relaxed log barrier (soft_inequality_constraint.hpp:98-105)
        if (z >= eps) return -k * log(z);
        const double t = (z - 2.0 * eps) / eps;
        return 0.5 * k * (t * t - 1.0) - k * log(eps);
    }
    // relaxed polynomial barrier (:131-190): quadratic below 0, cubic on [0, eps), 0 above
    const double a1 = k, b1 = -0.5 * k * eps;
    const double c1 = -(1.0 / 3.0) * (-b1 - a1 * eps) * eps - 0.5 * a1 * eps * eps - b1 * eps;
    if (z < 0.0) return 0.5 * a1 * z * z + b1 * z + c1;
    if (z < eps) {
        const double a2 = (-b1 - a1 * eps) / (eps * eps);
        return (1.0 / 3.0) * a2 * z * z * z + 0.5 * a1 * z * z + b1 * z + c1;
    }
    return 0.0;
}

__device__ __forceinline__ double BarrierD1(const BarrierParams& p, double z) {
    const double k = p.stiffness, eps = p.epsilon;
    if (p.type == 1) return z >= eps ? -k / z : k * (z - 2.0 * eps) / (eps * eps);
    const double a1 = k, b1 = -0.5 * k * eps;
    if (z < 0.0) return a1 * z + b1;
    if (z < eps) return (-b1 - a1 * eps) / (eps * eps) * z * z + a1 * z + b1;
    return 0.0;
}
__device__ __forceinline__ double BarrierD2(const BarrierParams& p, double z) {
    const double k = p.stiffness, eps = p.epsilon;
    if (p.type == 1) return z >= eps ? k / (z * z) : k / (eps * eps);
    const double a1 = k, b1 = -0.5 * k * eps;
    if (z < 0.0) return a1;
    if (z < eps) return 2.0 * (-b1 - a1 * eps) / (eps * eps) * z + a1;
    return 0.0;
}


/// Acceptance test of backtracking_line_search.hpp:116-151 for one candidate.
__device__ inline bool StepAcceptable(double theta, double phi, double slope, double thetaNext, double phiNext, double alpha, double thetaMin, double thetaMax, double eta,
                                      double gammaPhi, double gammaTheta) {
    if (thetaNext > thetaMax) return thetaNext < (1.0 - gammaTheta) * theta;
    if (fmax(theta, thetaNext) < thetaMin && slope < 0.0) return phiNext < phi + eta * alpha * slope;
    return phiNext < (1.0 - gammaPhi) * phi || thetaNext < (1.0 - gammaTheta) * theta;
}

}  // namespace ungar_amd::kernels
