// ungar_amd :: soft SQP for a BATCH of independent shooting problems, entirely on the device -- the batched counterpart of
// SoftSQPOptimizer (soft_sqp.hpp in this directory; reference include/ungar/optimization/soft_sqp.hpp:42-283).
//
// The reference solves ONE optimal-control problem per SoftSQPOptimizer::Optimize call: whole-horizon objective / equality /
// inequality functions, one OSQP instance (soft_sqp.hpp:62-112, 143-158).  An MPC fleet, a sampling-based planner or a parameter
// sweep solves thousands of instances of the SAME problem; here they advance together, one launch per step of the iteration:
//
//   user side     the problem is described by its STAGE functions -- the lambdas the reference's examples inline N times into
//                 their whole-horizon tapes (example/mpc/quadrotor.example.cpp:126-190, 196-291) -- as ordinary
//                 Ungar::Autodiff::Functions over one node's variables:
//                     dynamics    [x | u ; w | p] -> x+                         (JACOBIAN)
//                     carry       [x | u ; w | p] -> c+   (optional)            (JACOBIAN)
//                     cost        [c | x | u ; w | p] -> 1 value                (ALL)       row N: the terminal cost
//                     equality    [c | x | u ; w | p] -> ne rows (= 0)          (JACOBIAN)  optional
//                     inequality  [c | x | u ; w | p] -> nh rows (<= 0)         (JACOBIAN)  optional, behind the relaxed barrier
//                 with w = parameters of the knot (references, contact flags, weights that switch terms off at k = 0 / k = N) and
//                 p = parameters of the instance.  c carries a quantity of the PREVIOUS knot into the stage, which is how the
//                 cross-knot terms of the reference's OCPs become stage-local: the input-rate cost 1e-6 |u_k - u_{k-1}|^2
//                 (quadrotor.example.cpp:222-227, rc_car.example.cpp:216-220: carry the inputs) and the foot-contact rows
//                 pFoot_k - pFoot_{k-1} (quadruped.example.cpp:279-304: carry the foot positions).
//   device side   node rows [c|x|u|w|p] of all instances in one array; per iteration (SoftSQPOptimizer::Optimize's loop body):
//                 stage derivatives (ungar_function_*_nodes) -> ungar_shooting_assemble -> ungar_ocp_riccati_solve (the exact
//                 solution of the QP the reference hands to OSQP, stage equality rows included) -> merit terms -> all candidate steps
//                 of the backtracking search as one stacked batch -> per-instance selection and stopping rule.
// Host code is plain C++20 over the C ABI (include/ungar_amd.h); no HIP headers.
// HOST SYNCHRONISATION.  By default the line search is STAGED: the first two candidate steps go to every instance; if some instance accepted neither, Iterate()
// downloads one 4-byte counter (a stream synchronisation) and offers the remaining steps to the listed instances only -- 2.16 -> 1.45 ms per quadrotor iteration.  An
// iteration is therefore NOT free of host round trips by default (it was before ABI 5): it cannot be captured into a HIP graph as is, and its timing depends on the data.
// SetFirstLineSearchStage(0) restores the fully asynchronous schedule (all candidate steps for every instance in one stacked evaluation, no read-back).
#pragma once

#include <cstdint>
#include <cstdlib>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../autodiff/function.hpp"
#include "backtracking_line_search.hpp"
#include "soft_sqp.hpp"

namespace Ungar {

/// One shooting problem in stage form.  Sizes: state nx, input nu, carried quantity nc (0: none), knot parameters nw, instance
/// parameters np; every function takes nw + np parameters (whether it uses them or not), so that one node row serves all of them.
struct ShootingProblem {
    index_t horizon = 0, stateSize = 0, inputSize = 0, carrySize = 0, knotParameterSize = 0, instanceParameterSize = 0;
    bool carryInputs = false;  ///< c_{k+1} = u_k without a carry function (carrySize == inputSize)
    std::optional<Autodiff::Function> dynamics, carry, cost, equality, inequality;

    index_t RowSize() const {
        return carrySize + stateSize + inputSize + knotParameterSize + instanceParameterSize;
    }
    index_t CarryOffset() const { return 0; }
    index_t StateOffset() const { return carrySize; }
    index_t InputOffset() const { return carrySize + stateSize; }
    index_t KnotParameterOffset() const { return carrySize + stateSize + inputSize; }
    index_t InstanceParameterOffset() const { return carrySize + stateSize + inputSize + knotParameterSize; }
};

/// Stacked evaluation whose row image holds the VARIABLES only: `candidates` x `nodes` stacked nodes, candidate-major; the parameters of node i of every candidate
/// at parameters[e * stride + i] (ungar_function_*_nodes_split: instance := candidate, knot := node, parameter instance stride 0).  candidates = 0: whole rows.
struct ShootingSplitImage {
    index_t candidates = 0, nodes = 0;
    real_t* parameters = nullptr;
    index_t stride = 0;
};

class BatchedSoftSQPOptimizer {
  public:
    /// Same parameters as SoftSQPOptimizer (reference soft_sqp.hpp:44-60) plus the number of instances.
    BatchedSoftSQPOptimizer(ShootingProblem problem, const index_t batch, const bool verbose = false, const real_t constraintViolationMultiplier = 1.0,
                            const index_t maxIterations = 10, const real_t stiffness = 100.0, const real_t epsilon = 2e-5,
                            const RelaxedBarrierType relaxedBarrierType = RelaxedBarrierType::POLY, const BacktrackingLineSearch::Parameters& lineSearchParameters = {})
        : _p{std::move(problem)}, _batch{batch}, _verbose{verbose}, _multiplier{constraintViolationMultiplier}, _maxIterations{maxIterations}, _ls{lineSearchParameters} {
        if (ungar_abi_version() != UNGAR_AMD_ABI_VERSION)  // this header was compiled against another revision of the C ABI than the library that is loaded
            throw std::runtime_error("BatchedSoftSQPOptimizer: libungar_amd ABI version " + std::to_string(ungar_abi_version()) + ", header " + std::to_string(UNGAR_AMD_ABI_VERSION));
        _barrier.type = relaxedBarrierType == RelaxedBarrierType::LOG ? UNGAR_BARRIER_LOG : UNGAR_BARRIER_POLY;
        _barrier.reserved = 0;
        _barrier.stiffness = stiffness;
        _barrier.epsilon = epsilon;
        Validate();
        _dims = {_p.stateSize, _p.inputSize, _p.carrySize, _p.knotParameterSize, _p.instanceParameterSize, _p.horizon, _batch, _p.carryInputs ? 1 : 0, 0};
        _alphas = CandidateSteps(_ls);  // backtracking_line_search.hpp:116-151; more than 16 candidate steps are evaluated in groups of 16, largest first
        Allocate();
    }
    BatchedSoftSQPOptimizer(const BatchedSoftSQPOptimizer&) = delete;
    BatchedSoftSQPOptimizer& operator=(const BatchedSoftSQPOptimizer&) = delete;
    ~BatchedSoftSQPOptimizer() {
        for (void* ptr : _owned) (void)ungar_device_free(ptr);
    }

    inline static index_t maxStackedCandidates = 16;  // read at construction (see StackedCandidates)
    inline static bool fuseStageValues = true;        // read at construction: the values of all stage functions at the trial points in one launch (BuildStageValues)
    bool StageValuesFused() const { return _stageValues.has_value(); }
    const ShootingProblem& Problem() const { return _p; }
    /// Kernels the QP step of this problem runs (ungar_ocp_riccati_route / ungar_shooting_assemble_route: 0 run-time-size kernels, 1 compiled into the library,
    /// 2 instantiated for this problem's sizes by the kernel factory, 3 run-time-size one-wavefront assembly).
    int RiccatiRoute() const { return _riccatiRoute; }
    int AssembleRoute() const { return _assembleRoute; }
    index_t Batch() const { return _batch; }
    index_t RowSize() const { return _p.RowSize(); }
    /// Doubles of the host image of all node rows: batch x (horizon + 1) x RowSize(), row (b, k) at ((b (N + 1)) + k) RowSize().
    index_t RowsSize() const { return _batch * (_p.horizon + 1) * _p.RowSize(); }

    /// Uploads the node rows [c | x | u | w | p] of every instance and the measured states (batch x nx); the carried slots of rows
    /// 1..N are then made consistent with (x, u) of the previous row (row 0's carried slots are the caller's: the quantity measured
    /// before the horizon, e.g. measured foot positions).  All instances become active.
    void SetRows(const real_t* rows, const real_t* measuredStates) {
        Check(ungar_device_synchronize());  // the uploads below are synchronous copies on the null stream: nothing launched on the user's stream may still read the rows
        Check(ungar_device_upload(_rows, rows, Bytes(RowsSize())));
        Check(ungar_device_upload(_xm, measuredStates, Bytes(_batch * _p.stateSize)));
        RefreshCarried();
        std::vector<int32_t> ones(static_cast<std::size_t>(_batch), 1);
        Check(ungar_device_upload(_active, ones.data(), static_cast<int64_t>(ones.size() * sizeof(int32_t))));
    }
    void GetRows(real_t* rows) const { Check(ungar_device_download(rows, _rows, Bytes(RowsSize()))); }
    /// Device pointer of the node rows, for callers that fill them on the device (stream-ordered with Iterate()).  ANY write through it is picked up by the next
    /// Iterate(): the parameter image the stage functions read ([w | p] of every row) is refreshed from the rows first -- an MPC loop that moves its references or
    /// parameters on the device between iterations must not optimise against the old ones.  After writing (x, u) of a problem with carried quantities call
    /// RefreshCarried() (it rewrites the carried slots of rows 1..N, which the caller may not want on every hand-out of the pointer).
    real_t* DeviceRows() const {
        _parametersStale = true;
        return _rows;
    }
    real_t* DeviceMeasuredStates() const { return _xm; }

    void RefreshCarried() {
        if (_p.carrySize == 0) {
            RefreshParameterImage();
            return;
        }
        if (_p.carryInputs) {  // u_k into the carried slots of row k + 1, in place (the last search direction and the trial rows are left alone)
            Check(ungar_shooting_refresh_carried_inputs(&_dims, _rows, _stream));
        } else {
            CarryValues(_rows, _batch);
        }
        RefreshParameterImage();
    }
    /// The knot and instance parameters [w | p] of all node rows as ONE unit-fastest image (element e of node i at _parameters[e * _parameterStride + i]): the
    /// stage functions read their parameters from it at every candidate step of the line search and at the current point, so the per-iteration images of the rows
    /// hold the variables [c | x | u] only.  Refreshed with the rows (SetRows, RefreshCarried -- call the latter after writing rows on the device).
    void RefreshParameterImage() {
        _parametersStale = false;
        if (!_parameters) return;
        const index_t nd = Nz() + _p.inputSize;
        const real_t zero = 0.0;
        Check(ungar_shooting_trial_elements(&_dims, _rows, _dZ, _dU, &zero, 1, nullptr, 0, nd, _p.RowSize() - nd, _parameters, _parameterStride, _stream));
    }

    /// One SQP iteration of every active instance (the body of the reference's loop, soft_sqp.hpp:68-99).
    void Iterate() {
        const index_t N = _p.horizon, B = _batch, K = static_cast<index_t>(_alphas.size());
        if (_parametersStale) RefreshParameterImage();  // the rows were handed out for device-side writes since the image was taken (DeviceRows)
        // ---- derivatives at the current rows.  The stage kernels run one lane per node: they read a unit-fastest image of the rows (the trial-row kernel with
        // one candidate of step 0 writes it: a copy) with coalesced loads, as they do in the line search -- a node-major row is 64 separate 8-byte
        // transactions per load instruction
        real_t* at = _rows;
        index_t atStride = 0;
        ShootingSplitImage split;  // the image holds the variables only, the parameters come from their own image
        if (_trialStride > 0) {
            const real_t zero = 0.0;
            if (_parameters) split = {1, B * (N + 1), _parameters, _parameterStride};
            Check(ungar_shooting_trial_elements(&_dims, _rows, _dZ, _dU, &zero, 1, nullptr, 0, 0, split.candidates ? Nz() + _p.inputSize : 0, _trial, _trialStride, _stream));
            at = _trial;
            atStride = _trialStride;
        }
        Evaluate(*_p.dynamics, 0, at, _p.StateOffset(), _f, B * (N + 1), atStride, split);
        Evaluate(*_p.dynamics, 1, at, _p.StateOffset(), _fJ, B * (N + 1), atStride, split);
        if (_p.carry) Evaluate(*_p.carry, 1, at, _p.StateOffset(), _cJ, B * (N + 1), atStride, split);
        Evaluate(*_p.cost, 0, at, 0, _l, B * (N + 1), atStride, split);
        Evaluate(*_p.cost, 1, at, 0, _lg, B * (N + 1), atStride, split);
        Evaluate(*_p.cost, 2, at, 0, _lH, B * (N + 1), atStride, split);
        if (_p.inequality) {
            Evaluate(*_p.inequality, 0, at, 0, _h, B * (N + 1), atStride, split);
            Evaluate(*_p.inequality, 1, at, 0, _hJ, B * (N + 1), atStride, split);
        }
        if (_p.equality) {
            Evaluate(*_p.equality, 0, at, 0, _e, B * (N + 1), atStride, split);
            Evaluate(*_p.equality, 1, at, 0, _eJ, B * (N + 1), atStride, split);
        }
        // ---- QP data and solve (soft_sqp.hpp:143-158)
        ungar_shooting_assemble_args a{};
        a.dims = _dims;
        a.rows = _rows;
        a.xm = _xm;
        a.f = _f;
        a.f_jac = _fJ;
        a.carry_jac = _cJ;
        a.cost_grad = _lg;
        a.cost_hes = _lH;
        a.h = _h;
        a.h_jac = _hJ;
        a.eq_jac = _eJ;
        a.f_pattern = _pf;
        a.carry_pattern = _pc;
        a.cost_grad_pattern = _pg;
        a.cost_hes_pattern = _pH;
        a.h_pattern = _ph;
        a.eq_pattern = _pe;
        a.nh = Nh();
        a.ne = Ne();
        a.barrier = _barrier;
        a.regularization = 1e-6;  // soft_sqp.hpp:149-151
        a.AB = _AB;
        a.b = _b;
        a.W = _W;
        a.w = _w;
        a.E = _E;
        a.dz0 = _dz0;
        const bool eliminate = _eliminateEqualities && Ne() > 0;
        a.eliminate_equalities = eliminate ? 1 : 0;
        a.eq = _e;
        a.eq_reduced = _er;
        a.eq_pivots = _pivots;
        Check(ungar_shooting_assemble(&a, _stream));
        const index_t nz = Nz(), nu = _p.inputSize, nd = nz + nu;
        ungar_ocp_qp q{};
        q.nx = nz;
        q.nu = nu;
        q.horizon = N;
        q.batch = B;
        q.jac = {_AB, N * nz * nd, nz * nd, 1};
        q.b = {_b, N * nz, nz, 1};
        q.hess = {_W, (N + 1) * nd * nd, nd * nd, 1};
        q.grad = {_w, (N + 1) * nd, nd, 1};
        q.hess_terminal = {_W + N * nd * nd, (N + 1) * nd * nd, 0, 1};  // the z block of row N
        q.hess_terminal_ld = nd;
        q.grad_terminal = {_w + N * nd, (N + 1) * nd, 0, 1};
        q.dx0 = {_dz0, nz, 0, 1};
        q.dX = {_dZ, (N + 1) * nz, nz, 1};
        q.dU = {_dU, N * nu, nu, 1};
        q.workspace = _workspace;
        q.workspace_doubles = _workspaceDoubles;
        q.regularization = 0.0;  // already on the decision variables' diagonal (the carried slots take none)
        q.status = _status;
        q.ne = eliminate ? 0 : Ne();
        if (q.ne > 0) {
            q.eq = {_E, N * Ne() * nd, Ne() * nd, 1};
            q.eq_values = {_e, (N + 1) * Ne(), Ne(), 1};
        }
        Check(ungar_ocp_riccati_solve(&q, _stream));
        if (eliminate) Check(ungar_shooting_recover_inputs(&_dims, Ne(), _E, _er, _pivots, _dZ, _dU, _status, _stream));
        // ---- merit terms at the current point (soft_sqp.hpp:68-87) and along all candidate steps at once
        ungar_shooting_merit_args m{};
        m.dims = _dims;
        m.rows = _rows;
        m.xm = _xm;
        m.f = _f;
        m.cost = _l;
        m.h = _h;
        m.eq = _e;
        m.nh = Nh();
        m.ne = Ne();
        m.barrier = _barrier;
        m.violation_multiplier = _multiplier;
        m.cost_grad = _lg;
        m.cost_grad_pattern = _pg;
        m.dZ = _dZ;
        m.dU = _dU;
        m.theta = _theta0;
        m.phi = _phi0;
        m.objective = _obj0;
        m.slope = _slope;
        m.period = 0;
        Check(ungar_shooting_merit(&m, _stream));
        // candidate steps, largest first, in STAGES: the first few for every instance, the rest only for the instances that accepted none of them -- the
        // select kernel lists those, the later stages stack (candidates x listed instances) trial points instead of (candidates x all).  Most iterations
        // of a warm-started MPC take a full or half step: evaluating all 14 default candidates for everybody was 1.1-1.5 ms of the quadruped's iteration.
        // One 4-byte read-back per stage that leaves somebody unresolved.  The stages change what is evaluated, never which step an instance takes
        // (backtracking_line_search.hpp:116-151: the first acceptable candidate in descending order).
        // (more than kStacked candidates at once -- a gamma_alpha close to 1 or a tiny alpha_min: the reference accepts any parameters -- go in groups of kStacked)
        const ungar_line_search_parameters ls{_ls.alphaMin, _ls.thetaMin, _ls.thetaMax, _ls.eta, _ls.gammaPhi, _ls.gammaTheta, _ls.gammaAlpha};
        index_t listed = 0;  // 0: all instances
        int32_t *list = nullptr, *nextList = _listA;
        // Where most instances go on to the later stages anyway (a cold start, a problem whose full steps are rarely acceptable: nine tenths of 4096 RC-car
        // instances in every iteration), the first stage buys nothing and costs its launches and a read-back: after such an iteration the following ones offer all
        // candidates to everybody at once, and every eighth iteration looks again.  What is evaluated changes, never which step an instance takes.
        static const std::vector<index_t> allAtOnce;
        const bool probing = !_adaptiveStages || !_mostGoOn || _sinceProbe >= 8;
        const std::vector<index_t>& stages = probing ? _stages : allAtOnce;
        _sinceProbe = probing ? 0 : _sinceProbe + 1;
        std::size_t stage = 0;
        for (index_t begin = 0; begin < K; ++stage) {
            const index_t left = K - begin, sized = stage < stages.size() && stages[stage] > 0 ? stages[stage] : left;
            const index_t wanted = sized < left ? sized : left;
            const index_t count = wanted < kStacked ? wanted : kStacked;
            const bool last = begin + count == K;
            const index_t stacked = listed > 0 ? listed : B;  // instances with trial points in this stage
            // trial rows UNIT-FASTEST (element e of stacked node i at _trial[e * stride + i]): the stage functions read them coalesced and touch
            // only the elements they use
            const index_t stride = _trialStride;
            // The stacked (candidate, node) pairs of a stage share their parameters across the candidates: only the variables [c | x | u] of the trial points are
            // written (a third of the RC car's row), the stage functions take the parameters from an image of their own -- the image of all nodes for a stage
            // over all instances, an image gathered once per stage for the LISTED instances.
            ShootingSplitImage split;
            if (_parameters) {
                split = {count, stacked * (N + 1), _parameters, _parameterStride};
                if (listed > 0) {
                    const real_t zero = 0.0;
                    const index_t nd = Nz() + _p.inputSize;
                    Check(ungar_shooting_trial_elements(&_dims, _rows, _dZ, _dU, &zero, 1, list, listed, nd, _p.RowSize() - nd, _listedParameters, _parameterStride, _stream));
                    split.parameters = _listedParameters;
                }
            }
            Check(ungar_shooting_trial_elements(&_dims, _rows, _dZ, _dU, _alphas.data() + begin, count, list, listed, 0, split.candidates ? Nz() + _p.inputSize : 0, _trial, stride, _stream));
            if (_p.carry) CarryValues(_trial, count * stacked, stride, split);
            if (_stageValues) {
                Evaluate(*_stageValues, 0, _trial, 0, _fT, count * stacked * (N + 1), stride, split);
            } else {
                Evaluate(*_p.dynamics, 0, _trial, _p.StateOffset(), _fT, count * stacked * (N + 1), stride, split);
                Evaluate(*_p.cost, 0, _trial, 0, _lT, count * stacked * (N + 1), stride, split);
                if (_p.inequality) Evaluate(*_p.inequality, 0, _trial, 0, _hT, count * stacked * (N + 1), stride, split);
                if (_p.equality) Evaluate(*_p.equality, 0, _trial, 0, _eT, count * stacked * (N + 1), stride, split);
            }
            ungar_shooting_merit_args t = m;
            t.dims.batch = count * stacked;
            t.rows = _trial;
            t.f = _fT;
            t.cost = _lT;
            t.h = _hT;
            t.eq = _eT;
            t.value_stride = _stageValues ? _valueWidth : 0;
            t.cost_grad = nullptr;
            t.slope = nullptr;
            t.theta = _thetaT;
            t.phi = _phiT;
            t.objective = _objT;
            t.period = stacked;
            t.instances = list;
            t.rows_stride = stride;
            Check(ungar_shooting_merit(&t, _stream));
            if (!last) Check(ungar_device_zero(_unresolved, static_cast<int64_t>(sizeof(int32_t)), _stream));
            Check(ungar_shooting_select_listed(&_dims, &ls, _alphas.data() + begin, count, _theta0, _phi0, _obj0, _slope, _thetaT, _phiT, _objT, _accepted, _active, _status, _rows, _trial, stride,
                                               (begin > 0 ? UNGAR_SEARCH_NOT_FIRST : 0) | (last ? 0 : UNGAR_SEARCH_NOT_LAST), _unresolved, list, listed, last ? nullptr : nextList, _stream));
            begin += count;
            if (!last) {
                int32_t count = 0;  // (polled behind the stream: a device-wide wait and a pageable copy around these 4 bytes were 50 us of every staged iteration)
                Check(ungar_device_read_polled(&count, _unresolved, static_cast<int64_t>(sizeof count), _stream));
                const index_t unresolved = count;
                if (stage == 0) _mostGoOn = 3 * unresolved > 2 * B;
                if (unresolved == 0) break;  // everybody took one of the steps offered so far (or had stopped): the search is closed
                list = nextList;
                listed = unresolved;
                nextList = list == _listA ? _listB : _listA;
            }
        }
        ++_iterations;
    }

    /// Up to maxIterations iterations; an instance stops on its own when no step is acceptable or its objective decreases by less
    /// than 1e-6 (soft_sqp.hpp:88-99).  Returns the number of iterations launched (all of them when nobody asks: with thousands of
    /// instances somebody is always still active, and finished instances cost nothing but their share of the launches).
    index_t Optimize(const bool stopWhenAllInstancesAreDone = false) {
        _iterations = 0;
        for (index_t it = 0; it < _maxIterations; ++it) {
            if (_verbose) UNGAR_LOG(trace, "Starting batched soft SQP iteration {}...", it);
            Iterate();
            if (stopWhenAllInstancesAreDone) {
                bool any = false;
                for (const int32_t v : Active()) any = any || v != 0;
                if (!any) break;
            }
        }
        return _iterations;
    }

    // ---- results of the last iteration (each call synchronises) --------------------------------------------------------------
    std::vector<real_t> AcceptedStepSizes() const { return Download<real_t>(_accepted, _batch); }
    std::vector<int32_t> Active() const { return Download<int32_t>(_active, _batch); }
    std::vector<int32_t> QpStatus() const { return Download<int32_t>(_status, _batch); }
    /// Search direction of the last QP: dZ batch x (N + 1) x (nc + nx), dU batch x N x nu.
    std::vector<real_t> StateSteps() const { return Download<real_t>(_dZ, _batch * (_p.horizon + 1) * Nz()); }
    std::vector<real_t> InputSteps() const { return Download<real_t>(_dU, _batch * _p.horizon * _p.inputSize); }
    std::vector<real_t> ConstraintViolations() const { return Download<real_t>(_theta0, _batch); }
    std::vector<real_t> Objectives() const { return Download<real_t>(_obj0, _batch); }
    index_t Iterations() const { return _iterations; }
    /// The QP of the last iteration as assembled on the device (diagnostics; whole arrays, node-major): with nz = nc + nx, nd = nz + nu,
    /// AB batch x N x nz x nd, b batch x N x nz, W batch x (N + 1) x nd x nd (upper triangles), w batch x (N + 1) x nd,
    /// E batch x N x ne x nd, e batch x (N + 1) x ne, dz0 batch x nz.
    struct Qp {
        std::vector<real_t> AB, b, W, w, E, e, dz0;
    };
    Qp AssembledQp() const {
        const index_t N = _p.horizon, B = _batch, nz = Nz(), nd = nz + _p.inputSize;
        Qp q;
        q.AB = Download<real_t>(_AB, B * N * nz * nd);
        q.b = Download<real_t>(_b, B * N * nz);
        q.W = Download<real_t>(_W, B * (N + 1) * nd * nd);
        q.w = Download<real_t>(_w, B * (N + 1) * nd);
        if (Ne() > 0) {
            q.E = Download<real_t>(_E, B * N * Ne() * nd);
            q.e = Download<real_t>(_e, B * (N + 1) * Ne());
        }
        q.dz0 = Download<real_t>(_dz0, B * nz);
        return q;
    }
    void SetStream(void* hipStream) { _stream = hipStream; }
    /// Offer the first `candidates` step sizes (1, 1/2, ...) to every instance and evaluate the remaining ones only for the instances that accepted none
    /// of them -- every candidate costs a pass of the stage functions over all nodes of the instances it is offered to.  Costs one 4-byte
    /// read-back per iteration in which somebody needs a smaller step; 0 evaluates all candidates for everybody at once, without any host decision.
    void SetFirstLineSearchStage(const index_t candidates) {
        _stages.assign(1, candidates);
        _adaptiveStages = false;
    }
    /// The general form: stage s offers the next candidates[s] steps (to every instance for s = 0, to the instances still unresolved afterwards), a last stage the
    /// remaining ones.  Default {2}: 1 and 1/2 for everybody, the rest (down to alphaMin) for those who took neither.
    void SetLineSearchStages(std::vector<index_t> candidates) {
        _stages = std::move(candidates);
        _adaptiveStages = false;
    }
    /// Default on: an iteration whose first stage left more than two thirds of the instances unresolved is followed by iterations that offer all candidates to
    /// everybody at once (no first stage, no read-back), every eighth of which runs the stages again to see whether that is still so.  Setting the stages by hand
    /// switches it off.
    void SetAdaptiveLineSearchStages(const bool on) { _adaptiveStages = on; }
    /// Stage equality rows: eliminated node by node before the recursion (default; the sequential chain then carries no constraint
    /// block) or kept inside the Riccati recursion as the stage KKT block of every knot (false; same solution, for comparison).
    void EliminateEqualityRowsBeforeTheRecursion(const bool on) { _eliminateEqualities = on; }
    void SetLineSearchParameters(const BacktrackingLineSearch::Parameters& parameters) {
        const std::vector<real_t> alphas = CandidateSteps(parameters);
        const std::size_t cap = static_cast<std::size_t>(kStacked);
        // the stacked buffers hold _stackedCapacity = min(candidates at construction, 16) trial points per instance -- compared with what was ALLOCATED, not with the
        // current list (a call that shrank the list must not make a later call with the original parameters fail)
        if ((alphas.size() < cap ? alphas.size() : cap) > static_cast<std::size_t>(_stackedCapacity))
            throw std::invalid_argument("BatchedSoftSQPOptimizer: more candidate steps per stacked evaluation than the buffers allocated at construction hold");
        _ls = parameters;
        _alphas = alphas;
    }

    /// The candidate steps 1, gamma, gamma^2, ... >= alphaMin of the backtracking search (backtracking_line_search.hpp:116-151).  The reference's loop does not end
    /// for gammaAlpha >= 1 or alphaMin <= 0 (alpha never drops below alphaMin, or underflows to 0 >= 0); here such parameters are refused, and so is a list of more
    /// than 1024 steps.
    static std::vector<real_t> CandidateSteps(const BacktrackingLineSearch::Parameters& parameters) {
        if (!(parameters.gammaAlpha > 0.0 && parameters.gammaAlpha < 1.0) || !(parameters.alphaMin > 0.0))
            throw std::invalid_argument("BatchedSoftSQPOptimizer: the line search needs 0 < gammaAlpha < 1 and alphaMin > 0");
        std::vector<real_t> alphas;
        for (real_t alpha = 1.0; alpha >= parameters.alphaMin; alpha *= parameters.gammaAlpha) {
            if (alphas.size() == 1024) throw std::invalid_argument("BatchedSoftSQPOptimizer: more than 1024 candidate steps (gammaAlpha too close to 1 for this alphaMin)");
            alphas.push_back(alpha);
        }
        if (alphas.empty()) throw std::invalid_argument("BatchedSoftSQPOptimizer: the line search has no candidate step (alphaMin > 1)");
        return alphas;
    }

  private:
    static void Check(int code) {
        if (code != UNGAR_OK) throw std::runtime_error(std::string("ungar_amd: ") + ungar_last_error());
    }
    static int64_t Bytes(index_t doubles) { return static_cast<int64_t>(doubles) * static_cast<int64_t>(sizeof(real_t)); }
    index_t Nz() const { return _p.carrySize + _p.stateSize; }
    index_t Nh() const { return _p.inequality ? _p.inequality->DependentVariableSize() : 0; }
    index_t Ne() const { return _p.equality ? _p.equality->DependentVariableSize() : 0; }

    void Validate() const {
        const index_t par = _p.knotParameterSize + _p.instanceParameterSize, nxu = _p.stateSize + _p.inputSize, nd = _p.carrySize + nxu;
        auto bad = [](const char* what) { throw std::invalid_argument(std::string("BatchedSoftSQPOptimizer: ") + what); };
        if (_p.horizon < 1 || _p.stateSize < 1 || _p.inputSize < 1 || _p.carrySize < 0 || _batch < 1) bad("sizes must be positive");
        if (!_p.dynamics || !_p.cost) bad("dynamics and cost are required");
        if (_p.carryInputs && (_p.carrySize != _p.inputSize || _p.carry)) bad("carryInputs needs carrySize == inputSize and no carry function");
        if (!_p.carryInputs && (_p.carrySize > 0) != _p.carry.has_value()) bad("a carried quantity needs a carry function (or carryInputs)");
        auto check = [&](const Autodiff::Function& f, index_t n, index_t m, bool jac, bool hes, const char* name) {
            if (f.IndependentVariableSize() != n || f.ParameterSize() != par || (m > 0 && f.DependentVariableSize() != m) || (jac && !f.ImplementsJacobian()) ||
                (hes && !f.ImplementsHessian()))
                throw std::invalid_argument(std::string("BatchedSoftSQPOptimizer: stage function '") + name + "' has the wrong sizes or lacks a derivative");
        };
        check(*_p.dynamics, nxu, _p.stateSize, true, false, "dynamics");
        if (_p.carry) check(*_p.carry, nxu, _p.carrySize, true, false, "carry");
        check(*_p.cost, nd, 1, true, true, "cost");
        if (_p.equality) check(*_p.equality, nd, 0, true, false, "equality");
        if (_p.inequality) check(*_p.inequality, nd, 0, true, false, "inequality");
    }

    template <class T>
    T* Device(index_t count) {
        void* ptr = nullptr;
        Check(ungar_device_malloc(&ptr, static_cast<int64_t>(count > 0 ? count : 1) * static_cast<int64_t>(sizeof(T))));
        _owned.push_back(ptr);
        return static_cast<T*>(ptr);
    }
    ungar_stage_pattern UploadPattern(const Autodiff::Function& f, bool hessian) {
        const int32_t *rows = nullptr, *cols = nullptr;
        int64_t nnz = 0;
        Check(hessian ? ungar_function_hessian_sparsity(f.Handle(), &rows, &cols, &nnz) : ungar_function_jacobian_sparsity(f.Handle(), &rows, &cols, &nnz));
        int32_t* dr = Device<int32_t>(nnz);
        int32_t* dc = Device<int32_t>(nnz);
        Check(ungar_device_upload(dr, rows, nnz * static_cast<int64_t>(sizeof(int32_t))));
        Check(ungar_device_upload(dc, cols, nnz * static_cast<int64_t>(sizeof(int32_t))));
        return {dr, dc, nnz};
    }
    void Allocate() {
        const index_t N = _p.horizon, B = _batch, nv = _p.RowSize(), nx = _p.stateSize, nu = _p.inputSize, nz = Nz(), nd = nz + nu;
        const index_t K = static_cast<index_t>(_alphas.size()) < kStacked ? static_cast<index_t>(_alphas.size()) : kStacked;  // candidates evaluated at once
        _stackedCapacity = K;
        // the QP kernels of this problem's stage sizes: the register-resident Riccati recursion and the one-wavefront assembly kernel are compiled into the library
        // for the reference's own problems and instantiated by the kernel factory for any other size they fit -- here, once, not inside the first iteration
        _assembleRoute = ungar_shooting_assemble_route(nz, nu, _eliminateEqualities ? Ne() : 0, Nh(), 1);
        _riccatiRoute = ungar_ocp_riccati_route(nz, nu, _eliminateEqualities ? 0 : Ne(), 1);
        // A route of 0 where the sizes FIT the fast kernels means the kernel factory could not instantiate them on this machine (no hipcc, or the kernel sources
        // named by UNGAR_AMD_KERNEL_SOURCES are missing): the problem then runs the LDS-resident kernels at about half the rate.  Say so once, here.
        if (_riccatiRoute == 0 && ungar_ocp_riccati_route(nz, nu, _eliminateEqualities ? 0 : Ne(), 0) == 2)
            UNGAR_LOG(warn, "BatchedSoftSQPOptimizer: no register-resident Riccati kernel for stage sizes {} + {} on this machine (kernel factory: hipcc / kernel sources missing, see stderr); "
                            "taking the LDS-resident recursion", nz, nu);
        if (_assembleRoute == 0 && ungar_shooting_assemble_route(nz, nu, _eliminateEqualities ? Ne() : 0, Nh(), 0) == 2)
            UNGAR_LOG(warn, "BatchedSoftSQPOptimizer: no one-wavefront assembly kernel for stage sizes {} + {} with {} equality rows on this machine (kernel factory: hipcc / kernel sources "
                            "missing, see stderr); taking the workgroup kernel", nz, nu, Ne());
        const index_t nodes = B * (N + 1), stacked = K * nodes;
        _pf = UploadPattern(*_p.dynamics, false);
        if (_p.carry) _pc = UploadPattern(*_p.carry, false);
        _pg = UploadPattern(*_p.cost, false);
        _pH = UploadPattern(*_p.cost, true);
        if (_p.inequality) _ph = UploadPattern(*_p.inequality, false);
        if (_p.equality) _pe = UploadPattern(*_p.equality, false);
        _rows = Device<real_t>(nodes * nv);
        _xm = Device<real_t>(B * nx);
        _trialStride = _nodeMajorTrialRows ? 0 : ((stacked + 15) / 16 * 16 + 48);  // whole 128-byte segments, off the power-of-two channel strides
        _trial = Device<real_t>((_trialStride > 0 ? _trialStride : stacked) * nv);
        if (_trialStride > 0 && nv > nd) {  // one unit-fastest image of the parameter part of the rows
            _parameterStride = (nodes + 15) / 16 * 16 + 48;
            _parameters = Device<real_t>(_parameterStride * (nv - nd));
            _listedParameters = Device<real_t>(_parameterStride * (nv - nd));
        }
        _f = Device<real_t>(nodes * nx);
        _fJ = Device<real_t>(nodes * _pf.nnz);
        _cJ = _p.carry ? Device<real_t>(nodes * _pc.nnz) : nullptr;
        _l = Device<real_t>(nodes);
        _lg = Device<real_t>(nodes * _pg.nnz);
        _lH = Device<real_t>(nodes * _pH.nnz);
        _h = _p.inequality ? Device<real_t>(nodes * Nh()) : nullptr;
        _hJ = _p.inequality ? Device<real_t>(nodes * _ph.nnz) : nullptr;
        _e = _p.equality ? Device<real_t>(nodes * Ne()) : nullptr;
        _eJ = _p.equality ? Device<real_t>(nodes * _pe.nnz) : nullptr;
        // Values at the trial points of the line search: ONE function of the whole row that evaluates [f | cost | h | e] together -- the stage functions'
        // tapes stitched onto shared inputs (ungar_function_get_tape) and compiled like any other function -- so that a search stage is one value launch instead
        // of up to four (these kernels are launch-bound: ~10 us each whatever they compute).  Its output is one array of _valueWidth doubles per node;
        // the merit kernel reads the four parts through `value_stride`.
        BuildStageValues();
        if (_stageValues) {
            _valueWidth = nx + 1 + Nh() + Ne();
            _fT = Device<real_t>(stacked * _valueWidth);
            _lT = _fT + nx;
            _hT = _p.inequality ? _lT + 1 : nullptr;
            _eT = _p.equality ? _lT + 1 + Nh() : nullptr;
        } else {
            _fT = Device<real_t>(stacked * nx);
            _lT = Device<real_t>(stacked);
            _hT = _p.inequality ? Device<real_t>(stacked * Nh()) : nullptr;
            _eT = _p.equality ? Device<real_t>(stacked * Ne()) : nullptr;
        }
        _AB = Device<real_t>(B * N * nz * nd);
        _b = Device<real_t>(B * N * nz);
        _W = Device<real_t>(nodes * nd * nd);
        _w = Device<real_t>(nodes * nd);
        _E = _p.equality ? Device<real_t>(B * N * Ne() * nd) : nullptr;
        _er = _p.equality ? Device<real_t>(B * N * Ne()) : nullptr;
        _pivots = _p.equality ? Device<int32_t>(B * N * Ne()) : nullptr;
        _dz0 = Device<real_t>(B * nz);
        _dZ = Device<real_t>(nodes * nz);
        _dU = Device<real_t>(B * N * nu);
        _workspaceDoubles = ungar_ocp_riccati_workspace(nz, nu, N, B);
        if (_workspaceDoubles < 0) throw std::invalid_argument("BatchedSoftSQPOptimizer: sizes not supported by the batched QP solver");
        _workspace = Device<real_t>(_workspaceDoubles);
        _status = Device<int32_t>(B);
        _unresolved = Device<int32_t>(1);
        _listA = Device<int32_t>(_batch);
        _listB = Device<int32_t>(_batch);
        _active = Device<int32_t>(B);
        _theta0 = Device<real_t>(B);
        _phi0 = Device<real_t>(B);
        _obj0 = Device<real_t>(B);
        _slope = Device<real_t>(B);
        _accepted = Device<real_t>(B);
        _thetaT = Device<real_t>(K * B);
        _phiT = Device<real_t>(K * B);
        _objT = Device<real_t>(K * B);
        Check(ungar_device_zero(_W, Bytes(nodes * nd * nd), nullptr));  // the strict lower triangles are never written nor read
        Check(ungar_device_zero(_status, B * static_cast<int64_t>(sizeof(int32_t)), nullptr));
        Check(ungar_device_zero(_accepted, Bytes(B), nullptr));
        Check(ungar_device_synchronize());
    }

    /// what: 0 value, 1 sparse Jacobian, 2 sparse Hessian of `f` for `count` consecutive node rows starting at `rows`; the function's
    /// variables begin `offset` doubles into each row.
    /// Stitches the tapes of the stage functions onto the inputs of ONE function of the whole row ([c | x | u] independent, [w | p] parameters): input i of a stage
    /// function that reads the row from element `offset` on is input offset + i of the row.  Outputs [f (nx) | cost (1) | h (nh) | e (ne)].  Value only.
    /// inline static bool fuseStageValues = false switches it off (A/B, tests).
    void BuildStageValues() {
        if (!fuseStageValues || _trialStride <= 0) return;
        struct Member {
            const Autodiff::Function* f;
            index_t offset;
        };
        std::vector<Member> members{{&*_p.dynamics, _p.StateOffset()}, {&*_p.cost, 0}};
        if (_p.inequality) members.push_back({&*_p.inequality, 0});
        if (_p.equality) members.push_back({&*_p.equality, 0});
        const index_t nv = _p.RowSize(), nd = Nz() + _p.inputSize;
        std::vector<ungar_tape_node> nodes;
        std::vector<int32_t> outputs;
        std::string folder, name = "stage_values";
        for (const Member& mem : members) {
            const ungar_tape_node* src = nullptr;
            const int32_t* out = nullptr;
            const char* dir = nullptr;
            int64_t count = 0;
            Check(ungar_function_get_tape(mem.f->Handle(), &src, &count, &out, &dir));
            if (mem.offset + mem.f->IndependentVariableSize() + mem.f->ParameterSize() != nv) return;  // a stage function that does not read the row to its end: one by one
            if (folder.empty() && dir) folder = dir;
            const int32_t base = static_cast<int32_t>(nodes.size());
            for (int64_t i = 0; i < count; ++i) {
                ungar_tape_node nd = src[i];
                if (nd.op == 1) {  // input: index into the row
                    nd.a += static_cast<int32_t>(mem.offset);
                } else if (nd.op != 0) {
                    const int ar = nd.op <= 5 || nd.op == 18 || nd.op == 19 ? 2 : nd.op <= 17 ? 1 : 4;  // include/ungar_amd.h: op codes of ungar_tape_node
                    nd.a += base;
                    if (ar >= 2) nd.b += base;
                    if (ar == 4) {
                        nd.c += base;
                        nd.d += base;
                    }
                }
                nodes.push_back(nd);
            }
            for (index_t j = 0; j < mem.f->DependentVariableSize(); ++j) outputs.push_back(base + out[j]);
            name += "_" + std::to_string(count);
        }
        if (static_cast<index_t>(outputs.size()) != _p.stateSize + 1 + Nh() + Ne()) return;
        _stageValues.emplace(Autodiff::FunctionFactory::MakeFromTape(nodes, outputs, nd, nv - nd, name, EnabledDerivatives::NONE, folder));
    }

    void Evaluate(const Autodiff::Function& f, int what, real_t* rows, index_t offset, real_t* out, index_t count, index_t unitFastestStride = 0, const ShootingSplitImage& split = ShootingSplitImage{}) {
        const index_t nv = _p.RowSize();
        int64_t width = f.DependentVariableSize();
        if (what != 0) {
            const int32_t *r = nullptr, *c = nullptr;
            Check(what == 1 ? ungar_function_jacobian_sparsity(f.Handle(), &r, &c, &width) : ungar_function_hessian_sparsity(f.Handle(), &r, &c, &width));
        }
        const ungar_operand xp = unitFastestStride > 0 ? ungar_operand{rows + offset * unitFastestStride, 1, 0, unitFastestStride} : ungar_operand{rows + offset, nv, 0, 1};
        if (split.candidates > 0 && unitFastestStride > 0) {
            const index_t nodes = split.nodes;
            const ungar_operand x{rows + offset * unitFastestStride, nodes, 1, unitFastestStride}, par{split.parameters, 0, 1, split.stride}, ys{out, nodes * width, width, 1};
            Check(what == 0   ? ungar_function_forward_zero_nodes_split(f.Handle(), &x, &par, &ys, split.candidates * nodes, nodes, _stream)
                  : what == 1 ? ungar_function_sparse_jacobian_nodes_split(f.Handle(), &x, &par, &ys, split.candidates * nodes, nodes, _stream)
                              : ungar_function_sparse_hessian_nodes_split(f.Handle(), &x, &par, &ys, split.candidates * nodes, nodes, _stream));
            return;
        }
        const ungar_operand y{out, width, 0, 1};
        Check(what == 0   ? ungar_function_forward_zero_nodes(f.Handle(), &xp, &y, count, 1, _stream)
              : what == 1 ? ungar_function_sparse_jacobian_nodes(f.Handle(), &xp, &y, count, 1, _stream)
                          : ungar_function_sparse_hessian_nodes(f.Handle(), &xp, &y, count, 1, _stream));
    }
    /// c of row k + 1 <- carry(x, u, w, p of row k) for k < N of `instances` consecutive instances (rows in place).
    void CarryValues(real_t* rows, index_t instances, index_t unitFastestStride = 0, const ShootingSplitImage& split = ShootingSplitImage{}) {
        const index_t N = _p.horizon, nv = _p.RowSize();
        if (split.candidates > 0 && unitFastestStride > 0) {
            // variables from the stacked image (candidate c of instance i is stacked instance c * instances + i: one run of instances of N + 1 knots each), parameters
            // from the one image of the instances: instance index modulo their number -- one launch (one per candidate was 12 launches of a few microseconds each in
            // every later stage of the quadruped's line search)
            const index_t nodes = split.nodes, perCandidate = nodes / (N + 1);
            const ungar_operand x{rows + _p.StateOffset() * unitFastestStride, N + 1, 1, unitFastestStride}, par{split.parameters, N + 1, 1, split.stride},
                ys{rows + 1, N + 1, 1, unitFastestStride};
            Check(ungar_function_forward_zero_nodes_periodic(_p.carry->Handle(), &x, &par, perCandidate, &ys, split.candidates * perCandidate * N, N, _stream));
            return;
        }
        const ungar_operand xp = unitFastestStride > 0 ? ungar_operand{rows + _p.StateOffset() * unitFastestStride, N + 1, 1, unitFastestStride}
                                                       : ungar_operand{rows + _p.StateOffset(), (N + 1) * nv, nv, 1};
        const ungar_operand y = unitFastestStride > 0 ? ungar_operand{rows + 1, N + 1, 1, unitFastestStride} : ungar_operand{rows + nv, (N + 1) * nv, nv, 1};
        Check(ungar_function_forward_zero_nodes(_p.carry->Handle(), &xp, &y, instances * N, N, _stream));
    }
    template <class T>
    std::vector<T> Download(const T* device, index_t count) const {
        std::vector<T> host(static_cast<std::size_t>(count));
        Check(ungar_device_synchronize());
        Check(ungar_device_download(host.data(), device, static_cast<int64_t>(count) * static_cast<int64_t>(sizeof(T))));
        return host;
    }

    /// Candidates per stacked evaluation: the C ABI's bound of 16 (ungar_shooting_trial_rows / _select), or fewer when `maxStackedCandidates` says so before the
    /// optimiser is constructed (the test of the group logic runs the default 14 candidates in groups of 4 and must reproduce the single-group iterates bit for bit).
    static index_t StackedCandidates() { return maxStackedCandidates >= 1 && maxStackedCandidates <= 16 ? maxStackedCandidates : 16; }
    const index_t kStacked = StackedCandidates();
    ShootingProblem _p;
    index_t _batch;
    bool _verbose;
    real_t _multiplier;
    index_t _maxIterations, _iterations = 0;
    ungar_barrier _barrier{};
    ungar_shooting_dims _dims{};
    BacktrackingLineSearch::Parameters _ls{};
    std::vector<real_t> _alphas;
    void* _stream = nullptr;
    std::vector<void*> _owned;
    ungar_stage_pattern _pf{}, _pc{}, _pg{}, _pH{}, _ph{}, _pe{};
    real_t *_rows = nullptr, *_xm = nullptr, *_trial = nullptr, *_parameters = nullptr, *_listedParameters = nullptr;
    mutable bool _parametersStale = false;  // DeviceRows() handed the rows out since the parameter image was last refreshed
    index_t _parameterStride = 0;
    real_t *_f = nullptr, *_fJ = nullptr, *_cJ = nullptr, *_l = nullptr, *_lg = nullptr, *_lH = nullptr, *_h = nullptr, *_hJ = nullptr, *_e = nullptr, *_eJ = nullptr;
    real_t *_fT = nullptr, *_lT = nullptr, *_hT = nullptr, *_eT = nullptr;
    std::optional<Autodiff::Function> _stageValues;  // [f | cost | h | e] of a row in one function (BuildStageValues); empty: the stage functions one by one
    index_t _valueWidth = 0;
    real_t *_AB = nullptr, *_b = nullptr, *_W = nullptr, *_w = nullptr, *_E = nullptr, *_dz0 = nullptr, *_dZ = nullptr, *_dU = nullptr, *_workspace = nullptr;
    int64_t _workspaceDoubles = 0;
    real_t* _er = nullptr;
    int32_t *_status = nullptr, *_active = nullptr, *_pivots = nullptr;
    bool _eliminateEqualities = true;
    int _assembleRoute = 0, _riccatiRoute = 0;  // ungar_shooting_assemble_route / ungar_ocp_riccati_route of this problem
    index_t _stackedCapacity = 0;  // trial points per instance the stacked buffers were allocated for
    bool _adaptiveStages = true, _mostGoOn = false;
    index_t _sinceProbe = 0;
    std::vector<index_t> _stages{2};  // ({2, 4} measured: no gain -- quadrotor 1.22 -> 1.27 ms, RC car 0.84 -> 0.86: who needs less than 1/2 mostly needs much less)
    int32_t *_listA = nullptr, *_listB = nullptr;  // instances a stage of the line search left unresolved (read / written alternately)
    index_t _trialStride = 0;
    static constexpr bool _nodeMajorTrialRows = false;  // (node-major trial rows: measured in round 4, 1.5-2.6x slower stage functions; the code path stays for the record)
    int32_t* _unresolved = nullptr;
    real_t *_theta0 = nullptr, *_phi0 = nullptr, *_obj0 = nullptr, *_slope = nullptr, *_accepted = nullptr, *_thetaT = nullptr, *_phiT = nullptr, *_objT = nullptr;
};

}  // namespace Ungar
