// ungar_amd :: straight-line code emission from the expression tape.
//
// The reference lowers its tapes to C through CppADCodeGen and compiles them at run time with gcc
// (include/ungar/autodiff/function.hpp:468-503).  Here a tape is lowered to the body of a HIP
// device function, generic over an I/O policy (`io.x(i)`, `io.f(i, v)`, `io.j(k, r, c, v)`, ...)
// so that the hand-written kernel skeletons (csrc/kernels/node_kernel.hpp) decide how a wavefront
// maps onto shooting nodes, how operands are staged and how results are stored.
//
// A second dialect emits plain C with the reference's calling style (one instance per call); it
// exists ONLY to build the CPU baseline / checker under oracle/_gen (see oracle/README.md) and is
// never linked into the product library.
#pragma once

#include "../runtime/measurement.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "derive.hpp"

namespace ungar_amd::tape {

/// One value the generated code must deliver: `sink` is the store statement with "%s" standing
/// for the value expression.
struct OutputSlot {
    Id value;
    std::string sink;
    Id value2 = kNoId;  // second value of a paired sink ("%t" in the statement): two results leaving in one store instruction
    std::vector<Id> more;  // further values of a sink that gathers several results ("%2", "%3", ... in the statement, at most eight in all)
};

/// Replaces "%2".."%9" in a sink statement by the spelling of the slot's further values.
template <class Spell>
inline void SpellMoreValues(std::string& line, const OutputSlot& s, Spell&& spell) {
    for (std::size_t k = 0; k < s.more.size(); ++k) {
        const std::string tag = "%" + std::to_string(k + 2);
        const std::size_t pos = line.find(tag);
        if (pos != std::string::npos) line.replace(pos, tag.size(), spell(s.more[k]));
    }
}

/// Rewrites the program rooted at `roots` into `out` with its multiply-adds contracted EXPLICITLY (Op::Fma): a product with exactly one use, that use being a
/// sum or a difference, is fused into it -- the second operand of the sum first, otherwise the first; a - x y = fma(-x, y, a), x y - b = fma(x, y, -b) (the
/// negations are source modifiers of the instruction).  Compiled with contraction OFF, the generated code then performs exactly these roundings in every kernel
/// it is instantiated into, whatever the compiler's own heuristics would have picked next to different store code.  `roots` is remapped in place.
inline void FuseMultiplyAdd(const Graph& g, std::vector<Id>& roots, Graph& out) {
    const std::size_t n = g.Size();
    std::vector<char> live(n, 0);
    std::vector<int> uses(n, 0);
    for (Id r : roots) {
        live[static_cast<std::size_t>(r)] = 1;
        uses[static_cast<std::size_t>(r)] += 2;  // a value that leaves the program keeps its own name
    }
    for (std::size_t i = n; i-- > 0;) {
        if (!live[i]) continue;
        const Node& nd = g.At(static_cast<Id>(i));
        if (nd.op == Op::Input) continue;
        for (Id o : {nd.a, nd.b, nd.c, nd.d})
            if (o != kNoId) {
                live[static_cast<std::size_t>(o)] = 1;
                ++uses[static_cast<std::size_t>(o)];
            }
    }
    // A product is contracted when EVERY use of it is a sum / difference that takes it as its fused operand (a product used by three sums becomes three
    // multiply-adds and disappears: what the compiler's own aggressive fusion does, made explicit).  A sum of two candidate products takes its second operand; the
    // first then stays a product if it has no other taker -- iterated to a fixed point.
    std::vector<char> candidate(n, 0);
    std::vector<std::vector<Id>> consumers(n);
    for (std::size_t i = 0; i < n; ++i) {
        if (!live[i]) continue;
        const Node& nd = g.At(static_cast<Id>(i));
        if (nd.op == Op::Input || nd.op == Op::Const) continue;
        for (Id o : {nd.a, nd.b, nd.c, nd.d})
            if (o != kNoId) consumers[static_cast<std::size_t>(o)].push_back(static_cast<Id>(i));
    }
    std::vector<char> isRoot(n, 0);
    for (Id r : roots) isRoot[static_cast<std::size_t>(r)] = 1;
    for (std::size_t i = 0; i < n; ++i) {
        if (!live[i] || g.At(static_cast<Id>(i)).op != Op::Mul || isRoot[i] || consumers[i].empty()) continue;
        bool ok = true;
        for (Id c : consumers[i]) {
            const Node& nc = g.At(c);
            ok = ok && (nc.op == Op::Add || nc.op == Op::Sub) && nc.a != nc.b;
        }
        candidate[i] = ok;
    }
    std::vector<Id> fused(n, kNoId);    // sum / difference -> the product contracted into it
    for (bool changed = true; changed;) {
        changed = false;
        for (std::size_t i = 0; i < n; ++i) {
            if (!live[i]) continue;
            const Node& nd = g.At(static_cast<Id>(i));
            if (nd.op != Op::Add && nd.op != Op::Sub) continue;
            if (fused[i] != kNoId && candidate[static_cast<std::size_t>(fused[i])]) continue;
            fused[i] = candidate[static_cast<std::size_t>(nd.b)] ? nd.b : candidate[static_cast<std::size_t>(nd.a)] ? nd.a : kNoId;
        }
        for (std::size_t i = 0; i < n; ++i) {
            if (!candidate[i]) continue;
            for (Id c : consumers[i])
                if (fused[static_cast<std::size_t>(c)] != static_cast<Id>(i)) {
                    candidate[i] = 0;
                    changed = true;
                    break;
                }
        }
    }
    std::vector<char> absorbed(n, 0);   // products that disappear into their consumers
    for (std::size_t i = 0; i < n; ++i) {
        absorbed[i] = candidate[i];
        if (fused[i] != kNoId && !candidate[static_cast<std::size_t>(fused[i])]) fused[i] = kNoId;
    }
    out.Clear();
    std::vector<Id> map(n, kNoId);
    int inputs = 0;
    for (std::size_t i = 0; i < n; ++i)
        if (g.At(static_cast<Id>(i)).op == Op::Input) inputs = std::max(inputs, g.At(static_cast<Id>(i)).a + 1);
    std::vector<Id> inputNode(static_cast<std::size_t>(inputs));
    for (int k = 0; k < inputs; ++k) inputNode[static_cast<std::size_t>(k)] = out.Input();
    auto at = [&](Id o) { return map[static_cast<std::size_t>(o)]; };
    for (std::size_t i = 0; i < n; ++i) {
        if (!live[i] || absorbed[i]) continue;
        const Node& nd = g.At(static_cast<Id>(i));
        Id r;
        if (nd.op == Op::Const) r = out.Constant(nd.value);
        else if (nd.op == Op::Input) r = inputNode[static_cast<std::size_t>(nd.a)];
        else if (fused[i] != kNoId) {
            const Node& m = g.At(fused[i]);
            const Id x = at(m.a), y = at(m.b);
            if (nd.op == Op::Add) r = out.Fma(x, y, at(fused[i] == nd.b ? nd.a : nd.b));
            else if (fused[i] == nd.b) r = out.Fma(out.Unary(Op::Neg, x), y, at(nd.a));  // a - x y
            else r = out.Fma(x, y, out.Unary(Op::Neg, at(nd.b)));                          // x y - b
        } else if (IsUnary(nd.op)) r = out.Unary(nd.op, at(nd.a));
        else if (IsBinary(nd.op)) r = out.Binary(nd.op, at(nd.a), at(nd.b));
        else if (nd.op == Op::Fma) r = out.Fma(at(nd.a), at(nd.b), at(nd.c));
        else r = out.Cond(nd.op, at(nd.a), at(nd.b), at(nd.c), at(nd.d));
        map[i] = r;
    }
    for (Id& r : roots) r = at(r);
}

/// How input `i` of the tape is spelled in the generated code.
using InputNamer = std::string (*)(int index, const void* ctx);

struct EmitStats {
    std::size_t statements = 0, flops = 0, transcendentals = 0, divisions = 0;
};

class Emitter {
  public:
    Emitter(const Graph& g, std::vector<std::string> inputExpr) : g_{g}, inputExpr_{std::move(inputExpr)} {
    }

    /// Inputs flagged here are spelled by an expression that is evaluated wherever the value is used, in every phase
    /// that uses it (the phased emitter never gives them a home of their own): items of an LDS channel between two
    /// wavefronts, which must not be read before the statement that waits for them.
    void SetRereadInputs(std::vector<char> flags) {
        reread_ = std::move(flags);
    }

    /// Emits statements computing every slot, in slot order; each value's not-yet-emitted
    /// dependencies are emitted depth-first immediately before its first use, which keeps live
    /// ranges short (matters on the GPU where the "stack" is the VGPR file).
    std::string Emit(const std::vector<OutputSlot>& slots, const char* indent = "    ") {
        std::ostringstream os;
        name_.assign(g_.Size(), -1);
        for (const OutputSlot& s : slots) {
            EmitNode(s.value, os, indent);
            if (s.value2 != kNoId) EmitNode(s.value2, os, indent);
            for (Id m : s.more) EmitNode(m, os, indent);
            std::string line = s.sink;
            const std::size_t pos = line.find("%s");
            if (pos != std::string::npos) line.replace(pos, 2, Ref(s.value));
            const std::size_t pos2 = line.find("%t");
            if (pos2 != std::string::npos && s.value2 != kNoId) line.replace(pos2, 2, Ref(s.value2));
            SpellMoreValues(line, s, [&](Id v) { return Ref(v); });
            os << indent << line << "\n";
        }
        return os.str();
    }

    /// Phased emission with an explicit LDS home for long-lived values.
    ///
    /// The straight-line programs of the big models keep ~1000 values alive across the body
    /// (factorisation, kinematic state, shared partials); that exceeds the 512-VGPR file several
    /// times over and the compiler answers with scratch spills that turn an HBM-bound kernel into a
    /// scratch-bound one.  Here the emitter does its own allocation:
    ///   * the body is cut into PHASES (slot groups) separated by scheduling barriers;
    ///   * a value defined in one phase and used in a later one is either STORED -- it gets a
    ///     per-lane LDS slot (`io.st(slot, v)` at the definition, one `io.ld(slot)` per consuming
    ///     phase), slots being recycled when their value dies -- or REMATERIALISED: cheap values
    ///     with few consuming phases whose operands are stored/inputs/constants are simply
    ///     recomputed where needed (bounded cone depth), which shrinks the stored set ~3x;
    ///   * stored values beyond `ldsSlots` stay registers across phases.
    /// Returns the code; `slotsUsed` = LDS slots actually needed.
    /// `uniformInput` (optional, one flag per tape input) marks inputs whose value is the same in the four
    /// lanes of a quad; values derived only from those (and every QuadSum) are quad-uniform and may live in
    /// one of `uniformSlots` compact slots (io.ldu / io.stu: one copy per quad, a quarter of the LDS bytes).
    void SetInterleaveSinks(bool on) { interleaveSinks_ = on; }

    std::string EmitPhased(const std::vector<std::vector<OutputSlot>>& phases, int ldsSlots, int& slotsUsed, int rematMaxConsumers = 2,
                           int rematMaxDepth = 3, int prefetchDistance = 48, const char* indent = "    ",
                           const std::vector<char>* uniformInput = nullptr, int uniformSlots = 0, int* uniformSlotsUsed = nullptr,
                           bool prefetchAcrossPhases = false) {
        const std::size_t n = g_.Size();
        std::vector<char> uniform(n, 0);
        if (uniformInput && uniformSlots > 0)
            for (std::size_t i = 0; i < n; ++i) {  // operands precede their users in the graph
                const Node& nd = g_.At(static_cast<Id>(i));
                if (nd.op == Op::Const || nd.op == Op::QuadSum) uniform[i] = 1;
                else if (nd.op == Op::Input) uniform[i] = (*uniformInput)[static_cast<std::size_t>(nd.a)];
                else {
                    bool u = true;
                    for (Id o : {nd.a, nd.b, nd.c, nd.d})
                        if (o != kNoId && !uniform[static_cast<std::size_t>(o)]) u = false;
                    uniform[i] = u;
                }
            }
        // ---- analysis pass: first-definition phase, consumer phases -------------------------------------
        std::vector<int> defPhase(n, -1);
        std::vector<std::vector<Id>> order(phases.size());
        for (std::size_t ph = 0; ph < phases.size(); ++ph)
            for (const OutputSlot& s : phases[ph]) {
                CollectOrder(s.value, static_cast<int>(ph), defPhase, order[ph]);
                if (s.value2 != kNoId) CollectOrder(s.value2, static_cast<int>(ph), defPhase, order[ph]);
                for (Id m : s.more) CollectOrder(m, static_cast<int>(ph), defPhase, order[ph]);
            }
        const std::vector<std::vector<OutputSlot>>& ph_ = phases;
        std::vector<int> consumers(n, 0);
        for (std::size_t ph = 0; ph < ph_.size(); ++ph) {
            std::vector<char> seen(n, 0);
            auto use = [&](Id o) {
                if (o == kNoId) return;
                const std::size_t so = static_cast<std::size_t>(o);
                if (defPhase[so] >= 0 && defPhase[so] != static_cast<int>(ph) && !seen[so]) {
                    seen[so] = 1;
                    ++consumers[so];
                }
            };
            for (Id id : order[ph]) {
                const Node& nd = g_.At(id);
                use(nd.a);
                use(nd.b);
                use(nd.c);
                use(nd.d);
            }
            for (const OutputSlot& s : ph_[ph]) {
                use(s.value);
                use(s.value2);
                for (Id m : s.more) use(m);
            }
        }
        // ---- stored set: cross-phase values that are not worth recomputing ---------------------------------
        std::vector<char> stored(n, 0);
        std::vector<int> depth(n, 0);  // recompute-cone depth of non-stored cross values
        std::size_t crossTotal = 0, storedTotal = 0;
        for (std::size_t i = 0; i < n; ++i) {
            if (defPhase[i] < 0 || consumers[i] == 0) continue;
            ++crossTotal;
            const Node& nd = g_.At(static_cast<Id>(i));
            if (IsReread(nd)) {  // re-read from its channel in every consuming phase
                depth[i] = 1;
                continue;
            }
            bool cheap = nd.op != Op::Input && nd.op != Op::Div && nd.op != Op::Sin && nd.op != Op::Cos && nd.op != Op::Sqrt && nd.op != Op::Tan &&
                         nd.op != Op::Atan && nd.op != Op::Atan2 && nd.op != Op::Exp && nd.op != Op::Log && nd.op != Op::Pow && nd.op != Op::Asin &&
                         nd.op != Op::Acos;
            int dmax = 0;
            if (cheap && consumers[i] <= rematMaxConsumers) {
                for (Id o : {nd.a, nd.b, nd.c, nd.d}) {
                    if (o == kNoId) continue;
                    const std::size_t so = static_cast<std::size_t>(o);
                    const Node& no = g_.At(o);
                    if (no.op == Op::Const || no.op == Op::Input) continue;  // inputs are re-read from memory
                    if (stored[so]) continue;
                    if (consumers[so] > 0) {  // a rematerialised cross value: chain
                        dmax = std::max(dmax, depth[so]);
                    } else {
                        cheap = false;  // operand is private to its defining phase: would have to be kept
                    }
                }
            }
            if (cheap && consumers[i] <= rematMaxConsumers && dmax + 1 <= rematMaxDepth) {
                depth[i] = dmax + 1;
            } else {
                stored[i] = 1;
                ++storedTotal;
            }
        }
        // ---- generation (dry run first to learn the last phase that loads each stored value) ----------------
        std::vector<int> slotOf(n, -1);  // varying slot s >= 0; uniform slot u encoded as -2 - u
        std::vector<int> lastLoad(n, -1);
        auto hasSlot = [&](std::size_t i) { return slotOf[i] != -1; };
        auto ldExpr = [&](std::size_t i) {
            return slotOf[i] >= 0 ? "io.ld(" + std::to_string(slotOf[i]) + ")" : "io.ldu(" + std::to_string(-2 - slotOf[i]) + ")";
        };
        auto stStmt = [&](std::size_t i, const std::string& v) {
            return slotOf[i] >= 0 ? "io.st(" + std::to_string(slotOf[i]) + ", " + v + ");" : "io.stu(" + std::to_string(-2 - slotOf[i]) + ", " + v + ");";
        };
        std::string text;
        for (int pass = 0; pass < 2; ++pass) {
            const bool dry = pass == 0;
            std::ostringstream os;
            std::vector<int> availIn(n, -1);        // phase in which the value has a usable local
            std::vector<std::string> local(n);
            std::vector<char> defined(n, 0);
            int counter = 0;
            EmitStats stats;
            struct Line {
                std::string text;
                bool isLoad;
                std::size_t value;  // LDS load: the value it brings back
            };
            std::vector<Line> all;                // statements of all phases, in recording order
            std::vector<std::size_t> phaseBegin;  // index in `all` of the first statement of each phase (its marker)
            std::vector<std::size_t> storeLine(n, 0);
            struct LineSink {
                std::vector<Line>& all;
                std::size_t pendingValue = 0;
                void emplace_back(std::string t, bool isLoad) { all.push_back({std::move(t), isLoad, isLoad ? pendingValue : 0}); }
            } lines{all};
            for (std::size_t ph = 0; ph < ph_.size(); ++ph) {
                phaseBegin.push_back(all.size());
                if (ph && !dry) all.push_back({"io.phase();", false, 0});
                const int iph = static_cast<int>(ph);
                // sinks spelled "@begin:<statement>" open the phase (waiting for another wavefront's message, ...)
                if (!dry)
                    for (const OutputSlot& s : ph_[ph])
                        if (s.sink.rfind(kBeginTag, 0) == 0) all.push_back({s.sink.substr(std::strlen(kBeginTag)), false, 0});
                // explicit stack DFS producing statements for `root` in this phase
                auto produce = [&](Id root) {
                    std::vector<std::pair<Id, int>> stack{{root, 0}};
                    while (!stack.empty()) {
                        auto& [id, state] = stack.back();
                        const std::size_t si = static_cast<std::size_t>(id);
                        const Node& nd = g_.At(id);
                        if (nd.op == Op::Const || availIn[si] == iph) {
                            stack.pop_back();
                            continue;
                        }
                        const bool isStoredElsewhere = defined[si] && stored[si];
                        const bool regResident = isStoredElsewhere && !dry && !hasSlot(si);
                        if (isStoredElsewhere && (dry || hasSlot(si))) {
                            lastLoad[si] = std::max(lastLoad[si], iph);
                            availIn[si] = iph;
                            local[si] = "v" + std::to_string(counter++);
                            lines.pendingValue = si;
                            if (!dry) lines.emplace_back("const double " + local[si] + " = " + ldExpr(si) + ";", true);
                            stack.pop_back();
                            continue;
                        }
                        if (regResident) {  // kept in a register across phases: original name stays valid
                            availIn[si] = iph;
                            stack.pop_back();
                            continue;
                        }
                        // (re)compute here: operands first
                        bool pushed = false;
                        if (nd.op != Op::Input) {
                            const Id ops[4] = {nd.a, nd.b, nd.c, nd.d};
                            while (state < 4) {
                                const Id o = ops[state++];
                                if (o == kNoId) continue;
                                if (g_.At(o).op != Op::Const && availIn[static_cast<std::size_t>(o)] != iph) {
                                    stack.emplace_back(o, 0);
                                    pushed = true;
                                    break;
                                }
                            }
                        }
                        if (pushed) continue;
                        const Id me = id;
                        const std::size_t sm = static_cast<std::size_t>(me);
                        const Node& nm = g_.At(me);
                        stack.pop_back();
                        auto nameOf = [&](Id o) -> std::string {
                            if (o == kNoId) return "";
                            const Node& no = g_.At(o);
                            return no.op == Op::Const ? Lit(no.value) : local[static_cast<std::size_t>(o)];
                        };
                        std::string expr;
                        if (nm.op == Op::Input) expr = inputExpr_[static_cast<std::size_t>(nm.a)];
                        else {
                            EmitStats& keep = stats_;
                            (void)keep;
                            expr = ExprWith(nm, nameOf(nm.a), nameOf(nm.b), nameOf(nm.c), nameOf(nm.d));
                        }
                        local[sm] = "v" + std::to_string(counter++);
                        availIn[sm] = iph;
                        ++stats.statements;
                        if (!dry) lines.emplace_back("const double " + local[sm] + " = " + expr + ";", false);
                        if (!defined[sm]) {
                            defined[sm] = 1;
                            if (stored[sm] && !dry && hasSlot(sm)) {
                                storeLine[sm] = all.size();
                                lines.emplace_back(stStmt(sm, local[sm]), false);
                            }
                        }
                    }
                };
                // default: every value of the phase first, its sinks at the end (the machine scheduler then places the stores); interleaved: each
                // sink right behind the statements that produce its values, so that result stores are spread over the phase instead of leaving in
                // one burst at its end (the CU's store path takes 64 bytes per clock: a burst of 18 KiB per wavefront stalls the lone wavefront of a SIMD)
                if (!interleaveSinks_)
                    for (Id id : order[ph]) produce(id);
                for (const OutputSlot& s : ph_[ph]) {
                    if (s.sink.rfind(kBeginTag, 0) == 0) continue;
                    produce(s.value);
                    if (s.value2 != kNoId) produce(s.value2);
                    for (Id m : s.more) produce(m);
                    if (!dry) {
                        std::string line = s.sink;
                        auto spell = [&](Id v) {
                            const Node& nv = g_.At(v);
                            return nv.op == Op::Const ? Lit(nv.value) : local[static_cast<std::size_t>(v)];
                        };
                        const std::size_t pos = line.find("%s");
                        if (pos != std::string::npos) line.replace(pos, 2, spell(s.value));
                        const std::size_t pos2 = line.find("%t");
                        if (pos2 != std::string::npos && s.value2 != kNoId) line.replace(pos2, 2, spell(s.value2));
                        SpellMoreValues(line, s, spell);
                        lines.emplace_back(line, false);
                    }
                }
            }
            if (!dry) {
                // software prefetch: an LDS load depends only on the store of its value, so it is hoisted
                // `prefetchDistance` statements ahead of its first use to hide the ~100+ cycle LDS latency
                // (one wavefront per SIMD: nothing else would cover it) -- up to the start of its phase, or,
                // with prefetchAcrossPhases, into the tail of the previous phase (never above its store)
                phaseBegin.push_back(all.size());
                std::vector<std::pair<double, std::size_t>> key(all.size());
                std::size_t ph = 0;
                for (std::size_t i = 0; i < all.size(); ++i) {
                    while (ph + 1 < phaseBegin.size() && phaseBegin[ph + 1] <= i) ++ph;
                    double k = static_cast<double>(i);
                    if (all[i].isLoad) {
                        double lo = static_cast<double>(phaseBegin[ph]) + 0.25;  // after the phase marker
                        if (prefetchAcrossPhases && ph > 0)
                            lo = std::max(static_cast<double>(phaseBegin[ph - 1]), static_cast<double>(storeLine[all[i].value])) + 0.25;
                        k = std::max(lo, static_cast<double>(i) - prefetchDistance - 0.5);
                    }
                    key[i] = {k, i};
                }
                std::stable_sort(key.begin(), key.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
                for (const auto& [k, i] : key) os << indent << all[i].text << "\n";
            }
            if (dry) {
                // slot allocation from (defPhase, lastLoad) intervals
                std::vector<int> freeSlots, freeUniform;
                std::vector<std::vector<int>> expiring(ph_.size() + 2);
                int next = 0, nextUniform = 0;
                std::size_t peak = 0, live = 0, inLds = 0, inUniform = 0, uniformCross = 0;
                std::vector<std::size_t> liveDelta(ph_.size() + 2, 0);
                for (std::size_t ph = 0; ph < ph_.size(); ++ph) {
                    for (int sl : expiring[ph]) (sl >= 0 ? freeSlots : freeUniform).push_back(sl >= 0 ? sl : -2 - sl);
                    std::vector<Id> cross;
                    for (Id id : order[ph])
                        if (stored[static_cast<std::size_t>(id)] && lastLoad[static_cast<std::size_t>(id)] > static_cast<int>(ph)) cross.push_back(id);
                    std::stable_sort(cross.begin(), cross.end(),
                                     [&](Id x, Id y) { return lastLoad[static_cast<std::size_t>(x)] > lastLoad[static_cast<std::size_t>(y)]; });
                    live += cross.size();
                    peak = std::max(peak, live);

                    for (Id id : cross) {
                        const std::size_t si = static_cast<std::size_t>(id);
                        ++liveDelta[static_cast<std::size_t>(lastLoad[si]) + 1];
                        int slot = -1;
                        if (uniform[si]) {
                            ++uniformCross;
                            int us = -1;
                            if (!freeUniform.empty()) {
                                us = freeUniform.back();
                                freeUniform.pop_back();
                            } else if (nextUniform < uniformSlots) {
                                us = nextUniform++;
                            }
                            if (us >= 0) {
                                ++inUniform;
                                slotOf[si] = -2 - us;
                                expiring[static_cast<std::size_t>(lastLoad[si]) + 1].push_back(-2 - us);
                                continue;
                            }
                        }
                        if (!freeSlots.empty()) {
                            slot = freeSlots.back();
                            freeSlots.pop_back();
                        } else if (next < ldsSlots) {
                            slot = next++;
                        }
                        if (slot < 0) continue;
                        ++inLds;
                        slotOf[si] = slot;
                        expiring[static_cast<std::size_t>(lastLoad[si]) + 1].push_back(slot);
                    }
                    live -= liveDelta[ph + 1];
                }
                slotsUsed = next;
                if (UNGAR_MEASUREMENT_SWITCH("UNGAR_EMIT_DUMP_LIVE")) {  // codegen diagnostics: live ranges of the stored cross-phase values
                    std::vector<std::vector<int>> hist(ph_.size(), std::vector<int>(ph_.size(), 0));
                    for (std::size_t i = 0; i < n; ++i)
                        if (stored[i] && defPhase[i] >= 0 && lastLoad[i] > defPhase[i]) ++hist[static_cast<std::size_t>(defPhase[i])][static_cast<std::size_t>(lastLoad[i])];
                    for (std::size_t d = 0; d < ph_.size(); ++d)
                        for (std::size_t l = 0; l < ph_.size(); ++l)
                            if (hist[d][l]) std::fprintf(stderr, "[emit-live] defined in phase %zu, last used in phase %zu: %d values\n", d, l, hist[d][l]);
                }
                if (uniformSlotsUsed) *uniformSlotsUsed = nextUniform;
                if (uniformInput && uniformSlots > 0) {
                    std::size_t uniformStatements = 0, total = 0;
                    for (std::size_t i = 0; i < n; ++i) {
                        if (defPhase[i] < 0 || g_.At(static_cast<Id>(i)).op == Op::Input) continue;
                        ++total;
                        uniformStatements += uniform[i];
                    }
                    std::fprintf(stderr, "[emit] %zu of %zu statements are quad-uniform (computed identically by the four lanes of a node)\n", uniformStatements, total);
                }
                std::fprintf(stderr, "[emit] phased (%zu pieces): %zu cross-phase values, %zu stored (%zu in LDS + %zu of %zu quad-uniform in compact slots, peak live %zu), %zu rematerialised; %zu statements (+%zu recomputed)\n",
                             ph_.size(), crossTotal, storedTotal, inLds, inUniform, uniformCross, peak, crossTotal - storedTotal, stats.statements,
                             stats.statements - CountDefined(defPhase));
            } else {
                text = os.str();
                stats_.statements = stats.statements;
            }
        }
        return text;
    }

    static constexpr const char* kBeginTag = "@begin:";

    bool IsReread(const Node& nd) const {
        return nd.op == Op::Input && static_cast<std::size_t>(nd.a) < reread_.size() && reread_[static_cast<std::size_t>(nd.a)];
    }

    static std::size_t CountDefined(const std::vector<int>& defPhase) {
        std::size_t c = 0;
        for (int d : defPhase) c += d >= 0;
        return c;
    }

    const EmitStats& Stats() const {
        return stats_;
    }

  private:
    static std::string Lit(double v) {
        char buf[64];
        if (v == static_cast<double>(static_cast<long long>(v)) && std::fabs(v) < 1e15) {
            std::snprintf(buf, sizeof buf, "%.1f", v);
        } else {
            std::snprintf(buf, sizeof buf, "%.17g", v);
        }
        std::string s{buf};
        if (v < 0) s = "(" + s + ")";
        return s;
    }

    std::string Ref(Id id) const {
        const Node& nd = g_.At(id);
        if (nd.op == Op::Const) return Lit(nd.value);
        if (nd.op == Op::Input) return inputExpr_[static_cast<std::size_t>(nd.a)];
        return "v" + std::to_string(name_[static_cast<std::size_t>(id)]);
    }

    void EmitNode(Id root, std::ostringstream& os, const char* indent) {
        // Iterative post-order DFS (tapes are deep: thousands of chained adds).
        std::vector<std::pair<Id, int>> stack{{root, 0}};
        while (!stack.empty()) {
            auto& [id, state] = stack.back();
            const Node& nd = g_.At(id);
            if (Arity(nd.op) == 0 || name_[static_cast<std::size_t>(id)] >= 0) {
                stack.pop_back();
                continue;
            }
            const Id ops[4] = {nd.a, nd.b, nd.c, nd.d};
            bool pushed = false;
            while (state < 4) {
                const Id o = ops[state++];
                if (o == kNoId) continue;
                const Node& no = g_.At(o);
                if (Arity(no.op) != 0 && name_[static_cast<std::size_t>(o)] < 0) {
                    stack.emplace_back(o, 0);
                    pushed = true;
                    break;
                }
            }
            if (pushed) continue;
            const Id me = id;
            stack.pop_back();
            name_[static_cast<std::size_t>(me)] = next_++;
            os << indent << "const double v" << name_[static_cast<std::size_t>(me)] << " = " << Expr(me) << ";\n";
            ++stats_.statements;
        }
    }

    /// Post-order DFS that appends not-yet-ordered nodes (inputs included, constants excluded).
    void CollectOrder(Id root, int phase, std::vector<int>& defPhase, std::vector<Id>& out) {
        std::vector<std::pair<Id, int>> stack{{root, 0}};
        while (!stack.empty()) {
            auto& [id, state] = stack.back();
            const Node& nd = g_.At(id);
            if (nd.op == Op::Const || defPhase[static_cast<std::size_t>(id)] >= 0) {
                stack.pop_back();
                continue;
            }
            const Id ops[4] = {nd.a, nd.b, nd.c, nd.d};
            bool pushed = false;
            if (nd.op != Op::Input)
                while (state < 4) {
                    const Id o = ops[state++];
                    if (o == kNoId) continue;
                    if (g_.At(o).op != Op::Const && defPhase[static_cast<std::size_t>(o)] < 0) {
                        stack.emplace_back(o, 0);
                        pushed = true;
                        break;
                    }
                }
            if (pushed) continue;
            const Id me = id;
            stack.pop_back();
            defPhase[static_cast<std::size_t>(me)] = phase;
            out.push_back(me);
        }
    }

    std::string ExprWith(const Node& nd, const std::string& a, const std::string& b, const std::string& c, const std::string& d) {
        switch (nd.op) {
            case Op::Add: ++stats_.flops; return a + " + " + b;
            case Op::Sub: ++stats_.flops; return a + " - " + b;
            case Op::Mul: ++stats_.flops; return a + " * " + b;
            case Op::Div: ++stats_.flops; ++stats_.divisions; return a + " / " + b;
            case Op::Neg: return "-" + a;
            case Op::Sin: ++stats_.transcendentals; return "sin(" + a + ")";
            case Op::Cos: ++stats_.transcendentals; return "cos(" + a + ")";
            case Op::Tan: ++stats_.transcendentals; return "tan(" + a + ")";
            case Op::Asin: ++stats_.transcendentals; return "asin(" + a + ")";
            case Op::Acos: ++stats_.transcendentals; return "acos(" + a + ")";
            case Op::Atan: ++stats_.transcendentals; return "atan(" + a + ")";
            case Op::Exp: ++stats_.transcendentals; return "exp(" + a + ")";
            case Op::Log: ++stats_.transcendentals; return "log(" + a + ")";
            case Op::Sqrt: ++stats_.transcendentals; return "sqrt(" + a + ")";
            case Op::Abs: return "fabs(" + a + ")";
            case Op::Sign: return "(double)((" + a + " > 0.0) - (" + a + " < 0.0))";
            case Op::QuadSum: return "io.quad_sum(" + a + ")";
            case Op::QuadRot1: return "io.quad_rot1(" + a + ")";
            case Op::QuadRot2: return "io.quad_rot2(" + a + ")";
            case Op::QuadRot3: return "io.quad_rot3(" + a + ")";
            case Op::Pow: ++stats_.transcendentals; return "pow(" + a + ", " + b + ")";
            case Op::Atan2: ++stats_.transcendentals; return "atan2(" + a + ", " + b + ")";
            case Op::Fma: stats_.flops += 2; return "io.fma(" + a + ", " + b + ", " + c + ")";
            case Op::CondLt: return "(" + a + " < " + b + " ? " + c + " : " + d + ")";
            case Op::CondLe: return "(" + a + " <= " + b + " ? " + c + " : " + d + ")";
            case Op::CondEq: return "(" + a + " == " + b + " ? " + c + " : " + d + ")";
            case Op::CondGe: return "(" + a + " >= " + b + " ? " + c + " : " + d + ")";
            case Op::CondGt: return "(" + a + " > " + b + " ? " + c + " : " + d + ")";
            default: throw std::logic_error("Emitter: unexpected op");
        }
    }

    std::string Expr(Id id) {
        const Node& nd = g_.At(id);
        auto A = [&] { return Ref(nd.a); };
        auto B = [&] { return Ref(nd.b); };
        switch (nd.op) {
            case Op::Add: ++stats_.flops; return A() + " + " + B();
            case Op::Sub: ++stats_.flops; return A() + " - " + B();
            case Op::Mul: ++stats_.flops; return A() + " * " + B();
            case Op::Div: ++stats_.flops; ++stats_.divisions; return A() + " / " + B();
            case Op::Neg: return "-" + A();
            case Op::Sin: ++stats_.transcendentals; return "sin(" + A() + ")";
            case Op::Cos: ++stats_.transcendentals; return "cos(" + A() + ")";
            case Op::Tan: ++stats_.transcendentals; return "tan(" + A() + ")";
            case Op::Asin: ++stats_.transcendentals; return "asin(" + A() + ")";
            case Op::Acos: ++stats_.transcendentals; return "acos(" + A() + ")";
            case Op::Atan: ++stats_.transcendentals; return "atan(" + A() + ")";
            case Op::Exp: ++stats_.transcendentals; return "exp(" + A() + ")";
            case Op::Log: ++stats_.transcendentals; return "log(" + A() + ")";
            case Op::Sqrt: ++stats_.transcendentals; return "sqrt(" + A() + ")";
            case Op::Abs: return "fabs(" + A() + ")";
            case Op::Sign: return "(double)((" + A() + " > 0.0) - (" + A() + " < 0.0))";
            case Op::QuadSum: return "io.quad_sum(" + A() + ")";
            case Op::QuadRot1: return "io.quad_rot1(" + A() + ")";
            case Op::QuadRot2: return "io.quad_rot2(" + A() + ")";
            case Op::QuadRot3: return "io.quad_rot3(" + A() + ")";
            case Op::Pow: ++stats_.transcendentals; return "pow(" + A() + ", " + B() + ")";
            case Op::Atan2: ++stats_.transcendentals; return "atan2(" + A() + ", " + B() + ")";
            case Op::Fma: stats_.flops += 2; return "io.fma(" + A() + ", " + B() + ", " + Ref(nd.c) + ")";
            case Op::CondLt: return "(" + A() + " < " + B() + " ? " + Ref(nd.c) + " : " + Ref(nd.d) + ")";
            case Op::CondLe: return "(" + A() + " <= " + B() + " ? " + Ref(nd.c) + " : " + Ref(nd.d) + ")";
            case Op::CondEq: return "(" + A() + " == " + B() + " ? " + Ref(nd.c) + " : " + Ref(nd.d) + ")";
            case Op::CondGe: return "(" + A() + " >= " + B() + " ? " + Ref(nd.c) + " : " + Ref(nd.d) + ")";
            case Op::CondGt: return "(" + A() + " > " + B() + " ? " + Ref(nd.c) + " : " + Ref(nd.d) + ")";
            default: throw std::logic_error("Emitter: unexpected op");
        }
    }

    const Graph& g_;
    std::vector<std::string> inputExpr_;
    std::vector<char> reread_;
    bool interleaveSinks_ = false;
    std::vector<int> name_;
    int next_ = 0;
    EmitStats stats_;
};

}  // namespace ungar_amd::tape
