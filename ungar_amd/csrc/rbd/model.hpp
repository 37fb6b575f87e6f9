// ungar_amd :: rigid-body model (kinematic tree + lumped spatial inertias).
//
// Replaces, for the hot path only, what the reference obtains from Pinocchio v2.7.0 in
//   include/ungar/rbd/robot.hpp:43-50   pinocchio::urdf::buildModel(file, JointModelFreeFlyer, model)
// The build rules restated here are Pinocchio's published URDF semantics:
//   * root link attached to the universe through a free-flyer joint (nq 7 = [p, quat xyzw], nv 6);
//   * child joints of a link are visited depth-first in joint-NAME order (urdfdom keeps joints in a
//     std::map and fills child_links from it) -> ANYmal legs come out LF, LH, RF, RH
//     (test/rbd/robot.test.cpp:44-47) although the URDF lists LF, RF, LH, RH;
//   * a fixed joint adds no degree of freedom: its child body's inertia is transported by the
//     fixed placement and added to the supporting joint's body ("lumping"), and the placements of
//     joints further down are composed with it;
//   * default gravity (0, 0, -9.81) along world z.
// Two input formats are read: URDF (own minimal XML reader) and the flat ".robot" text that
// tools/urdf_to_robot.py emits (committed for ANYmal B because /root/reference does not travel).
#pragma once

#include <algorithm>
#include <array>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace ungar_amd::rbd {

using Mat3 = std::array<std::array<double, 3>, 3>;
using V3 = std::array<double, 3>;

inline Mat3 Identity3() {
    return {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
}
inline Mat3 MatMul(const Mat3& a, const Mat3& b) {
    Mat3 c{};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) c[i][j] += a[i][k] * b[k][j];
    return c;
}
inline Mat3 Transpose(const Mat3& a) {
    Mat3 t{};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) t[i][j] = a[j][i];
    return t;
}
inline V3 MatVec(const Mat3& a, const V3& v) {
    V3 r{};
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) r[i] += a[i][k] * v[k];
    return r;
}
/// URDF fixed-axis roll-pitch-yaw: R = Rz(yaw) Ry(pitch) Rx(roll).
inline Mat3 RpyToMatrix(const V3& rpy) {
    const double cr = std::cos(rpy[0]), sr = std::sin(rpy[0]);
    const double cp = std::cos(rpy[1]), sp = std::sin(rpy[1]);
    const double cy = std::cos(rpy[2]), sy = std::sin(rpy[2]);
    return {{{cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr},
             {sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr},
             {-sp, cp * sr, cp * cr}}};
}

/// Placement of a child frame in its parent frame: x_parent = R x_child + p.
struct Placement {
    Mat3 R = Identity3();
    V3 p{0, 0, 0};
    Placement operator*(const Placement& o) const {
        Placement r;
        r.R = MatMul(R, o.R);
        const V3 rp = MatVec(R, o.p);
        r.p = {p[0] + rp[0], p[1] + rp[1], p[2] + rp[2]};
        return r;
    }
};

/// Rigid-body inertia about the frame origin: mass, first moment h = m c, rotational inertia
/// about the ORIGIN (I_c - m c^ c^).  Additive under lumping.
struct BodyInertia {
    double mass = 0;
    V3 h{0, 0, 0};
    Mat3 I{};

    static BodyInertia FromCom(double m, const V3& c, const Mat3& Ic) {
        BodyInertia b;
        b.mass = m;
        b.h = {m * c[0], m * c[1], m * c[2]};
        const double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) b.I[i][j] = Ic[i][j] + m * ((i == j ? cc : 0.0) - c[i] * c[j]);
        return b;
    }
    /// Express in the parent frame of placement M (x_parent = R x + p).
    BodyInertia Transported(const Placement& M) const {
        if (mass == 0.0) return {};
        const V3 c{h[0] / mass, h[1] / mass, h[2] / mass};
        const double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
        Mat3 Ic{};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Ic[i][j] = I[i][j] - mass * ((i == j ? cc : 0.0) - c[i] * c[j]);
        const Mat3 IcP = MatMul(MatMul(M.R, Ic), Transpose(M.R));
        const V3 rc = MatVec(M.R, c);
        return FromCom(mass, {rc[0] + M.p[0], rc[1] + M.p[1], rc[2] + M.p[2]}, IcP);
    }
    BodyInertia& operator+=(const BodyInertia& o) {
        mass += o.mass;
        for (int i = 0; i < 3; ++i) {
            h[i] += o.h[i];
            for (int j = 0; j < 3; ++j) I[i][j] += o.I[i][j];
        }
        return *this;
    }
    /// 6x6 spatial inertia, motion/force ordering (linear, angular):
    ///   [ m 1    -h^  ]
    ///   [ h^      I   ]
    std::array<std::array<double, 6>, 6> Matrix() const {
        std::array<std::array<double, 6>, 6> Y{};
        const double hx[3][3] = {{0, -h[2], h[1]}, {h[2], 0, -h[0]}, {-h[1], h[0], 0}};
        for (int i = 0; i < 3; ++i) {
            Y[i][i] = mass;
            for (int j = 0; j < 3; ++j) {
                Y[i][3 + j] = -hx[i][j];
                Y[3 + i][j] = hx[i][j];
                Y[3 + i][3 + j] = I[i][j];
            }
        }
        return Y;
    }
};

enum class JointType { FreeFlyer, Revolute };

struct Joint {
    std::string name;
    JointType type = JointType::Revolute;
    int parent = 0;        // index into Model::joints; 0 = universe
    Placement placement;   // of the joint frame in the parent joint frame
    V3 axis{0, 0, 1};      // revolute only
    BodyInertia inertia;   // lumped, in the joint frame
    int idxQ = 0, idxV = 0, nq = 0, nv = 0;
};

/// A named frame rigidly attached to a moving joint (or to the universe): the universe, the root joint, and then every joint and every link of the
/// description (including the links lumped into their parents through fixed joints, e.g. the feet) in the order of the depth-first traversal -- a joint's
/// frame right before its child link's --, which is the frame numbering of the reference's model builder (test/rbd/robot.test.cpp:49-52 and
/// example/rbd/quantity.example.cpp:44-45 name the feet of ANYmal B by their indices 12, 22, 32, 42).
struct Frame {
    std::string name;
    int joint = 0;        // index into Model::joints of the supporting joint
    Placement placement;  // of the frame in that joint's frame
};

struct Model {
    std::string name;
    std::vector<Joint> joints;  // joints[0] is the universe
    std::vector<Frame> frames;  // universe, root joint, root link, then (joint, child link) pairs in visiting order
    V3 gravity{0.0, 0.0, -9.81};
    int nq = 0, nv = 0;
    int NumJoints() const {
        return static_cast<int>(joints.size());
    }
    double TotalMass() const {
        double m = 0;
        for (const Joint& j : joints) m += j.inertia.mass;
        return m;
    }
};

/// Format-independent robot description (what both readers produce).
struct RobotDescription {
    struct Link {
        std::string name;
        bool hasInertial = false;
        double mass = 0;
        V3 comXyz{0, 0, 0}, comRpy{0, 0, 0};
        double ixx = 0, ixy = 0, ixz = 0, iyy = 0, iyz = 0, izz = 0;
    };
    struct JointDesc {
        std::string name, type, parent, child;
        V3 xyz{0, 0, 0}, rpy{0, 0, 0}, axis{1, 0, 0};
    };
    std::string name;
    std::vector<Link> links;
    std::vector<JointDesc> joints;
};

namespace detail {

inline V3 ParseV3(const std::string& s, const V3& def) {
    if (s.empty()) return def;
    std::istringstream is(s);
    V3 v = def;
    is >> v[0] >> v[1] >> v[2];
    return v;
}

/// Minimal XML pull reader: elements, attributes, comments, declarations.  Enough for URDF.
struct XmlElement {
    std::string tag;
    std::map<std::string, std::string> attr;
    std::vector<XmlElement> children;
    const XmlElement* Child(const std::string& t) const {
        for (const XmlElement& c : children)
            if (c.tag == t) return &c;
        return nullptr;
    }
    std::string Attr(const std::string& k) const {
        auto it = attr.find(k);
        return it == attr.end() ? std::string{} : it->second;
    }
};

class XmlReader {
  public:
    explicit XmlReader(std::string text) : s_{std::move(text)} {
    }
    XmlElement ParseDocument() {
        XmlElement root;
        root.tag = "#document";
        ParseChildren(root);
        return root;
    }

  private:
    void SkipWs() {
        while (i_ < s_.size() && std::isspace(static_cast<unsigned char>(s_[i_]))) ++i_;
    }
    bool StartsWith(const char* lit) const {
        return s_.compare(i_, std::char_traits<char>::length(lit), lit) == 0;
    }
    void SkipUntil(const char* lit) {
        const std::size_t k = s_.find(lit, i_);
        if (k == std::string::npos) throw std::runtime_error("urdf: unterminated construct");
        i_ = k + std::char_traits<char>::length(lit);
    }
    void ParseChildren(XmlElement& parent) {
        for (;;) {
            const std::size_t lt = s_.find('<', i_);
            if (lt == std::string::npos) {
                i_ = s_.size();
                return;
            }
            i_ = lt;
            if (StartsWith("<!--")) {
                SkipUntil("-->");
            } else if (StartsWith("<?")) {
                SkipUntil("?>");
            } else if (StartsWith("<!")) {
                SkipUntil(">");
            } else if (StartsWith("</")) {
                SkipUntil(">");
                return;
            } else {
                ++i_;
                XmlElement e;
                while (i_ < s_.size() && !std::isspace(static_cast<unsigned char>(s_[i_])) && s_[i_] != '>' && s_[i_] != '/')
                    e.tag += s_[i_++];
                bool selfClosed = false;
                for (;;) {
                    SkipWs();
                    if (i_ >= s_.size()) throw std::runtime_error("urdf: truncated element");
                    if (s_[i_] == '/') {
                        selfClosed = true;
                        SkipUntil(">");
                        break;
                    }
                    if (s_[i_] == '>') {
                        ++i_;
                        break;
                    }
                    std::string key;
                    while (i_ < s_.size() && s_[i_] != '=' && !std::isspace(static_cast<unsigned char>(s_[i_]))) key += s_[i_++];
                    SkipWs();
                    if (i_ >= s_.size() || s_[i_] != '=') throw std::runtime_error("urdf: attribute without value");
                    ++i_;
                    SkipWs();
                    const char quote = s_[i_++];
                    std::string val;
                    while (i_ < s_.size() && s_[i_] != quote) val += s_[i_++];
                    ++i_;
                    e.attr[key] = val;
                }
                if (!selfClosed) ParseChildren(e);
                parent.children.push_back(std::move(e));
            }
        }
    }
    std::string s_;
    std::size_t i_ = 0;
};

inline std::string ReadFile(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open robot description '" + path + "'");
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

}  // namespace detail

inline RobotDescription ReadUrdf(const std::string& path) {
    detail::XmlReader reader{detail::ReadFile(path)};
    const detail::XmlElement doc = reader.ParseDocument();
    const detail::XmlElement* robot = doc.Child("robot");
    if (!robot) throw std::runtime_error("urdf: no <robot> element in '" + path + "'");
    RobotDescription d;
    d.name = robot->Attr("name");
    for (const detail::XmlElement& e : robot->children) {
        if (e.tag == "link") {
            RobotDescription::Link l;
            l.name = e.Attr("name");
            if (const detail::XmlElement* in = e.Child("inertial")) {
                l.hasInertial = true;
                if (const detail::XmlElement* o = in->Child("origin")) {
                    l.comXyz = detail::ParseV3(o->Attr("xyz"), {0, 0, 0});
                    l.comRpy = detail::ParseV3(o->Attr("rpy"), {0, 0, 0});
                }
                if (const detail::XmlElement* m = in->Child("mass")) l.mass = std::atof(m->Attr("value").c_str());
                if (const detail::XmlElement* I = in->Child("inertia")) {
                    l.ixx = std::atof(I->Attr("ixx").c_str());
                    l.ixy = std::atof(I->Attr("ixy").c_str());
                    l.ixz = std::atof(I->Attr("ixz").c_str());
                    l.iyy = std::atof(I->Attr("iyy").c_str());
                    l.iyz = std::atof(I->Attr("iyz").c_str());
                    l.izz = std::atof(I->Attr("izz").c_str());
                }
            }
            d.links.push_back(l);
        } else if (e.tag == "joint") {
            RobotDescription::JointDesc j;
            j.name = e.Attr("name");
            j.type = e.Attr("type");
            if (const detail::XmlElement* p = e.Child("parent")) j.parent = p->Attr("link");
            if (const detail::XmlElement* c = e.Child("child")) j.child = c->Attr("link");
            if (const detail::XmlElement* o = e.Child("origin")) {
                j.xyz = detail::ParseV3(o->Attr("xyz"), {0, 0, 0});
                j.rpy = detail::ParseV3(o->Attr("rpy"), {0, 0, 0});
            }
            if (const detail::XmlElement* a = e.Child("axis")) j.axis = detail::ParseV3(a->Attr("xyz"), {1, 0, 0});
            d.joints.push_back(j);
        }
    }
    return d;
}

/// ".robot" flat text: `robot <name>` / `link <name> <has> m cx cy cz r p y ixx ixy ixz iyy iyz izz`
/// / `joint <name> <type> <parent> <child> x y z r p y ax ay az`.
inline RobotDescription ReadRobotText(const std::string& path) {
    std::istringstream in{detail::ReadFile(path)};
    RobotDescription d;
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ls{line};
        std::string kind;
        ls >> kind;
        if (kind == "robot") {
            ls >> d.name;
        } else if (kind == "link") {
            RobotDescription::Link l;
            int has = 0;
            ls >> l.name >> has >> l.mass >> l.comXyz[0] >> l.comXyz[1] >> l.comXyz[2] >> l.comRpy[0] >> l.comRpy[1] >>
                l.comRpy[2] >> l.ixx >> l.ixy >> l.ixz >> l.iyy >> l.iyz >> l.izz;
            l.hasInertial = has != 0;
            d.links.push_back(l);
        } else if (kind == "joint") {
            RobotDescription::JointDesc j;
            ls >> j.name >> j.type >> j.parent >> j.child >> j.xyz[0] >> j.xyz[1] >> j.xyz[2] >> j.rpy[0] >> j.rpy[1] >>
                j.rpy[2] >> j.axis[0] >> j.axis[1] >> j.axis[2];
            d.joints.push_back(j);
        } else {
            throw std::runtime_error("robot text: unknown record '" + kind + "'");
        }
        if (!ls) throw std::runtime_error("robot text: malformed line: " + line);
    }
    return d;
}

inline RobotDescription ReadRobotDescription(const std::string& path) {
    const bool urdf = path.size() >= 5 && path.compare(path.size() - 5, 5, ".urdf") == 0;
    return urdf ? ReadUrdf(path) : ReadRobotText(path);
}

/// Builds the free-flyer-rooted, fixed-joint-lumped model (see file header for the rules).
inline Model BuildModel(const RobotDescription& d) {
    std::map<std::string, const RobotDescription::Link*> links;
    for (const auto& l : d.links) links[l.name] = &l;
    std::map<std::string, const RobotDescription::JointDesc*> jointsByName;  // name-sorted, as urdfdom
    std::map<std::string, bool> isChild;
    for (const auto& j : d.joints) {
        jointsByName[j.name] = &j;
        isChild[j.child] = true;
    }
    std::string root;
    for (const auto& l : d.links)
        if (!isChild.count(l.name)) {
            if (!root.empty()) throw std::runtime_error("robot description has several root links");
            root = l.name;
        }
    if (root.empty()) throw std::runtime_error("robot description has no root link");

    auto linkInertia = [](const RobotDescription::Link& l) {
        if (!l.hasInertial) return BodyInertia{};
        const Mat3 I{{{l.ixx, l.ixy, l.ixz}, {l.ixy, l.iyy, l.iyz}, {l.ixz, l.iyz, l.izz}}};
        const Mat3 R = RpyToMatrix(l.comRpy);
        return BodyInertia::FromCom(l.mass, l.comXyz, MatMul(MatMul(R, I), Transpose(R)));
    };

    Model m;
    m.name = d.name;
    m.joints.emplace_back();
    m.joints[0].name = "universe";
    m.joints[0].type = JointType::FreeFlyer;
    m.joints[0].nq = m.joints[0].nv = 0;
    Joint rootJoint;
    rootJoint.name = "root_joint";
    rootJoint.type = JointType::FreeFlyer;
    rootJoint.parent = 0;
    rootJoint.nq = 7;
    rootJoint.nv = 6;
    rootJoint.inertia = linkInertia(*links.at(root));
    m.joints.push_back(rootJoint);
    m.frames.push_back({"universe", 0, Placement{}});
    m.frames.push_back({"root_joint", 1, Placement{}});
    m.frames.push_back({root, 1, Placement{}});

    // Depth-first over links in joint-name order; `support` = index of the moving joint carrying
    // the link, `toSupport` = placement of the link frame in that joint's frame.  A moving joint
    // receives its index when it is first visited, so numbering is Pinocchio's DFS numbering.
    struct Visitor {
        const std::map<std::string, const RobotDescription::Link*>& links;
        const std::map<std::string, const RobotDescription::JointDesc*>& jointsByName;
        Model& m;
        decltype(linkInertia)& inertiaOf;
        void Visit(const std::string& link, int support, const Placement& toSupport) {
            for (const auto& [jname, j] : jointsByName) {
                if (j->parent != link) continue;
                Placement jp;
                jp.R = RpyToMatrix(j->rpy);
                jp.p = j->xyz;
                const Placement inSupport = toSupport * jp;
                const RobotDescription::Link& child = *links.at(j->child);
                if (j->type == "fixed") {
                    m.joints[static_cast<std::size_t>(support)].inertia += inertiaOf(child).Transported(inSupport);
                    m.frames.push_back({jname, support, inSupport});
                    m.frames.push_back({j->child, support, inSupport});
                    Visit(j->child, support, inSupport);
                } else if (j->type == "revolute" || j->type == "continuous") {
                    Joint nj;
                    nj.name = jname;
                    nj.type = JointType::Revolute;
                    nj.parent = support;
                    nj.placement = inSupport;
                    nj.axis = j->axis;
                    nj.nq = nj.nv = 1;
                    nj.inertia = inertiaOf(child);
                    m.joints.push_back(nj);
                    m.frames.push_back({jname, static_cast<int>(m.joints.size()) - 1, Placement{}});
                    m.frames.push_back({j->child, static_cast<int>(m.joints.size()) - 1, Placement{}});
                    Visit(j->child, static_cast<int>(m.joints.size()) - 1, Placement{});
                } else {
                    throw std::runtime_error("robot description: unsupported joint type '" + j->type + "'");
                }
            }
        }
    } visitor{links, jointsByName, m, linkInertia};
    visitor.Visit(root, 1, Placement{});
    for (Joint& j : m.joints) {
        j.idxQ = m.nq;
        j.idxV = m.nv;
        m.nq += j.nq;
        m.nv += j.nv;
    }
    return m;
}

}  // namespace ungar_amd::rbd
