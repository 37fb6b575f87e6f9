#!/usr/bin/env python3
"""'.robot' text (tools/urdf_to_robot.py) -> a minimal URDF with the same kinematic / inertial DATA: links with <inertial>, joints with <origin>, <axis>, parent and
child.  The reference's own rbd test and examples open `UNGAR_DATA_FOLDER "/robots/anymal_b_description/robots/anymal.urdf"` at run time
(test/rbd/robot.test.cpp:94-95); /root/reference does not exist on the GPU box, so oracle/ref_tests/build_ref_tests.sh writes this file under oracle/_ref/data
(git-ignored, travels with the snapshot) from ungar_amd/data/anymal_b.robot.  It also exercises the URDF reader of csrc/rbd/model.hpp on the GPU box.
usage: robot_to_urdf.py <in.robot> <out.urdf>"""
import os
import sys


def main(src, dst):
    name, links, joints = "robot", [], []
    for line in open(src):
        t = line.split()
        if not t or t[0].startswith("#"):
            continue
        if t[0] == "robot":
            name = t[1]
        elif t[0] == "link":
            links.append(t[1:])
        elif t[0] == "joint":
            joints.append(t[1:])
    out = ['<?xml version="1.0"?>', f'<robot name="{name}">']
    for l in links:
        if l[1] == "0":
            out.append(f'  <link name="{l[0]}"/>')
            continue
        mass, xyz, rpy, I = l[2], " ".join(l[3:6]), " ".join(l[6:9]), l[9:15]
        out += [f'  <link name="{l[0]}">', "    <inertial>", f'      <origin xyz="{xyz}" rpy="{rpy}"/>', f'      <mass value="{mass}"/>',
                f'      <inertia ixx="{I[0]}" ixy="{I[1]}" ixz="{I[2]}" iyy="{I[3]}" iyz="{I[4]}" izz="{I[5]}"/>', "    </inertial>", "  </link>"]
    for j in joints:
        out += [f'  <joint name="{j[0]}" type="{j[1]}">', f'    <parent link="{j[2]}"/>', f'    <child link="{j[3]}"/>',
                f'    <origin xyz="{" ".join(j[4:7])}" rpy="{" ".join(j[7:10])}"/>', f'    <axis xyz="{" ".join(j[10:13])}"/>', "  </joint>"]
    out.append("</robot>")
    os.makedirs(os.path.dirname(os.path.abspath(dst)), exist_ok=True)
    with open(dst, "w") as f:
        f.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
