// ungar_amd :: what example/rbd/quantity.example.cpp:28 includes by this name -- forward kinematics of the frames, on ungar_amd's own rigid-body code
// (the `pinocchio` names live at the end of ungar/rbd/robot.hpp).
#pragma once

#include "../../ungar/rbd/robot.hpp"
