// ungar_amd :: reference include path ungar/rbd/quantities/composite_rigid_body_inertia.hpp; all quantities live in quantities.hpp.
#pragma once
#include "quantities.hpp"
