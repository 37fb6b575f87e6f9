// ungar_amd :: reference include path ungar/rbd/quantities/generalized_gravity.hpp; all quantities live in quantities.hpp.
#pragma once
#include "quantities.hpp"
