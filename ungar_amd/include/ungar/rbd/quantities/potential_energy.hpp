// ungar_amd :: reference include path ungar/rbd/quantities/potential_energy.hpp; all quantities live in quantities.hpp.
#pragma once
#include "quantities.hpp"
