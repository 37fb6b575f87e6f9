// ungar_amd :: reference include path ungar/rbd/quantities/kinetic_energy.hpp; all quantities live in quantities.hpp.
#pragma once
#include "quantities.hpp"
