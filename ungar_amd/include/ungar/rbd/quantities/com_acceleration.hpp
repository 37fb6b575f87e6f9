// ungar_amd :: reference include path ungar/rbd/quantities/com_acceleration.hpp; all quantities live in quantities.hpp.
#pragma once
#include "quantities.hpp"
