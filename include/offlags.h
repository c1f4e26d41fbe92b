// offlags.h -- forwarding header: the reference's host code includes "offlags.h"; the definitions live in octane_types.hpp.
#pragma once
#include "octane_types.hpp"
