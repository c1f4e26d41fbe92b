// image.h -- forwarding header: the reference's host code includes "image.h"; the definitions live in octane_types.hpp.
#pragma once
#include "octane_types.hpp"
