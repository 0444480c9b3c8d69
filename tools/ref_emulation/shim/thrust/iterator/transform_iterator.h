// the reference includes thrust on the hot path but never calls it there; marching cubes gets its scan from cuemu_thrust.h
#pragma once
#include <cuemu_thrust.h>
