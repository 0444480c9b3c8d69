// Forwarding header: the reference include path <sobfu/reductor.hpp> resolves to the MI355X shells.
#pragma once
#include <sobfu_amd/sobfu.hpp>
