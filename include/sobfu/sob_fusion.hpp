// Forwarding header: the reference include path <sobfu/sob_fusion.hpp> resolves to the MI355X shells.
#pragma once
#include <sobfu_amd/sobfu.hpp>
