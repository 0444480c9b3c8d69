// Forwarding header: the reference include path <kfusion/kinfu.hpp> resolves to the MI355X shells.
#pragma once
#include <sobfu_amd/sobfu.hpp>
