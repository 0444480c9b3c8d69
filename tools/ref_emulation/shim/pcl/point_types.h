// forwards to tools/ref_emulation/shim/cuemu_pcl.h
#pragma once
#include <cuemu_pcl.h>
