// the reference includes OpenCV viz in kfusion/types.hpp but the hot path uses nothing from it
#pragma once
#include <opencv2/core/core.hpp>
