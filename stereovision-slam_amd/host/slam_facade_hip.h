// slam_facade_hip.h — the facade (slam_facade.h) bound to the HIP kernels: the classes a user of the
// reference switches to.  Link with lib/libsvslam_hip.so (and -lz for the PNG reader).
#pragma once
#include "kernels_hip.h"
#include "slam_facade.h"

namespace svs {
namespace facade {
using Frontend = FrontendT<HipKernels>;
using VisualOdometry = VisualOdometryT<HipKernels>;
} // namespace facade
} // namespace svs
