// The matching kernels of the library (csrc/metrics.hip), compiled for the host against the
// emulation header in this directory: the same source text, executed by OS threads.
#include "../../2d_object_detection_amd/csrc/metrics.hip"
