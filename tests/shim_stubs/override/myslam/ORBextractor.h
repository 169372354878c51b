// the reference's include/myslam/ORBextractor.h is REPLACED by the shim header (same class, same
// members); this forwarding file stands in for that replacement when the reference's own headers and callers are parsed
// in place (tests/test_shims_vs_reference.py).
#include "myslam_shim/ORBextractor.h"
