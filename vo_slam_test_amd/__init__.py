"""MI355X-native ORB front end + bundle adjustment hot path (drop-in for the
ORBextractor / Matcher / Optimizer path of guisongchen/vo_slam_test).

The product is the C-ABI library built from ``csrc/`` (``include/vo_hip.h``);
this package is the thin Python harness used by tests and ``bench.py``.
"""
__all__ = ["synth"]
