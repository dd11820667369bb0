"""sonar_slam_amd -- MI355X (gfx950) native sonar-SLAM front end.

Drop-in replacements for the two native modules of jake3991/sonar-SLAM (``bruce_slam.cfar``
and ``bruce_slam.pcl``) plus the device-resident feature-extraction / scan-matching pipeline
behind them.  Host code is Python + ctypes over the C ABI of ``libsonarfe.so``
(include/sonarfe.h); all compute is hand-written HIP.  No PyTorch, no CPU fallback.

    sonar_slam_amd.cfar                -> bruce_slam.cfar   (cpp/cfar.cpp)
    sonar_slam_amd.pcl                 -> bruce_slam.pcl    (cpp/pcl.cpp)
    sonar_slam_amd.CFAR                -> bruce_slam.CFAR   (CFAR.py, host-side tau solving)
    sonar_slam_amd.feature_extraction  -> ROS-free core of bruce_slam.feature_extraction
    sonar_slam_amd.farm                -> multi-GPU job farm (one worker per device)
"""
__all__ = ["cfar", "pcl", "CFAR", "feature_extraction", "farm", "icp_config"]
