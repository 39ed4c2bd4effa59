"""MI355X-native EKF-SLAM predict/update core (host side).  Import as ``slam_duckietown_amd``."""
