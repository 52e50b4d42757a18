"""MI355X-native hot path of 3D-Pose-Estimation-with-Previleged-Information."""
