"""MI355X-native PointNet hot path behind the API of MAPieschl/PointCloudProcessing's
point_cloud_analysis package (pointnet.PointNet, pointcloud.PointCloudSet, pointnet_train).

Compute lives in libpointnet_hip.so (hand-written HIP for gfx950, C ABI in include/pointnet_hip.h);
PyTorch is used for device memory, streams and torch.distributed only.
"""
__all__ = ["_lib", "ops"]
