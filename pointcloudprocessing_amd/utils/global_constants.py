# SE3 matrix constants (reference: point_cloud_analysis/utils/global_constants.py:1-4)
SE3_ROWS = 4
SE3_COLS = 4
SE3_SIZE = SE3_ROWS * SE3_COLS
