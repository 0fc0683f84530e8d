"""`simple_knn` as the reference imports it (scene/c_gaussian_model.py:20) -> ex4dgs_amd's HIP distCUDA2."""
