"""Drop-in for the reference's `simple_knn` package (submodules/simple-knn): `from simple_knn._C import distCUDA2`
(scene/c_gaussian_model.py:20) resolves here when ex4dgs_amd/ is on the import path."""
