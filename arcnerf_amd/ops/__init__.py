"""Operator layer: ctypes-backed kernels (functional), their autograd wrappers, and the reference-named op modules."""
