"""Host side of the GPU backend: scene packing, flat BVH construction and the ctypes binding."""
