"""
MI355X-native implementation of the vectorised render path of
ninezerozeronine/raytracing-one-weekend.  Same Python surface as the
reference package of the same name (Camera, SphereGroup, TriangleGroup,
Disk, OBJTriMesh, materials.*, scenes.*, main.render); the arithmetic runs in
hand-written HIP kernels (libwrt.so) reached through ctypes.
"""
