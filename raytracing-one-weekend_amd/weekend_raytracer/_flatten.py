"""Scene objects -> the flat record Context.upload() takes."""
import numpy


def flatten(camera, object_groups, material_map, horizon, zenith):
    return dict(
        camera=camera._flat(),
        groups=[g._flat() for g in object_groups],
        materials={int(index): m._flat() for index, m in material_map.items()},
        sky=dict(horizon=numpy.asarray(horizon, dtype=numpy.single),
                 zenith=numpy.asarray(zenith, dtype=numpy.single)),
    )
