"""Module path of the reference (projects/mmdet3d_plugin/core/points/radar_points.py, imported by its loader at
datasets/pipelines/loading.py:9); the container itself lives next to the loader that fills it."""
from projects.mmdet3d_plugin.datasets.pipelines.loading import RadarPoints  # noqa: F401

__all__ = ["RadarPoints"]
