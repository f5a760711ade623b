from .loading import LoadRadarPointsMultiSweeps, RadarPoints, merge_radar_sweeps  # noqa: F401
