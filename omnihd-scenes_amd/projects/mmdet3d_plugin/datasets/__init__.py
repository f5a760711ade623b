from .newscenes_dataset import NewScenesDataset, camera_matrices, output_to_newsc_box  # noqa: F401
from .pipelines import LoadRadarPointsMultiSweeps, RadarPoints  # noqa: F401
