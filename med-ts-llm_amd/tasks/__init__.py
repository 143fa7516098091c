"""Trainer registry with the reference's surface (R:tasks/__init__.py:9-20)."""
from .base import BaseTask
from .tasks import (ForecastTask, ReconstructionTask, AnomalyDetectionTask, SegmentationTask, SemanticSegmentationTask,
                    PretrainingTask)

task_lookup = {
    "forecasting": ForecastTask,
    "anomaly_detection": AnomalyDetectionTask,
    "reconstruction": ReconstructionTask,
    "segmentation": SegmentationTask,
    "semantic_segmentation": SemanticSegmentationTask,
    "pretraining": PretrainingTask,
}


def get_trainer(run_id, config):
    return task_lookup[config.task](run_id, config)
