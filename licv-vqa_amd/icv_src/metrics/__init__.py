from .vqa_metric import compute_vqa_accuracy, postprocess_vqa_generation, vqa_postprocess  # noqa: F401
