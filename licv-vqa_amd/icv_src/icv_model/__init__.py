from .icv_intervention import LearnableICVInterventionLMM  # noqa: F401
