from .body_models import SMPLX, create  # noqa: F401
