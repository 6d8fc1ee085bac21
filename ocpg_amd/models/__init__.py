"""`from models import build_model` -- the reference's model API (models/__init__.py:3-5)."""
from .ocpg import build


def build_model(args):
    return build(args)
