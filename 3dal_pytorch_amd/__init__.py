"""3dal_pytorch_amd — MI355X-native Frustum-PointNet auto-labeling heads (3DAL static/dynamic).

The directory name starts with a digit, so import it with importlib:

    import importlib
    static_model = importlib.import_module("3dal_pytorch_amd.static_model")

or put its `dropin/` sub-directory on sys.path and `from static_model import ...` / `from dynamic_model import ...`
exactly as the reference's tools/ scripts do.
"""
__version__ = "0.1.0"
