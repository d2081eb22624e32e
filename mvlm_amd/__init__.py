"""mvlm_amd - MI355X-native implementation of cvjena/mvlm's ``predict_one_file`` hot path.

Import surface mirrors the reference package (src/mvlm/__init__.py):
``mvlm_amd.pipeline.create_pipeline(name, **kw).predict_one_file(path)``.
Submodules are imported lazily so ``arch`` / ``weights`` / ``config`` stay usable
on machines without a GPU (packing weights, generating fixtures).
"""
__all__ = ["pipeline", "utils", "prediction", "arch", "weights", "config", "parallel"]


def __getattr__(name):
    if name in __all__:
        import importlib

        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
